"""python tools/mode_dump.py OUT.pkl [case indices] -- the records of a few frames per case under the G1S_K3 of the environment;
python tools/mode_dump.py --cmp A.pkl B.pkl -- where two dumps differ (plane, entries), case by case.  A debugging aid for
tests/test_gpu_parity.py::test_accumulation_modes_agree."""
import os
import pickle
import sys
from fractions import Fraction

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from grav1synth_amd.synth import SynthSpec  # noqa: E402

CASES = [
    (SynthSpec(320, 192, 8), 3, True, 2),
    (SynthSpec(320, 200, 8), 3, True, 2),
    (SynthSpec(326, 198, 10), 2, True, 2),
    (SynthSpec(256, 160, 10, xdec=0, ydec=0), 3, True, 2),
    (SynthSpec(320, 192, 10, xdec=1, ydec=0), 1, True, 2),
    (SynthSpec(320, 192, 8, xdec=0, ydec=1), 3, True, 2),
    (SynthSpec(320, 192, 8), 2, False, 2),
    (SynthSpec(1280, 720, 12), 3, True, 2),
    (SynthSpec(960, 544, 10, gain_scale=8), 3, True, 2),
    (SynthSpec(1920, 1080, 10), 3, True, 2),
    (SynthSpec(3840, 2160, 10), 3, True, 2),
    (SynthSpec(300, 180, 8, textured=False), 3, True, 2),
    (SynthSpec(64, 64, 8), 3, True, 2),
]


def dump(path, which):
    from grav1synth_amd.diff import DiffGenerator, format_tbl
    from grav1synth_amd.synth import make_pair

    out = {}
    for ci in which:
        spec, lag, chroma, n = CASES[ci]
        g = DiffGenerator(Fraction(24, 1), spec.bit_depth, spec.bit_depth, ar_coeff_lag=lag, luma_only=not chroma, batch_frames=2)
        frames = []
        for k in range(n):
            s, d = make_pair(spec, 7 + k, device="cuda")
            g.diff_frame(s, d, spec.xdec, spec.ydec)
            g.sync()
            r = g.last_record()
            fr = {"mask": r.flat_mask().copy(), "scores": r.scores().view(np.uint32).copy()}
            for c in range(3 if chroma else 1):
                S, Sb, nobs = r.ar_sums(c)
                fr["S%d" % c] = np.array(S)
                fr["Sb%d" % c] = np.array(Sb)
                fr["nobs%d" % c] = int(nobs)
                fr["stats%d" % c] = [np.array(a) for a in r.block_stats(c)]
            frames.append(fr)
        out[ci] = {"frames": frames, "tbl": format_tbl(g.finish())}
        print("case", ci, spec.width, spec.height, spec.bit_depth, spec.xdec, spec.ydec, "lag", lag, "done", flush=True)
    with open(path, "wb") as f:
        pickle.dump(out, f)


def cmp(a, b):
    A, B = pickle.load(open(a, "rb")), pickle.load(open(b, "rb"))
    bad = 0
    for ci in sorted(set(A) & set(B)):
        spec, lag, chroma, n = CASES[ci]
        msgs = []
        for k, (fa, fb) in enumerate(zip(A[ci]["frames"], B[ci]["frames"])):
            for key in fa:
                va, vb = fa[key], fb[key]
                if key.startswith("stats"):
                    flat = fa["mask"].ravel() != 0
                    for j, (x, y) in enumerate(zip(va, vb)):
                        x, y = x.ravel()[flat], y.ravel()[flat]
                        if not np.array_equal(x, y):
                            w = np.flatnonzero(x != y)
                            msgs.append(f"frame {k} {key}[{j}]: {len(w)} flat blocks differ, first (flat index) {w[:6].tolist()} {x[w[:3]].tolist()} vs {y[w[:3]].tolist()}")
                elif isinstance(va, int):
                    if va != vb:
                        msgs.append(f"frame {k} {key}: {va} vs {vb}")
                elif not np.array_equal(va, vb):
                    w = np.argwhere(va != vb)
                    msgs.append(f"frame {k} {key}: {len(w)} of {va.size} entries differ, first {w[:8].tolist()} : {va[tuple(w[0])]} vs {vb[tuple(w[0])]}")
        if A[ci]["tbl"] != B[ci]["tbl"]:
            msgs.append("tables differ")
        print(f"case {ci} ({spec.width}x{spec.height} {spec.bit_depth}b {spec.xdec}{spec.ydec} lag {lag}):", "EQUAL" if not msgs else "DIFFERENT")
        for m in msgs[:14]:
            print("   ", m)
        bad += bool(msgs)
    print("cases that differ:", bad)
    return bad


if __name__ == "__main__":
    if sys.argv[1] == "--cmp":
        sys.exit(1 if cmp(sys.argv[2], sys.argv[3]) else 0)
    which = [int(x) for x in sys.argv[2:]] or list(range(len(CASES)))
    dump(sys.argv[1], which)
