#!/usr/bin/env python3
"""tools/make_y4m_fixture.py OUTDIR -- the Y4M pairs of the reference pin (oracle/ref_harness, tests/test_reference_pin.py).

Small seeded source / denoised pairs from the repository's own integer generator (grav1synth_amd/synth.py), written with
the repository's Y4M writer: <name>_source.y4m, <name>_denoised.y4m.  Deterministic: the same bytes on every machine, so
a table the real av1-grain wrote for them elsewhere (oracle/ref_harness/build.sh) can be committed as a golden and
checked here without the files themselves travelling."""
import os
import sys
from fractions import Fraction

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from grav1synth_amd.ingest import write_y4m  # noqa: E402
from grav1synth_amd.synth import SynthSpec, make_pair  # noqa: E402

FIXTURES = {
    # name: (spec, frames, fps)
    "320x192_8b_420": (SynthSpec(320, 192, 8), 4, Fraction(24, 1)),
    "320x200_10b_420": (SynthSpec(320, 200, 10), 3, Fraction(30000, 1001)),
    "256x160_10b_444": (SynthSpec(256, 160, 10, xdec=0, ydec=0), 2, Fraction(24, 1)),
}


def frames_of(name):
    spec, n, fps = FIXTURES[name]
    src, den = [], []
    for k in range(n):
        s, d = make_pair(spec, k, device="cpu")
        src.append([p.numpy() for p in s])
        den.append([p.numpy() for p in d])
    return spec, fps, src, den


def main(outdir):
    os.makedirs(outdir, exist_ok=True)
    for name in FIXTURES:
        spec, fps, src, den = frames_of(name)
        write_y4m(os.path.join(outdir, name + "_source.y4m"), src, spec.bit_depth, spec.xdec, spec.ydec, fps)
        write_y4m(os.path.join(outdir, name + "_denoised.y4m"), den, spec.bit_depth, spec.xdec, spec.ydec, fps)
        print("wrote", name)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "oracle", "_ref", "fixtures"))
