"""Generates tests/golden/oracle_*.tbl with the CPU oracle on seeded synthetic
frames.  These pin the ORACLE against regressions (and the GPU path against the
oracle); they do NOT pin the oracle against the reference: the reference has
no `diff` vectors and its arithmetic (crate av1-grain 0.4.2) cannot be built
here.  reference-example-table.tbl is the reference's own data
file tests/example-table.tbl (format anchor for the writer/parser)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from fractions import Fraction  # noqa: E402

from grav1synth_amd.synth import SynthSpec  # noqa: E402
from tests.helpers import oracle_run  # noqa: E402

GOLDEN = {
    "oracle_320x192_8b_420_lag3.tbl": dict(spec=SynthSpec(320, 192, 8), frames=3, lag=3, chroma=True),
    "oracle_320x200_10b_420_lag3.tbl": dict(spec=SynthSpec(320, 200, 10), frames=2, lag=3, chroma=True),
    "oracle_256x160_10b_444_lag3.tbl": dict(spec=SynthSpec(256, 160, 10, xdec=0, ydec=0), frames=2, lag=3, chroma=True),
    "oracle_320x192_8b_lag2_luma.tbl": dict(spec=SynthSpec(320, 192, 8), frames=2, lag=2, chroma=False),
    "oracle_scenecut_30000_1001.tbl": dict(spec=SynthSpec(320, 192, 8), frames=6, lag=3, chroma=True, cut=3),
}


# BASELINE.json's configurations at their full sizes (configs[1], configs[0]'s 1080p 4:2:0 sibling, configs[4]'s format):
# minutes of oracle time, so the CPU suite does not regenerate them -- `python -m tests.golden.make_golden full` does --
# and the `-m gpu` suite compares the HIP path's table with the committed bytes (tests/test_gpu_parity.py).
FULL_SIZE = {
    "oracle_full_1920x1080_8b_lag2_luma.tbl": dict(spec=SynthSpec(1920, 1080, 8), frames=3, lag=2, chroma=False),
    "oracle_full_1920x1080_8b_420_lag3.tbl": dict(spec=SynthSpec(1920, 1080, 8), frames=3, lag=3, chroma=True),
    "oracle_full_7680x4320_10b_444_lag3.tbl": dict(spec=SynthSpec(7680, 4320, 10, xdec=0, ydec=0), frames=2, lag=3, chroma=True),
    # configs[0] as BASELINE.json states it: 1080p 8-bit 4:2:0, 30 frames, lag 3, chroma
    "oracle_full_1920x1080_8b_420_lag3_30frames.tbl": dict(spec=SynthSpec(1920, 1080, 8), frames=30, lag=3, chroma=True),
    # configs[2]'s format with a scene cut (the noise gain triples from frame 4 on): is_different at the bench workload's size
    "oracle_full_3840x2160_10b_420_lag3_cut.tbl": dict(spec=SynthSpec(3840, 2160, 10), frames=8, lag=3, chroma=True, cut=4, fps=(24, 1)),
}

# configs[3] at its stated size: 3840x2160 10-bit 4:2:0, lag 3, chroma, 1000 frames (125 a shard x 8).  Two scene cuts: frame 500
# is a boundary of the 125-frame shards (and lies inside a 64-frame batch), frame 768 is a boundary of the 64-frame batches the
# streaming job deals (and lies inside a 125-frame shard).  About an hour of oracle time on one core:
# `python -m tests.golden.make_golden long` writes it; the `-m gpu` suite compares the sharded HIP jobs with the committed bytes.
LONG = {
    "oracle_full_3840x2160_10b_420_lag3_1000frames.tbl": dict(spec=SynthSpec(3840, 2160, 10), frames=1000, lag=3, chroma=True,
                                                              cuts=(500, 768), fps=(24, 1)),
    # configs[4]'s format in eight shards: 7680x4320 10-bit 4:4:4, lag 3, chroma, 32 frames (4 a shard x 8; 398 MB a frame pair).  Cuts at
    # frame 16 (a boundary of the 4-frame shards and batches) and at frame 22 (inside one).  Ten minutes of oracle time.
    "oracle_full_7680x4320_10b_444_lag3_32frames.tbl": dict(spec=SynthSpec(7680, 4320, 10, xdec=0, ydec=0), frames=32, lag=3, chroma=True,
                                                            cuts=(16, 22), fps=(24, 1), batch=4),
}


def frame_specs(g):
    """the SynthSpec of every frame of a golden's job: the noise gain triples between an odd and the next even cut"""
    spec = g["spec"]
    cuts = tuple(g["cuts"]) if "cuts" in g else ((g["cut"],) if "cut" in g else ())
    b = SynthSpec(spec.width, spec.height, spec.bit_depth, xdec=spec.xdec, ydec=spec.ydec, gain_scale=3)
    return [b if sum(k >= c for c in cuts) & 1 else spec for k in range(g["frames"])]


def generate_long(name, workers=3, checkpoint_every=25):
    """a LONG golden: the frames are made ahead on a few threads (a 4K pair is a second of integer torch work), the oracle takes
    them in order.  Two aids for an hour-long job: the oracle compiled for this machine's vector unit (oracle/Makefile
    `_variants/liborc_native.so`: the same IEEE operations packed, no contraction, no reassociation -- checked here against every
    committed small golden and the 4K scene-cut golden's first frames before it is trusted), and a checkpoint of the generator's
    state every few frames (/tmp/<name>.ckpt: orc_diff_save) from which an interrupted run carries on."""
    import pickle
    import time
    from concurrent.futures import ThreadPoolExecutor

    from tests.helpers import np_pair
    from tests.oracle_binding import OracleDiff, format_tbl, load_variant

    L = load_variant("native")
    for gname, gg in GOLDEN.items():  # the native build writes the committed bytes
        o = OracleDiff(*( gg.get("fps", (30000, 1001)) if "cut" in gg else (24, 1)), gg["spec"].bit_depth, gg["spec"].bit_depth, gg["lag"], gg["chroma"], library=L)
        for k, sp in enumerate(frame_specs(gg)):
            s, d = np_pair(sp, k)
            o.diff_frame(s if gg["chroma"] else s[:1], d if gg["chroma"] else d[:1], sp.xdec, sp.ydec)
        with open(os.path.join(HERE, gname), "rb") as f:
            assert format_tbl(o.finish()) == f.read(), f"the native build of the oracle differs on {gname}"
    g = LONG[name]
    specs = frame_specs(g)
    o = OracleDiff(g["fps"][0], g["fps"][1], g["spec"].bit_depth, g["spec"].bit_depth, g["lag"], g["chroma"], library=L)
    # ... and the same states as the base build after two 4K frames (every f64 of the noise model)
    ob = OracleDiff(g["fps"][0], g["fps"][1], g["spec"].bit_depth, g["spec"].bit_depth, g["lag"], g["chroma"])
    checkpoint_every = min(checkpoint_every, max(1, len(specs) // 8))
    ckpt = os.path.join("/tmp", name + ".ckpt")
    start = 0
    if os.path.exists(ckpt):
        with open(ckpt, "rb") as f:
            start, state = pickle.load(f)
        o.restore(state)
        print(f"  resuming behind frame {start}", flush=True)
    else:
        for k in range(2):
            s, d = np_pair(specs[k], k)
            o.diff_frame(s, d, specs[k].xdec, specs[k].ydec)
            ob.diff_frame(s, d, specs[k].xdec, specs[k].ydec)
        assert o.save() == ob.save(), "the native build of the oracle differs from the base build on 4K frames"
        start = 2
    ob.close()
    t0 = time.time()
    with ThreadPoolExecutor(workers) as ex:
        ahead = workers + 1
        futs = {k: ex.submit(np_pair, specs[k], k) for k in range(start, min(start + ahead, len(specs)))}
        for k in range(start, len(specs)):
            s, d = futs.pop(k).result()
            if k + ahead < len(specs):
                futs[k + ahead] = ex.submit(np_pair, specs[k + ahead], k + ahead)
            o.diff_frame(s, d, specs[k].xdec, specs[k].ydec)
            if (k + 1) % checkpoint_every == 0:
                with open(ckpt + ".tmp", "wb") as f:
                    pickle.dump((k + 1, o.save()), f)
                os.replace(ckpt + ".tmp", ckpt)
                print(f"  frame {k + 1}/{len(specs)}  {time.time() - t0:.0f} s  segments so far {o.num_segments()}", flush=True)
    return format_tbl(o.finish())


def generate(name):
    if name in LONG:
        return generate_long(name)
    g = GOLDEN[name] if name in GOLDEN else FULL_SIZE[name]
    spec = g["spec"]
    specs = None
    fps = Fraction(24, 1)
    if "cut" in g:
        b = SynthSpec(spec.width, spec.height, spec.bit_depth, gain_scale=3)
        specs = [spec if k < g["cut"] else b for k in range(g["frames"])]
        fps = Fraction(*g.get("fps", (30000, 1001)))
    tbl, _ = oracle_run(spec, range(g["frames"]), g["lag"], g["chroma"], fps=fps, specs_per_frame=specs)
    return tbl


if __name__ == "__main__":
    names = FULL_SIZE if sys.argv[1:2] == ["full"] else LONG if sys.argv[1:2] == ["long"] else GOLDEN
    if len(sys.argv) > 2:  # `full NAME ...`: only those
        names = sys.argv[2:]
    for name in names:
        tbl = generate(name)  # (the file is opened when there is something to write: a long job must not leave an empty golden behind)
        with open(os.path.join(HERE, name), "wb") as f:
            f.write(tbl)
        print("wrote", name)
