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


def generate(name):
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
    names = FULL_SIZE if sys.argv[1:2] == ["full"] else GOLDEN
    if len(sys.argv) > 2:  # `full NAME ...`: only those
        names = sys.argv[2:]
    for name in names:
        with open(os.path.join(HERE, name), "wb") as f:
            f.write(generate(name))
        print("wrote", name)
