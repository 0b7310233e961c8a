#!/usr/bin/env python3
"""w_check.py -- which accumulation kernels run for a few formats (set_timing names), and parity of the records with the oracle."""
import sys, os
from fractions import Fraction
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from grav1synth_amd.diff import DiffGenerator, Frame, format_tbl
from grav1synth_amd.synth import SynthSpec, make_pair
from tests.helpers import oracle_run

def run(spec, lag, chroma, nframes, batch=None, check=True):
    g = DiffGenerator(Fraction(24, 1), spec.bit_depth, spec.bit_depth, ar_coeff_lag=lag, luma_only=not chroma,
                      batch_frames=batch or nframes)
    g.set_timing(True)
    for k in range(nframes):
        s, d = make_pair(spec, k, device="cuda")
        g.diff_frame(Frame(s, spec.xdec, spec.ydec), Frame(d, spec.xdec, spec.ydec))
    g.sync()
    kt = g.kernel_times()
    mine = format_tbl(g.finish())
    ref = oracle_run(spec, range(nframes), lag=lag, chroma=chroma)[0] if check else mine
    names = {k: (round(v[0] * 1e3 / max(v[1], 1), 1), v[1]) for k, v in kt.items() if k.startswith("k3")}
    print(f"{spec.width}x{spec.height} {spec.bit_depth}b {spec.xdec}{spec.ydec} lag{lag} {'yuv' if chroma else 'y'}: table {'==' if mine == ref else '!='} oracle; us per launch: {names}", flush=True)
    return mine == ref

if __name__ == "__main__":
    ok = True
    ok &= run(SynthSpec(320, 192, 8), 3, True, 3)
    ok &= run(SynthSpec(352, 208, 10), 3, True, 2)
    ok &= run(SynthSpec(320, 192, 8), 2, False, 2)
    ok &= run(SynthSpec(1920, 1080, 8), 3, True, 16, check=False)
    ok &= run(SynthSpec(3840, 2160, 10), 3, True, 32, check=False)
    ok &= run(SynthSpec(3840, 2160, 10), 3, True, 32, check=False)
    ok &= run(SynthSpec(3840, 2160, 10, textured=False), 3, True, 32, check=False)
    print("ALL OK" if ok else "MISMATCH")
