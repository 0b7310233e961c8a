"""Shared helpers for the parity tests (test infrastructure)."""
from __future__ import annotations

from fractions import Fraction
from typing import List, Sequence

import numpy as np

from grav1synth_amd.synth import SynthSpec, make_pair
from tests.oracle_binding import OracleDiff, format_tbl as oracle_format_tbl


def np_pair(spec: SynthSpec, frame: int):
    s, d = make_pair(spec, frame)
    return [p.numpy() for p in s], [p.numpy() for p in d]


def oracle_run(spec: SynthSpec, frames: Sequence[int], lag=3, chroma=True, fps=Fraction(24, 1),
               specs_per_frame=None, collect=None):
    """Run the CPU oracle over synthetic frames; returns (.tbl bytes, segments).
    `collect(oracle, frame_index)` is called after each frame when given."""
    o = OracleDiff(fps.numerator, fps.denominator, spec.bit_depth, spec.bit_depth, lag, chroma)
    for k, f in enumerate(frames):
        sp = specs_per_frame[k] if specs_per_frame else spec
        s, d = np_pair(sp, f)
        if not chroma:
            s, d = s[:1], d[:1]
        o.diff_frame(s, d, sp.xdec, sp.ydec)
        if collect:
            collect(o, k)
    segs = o.finish()
    return oracle_format_tbl(segs), segs


def record_from_oracle(o: OracleDiff, spec: SynthSpec, lag: int, nplanes: int):
    """Assemble a product record from the oracle's exact integer shadows."""
    from grav1synth_amd.diff import Record

    r = Record.blank(spec.width, spec.height, spec.xdec, spec.ydec, nplanes, lag)
    for c in range(nplanes):
        v = r.views(c)
        S, Sb, nobs = o.ar_sums(c)
        v["S"][:] = S
        v["Sb_nobs"][:-1] = Sb
        v["Sb_nobs"][-1] = nobs
        ls, sd, sd2 = o.block_stats(c)
        if c == 0:
            v["luma_sum"][:] = ls
            v["mask"][:] = o.flat_mask().ravel()
            v["scores"][:] = o.scores().ravel()
        v["sum_d"][:] = sd
        v["sum_d2"][:] = sd2
    return r
