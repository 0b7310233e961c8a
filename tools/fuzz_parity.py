#!/usr/bin/env python3
"""tools/fuzz_parity.py [N] [SEED] [WMAX HMAX] -- random geometries / depths / subsamplings / lags through the oracle comparison
of tests/test_gpu_parity.py (records and table, bit for bit).  Prints the failing specs, if any.
FUZZ_ALIGN=16: widths rounded down to multiples of 16 -- every case then runs the WIDE chain (whole 8-sample words in every
plane: engine.hip wide_ok); with random widths one case in sixteen does, the others take the fallback chain."""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grav1synth_amd.synth import SynthSpec
from tests import test_gpu_parity as T

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
WMAX, HMAX = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (420, 300)
ALIGN = int(os.environ.get("FUZZ_ALIGN", "1"))
bad = 0
t0 = time.time()
for k in range(n):
    w, h = max(80, rng.randint(66, WMAX) // ALIGN * ALIGN), rng.randint(66, HMAX)
    bd = rng.choice([8, 10, 12])
    xd, yd = rng.choice([(1, 1), (1, 1), (1, 0), (0, 0)])
    lag = rng.choice([3, 3, 2, 1])
    chroma = rng.random() < 0.8
    spec = SynthSpec(w, h, bd, xdec=xd, ydec=yd, textured=rng.random() < 0.6, gain_scale=rng.choice([1, 1, 2, 4]))
    case = (spec, lag, chroma, 2, rng.random() < 0.7)
    try:
        T.test_records_and_table_match_oracle(case)
    except RuntimeError as e:  # "Not enough flat blocks": the reference's error, raised by the oracle side first
        if "flat blocks" not in str(e):
            bad += 1
            print("FAIL", case, repr(e)[:300])
    except BaseException as e:
        bad += 1
        print("FAIL", case, repr(e)[:300])
print(f"{n} single-batch cases, {bad} failures, {time.time() - t0:.0f} s")

# second sweep: several frames in ragged batches, mixed source / denoised depths, table only
from fractions import Fraction
import numpy as np
from grav1synth_amd.diff import DiffGenerator, Frame, format_tbl
from tests.helpers import np_pair
from tests.oracle_binding import OracleDiff, format_tbl as oracle_tbl

bad2 = 0
t0 = time.time()
for k in range(n // 4):
    w, h = max(80, rng.randint(66, WMAX) // ALIGN * ALIGN), rng.randint(66, HMAX)
    sbd, dbd = rng.choice([(8, 8), (10, 10), (10, 8), (8, 10), (12, 10)])
    xd, yd = rng.choice([(1, 1), (1, 0), (0, 0)])
    lag = rng.choice([3, 3, 2, 1])
    nf, bf = rng.randint(3, 7), rng.randint(1, 4)
    ss = SynthSpec(w, h, sbd, xdec=xd, ydec=yd, textured=rng.random() < 0.6)
    ds = SynthSpec(w, h, dbd, xdec=xd, ydec=yd, textured=ss.textured)
    try:
        o = OracleDiff(30000, 1001, sbd, dbd, lag, True)
        g = DiffGenerator(Fraction(30000, 1001), sbd, dbd, ar_coeff_lag=lag, batch_frames=bf)
        damage = rng.random() < 0.35   # outliers |src - den| > 127: the deferred exact path
        cut = rng.randint(1, nf - 1) if rng.random() < 0.35 else nf   # noise gain changes there: a new segment
        ss2 = SynthSpec(w, h, sbd, xdec=xd, ydec=yd, textured=ss.textured, gain_scale=3)
        for f in range(nf):
            s, _ = np_pair(ss if f < cut else ss2, f)
            _, d = np_pair(ds, f)
            if damage:
                nr = np.random.default_rng(rng.randint(0, 1 << 30))
                d = [p.copy() for p in d]
                for c in range(len(d)):
                    hh, ww = d[c].shape
                    for _ in range(nr.integers(1, 10)):
                        y, x = int(nr.integers(0, hh)), int(nr.integers(0, ww))
                        d[c][y, x] = 0 if (int(s[c][y, x]) >> (sbd - 8)) > 140 else ((255 << (dbd - 8)))
            o.diff_frame(s, d, xd, yd)
            g.diff_frame(Frame(s, xd, yd), Frame(d, xd, yd))
        a, b = format_tbl(g.finish()), oracle_tbl(o.finish())
        g.close()
        if a != b:
            bad2 += 1
            print("FAIL tbl", (w, h, sbd, dbd, xd, yd, lag, nf, bf))
    except RuntimeError as e:
        if "flat blocks" not in str(e) and "G1S_ERR_NOT_ENOUGH_FLAT" not in str(e):
            bad2 += 1
            print("FAIL", (w, h, sbd, dbd, xd, yd, lag, nf, bf), repr(e)[:200])
    except BaseException as e:
        if "NOT_ENOUGH_FLAT" not in repr(e):
            bad2 += 1
            print("FAIL", (w, h, sbd, dbd, xd, yd, lag, nf, bf), repr(e)[:200])
print(f"{n // 4} multi-frame mixed-depth cases, {bad2} failures, {time.time() - t0:.0f} s")
