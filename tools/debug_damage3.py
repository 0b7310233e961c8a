#!/usr/bin/env python3
"""tools/debug_damage3.py [N] [SEED] -- failing damaged frames reduced to single hits; S error structure printed."""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fractions import Fraction
import numpy as np
from grav1synth_amd.diff import DiffGenerator, Frame
from grav1synth_amd.synth import SynthSpec
from tests.helpers import np_pair
from tests.oracle_binding import OracleDiff

def run(s, d, bd, xd, yd, lag):
    o = OracleDiff(24, 1, bd, bd, lag, True)
    g = DiffGenerator(Fraction(24, 1), bd, bd, ar_coeff_lag=lag, batch_frames=1)
    try:
        o.diff_frame(s, d, xd, yd)
    except RuntimeError:
        g.close(); return None
    g.diff_frame(Frame(s, xd, yd), Frame(d, xd, yd))
    try:
        g.sync()
    except Exception as e:
        g.close(); return None
    r = g.last_record()
    out = []
    for c in range(3):
        S, Sb, nobs = o.ar_sums(c); S2, Sb2, nobs2 = r.ar_sums(c)
        out.append((np.array_equal(S, S2) and np.array_equal(Sb, Sb2) and nobs == nobs2, (S2 - S).astype(np.int64), (np.asarray(Sb2) - np.asarray(Sb)).astype(np.int64), nobs2 - nobs))
    fl = o.flat_mask().copy()
    g.close()
    return out, fl

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
for k in range(n):
    w, h = rng.randint(66, 300), rng.randint(66, 300)
    bd = rng.choice([8, 10])
    xd, yd = rng.choice([(1, 1), (1, 0), (0, 0)])
    lag = rng.choice([3, 2, 1])
    spec = SynthSpec(w, h, bd, xdec=xd, ydec=yd, textured=rng.random() < 0.6)
    s, d0 = np_pair(spec, k)
    nr = np.random.default_rng(rng.randint(0, 1 << 30))
    hits = []
    for c in range(3):
        hh, ww = d0[c].shape
        for _ in range(nr.integers(1, 4)):
            y, x = int(nr.integers(0, hh)), int(nr.integers(0, ww))
            hits.append((c, x, y))
    def damaged(sel):
        d = [p.copy() for p in d0]
        for c, x, y in sel:
            d[c][y, x] = 0 if (int(s[c][y, x]) >> (bd - 8)) > 140 else (255 << (bd - 8))
        return d
    res = run(s, damaged(hits), bd, xd, yd, lag)
    if res is None or all(o[0] for o in res[0]): continue
    print(f"case {k}: {w}x{h} {bd}b xd{xd} yd{yd} lag{lag} planes failing {[c for c in range(3) if not res[0][c][0]]}")
    res0 = run(s, damaged([]), bd, xd, yd, lag)
    print("   no hits:", "ok" if all(o[0] for o in res0[0]) else "FAIL")
    for hsel in hits:
        r1 = run(s, damaged([hsel]), bd, xd, yd, lag)
        if r1 is None: continue
        ok = all(o[0] for o in r1[0])
        c, x, y = hsel
        bwc, bhc = (32 >> xd, 32 >> yd) if c else (32, 32)
        fl = r1[1]
        dv = (int(s[c][y, x]) >> (bd - 8)) - (int(damaged([hsel])[c][y, x]) >> (bd - 8))
        print(f"   hit plane {c} px ({x},{y}) block ({x // bwc},{y // bhc}) in-block ({x % bwc},{y % bhc}) d={dv}: {'ok' if ok else 'FAIL'}")
        if not ok:
            print("   flat map:\n" + "\n".join("       " + "".join("#" if v else "." for v in row) for row in fl))
            for cc in range(3):
                okc, E, Eb, dn = r1[0][cc]
                if okc: continue
                nzr = sorted(set(np.argwhere(E != 0)[:, 0].tolist())); nzc = sorted(set(np.argwhere(E != 0)[:, 1].tolist()))
                print(f"     plane {cc}: dnobs {dn}; S err nonzero rows {nzr} cols {nzc}; Sb err nonzero {np.flatnonzero(Eb).tolist()}")
                print("     S err diag:", np.diag(E).tolist())
                print("     Sb err:", Eb.tolist())
