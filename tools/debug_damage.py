#!/usr/bin/env python3
"""tools/debug_damage.py [N] [SEED] -- damaged frames (isolated |src - den| > 127) through the per-frame record comparison."""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fractions import Fraction
import numpy as np
from grav1synth_amd.diff import DiffGenerator, Frame
from grav1synth_amd.synth import SynthSpec
from tests.helpers import np_pair
from tests.oracle_binding import OracleDiff

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
bad = 0
for k in range(n):
    w, h = rng.randint(66, 300), rng.randint(66, 300)
    bd = rng.choice([8, 10])
    xd, yd = rng.choice([(1, 1), (1, 0), (0, 0)])
    lag = rng.choice([3, 2, 1])
    spec = SynthSpec(w, h, bd, xdec=xd, ydec=yd, textured=rng.random() < 0.6)
    o = OracleDiff(24, 1, bd, bd, lag, True)
    g = DiffGenerator(Fraction(24, 1), bd, bd, ar_coeff_lag=lag, batch_frames=1)
    s, d = np_pair(spec, k)
    nr = np.random.default_rng(rng.randint(0, 1 << 30))
    d = [p.copy() for p in d]
    hits = []
    for c in range(1 if os.environ.get("ONE_HIT") else 3):
        hh, ww = d[c].shape
        for _ in range(1 if os.environ.get("ONE_HIT") else nr.integers(1, 4)):
            y, x = int(nr.integers(0, hh)), int(nr.integers(0, ww))
            d[c][y, x] = 0 if (int(s[c][y, x]) >> (bd - 8)) > 140 else (255 << (bd - 8))
            hits.append((c, x, y, (int(s[c][y, x]) >> (bd - 8)) - (int(d[c][y, x]) >> (bd - 8))))
    try:
        o.diff_frame(s, d, xd, yd)
    except RuntimeError:
        continue
    g.diff_frame(Frame(s, xd, yd), Frame(d, xd, yd))
    try:
        g.sync()
    except Exception as e:
        print("skip", repr(e)[:80]); continue
    r = g.last_record()
    msgs = []
    flat = o.flat_mask().ravel() != 0
    nbw = o.flat_mask().shape[1]
    for c in range(3):
        S, Sb, nobs = o.ar_sums(c)
        S2, Sb2, nobs2 = r.ar_sums(c)
        if nobs != nobs2: msgs.append(f"plane {c}: nobs {nobs} vs {nobs2}")
        if not np.array_equal(S, S2): msgs.append(f"plane {c}: S differs in {int((S != S2).sum())} entries, max |diff| {int(np.abs(S - S2).max())}")
        if not np.array_equal(Sb, Sb2): msgs.append(f"plane {c}: Sb differs")
        ls, sd, sd2 = o.block_stats(c)
        ls2, sd_2, sd2_2 = r.block_stats(c)
        meas = flat & ((sd2 != 0) | (sd != 0) | ((ls != 0) if c == 0 else False))
        for nm, a, b in (("luma_sum", ls, ls2), ("sum_d", sd, sd_2), ("sum_d2", sd2, sd2_2)):
            if c and nm == "luma_sum": continue
            w_ = np.flatnonzero((a != b) & meas)
            if len(w_): msgs.append(f"plane {c}: {nm} differs at blocks {[(int(i % nbw), int(i // nbw)) for i in w_[:4]]}: {a[w_[:4]].tolist()} vs {b[w_[:4]].tolist()}")
    if msgs and os.environ.get("QUIET"):
        bad += 1
        print("F", k, end=" ")
    elif msgs:
        bad += 1
        bwc, bhc = 32 >> xd, 32 >> yd
        hb = [(c, x // (bwc if c else 32), y // (bhc if c else 32), x % (bwc if c else 32), y % (bhc if c else 32), "d=%d" % dv) for c, x, y, dv in hits]
        fl = o.flat_mask()
        S, Sb, nobs = o.ar_sums(0); S2, Sb2, _ = r.ar_sums(0)
        D = (S - S2); nz = np.argwhere(D != 0)
        # which block's contribution is the GPU off by? (luma, numpy brute force)
        D = (s[0].astype(np.int64) >> (bd - 8)) - (d[0].astype(np.int64) >> (bd - 8))
        offs = [(cy, cx) for cy in range(-lag, 1) for cx in range(-lag, lag + 1) if (cy, cx) < (0, 0)]
        nbh_, nbw_ = fl.shape
        def contrib(bx, by, dmat):
            ys = 0 if (by > 0 and fl[by - 1, bx]) else lag
            xs = 0 if (bx > 0 and fl[by, bx - 1]) else lag
            ye = min(h - by * 32, 32)
            xe = min(w - bx * 32 - lag, 32 if (bx + 1 < nbw_ and fl[by, bx + 1]) else 32 - lag)
            C = np.zeros((len(offs), len(offs)), np.int64)
            if xe <= xs or ye <= ys: return C
            Y, X = np.mgrid[by * 32 + ys: by * 32 + ye, bx * 32 + xs: bx * 32 + xe]
            V = np.stack([dmat[Y + cy, X + cx].ravel() for cy, cx in offs])
            return V @ V.T
        E = np.triu((S2 - S).astype(np.int64))
        wrap = ((D + 128) % 256) - 128
        found = False
        for by_ in range(nbh_):
            for bx_ in range(nbw_):
                if not fl[by_, bx_]: continue
                C, Cw = contrib(bx_, by_, D), contrib(bx_, by_, wrap)
                for name, M in (("+exact", C), ("+wrapped", Cw), ("wrapped - exact", Cw - C), ("-exact", -C)):
                    if M.any() and np.array_equal(np.triu(M), E): print("     gpu - oracle == block", (bx_, by_), name); found = True
        if os.environ.get("G1S_DBG_SKIP") == "generic":
            # planes mode, no generic kernel: the GPU holds the matrix-core part only = the blocks outside the 6-neighbourhood of a bad block
            badb = set()
            for c_, x_, y_, dv in hits:
                if c_ == 0 and abs(dv) > 127: badb.add((x_ // 32, y_ // 32))
            deferred = set((bx_ + dx, by_ + dy) for bx_, by_ in badb for dx in (-1, 0, 1) for dy in (0, 1))
            exp = sum((contrib(bx_, by_, D) for by_ in range(nbh_) for bx_ in range(nbw_) if fl[by_, bx_] and (bx_, by_) not in deferred), np.zeros((len(offs), len(offs)), np.int64))
            G_ = np.triu(S2.astype(np.int64)); X_ = np.triu(exp)
            print("     matrix-core part == numpy over the non-deferred blocks:", np.array_equal(G_, X_), "bad blocks", badb, "max |diff|", int(np.abs(G_ - X_).max()))
            for by_ in range(nbh_):
                for bx_ in range(nbw_):
                    if fl[by_, bx_] and np.array_equal(np.triu(contrib(bx_, by_, D)), X_ - G_): print("       numpy - gpu == block", (bx_, by_), "deferred" if (bx_, by_) in deferred else "NOT deferred")
                    if fl[by_, bx_] and np.array_equal(np.triu(contrib(bx_, by_, D)), G_ - X_): print("       gpu - numpy == block", (bx_, by_), "deferred" if (bx_, by_) in deferred else "NOT deferred")
        if not found:
            blocks = [(bx_, by_) for by_ in range(nbh_) for bx_ in range(nbw_) if fl[by_, bx_]]
            iu = np.triu_indices(len(offs))
            A = np.stack([contrib(bx_, by_, D)[iu].astype(np.float64) for bx_, by_ in blocks] + [(contrib(bx_, by_, wrap) - contrib(bx_, by_, D))[iu].astype(np.float64) for bx_, by_ in blocks], axis=1)
            sol, res, rk, _ = np.linalg.lstsq(A, E[iu].astype(np.float64), rcond=None)
            nb = len(blocks)
            print("     lstsq: exact-part coefficients", {blocks[i]: round(float(sol[i]), 3) for i in range(nb) if abs(sol[i]) > 1e-3}, " wrapped-minus-exact coefficients", {blocks[i]: round(float(sol[nb + i]), 3) for i in range(nb) if abs(sol[nb + i]) > 1e-3}, "residual", float(np.abs(A @ sol - E[iu]).max()))
        if not found: print("     gpu - oracle matches no single block; flat map:\n" + "\n".join("       " + "".join("#" if v else "." for v in row) for row in fl))
        print("     luma diff entries (i, j, oracle - gpu):", [(int(i), int(j), int(D[i, j])) for i, j in nz[:12]], "nblocks", fl.shape, "flat", int((fl != 0).sum()))
        fl = o.flat_mask()
        print(f"FAIL {w}x{h} {bd}b xd{xd} yd{yd} lag{lag}: hits (plane, bx, by, x in block, y in block) {hb}; flat of hit blocks {[int(fl[min(b[2], fl.shape[0]-1), min(b[1], fl.shape[1]-1)]) for b in hb]}")
        for m in msgs: print("    ", m)
    g.close()
print(n, "cases,", bad, "failures")
