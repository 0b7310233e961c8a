#!/usr/bin/env python3
"""tools/pin_sensitivity.py [--fuzz N] [--full] [--procs P] [--out FILE] -- how fragile is the oracle's pin?

The oracle (oracle/diff_oracle.c) restates av1-grain 0.4.2's `diff` with every f64 product and sum rounded on its own and the
frame's normal equations accumulated per sample.  Two things about the real crate cannot be checked in this image (no rustc,
no crate source: SURVEY.md 8(c)): where it writes `f64::mul_add` (a fused multiply-add rounds once where a * b + c rounds
twice), and -- on the HIP side -- what it costs that the kernels sum exact integers and the fold divides ONCE per frame where
the crate divides per sample.  This script sizes both instead of asserting "razor edge":

  * builds the oracle's variants (oracle/Makefile `variants`): fused multiply-adds at six groups of sites one group at a time
    and all together, contraction left to the compiler (-ffp-contract=fast), one division of the exact sums a frame, and that
    with every fused site as well;
  * runs every committed golden job (tests/golden/make_golden.py GOLDEN; --full: the full-size ones too) and N seeded fuzz jobs
    (random sizes, depths, subsamplings, lags, frame counts, scene cuts, flat / textured content) through the base oracle and
    through every variant;
  * counts, per variant, what moved against the base: mask bytes, f32 score bits, segment cuts, and fields of the `.tbl`.

CPU only; test infrastructure (it loads the oracle).  The summary of a run is committed as profiles/r06_pin_sensitivity.txt.
"""
from __future__ import annotations

import argparse
import os
import random
import sys
import time
from fractions import Fraction

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

VARIANTS = ["contract", "fma_linsolve", "fma_matmul", "fma_finder", "fma_strength", "fma_noise", "fma_model", "fma_all", "divide_once",
            "divide_once_fma_all"]


def fuzz_case(i: int, seed: int):
    """job i of the fuzz: (name, spec kwargs, frames, lag, chroma, cut or None)"""
    r = random.Random(seed * 1000003 + i)
    bd = r.choice([8, 8, 10, 10, 12])
    xdec, ydec = r.choice([(1, 1), (1, 1), (0, 0), (1, 0)])
    w = r.randrange(64, 417, 2 << xdec if xdec else 2)
    h = r.randrange(64, 289, 2 << ydec if ydec else 2)
    frames = r.choice([1, 2, 2, 3, 4])
    lag = r.choice([3, 3, 3, 2, 1])
    chroma = r.random() < 0.8
    textured = r.random() < 0.7
    cut = r.randrange(1, frames) if frames >= 3 and r.random() < 0.3 else None
    sd = r.randrange(1 << 30)
    return (f"fuzz{i}:{w}x{h}_{bd}b_{xdec}{ydec}_lag{lag}_{'yuv' if chroma else 'y'}_{frames}f{'_cut%d' % cut if cut else ''}{'' if textured else '_flat'}",
            dict(width=w, height=h, bit_depth=bd, xdec=xdec, ydec=ydec, textured=textured, seed=sd), frames, lag, chroma, cut, (24, 1))


def golden_cases(full: bool):
    from tests.golden import make_golden as mg

    out = []
    for name, g in list(mg.GOLDEN.items()) + (list(mg.FULL_SIZE.items()) if full else []):
        sp = g["spec"]
        fps = g.get("fps", (30000, 1001)) if "cut" in g else (24, 1)
        out.append((name, dict(width=sp.width, height=sp.height, bit_depth=sp.bit_depth, xdec=sp.xdec, ydec=sp.ydec, textured=sp.textured, seed=sp.seed),
                    g["frames"], g["lag"], g["chroma"], g.get("cut"), fps))
    return out


_LIBS = None


def _libs():
    global _LIBS
    if _LIBS is None:
        from tests import oracle_binding as ob

        _LIBS = {"base": ob.lib()}
        for v in VARIANTS:
            _LIBS[v] = ob.load_variant(v)
    return _LIBS


def run_case(case):
    """one job through the base oracle and every variant; returns (name, {variant: differences against the base})"""
    from grav1synth_amd.synth import SynthSpec
    from tests import oracle_binding as ob
    from tests.helpers import np_pair

    name, kw, frames, lag, chroma, cut, fps = case
    spec = SynthSpec(**kw)
    specb = SynthSpec(**dict(kw, gain_scale=3))
    pairs = []
    for k in range(frames):
        s, d = np_pair(specb if (cut is not None and k >= cut) else spec, k)
        pairs.append((s, d) if chroma else (s[:1], d[:1]))
    res = {}
    for vname, L in _libs().items():
        o = ob.OracleDiff(fps[0], fps[1], spec.bit_depth, spec.bit_depth, lag, chroma, library=L)
        masks, scores, err = [], [], None
        try:
            for s, d in pairs:
                o.diff_frame(s, d, spec.xdec, spec.ydec)
                masks.append(o.flat_mask().ravel().copy())
                scores.append(o.scores().ravel().view(np.uint32).copy())
            segs = o.finish()
            arr = (ob.OrcSegment * len(segs))(*segs)
            import ctypes as C

            buf = C.create_string_buffer(1 << 20)
            n = L.orc_format_tbl(arr, len(segs), buf, len(buf))
            tbl = buf.raw[:n]
        except RuntimeError as e:  # (a refused frame: the message is the result)
            err, tbl, segs = str(e), b"", []
        o.close()
        res[vname] = (masks, scores, len(segs), tbl, err)
    base = res["base"]
    out = {}
    for v in VARIANTS:
        m, sc, ns, tbl, err = res[v]
        d = dict(mask=0, score=0, cuts=int(ns != base[2]), tbl_fields=0, err=int(err != base[4]), blocks=sum(len(x) for x in base[0]))
        for a, b in zip(m, base[0]):
            d["mask"] += int((a != b).sum())
        for a, b in zip(sc, base[1]):
            d["score"] += int((a != b).sum())
        if len(m) != len(base[0]):
            d["err"] = 1
        ta, tb = tbl.split(), base[3].split()
        d["tbl_fields"] = (sum(x != y for x, y in zip(ta, tb)) + abs(len(ta) - len(tb))) if tbl != base[3] else 0
        d["tbl_total"] = len(tb)
        out[v] = d
    return name, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fuzz", type=int, default=2000)
    ap.add_argument("--seed", type=int, default=6)
    ap.add_argument("--full", action="store_true", help="the full-size goldens too (1080p x 30 frames, 4K x 8, 8K x 2: minutes a variant)")
    ap.add_argument("--procs", type=int, default=max(1, (os.cpu_count() or 2) - 2))
    ap.add_argument("--out", default=None)
    ap.add_argument("--cache", default=None, help="keep finished jobs' results here and resume from them")
    args = ap.parse_args()
    import subprocess

    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all", "variants"])
    cases = golden_cases(args.full) + [fuzz_case(i, args.seed) for i in range(args.fuzz)]
    t0 = time.time()
    # (--cache FILE: finished jobs are kept there, a chunk at a time, and an interrupted run carries on behind them)
    import pickle

    done = {}
    if args.cache and os.path.exists(args.cache):
        with open(args.cache, "rb") as f:
            done = pickle.load(f)
    todo = [c for c in cases if c[0] not in done]
    pool = None
    if args.procs > 1:
        import multiprocessing as mp

        pool = mp.get_context("fork").Pool(args.procs)
    for i in range(0, len(todo), 40):
        chunk = todo[i:i + 40]
        for nm, out in (pool.map(run_case, chunk, chunksize=2) if pool else map(run_case, chunk)):
            done[nm] = out
        if args.cache:
            with open(args.cache + ".tmp", "wb") as f:
                pickle.dump(done, f)
            os.replace(args.cache + ".tmp", args.cache)
        print(f"# {len(done)} of {len(cases)} jobs done, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
    if pool:
        pool.close()
    results = [(c[0], done[c[0]]) for c in cases]
    lines = []
    P = lines.append
    ngold = len(cases) - args.fuzz
    P(f"# tools/pin_sensitivity.py --fuzz {args.fuzz} --seed {args.seed}{' --full' if args.full else ''}: {ngold} golden jobs + {args.fuzz} fuzz jobs, "
      f"{sum(r[1][VARIANTS[0]]['blocks'] for r in results)} blocks (frame x block), {time.time() - t0:.0f} s on {args.procs} processes")
    P("# against the base oracle (every product and sum its own rounding, normal equations accumulated per sample): what each variant moves")
    P(f"# {'variant':24s} {'jobs: mask':>10s} {'mask bytes':>11s} {'jobs: score':>11s} {'score bits':>11s} {'jobs: cuts':>10s} {'jobs: .tbl':>10s} {'.tbl fields':>11s} {'jobs: error':>11s}")
    for v in VARIANTS:
        ds = [r[1][v] for r in results]
        P(f"  {v:24s} {sum(d['mask'] > 0 for d in ds):10d} {sum(d['mask'] for d in ds):11d} {sum(d['score'] > 0 for d in ds):11d} {sum(d['score'] for d in ds):11d} "
          f"{sum(d['cuts'] for d in ds):10d} {sum(d['tbl_fields'] > 0 for d in ds):10d} {sum(d['tbl_fields'] for d in ds):11d} {sum(d['err'] for d in ds):11d}")
    P(f"# (.tbl fields of all jobs together: {sum(r[1][VARIANTS[0]]['tbl_total'] for r in results)})")
    P("# jobs whose table moved, per variant (first 12):")
    for v in VARIANTS:
        moved = [(r[0], r[1][v]) for r in results if r[1][v]["tbl_fields"] or r[1][v]["cuts"] or r[1][v]["mask"] or r[1][v]["err"]]
        for nm, d in moved[:12]:
            P(f"#   {v:22s} {nm}: mask bytes {d['mask']}, score bits {d['score']}, cuts {'moved' if d['cuts'] else 'same'}, .tbl fields {d['tbl_fields']} of {d['tbl_total']}")
    text = "\n".join(lines) + "\n"
    sys.stdout.write(text)
    if args.out:
        with open(args.out, "w") as f:
            f.write(text)


if __name__ == "__main__":
    main()
