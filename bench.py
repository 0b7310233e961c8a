#!/usr/bin/env python3
"""bench.py -- `diff` throughput on MI355X (BASELINE.json metric).

A "step" is one pass of the whole hot path (flat-block finder, AR accumulation,
block statistics, ordered fold) over one batch of synthetic frame pairs that
are already resident in HBM.  Workload at any N: 3840x2160 10-bit 4:2:0,
ar_coeff_lag 3, chroma (BASELINE.json configs[2], the one the metric is quoted
on); each rank owns its own `--frames` frame pairs per step (weak scaling,
frame sharding), exchanges the per-frame integer records with ONE RCCL
all-gather per step, and rank 0 runs the ordered fold over all N*frames records.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from fractions import Fraction

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)

WORKLOADS = {
    # name: (W, H, bit_depth, xdec, ydec, lag, chroma, bytes per luma pixel)
    "4k10": (3840, 2160, 10, 1, 1, 3, True, 6),
    "1080p8_lag2_luma": (1920, 1080, 8, 1, 1, 2, False, 2),
    "1080p8": (1920, 1080, 8, 1, 1, 3, True, 3),
    "8k10_444": (7680, 4320, 10, 0, 0, 3, True, 12),
}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=256, help="frame pairs per rank per step (one video chunk per step)")
    ap.add_argument("--cycles", type=int, default=8,
                    help="a step feeds the resident frames this many times over: one job of frames x cycles frame pairs")
    ap.add_argument("--batch", type=int, default=32, help="frames per kernel launch group (<= 256)")
    ap.add_argument("--workload", default="4k10", choices=sorted(WORKLOADS))
    ap.add_argument("--flat", action="store_true", help="all-flat stress variant (no textured region)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=4)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the diff path has no CPU fallback)")
    # one rank per GPU; G1S_BENCH_SHARE_GPU=1 lets several ranks share a device (single-GPU smoke test of
    # the N > 1 code path, with the gloo backend: RCCL refuses two ranks on one device)
    share = os.environ.get("G1S_BENCH_SHARE_GPU") == "1"
    dev_index = local_rank % torch.cuda.device_count() if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist  # noqa: F811

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if world > 1:  # the host fold pools of the ranks share the node's cores
        lws = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        os.environ.setdefault("G1S_FOLD_THREADS", str(max(2, min(32, (os.cpu_count() or 8) // max(lws, 1)))))
    from grav1synth_amd.diff import DiffGenerator, format_tbl
    from grav1synth_amd.dist import ShardedDiff, StreamingShardedDiff
    from grav1synth_amd.synth import SynthSpec, make_pair

    W, H, bd, xdec, ydec, lag, chroma, bpp = WORKLOADS[args.workload]
    spec = SynthSpec(W, H, bd, xdec, ydec, textured=not args.flat)
    F = args.frames
    FJ = F * args.cycles  # frame pairs of one step (job)
    fps = Fraction(24, 1)

    # ---- synthetic frame pairs, resident in HBM before any timed region ----
    # N > 1: the video is dealt to the ranks batch by batch (global batch j -> rank j % N)
    B = max(1, min(args.batch, 32))
    frames = []
    for k in range(F):
        gid = ((k // B) * world + rank) * B + (k % B) if world > 1 else k
        s, d = make_pair(spec, gid, device=dev)
        if not chroma:
            s, d = s[:1], d[:1]
        frames.append((s, d))
    torch.cuda.synchronize()
    # FFI frame descriptors (pointers, strides) are built once, outside the timed region:
    # a native caller hands over an array of frame structs just like this
    prepared = DiffGenerator.prepare_frames(frames, xdec, ydec)
    prepared_batches = [DiffGenerator.prepare_frames(frames[i:i + B], xdec, ydec) for i in range(0, F, B)] if world > 1 else []
    nplanes = 3 if chroma else 1

    stats_total = None
    last_tbl = None
    window_samples = None  # per plane, of the last frame of the timed-kernels step

    def one_step(timing: bool):
        nonlocal stats_total, last_tbl, window_samples
        if world > 1:
            # streaming frame shards: per batch one small all-gather of latest states, rank 0 merges in order
            sd = StreamingShardedDiff(fps, bd, bd, ar_coeff_lag=lag, luma_only=not chroma, device=dev_index,
                                      batch_frames=B, group=dist)
            sd.generator.set_timing(timing)
            for _ in range(args.cycles):
                for pb in prepared_batches:
                    sd.diff_prepared(pb, sync_torch=False)
        else:
            sd = ShardedDiff(fps, bd, bd, ar_coeff_lag=lag, luma_only=not chroma, device=dev_index,
                             batch_frames=args.batch, group=None)
            sd.generator.set_timing(timing)
            for _ in range(args.cycles):
                sd.diff_prepared(prepared, W, H, nplanes, sync_torch=False)
        segs = sd.finish()  # (exchange +) ordered fold; rank 0 holds the table
        st = sd.generator.stats()
        if segs is not None:
            last_tbl = format_tbl(segs)
        if timing:
            try:
                r = sd.generator.last_record()
                window_samples = [int(r.ar_sums(c)[2]) for c in range(nplanes)]
            except Exception:
                window_samples = None
        sd.close()
        return st

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step(False)
    barrier()
    t0 = time.perf_counter()
    step_ms = []
    for _ in range(args.steps):
        ts = time.perf_counter()
        one_step(False)
        step_ms.append((time.perf_counter() - ts) * 1e3)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=torch.device("cpu") if share else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- per-kernel HIP-event timing, in separate (untimed) steps: events between
    # kernels serialise nothing here but we keep them out of the headline number ----
    st = one_step(True)
    kernels = {
        "k1_flat_features": (st.ms_flat_features, st.launches_flat_features),
        "k2_flat_select": (st.ms_flat_select, st.launches_flat_select),
        "k3_ar_accumulate": (st.ms_ar_accumulate, st.launches_ar_accumulate),
    }
    dom = max(kernels, key=lambda k: kernels[k][0])
    dom_ms, dom_launches = kernels[dom]
    frames_per_launch = FJ / max(dom_launches, 1)
    alg_bytes_per_launch = bpp * W * H * frames_per_launch
    avg_launch_ms = dom_ms / max(dom_launches, 1)
    achieved = alg_bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9 if avg_launch_ms > 0 else 0.0
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_path):
        try:
            with open(pmc_path) as f:
                traffic = json.load(f).get(args.workload, {}).get(dom)
        except Exception:
            traffic = None

    total_px = float(W) * H * F * args.cycles * args.steps * world
    value = total_px / elapsed / 1e6
    valu = None
    if window_samples and lag == 3:
        macs = window_samples[0] * 324 + sum(window_samples[1:]) * 350
        fps_all = value * 1e6 / (W * H)
        valu = {"window_samples_per_frame": window_samples, "gmac_per_frame": macs / 1e9,
                "achieved_tmac_s": macs * fps_all / world / 1e12, "dot4_peak_tmac_s": 133.0,
                "frac": macs * fps_all / world / 1e12 / 133.0}
    out = {
        "metric": "diff Mpixels/s (luma pixels of frame pairs fully processed: flat-block finder + AR accumulation + block stats + ordered fold)",
        "value": value,
        "unit": "Mpixels/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "step_ms": [round(x, 3) for x in step_ms],
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8/i32/i64 exact-integer accumulation + f64 flat-block features",
        "data": "synthetic (deterministic integer generator, grav1synth_amd/synth.py), device-resident",
        "config": {
            "workload": f"diff {W}x{H} {bd}-bit {'4:2:0' if (xdec, ydec) == (1, 1) else '4:4:4' if (xdec, ydec) == (0, 0) else '4:2:2'}, ar_coeff_lag={lag}, {'chroma' if chroma else 'luma-only'} ({args.workload}{', all-flat' if args.flat else ''})",
            "frames_per_rank_per_step": FJ,
            "resident_frames_per_rank": F,
            "batch_frames": args.batch,
            "flat_fraction": (st.flat_blocks / st.blocks) if st.blocks else None,
            "flat_finder_literal_fraction": (st.literal_blocks / st.blocks) if st.blocks else None,
            "parallelism": f"frame-shard x{world} (batches dealt round-robin), one small RCCL all-gather of per-frame latest states per batch, ordered merge on rank 0" if world > 1 else "single GPU",
        },
        "hbm_roofline_frac_whole_job": (value * bpp * 1e6 / 1e9) / (HBM_PEAK_GBS * world),
        "roofline": {
            "bound": "hbm",
            "kernel": dom,
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "avg_launch_ms": avg_launch_ms,
            "alg_bytes_per_launch": alg_bytes_per_launch,
            "all_kernels_ms_per_frame": {k: v[0] / FJ for k, v in kernels.items()},
            "host_fold_ms_per_frame": st.ms_host_fold / FJ,
            # SURVEY 8(d), caveat H1: the accumulation is VALU work.  Algorithmic MACs of a frame = window samples x
            # (324 luma / 350 chroma: unique products + right-hand sides of add_block_observations); the kernels execute
            # about a sixth of them (46 lag sums instead of 324 products on full groups).  Peak = measured v_dot4 rate.
            "valu_algorithmic": valu,
            # inside k3_ar_accumulate: K0, the one pass over the source / denoised planes (the HBM-streaming kernel)
            "k0_residual": {
                "ms_per_frame": st.ms_residual / FJ,
                "achieved": (bpp * W * H * FJ / (st.ms_residual * 1e-3) / 1e9) if st.ms_residual > 0 else None,
                "unit": "GB/s",
                "frac": (bpp * W * H * FJ / (st.ms_residual * 1e-3) / 1e9 / HBM_PEAK_GBS) if st.ms_residual > 0 else None,
            },
        },
    }

    # ---- CPU baseline: the oracle (a port, scalar f64, 1 thread) on a bounded sample ----
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from tests.oracle_binding import OracleDiff

        n_cpu = max(1, args.cpu_frames)
        o = OracleDiff(fps.numerator, fps.denominator, bd, bd, lag, chroma)
        host = []
        for k in range(n_cpu):
            s, d = frames[k]
            host.append(([p.cpu().numpy() for p in s], [p.cpu().numpy() for p in d]))
        t0 = time.perf_counter()
        for s, d in host:
            o.diff_frame(s, d, xdec, ydec)
        o.finish()
        cpu_s = time.perf_counter() - t0
        out["cpu_baseline"] = {
            "value": W * H * n_cpu / cpu_s / 1e6,
            "unit": "Mpixels/s",
            "cores": 1,
            "kind": "port",
            "sample": f"{n_cpu} frame pair(s) of the same workload through oracle/liborc_diff.so (scalar f64, reference operation order), {cpu_s:.1f} s",
        }
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
