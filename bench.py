#!/usr/bin/env python3
"""bench.py -- `diff` throughput on MI355X (BASELINE.json metric).

A "step" is one pass of the whole hot path (flat-block finder, AR accumulation,
block statistics, ordered fold) over one job of synthetic frame pairs that are
already resident in HBM.  Workload at any N: 3840x2160 10-bit 4:2:0,
ar_coeff_lag 3, chroma (BASELINE.json configs[2], the one the metric is quoted
on).  N > 1: the video is dealt to the ranks batch by batch (frame sharding,
weak scaling: every rank owns `--frames` x `--cycles` frame pairs per step);
per batch round the ranks all-gather their frames' small latest-state blobs
(RCCL) and rank 0 merges them in order -- no collective on the pixel path.

The timed region is `--steps` such jobs between barrier + synchronize pairs.
After it, one more (untimed) job runs with HIP events around every kernel
launch: the `roofline` object comes from those, the `cpu_baseline` object from
the oracle (tests' checker) timed on the host cores over a bounded sample.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time
from fractions import Fraction

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# The HIP runtime gives a process 4 hardware queues by default; the generator's four streams and torch's own then share them
# and kernels of different streams wait for each other (the same build: 544 - 593 k Mpx/s from run to run; with 8 queues
# 590 - 595 k: profiles/r04_streams.txt).  Read by the runtime when it starts: set before anything touches the GPU.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

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


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` with no launcher around it: re-run this command line as N ranks under torch.distributed.run
    (what the driver's own N > 1 command is), one rank per GPU, and pass the children's output through."""
    import socket
    import subprocess

    with socket.socket() as s:  # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("GPU_MAX_HW_QUEUES", "8")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # (dmabuf IPC: what RCCL needs on this host driver)
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def bind_to_gpu_numa_node(dev_index: int):
    """Pin this rank (its launch threads, the fold pools made later, its pinned rings' first touch) to the CPUs of the NUMA
    node its GPU hangs off: torch.distributed.run binds nothing.  Returns the node, or None when the host does not say."""
    try:
        pr = torch.cuda.get_device_properties(dev_index)
        bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        base = "/sys/bus/pci/devices/" + bdf
        node = int(open(base + "/numa_node").read())
        if node < 0:
            return None
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= set(os.sched_getaffinity(0))
        if not cpus:
            return None
        os.sched_setaffinity(0, cpus)
        return node
    except Exception:
        return None


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=256, help="frame pairs per rank per step (one video chunk per step)")
    ap.add_argument("--cycles", type=int, default=120,
                    help="a step feeds the resident frames this many times over: one job of frames x cycles frame pairs "
                         "(default: a step of about half a second, so that the timed region of --steps 20 is 10 s)")
    ap.add_argument("--batch", type=int, default=0, help="frames per kernel launch group (<= 256; default: the engine's own choice, "
                    "about 530 Mpixels a group: 64 at 4K, 128 at 1080p, 32 at 8K)")
    ap.add_argument("--workload", default="4k10", choices=sorted(WORKLOADS))
    ap.add_argument("--flat", action="store_true", help="all-flat stress variant (no textured region)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-all-flat", action="store_true", help="skip the all-flat variant's timed-kernels job (roofline.frac_all_flat)")
    ap.add_argument("--cpu-frames", type=int, default=2)
    ap.add_argument("--no-table-check", action="store_true", help="skip the untimed half-batch job the timed job's table is checked against "
                    "(profiling runs: its launches are half the size and would halve the profiler's per-kernel averages)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` by itself: start the N ranks the way the driver would (one process per GPU under
        # torch.distributed.run, rendezvous on 127.0.0.1), hand the one JSON line of rank 0 through, leave with its exit code
        raise SystemExit(self_launch(args.gpus))
    # (G1S_BENCH_FORCE_DIST=1, a test aid: ONE rank through the N > 1 code path -- process group over RCCL, the streaming frame
    #  shards, the rounds' transport, the collectives of the timed region -- which a single-GPU box can otherwise not run)
    multi = world > 1 or os.environ.get("G1S_BENCH_FORCE_DIST") == "1"
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with `python bench.py --gpus N` or "
                         "`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the diff path has no CPU fallback)")
    # The line this command prints is the ONLY thing on its standard output: libraries write there too (RCCL prints a five-line
    # version banner from its first communicator), so from here on file descriptor 1 is the error stream and the line goes to
    # the saved descriptor at the end.
    sys.stdout.flush()
    line_fd = os.dup(1)
    os.dup2(2, 1)
    # one rank per GPU; G1S_BENCH_SHARE_GPU=1 lets several ranks share a device (single-GPU smoke test of
    # the N > 1 code path, with the gloo backend: RCCL refuses two ranks on one device)
    share = os.environ.get("G1S_BENCH_SHARE_GPU") == "1"
    dev_index = local_rank % torch.cuda.device_count() if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    numa = bind_to_gpu_numa_node(dev_index) if multi and not share else None
    dist = None
    if multi:
        import torch.distributed as dist  # noqa: F811

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:  # (no launcher around a forced one-rank job)
            import socket

            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    if multi:  # the host fold pools of the ranks share the node's cores
        lws = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        from grav1synth_amd import _lib

        # (usable = hardware threads cut to the cgroup CPU quota: the ranks of a node share it)
        # (a name of its own: `share` above says whether the ranks share ONE device -- it picks gloo / CPU tensors further down)
        cores_per_rank = int(_lib.lib().g1s_usable_cpus()) // max(lws, 1)
        os.environ.setdefault("G1S_FOLD_THREADS", str(max(2, min(32, cores_per_rank))))
        # A rank's per-frame half keeps ~9 cores busy at 4K next to the launches, the copies and the merge (profiles/
        # r05_host_budget_8ranks.txt); a job whose CPU quota gives a rank fewer is bound by the host.  The half on the device
        # (k4_latest, rebuilt in round 5: profiles/r05_device_latest.txt) runs a GPU at 0.90 - 0.95 x of its unconstrained
        # host-half rate and leaves the host the launches, 27 KB a frame and the merge.  Measured with one rank held to n CPUs
        # (profiles/r05z_half_by_cores.txt): 8 CPUs 453 k (host) / 549 k (device) Mpx/s, 10 CPUs 542 / 548, 12 CPUs 553 / 485 - 549:
        # the device half below 10 cores a rank (two ranks on a 16-core quota have 8).
        # (not when the ranks SHARE one device -- the single-GPU rehearsal of this code path: two processes' low-priority k4_latest
        #  streams on one GPU starve each other, 109 k Mpx/s where the host half reads 606 k, profiles/r06_other_workloads.txt; a
        #  rehearsal that wants the device half says G1S_LATEST=device)
        if cores_per_rank < 10 and not share:
            os.environ.setdefault("G1S_LATEST", "device")
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
    if args.batch <= 0:  # the engine's own choice (engine.hip: set_geometry_alloc), said out loud so that the line can report it
        args.batch = int(min(128, max(32, (530000000 + W * H // 2) // (W * H))))
    B = max(1, args.batch)  # (frames per launch, the same at any N; what the line reports as config.batch_frames)
    frames = []
    for k in range(F):
        gid = ((k // B) * world + rank) * B + (k % B) if multi else k
        s, d = make_pair(spec, gid, device=dev)
        if not chroma:
            s, d = s[:1], d[:1]
        frames.append((s, d))
    torch.cuda.synchronize()
    # FFI frame descriptors (pointers, strides) are built once, outside the timed region:
    # a native caller hands over an array of frame structs just like this
    prepared = DiffGenerator.prepare_frames(frames, xdec, ydec)
    prepared_batches = [DiffGenerator.prepare_frames(frames[i:i + B], xdec, ydec) for i in range(0, F, B)] if multi else []
    nplanes = 3 if chroma else 1

    stats_total = None
    kernel_times = {}
    last_tbl = None
    tbl_digests = []  # SHA-256 of the table of every job that ended in one (rank 0 at N > 1), in order
    window_samples = None  # per plane, of the last frame of the timed-kernels step
    exchange_s, exchange_rounds = 0.0, 0  # N > 1: the feeding thread's time in the rounds' exchange

    def one_step(timing, cycles: int = 0, prep=None, batch: int = 0):  # timing: False | True (an event a kernel) | 2 (one pair a batch)
        """one job: the resident frames `cycles` times over through a fresh generator, up to the finished table"""
        nonlocal stats_total, last_tbl, window_samples, kernel_times, exchange_s, exchange_rounds
        cycles = cycles or args.cycles
        prep = prep if prep is not None else prepared
        if multi:
            # streaming frame shards: per batch one small all-gather of latest states, rank 0 merges in order
            sd = StreamingShardedDiff(fps, bd, bd, ar_coeff_lag=lag, luma_only=not chroma, device=dev_index,
                                      batch_frames=B, group=dist)
            sd.generator.set_timing(timing)
            for _ in range(cycles):
                for pb in prepared_batches:
                    sd.diff_prepared(pb, sync_torch=False)
        else:
            sd = ShardedDiff(fps, bd, bd, ar_coeff_lag=lag, luma_only=not chroma, device=dev_index,
                             batch_frames=batch or args.batch, group=None)
            if timing:
                # (one untimed pass first: the first batch of a generator carries one-off costs -- 1.3 ms in the last kernel
                #  of its chain -- that are not the kernels')
                sd.generator.set_timing(False)
                sd.diff_prepared(prep, W, H, nplanes, sync_torch=False)
            sd.generator.set_timing(timing)
            for _ in range(cycles):
                sd.diff_prepared(prep, W, H, nplanes, sync_torch=False)
        segs = sd.finish()  # (exchange +) ordered fold; rank 0 holds the table
        if multi and not timing:
            exchange_s += sd.exchange_s
            exchange_rounds += sd.exchange_rounds  # (feeds + flush rounds + the last group's padding: counted where they happen)
        st = sd.generator.stats()
        if timing is True:
            kernel_times = sd.generator.kernel_times()
        if segs is not None:
            last_tbl = format_tbl(segs)
            tbl_digests.append(hashlib.sha256(last_tbl).hexdigest())
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
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    # The table of the timed job is checked, not dropped: behind the timed steps the same 30 720 frames go once more through a generator
    # with HALF the frames a launch (another partition of the video into launches, other slices a workgroup: the integer records do not
    # depend on it), untimed; every timed step must have ended in those bytes.
    check_tbl = None
    for _ in range(args.warmup):
        one_step(False)
    barrier()
    n_untimed = len(tbl_digests)
    t0 = time.perf_counter()
    cpu0 = sum(os.times()[:2])  # (this process' user + system seconds, every thread: the host cores a rank keeps busy)
    step_ms, step_fold_ms = [], []
    for _ in range(args.steps):
        ts = time.perf_counter()
        st_ = one_step(False)
        step_ms.append((time.perf_counter() - ts) * 1e3)
        step_fold_ms.append(st_.ms_host_fold)
    barrier()
    elapsed = time.perf_counter() - t0
    host_cores_busy = (sum(os.times()[:2]) - cpu0) / max(elapsed, 1e-9)
    timed_digests = tbl_digests[n_untimed:]
    if not multi and not args.no_table_check:  # (behind the timed region: nothing but the W warm-up steps runs in front of it)
        one_step(False, batch=max(1, B // 2))
        check_tbl = tbl_digests.pop()
    if rank == 0:
        assert len(timed_digests) == args.steps and len(set(timed_digests)) == 1, "the timed steps did not all end in one table"
        assert check_tbl is None or timed_digests[0] == check_tbl, "the timed job's table differs from the untimed half-batch job's"
        assert last_tbl.startswith(b"filmgrn1\n") and last_tbl.count(b"\nE ") + last_tbl.startswith(b"E ") >= 1
    if multi:
        t = torch.tensor([elapsed], device=torch.device("cpu") if share else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- per-kernel HIP-event timing, in separate (untimed) steps: events between
    # kernels serialise nothing here but we keep them out of the headline number ----
    TC = min(args.cycles, 8)  # (the timed-kernels job: 8 passes over the resident frames are plenty)
    st = one_step(True, TC)
    FT = F * TC
    families = {
        "k1_flat_features": (st.ms_flat_features, st.launches_flat_features),
        "k2_flat_select": (st.ms_flat_select, st.launches_flat_select),
        "k3_ar_accumulate": (st.ms_ar_accumulate, st.launches_ar_accumulate),
    }
    # HIP events around every launch of the timed job (one stream): kernel name -> (ms, launches)
    kt = {k: v for k, v in kernel_times.items() if v[1] > 0}
    dom = max(kt, key=lambda k: kt[k][0]) if kt else max(families, key=lambda k: families[k][0])
    dom_ms, dom_launches = kt[dom] if kt else families[dom]
    n_batches = max(dom_launches if kt else st.launches_ar_accumulate, 1)  # (the timed batches: the job's first pass is not)
    frames_per_launch = FT / n_batches
    # bpp bytes per luma pixel of a frame pair = every source and denoised sample once
    alg_bytes_per_launch = bpp * W * H * frames_per_launch
    # the whole pass over one batch: every kernel from the finder's first to the accumulation's last, alone on the chip
    batch_ms_events = (sum(v[0] for v in kt.values()) if kt else st.ms_total_gpu) / n_batches
    # ... and the same batches with ONE pair of events, around the chain (first kernel's start -> last kernel's end; nothing between
    # two launches but their own dependency): what a batch takes alone on the chip.  The per-kernel events above each put a
    # barrier packet and a signal between two launches -- 5 - 7 us a kernel that rocprofv3's kernel trace does not see
    # (profiles/r05_kernel_stats_one_stream.txt); this figure is the one that agrees with the profiler's sum.
    stc = one_step(2, TC)
    batch_ms = (stc.ms_chain / stc.chain_batches) if stc.chain_batches else batch_ms_events
    achieved = alg_bytes_per_launch / (batch_ms * 1e-3) / 1e9 if batch_ms > 0 else 0.0
    bps = 1 if bd == 8 else 2
    cpx = (W >> xdec) * (H >> ydec) if chroma else 0

    def own_bytes(name):
        """algorithmic bytes one launch of this kernel exists to read, per frame pair (None: not a pixel kernel)"""
        if name.startswith("k3w_pass"):
            t = [x.strip(" >") for x in name.split("<")[1].split(",")]
            return 2 * bps * W * H if t[0] == "0" else 2 * bps * 2 * cpx
        if name.startswith("k3s_fused"):
            t = [x.strip(" >") for x in name.split("<")[1].split(",")]
            return 2 * bps * W * H if t[3] == "0" else 2 * bps * (2 if t[3] == "1" else 1) * cpx  # (launch 2 / 3: one chroma plane)
        if name.startswith("k3f_fused"):
            t = [x.strip(" >") for x in name.split("<")[1].split(",")]
            if t[4] == "1":  # staging the int8 planes of the pixel pass
                return (W * H if t[3] == "0" else 2 * cpx + cpx)
            return 2 * bps * W * H if t[3] == "0" else 2 * bps * 2 * cpx
        if name.startswith("k1_moments"):
            return bps * W * H
        if name.startswith("k0_residual"):
            return bpp * W * H
        return None

    ob = own_bytes(dom)
    dom_flat = (st.flat_blocks / st.blocks) if st.blocks else 1.0
    dom_avg_ms = dom_ms / max(dom_launches, 1)
    traffic = None
    traffic_source = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_path):
        try:
            with open(pmc_path) as f:
                tj = json.load(f).get(args.workload, {})
                mode = os.environ.get("G1S_K3", "wide")
                # measured HBM bytes per frame pair (PMC passes, tools/profile_round.sh) x the frames of a launch
                if mode + "_per_frame" in tj:
                    traffic = tj[mode + "_per_frame"] * frames_per_launch
                else:
                    traffic = tj.get(mode)
                if traffic is not None:
                    traffic_source = ("static: profiles/pmc_traffic.json (rocprofv3 --pmc passes of tools/diff_pmc.py, FETCH_SIZE x 2 + "
                                      "WRITE_SIZE per the guide's gfx950 correction), " + str(tj.get("measured", "commit not recorded")) +
                                      "; not collected in this run")
        except Exception:
            traffic = None

    # ---- SURVEY 8(d) asks for both content variants: the timed-kernels job once more on the all-flat stress variant (no
    #      textured region: every unit of every frame is staged and multiplied), 64 resident frame pairs, rank 0 of N = 1 ----
    frac_all_flat = flat_fraction_all_flat = batch_ms_all_flat = None
    if not multi and not args.flat and not args.no_all_flat:
        fl_spec = SynthSpec(W, H, bd, xdec, ydec, textured=False)
        nfl = min(F, max(args.batch, 64))
        fl_frames = []
        for k in range(nfl):
            s_, d_ = make_pair(fl_spec, k, device=dev)
            if not chroma:
                s_, d_ = s_[:1], d_[:1]
            fl_frames.append((s_, d_))
        torch.cuda.synchronize()
        fl_prep = DiffGenerator.prepare_frames(fl_frames, xdec, ydec)
        keep_kt = dict(kernel_times)
        one_step(False, 2, fl_prep)  # (warm)
        fst = one_step(True, 8, fl_prep)
        fkt = {k: v for k, v in kernel_times.items() if v[1] > 0}
        kernel_times = keep_kt
        fb = max(fst.launches_ar_accumulate, 1)
        fstc = one_step(2, 8, fl_prep)  # (one pair of events a batch, as above)
        if fstc.chain_batches:  # (8 passes over nfl frames in chain_batches timed batches: the job's untimed first pass is not among them)
            batch_ms_all_flat = fstc.ms_chain / fstc.chain_batches
            frac_all_flat = (bpp * W * H * (nfl * 8 / fstc.chain_batches)) / (batch_ms_all_flat * 1e-3) / 1e9 / HBM_PEAK_GBS
        else:  # (fb counts the untimed first pass' batches too: time and frames are both per fb)
            batch_ms_all_flat = sum(v[0] for v in fkt.values()) / fb
            frac_all_flat = (bpp * W * H * (nfl * 8 / fb)) / (batch_ms_all_flat * 1e-3) / 1e9 / HBM_PEAK_GBS
        flat_fraction_all_flat = (fst.flat_blocks / fst.blocks) if fst.blocks else None
        del fl_frames, fl_prep
        torch.cuda.empty_cache()

    total_px = float(W) * H * F * args.cycles * args.steps * world
    value = total_px / elapsed / 1e6
    out = {
        "metric": "diff Mpixels/s (luma pixels of frame pairs fully processed: flat-block finder + AR accumulation + block stats + ordered fold)",
        "value": value,
        "unit": "Mpixels/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "step_ms": [round(x, 3) for x in step_ms],
        "step_host_fold_ms": [round(x, 3) for x in step_fold_ms],  # (the ordered merge's wall time inside each step: a thread of its own)
        "host_cores_busy": round(host_cores_busy, 2),  # (rank 0's process over the timed steps: CPU seconds / wall seconds)
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        # the timed job really ends in a grain table: SHA-256 of its .tbl bytes (every timed step: one value, asserted), and of the
        # same frames' table through an untimed job with half the frames a launch (N = 1; asserted equal)
        "tbl_sha256": timed_digests[0] if timed_digests else None,
        "tbl_sha256_untimed_half_batch_job": check_tbl,
        "tbl_bytes": len(last_tbl) if last_tbl is not None else None,
        "dtype": "i8 x i8 -> i32 matrix-core products, i64 sums (exact integers) + f64 flat-block features",
        "data": "synthetic (deterministic integer generator, grav1synth_amd/synth.py), device-resident",
        "config": {
            "workload": f"diff {W}x{H} {bd}-bit {'4:2:0' if (xdec, ydec) == (1, 1) else '4:4:4' if (xdec, ydec) == (0, 0) else '4:2:2'}, ar_coeff_lag={lag}, {'chroma' if chroma else 'luma-only'} ({args.workload}{', all-flat' if args.flat else ''})",
            "frames_per_rank_per_step": FJ,
            "resident_frames_per_rank": F,
            "batch_frames": args.batch,
            "accumulation": {"wide": "exact int8 SYRK on the matrix cores (v_mfma_i32_16x16x64_i8 on operand pairs), 128-sample units, residuals in 32-bit SWAR fused into the consumer, windows as masks on the A operand, one tile buffer -- a deviation from north_star's 'no MFMA', signed off in VERDICT r01",
                             "stream": "exact int8 SYRK on the matrix cores (v_mfma_i32_16x16x64_i8 on operand pairs, one LDS operand read per 64 samples), residual fused into the consumer, two tile buffers -- a deviation from north_star's 'no MFMA', signed off in VERDICT r01"}[os.environ.get("G1S_K3", "wide")],
            "flat_fraction": (st.flat_blocks / st.blocks) if st.blocks else None,
            "flat_finder_literal_fraction": (st.literal_blocks / st.blocks) if st.blocks else None,
            "rccl_ranks": (dist.get_world_size() if (multi and not share) else (1 if not multi else 0)),
            "backend": (dist.get_backend() if multi else "none (one process)"),
            "numa_node_rank0": numa,  # (N > 1: the NUMA node rank 0 bound itself to -- its GPU's; None: the host does not say)
            "exchange_ms_per_round": (exchange_s * 1e3 / exchange_rounds) if exchange_rounds else None,  # (N > 1: pack + gather + hand-over, on the feeding thread of this rank)
            "per_frame_fold_half": ("device (k4_latest)" if os.environ.get("G1S_LATEST") == "device" else "host pool"),
            "parallelism": f"frame-shard x{world} (batches dealt round-robin), one small RCCL all-gather of per-frame latest states per batch, ordered merge on rank 0" if multi else "single GPU",
        },
        "hbm_roofline_frac_whole_job": (value * bpp * 1e6 / 1e9) / (HBM_PEAK_GBS * world),
        "roofline": {
            "bound": "hbm",
            # the kernel with the most time in a batch, by the name rocprofv3 prints; its family; and what `achieved` covers
            "kernel": dom,
            "family": "k3_ar_accumulate" if dom.startswith(("k3", "k0")) else "k1_flat_features" if dom.startswith("k1") else "k2_flat_select",
            "scope": "all kernels of one batch (the pass needs every one of them to have read a frame pair once): algorithmic bytes "
                     "of the batch / HIP-event time from its first kernel's start to its last kernel's end, the batch alone on the "
                     "chip, one stream (frac_kernel_events: / the sum of per-kernel event pairs, each of which adds a "
                     "barrier packet between two launches)",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_source": traffic_source,
            "timing_method": "one HIP-event pair around a batch's chain of kernels (g1s_diff_set_timing 2; since the end of round 5); "
                             "rounds 1-5 summed per-kernel event pairs: compare rounds on frac_kernel_events",
            # the same fraction on the all-flat stress variant (every unit staged and multiplied), its flat fraction, its batch
            "frac_all_flat": frac_all_flat,
            "flat_fraction_all_flat": flat_fraction_all_flat,
            "avg_launch_ms_all_flat": batch_ms_all_flat,
            "flat_fraction": (st.flat_blocks / st.blocks) if st.blocks else None,
            "avg_launch_ms": batch_ms,
            "avg_launch_ms_kernel_events": batch_ms_events,
            "frac_kernel_events": (alg_bytes_per_launch / (batch_ms_events * 1e-3) / 1e9 / HBM_PEAK_GBS) if batch_ms_events > 0 else None,
            "alg_bytes_per_launch": alg_bytes_per_launch,
            "frames_per_launch": frames_per_launch,
            "dominant_kernel": {
                "name": dom,
                "avg_launch_ms": dom_avg_ms,
                "share_of_batch": dom_avg_ms / batch_ms_events if batch_ms_events > 0 else None,
                # the planes this one kernel is launched over, and the share of them it loads: an accumulation launch only reads the
                # units that hold a flat block, so its bandwidth is planes x flat fraction / duration (a lower bound: a unit of four
                # blocks is loaded when ONE of them is flat; the PMC figure per kernel is in profiles/r06_pmc_traffic.txt)
                "plane_bytes_per_launch": ob * frames_per_launch if ob else None,
                "achieved_flat_weighted": (ob * frames_per_launch * (dom_flat if dom.startswith("k3") else 1.0) / (dom_avg_ms * 1e-3) / 1e9)
                if ob and dom_avg_ms > 0 else None,
            },
            "kernels_us_per_launch": {k: round(v[0] / v[1] * 1e3, 2) for k, v in sorted(kt.items(), key=lambda kv: -kv[1][0])},
            "families_ms_per_frame": {k: v[0] / FT for k, v in families.items()},
            "host_fold_ms_per_frame": st.ms_host_fold / (F * (TC + (0 if multi else 1))),  # (every frame the job fed, its untimed first pass too)
            "accumulation": os.environ.get("G1S_K3", "wide"),
        },
    }

    # ---- CPU baseline: the oracle (a port: scalar f64, the reference's operation order) on a bounded sample,
    #      one thread, then T = all host cores (T independent generators, one per thread -- the reference's `diff`
    #      is sequential per video, so all cores means T videos / frame shards at once) ----
    if rank == 0 and not multi and not args.no_cpu_baseline:
        from concurrent.futures import ThreadPoolExecutor

        from tests.oracle_binding import OracleDiff

        n_cpu = max(1, args.cpu_frames)
        host = []
        for k in range(n_cpu):
            s, d = frames[k]
            host.append(([p.cpu().numpy() for p in s], [p.cpu().numpy() for p in d]))

        def oracle_job(rows):
            o = OracleDiff(fps.numerator, fps.denominator, bd, bd, lag, chroma)
            for s, d in host:
                if rows:  # a strip of full-width block rows (same row length, format and flat fraction; fewer rows)
                    s = [np.ascontiguousarray(p[: rows >> (ydec if i else 0)]) for i, p in enumerate(s)]
                    d = [np.ascontiguousarray(p[: rows >> (ydec if i else 0)]) for i, p in enumerate(d)]
                o.diff_frame(s, d, xdec, ydec)  # (ctypes releases the GIL for the call)
            o.finish()

        t0 = time.perf_counter()
        oracle_job(0)
        one_s = time.perf_counter() - t0
        nproc = os.cpu_count() or 1
        try:
            nproc = len(os.sched_getaffinity(0))
        except Exception:
            pass
        from grav1synth_amd import _lib as _l

        nproc = max(1, min(nproc, int(_l.lib().g1s_usable_cpus())))  # (a cgroup CPU quota below the hardware threads: what runs at once)
        # all cores: every thread a strip, sized so that the leg stays near 20 s even if the threads scale no better than 8x
        strip = H if nproc <= 8 else max(32, min(H, (int(H * 8 * 20.0 / (one_s * nproc)) // 32) * 32))
        t0 = time.perf_counter()
        with ThreadPoolExecutor(nproc) as ex:
            list(ex.map(oracle_job, [strip if strip < H else 0] * nproc))
        all_s = time.perf_counter() - t0
        out["cpu_baseline"] = {
            "value": W * strip * n_cpu * nproc / all_s / 1e6,
            "unit": "Mpixels/s",
            "cores": nproc,
            "nproc": nproc,
            "kind": "port",
            "kind_note": "oracle/diff_oracle.c: a C restatement of the algorithm av1-grain 0.4.2 ports; there is no Rust toolchain in the image to build the reference itself",
            "sample": f"{nproc} threads x {n_cpu} frame pair(s) of the same workload cut to a {W}x{strip} strip, one oracle generator (oracle/liborc_diff.so: scalar f64, "
                      f"reference operation order) per thread, {all_s:.1f} s",
            "one_thread": {"value": W * H * n_cpu / one_s / 1e6, "cores": 1, "sample": f"{n_cpu} frame pair(s), {one_s:.1f} s"},
        }
    if multi:
        dist.destroy_process_group()
    sys.stdout.flush()
    if rank == 0:
        os.write(line_fd, (json.dumps(out) + "\n").encode())
    os.close(line_fd)


if __name__ == "__main__":
    main()
