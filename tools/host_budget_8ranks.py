#!/usr/bin/env python3
"""tools/host_budget_8ranks.py [seconds] [local_ranks] -- the HOST side of an 8-GPU frame-shard job, replayed without the GPUs.

What an 8-rank job asks of the node's cores and memory, all at once:
  * eight ranks, each turning its frames' records (285 KB a 4K frame, the integer sums the kernels leave) into latest states
    (27 KB) on its OWN per-frame pool -- g1s_latest_from_record on G1S_FOLD_THREADS threads, as bench.py sets them for
    eight local ranks (hardware threads / 8, at most 32) -- from a buffer that is rewritten per batch the way a D2H copy
    rewrites the pinned ring (a memcpy of the batch's records: the memory traffic of the copy landing);
  * rank 0's ORDERED MERGE of all eight ranks' states (g1s_shard_merge, merge pool = G1S_MERGE_THREADS, default 8), fed with
    real round messages (eight ranks x one batch a round, batch indices in the global order);
  * optionally (a GPU on the box) one device-to-host stream of records into pinned memory next to it.
Every part runs for the same wall-clock window in its own process, PACED at the rate an 8-rank job at the single-GPU speed
asks of it (a rank's half: the bench line's frames/s; the merge: 8 x that; PACE=0: flat out); printed: frames/s each part
sustained, its CPU-seconds per frame, the cores the box gives (cgroup quota) and the cores the whole job would need.
`local_ranks` (default 8) = how many ranks' halves run here: 8 replays the node, 1 replays RANK 0's share of it (its own
half + the whole merge + a copy stream) -- what fits a box whose quota is one GPU's share of the node.
No transport: the eight 1.7 MB messages a round are nothing next to the rest."""
import json
import multiprocessing as mp
import os
import sys
import time
from fractions import Fraction

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

RANKS = 8
BATCH = 64
LAG = 3


def make_records():
    import numpy as np

    from grav1synth_amd.synth import SynthSpec
    from tests.helpers import oracle_run, record_from_oracle

    spec = SynthSpec(3840, 2160, 10)
    recs = []
    oracle_run(spec, [0, 1], LAG, True, collect=lambda o, k: recs.append(record_from_oracle(o, spec, LAG, 3).buf.copy()))
    return np.stack(recs)


def pace(t0, done, rate):
    """sleep until `done` units are due at `rate` units/s (rate 0: never)"""
    if rate > 0:
        ahead = done / rate - (time.perf_counter() - t0)
        if ahead > 0:
            time.sleep(ahead)


def rank_main(rank, rec_path, seconds, threads, out_q, start_evt, rate):
    os.environ["G1S_FOLD_THREADS"] = str(threads)
    import numpy as np

    from grav1synth_amd.diff import latest_from_records

    from grav1synth_amd import _lib

    L = _lib.lib()
    R = np.load(rec_path)
    src = np.concatenate([R] * (BATCH // len(R)))  # one batch of records (what the kernels of a batch leave)
    ring = [np.empty_like(src) for _ in range(2)]  # the pinned ring a D2H copy lands in: one batch lands while one is read
    bs = int(L.g1s_latest_size(LAG))
    blobs = np.zeros((BATCH, bs), dtype=np.uint8)
    latest_from_records(src[:2], LAG)              # (pool made -- G1S_FOLD_THREADS threads, native -- code warm)
    import threading

    landed = [threading.Semaphore(0), threading.Semaphore(0)]
    free = [threading.Semaphore(1), threading.Semaphore(1)]
    stop = threading.Event()

    def lander():  # the copy landing: 18 MB written per batch (a DMA in the real job: here a core's memcpy, GIL released)
        k = 0
        while not stop.is_set():
            free[k & 1].acquire()
            np.copyto(ring[k & 1], src)
            landed[k & 1].release()
            k += 1

    th = threading.Thread(target=lander, daemon=True)
    start_evt.wait()
    th.start()
    t0 = time.perf_counter()
    c0 = time.process_time()
    frames = 0
    k = 0
    while time.perf_counter() - t0 < seconds:
        landed[k & 1].acquire()
        r = ring[k & 1]
        # the per-frame half of the batch on this rank's pool: what the generator's drainer does (Pool::parallel_for)
        rc = L.g1s_latest_from_records(r.ctypes.data, r.shape[1], BATCH, LAG, blobs.ctypes.data, bs)
        assert rc == 0
        free[k & 1].release()
        k += 1
        frames += BATCH
        pace(t0, frames, rate)
    stop.set()
    free[0].release()
    free[1].release()
    wall = time.perf_counter() - t0
    out_q.put({"rank": rank, "frames": frames, "wall_s": wall, "cpu_s": time.process_time() - c0})


def merge_main(rec_path, seconds, threads, out_q, start_evt, rate):
    os.environ["G1S_FOLD_THREADS"] = str(threads)
    os.environ["G1S_MERGE_POOL"] = str(threads)  # (the merge pool's own size; its default is min(8, usable / 2))
    import numpy as np

    from grav1synth_amd import _lib
    from grav1synth_amd.diff import RecordFold, latest_from_records

    L = _lib.lib()
    R = np.load(rec_path)
    blobs = latest_from_records(np.concatenate([R] * (BATCH // len(R))), LAG)  # one batch of latest states
    mb = int(L.g1s_shard_msg_size(LAG, BATCH))
    msgs = np.zeros((RANKS, mb), dtype=np.uint8)
    fold = RecordFold(Fraction(24, 1), LAG)
    start_evt.wait()
    t0 = time.perf_counter()
    c0 = time.process_time()
    rounds = 0
    for r in range(RANKS):  # the eight messages of a round, built once (the ranks build theirs, not rank 0) ...
        rc = L.g1s_shard_msg_from_latest_at(blobs.ctypes.data, BATCH, LAG, BATCH, 0, msgs[r].ctypes.data, mb)
        assert rc == 0
    idx = msgs.view(np.uint32)  # ... and re-labelled per round: header word 4 = the sending rank's local batch index
    while time.perf_counter() - t0 < seconds:
        idx[:, 4] = rounds      # rank r's local batch `rounds` = global batch rounds * RANKS + r
        rc = L.g1s_shard_merge(fold._h, msgs.ctypes.data, msgs.strides[0], RANKS)
        assert rc == 0, L.g1s_fold_last_error(fold._h)
        rounds += 1
        pace(t0, rounds * RANKS * BATCH, rate)
    wall = time.perf_counter() - t0
    cpu = time.process_time() - c0
    merged = int(L.g1s_fold_frames(fold._h))
    fold.finish()
    fold.close()
    out_q.put({"rank": "merge", "frames": merged, "wall_s": wall, "cpu_s": cpu, "rounds": rounds})


def d2h_main(rec_bytes, seconds, out_q, start_evt):
    try:
        if os.environ.get("NO_D2H"):
            raise RuntimeError("switched off (NO_D2H)")
        import torch

        if not torch.cuda.is_available():
            raise RuntimeError("no GPU")
        n = BATCH * rec_bytes
        dev = [torch.zeros(n, dtype=torch.uint8, device="cuda") for _ in range(2)]
        host = [torch.zeros(n, dtype=torch.uint8).pin_memory() for _ in range(4)]
        streams = [torch.cuda.Stream() for _ in range(4)]
        start_evt.wait()
        t0 = time.perf_counter()
        copied = 0
        k = 0
        while time.perf_counter() - t0 < seconds:
            for s in range(4):
                with torch.cuda.stream(streams[s]):
                    host[s].copy_(dev[k & 1], non_blocking=True)
                k += 1
            torch.cuda.synchronize()
            copied += 4 * n
        wall = time.perf_counter() - t0
        out_q.put({"rank": "d2h", "bytes": copied, "wall_s": wall})
    except Exception as e:  # no GPU: the replay runs without the copy stream
        start_evt.wait()
        out_q.put({"rank": "d2h", "bytes": 0, "wall_s": 0.0, "note": str(e)[:80]})


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
    local = int(sys.argv[2]) if len(sys.argv) > 2 else RANKS
    paced = os.environ.get("PACE", "1") != "0"
    single_gpu_fps = float(os.environ.get("SINGLE_GPU_FPS", "62900"))  # frames/s of one GPU (bench line: Mpx/s / 8.2944)
    import numpy as np

    R = make_records()
    rec_path = "/tmp/g1s_host_budget_records.npy"
    np.save(rec_path, R)
    from grav1synth_amd import _lib

    ncpu = int(_lib.lib().g1s_usable_cpus())
    per_rank = int(os.environ.get("HALF_THREADS", "0")) or max(2, min(32, ncpu // max(1, local)))  # (bench.py: G1S_FOLD_THREADS of a local rank)
    merge_threads = int(os.environ.get("G1S_MERGE_THREADS", "8"))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ev = ctx.Event()
    half_rate = single_gpu_fps if paced else 0.0
    procs = [ctx.Process(target=rank_main, args=(r, rec_path, seconds, per_rank, q, ev, half_rate)) for r in range(local)]
    procs.append(ctx.Process(target=merge_main, args=(rec_path, seconds, merge_threads, q, ev, RANKS * half_rate)))
    procs.append(ctx.Process(target=d2h_main, args=(int(R.shape[1]), seconds, q, ev)))
    for p in procs:
        p.start()
    time.sleep(8.0)  # (imports, pools, warm-up in every process)
    ev.set()
    res = [q.get() for _ in procs]
    for p in procs:
        p.join()
    ranks = sorted((r for r in res if isinstance(r["rank"], int)), key=lambda r: r["rank"])
    merge = next(r for r in res if r["rank"] == "merge")
    d2h = next(r for r in res if r["rank"] == "d2h")
    rank_fps = [r["frames"] / r["wall_s"] for r in ranks] or [float("inf")]  # (local_ranks 0: the merge alone)
    out = {
        "host": {"hw_threads": os.cpu_count(), "usable_cores_cgroup": ncpu, "ranks_replayed_here": local, "paced": paced, "ranks": RANKS, "per_rank_pool_threads": per_rank, "merge_pool_threads": merge_threads, "window_s": seconds},
        "record_bytes": int(R.shape[1]),
        "per_frame_half": {"frames_per_s_per_rank": [round(x) for x in rank_fps] if ranks else [], "min": round(min(rank_fps)) if ranks else None, "sum": round(sum(rank_fps)) if ranks else 0,
                           "cpu_us_per_frame": round(1e6 * sum(r["cpu_s"] for r in ranks) / max(1, sum(r["frames"] for r in ranks)), 1),
                           "host_memory_GBps_read_plus_written": round(2 * sum(rank_fps) * R.shape[1] / 1e9, 1) if ranks else 0},
        "ordered_merge": {"frames_per_s": round(merge["frames"] / merge["wall_s"]), "cpu_us_per_frame": round(1e6 * merge["cpu_s"] / max(1, merge["frames"]), 2),
                          "rounds": merge["rounds"]},
        "d2h_next_to_it": {"GBps": round(d2h["bytes"] / d2h["wall_s"] / 1e9, 1) if d2h["wall_s"] else None, "note": d2h.get("note")},
        "needed_at_8_ranks": {"single_gpu_frames_per_s": single_gpu_fps, "per_rank_half": single_gpu_fps, "merge": 8 * single_gpu_fps,
                              "six_x_target_merge": 6 * single_gpu_fps},
    }
    half_cpu = out["per_frame_half"]["cpu_us_per_frame"]
    merge_cpu = out["ordered_merge"]["cpu_us_per_frame"]
    out["cores_needed"] = {"a_rank_half": round(single_gpu_fps * half_cpu * 1e-6, 1), "merge_of_8": round(8 * single_gpu_fps * merge_cpu * 1e-6, 1),
                           "node_of_8": round(8 * single_gpu_fps * (half_cpu + merge_cpu) * 1e-6, 1)}
    out["verdict"] = {
        "per_frame_half_holds_a_rank": min(rank_fps) >= 0.97 * single_gpu_fps,  # (0.97: a paced loop ends a batch short of its target)
        "merge_holds_8x": out["ordered_merge"]["frames_per_s"] >= 0.97 * 8 * single_gpu_fps,
        "merge_holds_6x": out["ordered_merge"]["frames_per_s"] >= 0.97 * 6 * single_gpu_fps,
        "scaling_ceiling_from_the_host": round(min(8.0, out["ordered_merge"]["frames_per_s"] / single_gpu_fps, 8.0 * min(rank_fps) / single_gpu_fps), 2),
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
