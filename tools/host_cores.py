#!/usr/bin/env python3
"""tools/host_cores.py -- what the box really gives: cgroup CPU quota, and the aggregate rate of N spinning processes."""
import multiprocessing as mp
import os
import time


def spin(sec, q):
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < sec:
        for _ in range(20000):
            n += 1
    q.put(n)


if __name__ == "__main__":
    for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us",
              "/sys/fs/cgroup/cpuset.cpus.effective", "/sys/fs/cgroup/cpuset/cpuset.cpus"):
        try:
            print(p, open(p).read().strip())
        except OSError:
            pass
    print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
    base = None
    for n in (1, 8, 32, 64, 128, 256):
        q = mp.Queue()
        ps = [mp.Process(target=spin, args=(2.0, q)) for _ in range(n)]
        for p in ps:
            p.start()
        tot = sum(q.get() for _ in ps)
        for p in ps:
            p.join()
        base = base or tot
        print("%3d spinning processes: %.1f x one" % (n, tot / base))
