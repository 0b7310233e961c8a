#!/usr/bin/env python3
"""tools/thread_cpu.py SECONDS -- CMD...: runs CMD and, SECONDS after its start, samples every thread of the process tree twice a second apart
(/proc/PID/task/TID/stat: utime + stime): which threads of a rank burn the host's cores (cores busy per thread name over the window)."""
import os, subprocess, sys, time, collections

delay = float(sys.argv[1])
cmd = sys.argv[sys.argv.index("--") + 1:]
p = subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
time.sleep(delay)
tck = os.sysconf("SC_CLK_TCK")

def pids(root):
    out, todo = [], [root]
    while todo:
        q = todo.pop()
        out.append(q)
        try:
            for t in os.listdir(f"/proc/{q}/task"):
                try:
                    todo += [int(c) for c in open(f"/proc/{q}/task/{t}/children").read().split()]
                except OSError:
                    pass
        except OSError:
            pass
    return sorted(set(out))

def snap():
    s = {}
    for q in pids(p.pid):
        try:
            for t in os.listdir(f"/proc/{q}/task"):
                f = open(f"/proc/{q}/task/{t}/stat").read()
                name = f[f.index("(") + 1:f.rindex(")")]
                rest = f[f.rindex(")") + 2:].split()
                s[(q, int(t))] = (name, (int(rest[11]) + int(rest[12])) / tck)
        except OSError:
            pass
    return s

W = 2.0
a = snap(); t0 = time.perf_counter(); time.sleep(W); b = snap(); dt = time.perf_counter() - t0
rows = []
for k, (name, cpu) in b.items():
    if k in a:
        rows.append(((cpu - a[k][1]) / dt, k, name))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"threads {len(rows)}  cores busy {tot:.2f} over {dt:.2f} s")
for c, (q, t), name in rows[:14]:
    print(f"  {c:5.2f}  pid {q} tid {t} {name}{'  (main thread)' if q == t else ''}")
p.wait()
