#!/usr/bin/env python3
"""tools/gpu_sampler.py OUT [period_s] -- samples the GPU's clocks, power, busy percent and the host's load from sysfs until
killed (SIGTERM): one line per sample, "t sclk_mhz mclk_mhz fclk_mhz power_w busy_pct temp_c cpu_busy_cores".  No HIP call:
it does not touch the runtime or the device's queues."""
import glob
import os
import signal
import sys
import time

out = open(sys.argv[1], "w")
period = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
stop = False


def _term(*_):
    global stop
    stop = True


signal.signal(signal.SIGTERM, _term)
cards = [c for c in sorted(glob.glob("/sys/class/drm/card*/device")) if os.path.exists(c + "/pp_dpm_sclk")]
card = cards[0] if cards else None


def cur(path):
    try:
        for ln in open(path):
            if "*" in ln:
                return ln.split(":")[1].strip().split("Mhz")[0].strip()
    except Exception:
        pass
    return "-"


def rd(path):
    try:
        return open(path).read().strip()
    except Exception:
        return "-"


def cpu_total():
    f = open("/proc/stat").readline().split()
    busy = sum(int(x) for x in f[1:4]) + sum(int(x) for x in f[6:9])
    return busy


hw = glob.glob(card + "/hwmon/hwmon*")[0] if card and glob.glob(card + "/hwmon/hwmon*") else None
print("# card", card, "hwmon", hw, file=out)
t0 = time.time()
last_cpu, last_t = cpu_total(), t0
hz = os.sysconf("SC_CLK_TCK")
while not stop:
    t = time.time()
    c = cpu_total()
    cores = (c - last_cpu) / hz / max(t - last_t, 1e-6)
    last_cpu, last_t = c, t
    p = rd(hw + "/power1_average") if hw else "-"
    if p == "-" and hw:
        p = rd(hw + "/power1_input")
    try:
        p = "%.0f" % (int(p) / 1e6)
    except Exception:
        pass
    temp = rd(hw + "/temp1_input") if hw else "-"
    print("%.2f %s %s %s %s %s %s %.1f" % (t - t0, cur(card + "/pp_dpm_sclk") if card else "-", cur(card + "/pp_dpm_mclk") if card else "-",
                                         cur(card + "/pp_dpm_fclk") if card else "-", p, rd(card + "/gpu_busy_percent") if card else "-", temp, cores),
          file=out, flush=True)
    time.sleep(period)
