#!/bin/bash
# fuzz of the default (stream) chain against the oracle: small frames, large frames (long runs: the fast path), few workgroups
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{
echo "# tools/fuzz_parity.py / tools/debug_damage3.py on the round-3 build (G1S_K3=stream default), 1x MI355X"
echo "## 600 small cases (<= 420 x 300), seed 41"; timeout 1500 python tools/fuzz_parity.py 600 41 2>&1 | tail -4
echo "## 160 large cases (<= 1500 x 700: runs of units, the fast path), seed 42"; timeout 2400 python tools/fuzz_parity.py 160 42 1500 700 2>&1 | tail -4
echo "## 120 large cases with 8 workgroups a frame (G1S_F_WGS=8: long runs per workgroup), seed 33"; G1S_F_WGS=8 timeout 2400 python tools/fuzz_parity.py 120 33 1500 700 2>&1 | tail -4
echo "## 120 large cases without the reuse / fast path (G1S_F_REUSE=0), seed 34"; G1S_F_REUSE=0 timeout 2400 python tools/fuzz_parity.py 120 34 1500 700 2>&1 | tail -4
echo "## 300 small + 80 large cases with the per-frame half of the fold on the device (G1S_LATEST=device: k4_latest), seeds 36, 37"
G1S_LATEST=device timeout 1500 python tools/fuzz_parity.py 300 36 2>&1 | tail -2
G1S_LATEST=device timeout 2400 python tools/fuzz_parity.py 80 37 1500 700 2>&1 | tail -2
echo "## damaged frames (isolated residuals outside int8): 300 cases, seed 35"; timeout 1500 python tools/debug_damage3.py 300 35 2>&1 | tail -6
} > gpurun_out/r03_fuzz_parity.txt 2>&1
cat gpurun_out/r03_fuzz_parity.txt
