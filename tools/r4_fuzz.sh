#!/bin/bash
# fuzz of the default (wide) chain against the oracle: small frames, large frames (long runs of units), few workgroups, the device half
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{
echo "# tools/fuzz_parity.py / tools/debug_damage3.py on the round-4 build (G1S_K3=wide default), 1x MI355X"
echo "## 600 small cases (<= 420 x 300), seed 51"; timeout 1500 python tools/fuzz_parity.py 600 51 2>&1 | tail -4
echo "## 160 large cases (<= 1500 x 700), seed 52"; timeout 2400 python tools/fuzz_parity.py 160 52 1500 700 2>&1 | tail -4
echo "## 120 large cases with 8 workgroups a frame (G1S_W_WGS=8 G1S_W_WGS_C=8: long slices per workgroup), seed 53"; G1S_W_WGS=8 G1S_W_WGS_C=8 timeout 2400 python tools/fuzz_parity.py 120 53 1500 700 2>&1 | tail -4
echo "## 120 large cases down the fallback chain (G1S_K3=stream), seed 54"; G1S_K3=stream timeout 2400 python tools/fuzz_parity.py 120 54 1500 700 2>&1 | tail -4
echo "## 300 small + 80 large cases with the per-frame half of the fold on the device (G1S_LATEST=device: k4_latest), seeds 56, 57"
G1S_LATEST=device timeout 1500 python tools/fuzz_parity.py 300 56 2>&1 | tail -2
G1S_LATEST=device timeout 2400 python tools/fuzz_parity.py 80 57 1500 700 2>&1 | tail -2
echo "## damaged frames (isolated residuals outside int8): 300 cases, seed 55"; timeout 1500 python tools/debug_damage3.py 300 55 2>&1 | tail -6
} > gpurun_out/r04_fuzz_parity.txt 2>&1
cat gpurun_out/r04_fuzz_parity.txt
