#!/bin/bash
# tools/r5_fuzz.sh [scale] -- fuzz of the build in the tree against the oracle (tools/fuzz_parity.py, tools/debug_damage3.py, the flat-block
# finder's stress): small frames, large frames (long runs of units), few workgroups a frame, the fallback chain, the per-frame half on
# the device (the rebuilt k4_latest), mixed depths (in every run: the wide chain's general residual form).  The file names the commit.
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
S=${1:-1}
{
echo "# tools/r5_fuzz.sh $S (OUT=${OUT:-r06}) on commit $(cat .gpurun_head 2>/dev/null || echo unknown), 1x MI355X"
echo "## $((600*S)) small cases (<= 420 x 300) with widths that are multiples of 16 (FUZZ_ALIGN=16: the wide chain every time), seed 60"; FUZZ_ALIGN=16 timeout 2400 python tools/fuzz_parity.py $((600*S)) 60 2>&1 | tail -3
echo "## $((300*S)) small cases, random widths (one in sixteen the wide chain, the others the fallback), seed 61"; timeout 2400 python tools/fuzz_parity.py $((300*S)) 61 2>&1 | tail -3
echo "## $((160*S)) large cases (<= 1500 x 700), widths multiples of 16, seed 62"; FUZZ_ALIGN=16 timeout 2400 python tools/fuzz_parity.py $((160*S)) 62 1500 700 2>&1 | tail -3
echo "## $((120*S)) large cases with 8 workgroups a frame (G1S_W_WGS=8 G1S_W_WGS_C=8), seed 63"; FUZZ_ALIGN=16 G1S_W_WGS=8 G1S_W_WGS_C=8 timeout 2400 python tools/fuzz_parity.py $((120*S)) 63 1500 700 2>&1 | tail -3
echo "## $((120*S)) large cases down the fallback chain (G1S_K3=stream), seed 64"; G1S_K3=stream timeout 2400 python tools/fuzz_parity.py $((120*S)) 64 1500 700 2>&1 | tail -3
echo "## $((400*S)) small + $((100*S)) large cases with the per-frame half on the device (G1S_LATEST=device), seeds 66, 67"
FUZZ_ALIGN=16 G1S_LATEST=device timeout 2400 python tools/fuzz_parity.py $((400*S)) 66 2>&1 | tail -2
G1S_LATEST=device timeout 2400 python tools/fuzz_parity.py $((100*S)) 67 1500 700 2>&1 | tail -2
echo "## damaged frames (isolated residuals outside int8): $((300*S)) cases, seed 65"; timeout 2400 python tools/debug_damage3.py $((300*S)) 65 2>&1 | grep -v amdgpu.ids | tail -4
echo "(end)"
} > gpurun_out/${OUT:-r06}_fuzz_parity.txt 2>&1
grep -v amdgpu.ids gpurun_out/${OUT:-r06}_fuzz_parity.txt
