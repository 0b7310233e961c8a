#!/bin/bash
# final-build checks that the test suite cannot afford: full-size parity against the oracle, lifecycle soak, a long fuzz
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
{
echo "# final build (OUT=${OUT:-r06}; commit $(cat .gpurun_head 2>/dev/null || echo unknown)), 1x MI355X"
echo "## tools/check_large.py 3840 2160 10 1 1 3 (the bench workload's format, three frames, every record field and the table vs the oracle)"
timeout 900 python tools/check_large.py 3840 2160 10 1 1 3 2>&1 | tail -1
echo "## tools/check_large.py (8K 10-bit 4:4:4, two frames)"
timeout 1500 python tools/check_large.py 2>&1 | tail -1
echo "## tools/soak.py 300"
timeout 900 python tools/soak.py 300 2>&1 | tail -2
echo "## tools/fuzz_parity.py 1500 71 (small), 300 72 1500 700 (large), with G1S_LATEST=device 300 73"
timeout 1800 python tools/fuzz_parity.py 1500 71 2>&1 | tail -2
timeout 2400 python tools/fuzz_parity.py 300 72 1500 700 2>&1 | tail -2
G1S_LATEST=device timeout 1500 python tools/fuzz_parity.py 300 73 2>&1 | tail -2
echo "## tools/debug_damage3.py 400 75 (prints failures only)"
timeout 1500 python tools/debug_damage3.py 400 75 2>&1 | grep -v amdgpu.ids | tail -5
echo "(end)"
} > gpurun_out/${OUT:-r06}_final_checks.txt 2>&1
grep -v "amdgpu.ids" gpurun_out/${OUT:-r06}_final_checks.txt
