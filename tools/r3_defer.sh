#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 1200 python -m pytest tests -m gpu -x -q -k "large_residuals or records_and_table or modes_agree or tuning or ceiling or goldens or halo or damaged or int8" 2>&1 | tail -3
timeout 900 python tools/debug_damage3.py 150 51 2>&1 | tail -3; echo "damage done"
timeout 900 python tools/fuzz_parity.py 200 52 2>&1 | tail -2
