#!/bin/bash
# quick check of a build: the record parity cases, then per-kernel times of the bench workload (and its all-flat variant)
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_records_and_table_match_oracle or large_residuals or ragged or partial_last or int32_acc" 2>&1 | tail -3
python tools/ktime.py 2 > /dev/null 2>&1  # (the first process on a fresh box runs 5 - 10 % slow)
python tools/ktime.py 4 2>/dev/null | tail -1
FLAT=1 python tools/ktime.py 4 2>/dev/null | tail -1
