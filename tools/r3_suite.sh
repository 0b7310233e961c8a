#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -m gpu -x -q --durations=12 ) > gpurun_out/suite.txt 2>&1
tail -25 gpurun_out/suite.txt
( time python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) 2>&1 | tail -5
