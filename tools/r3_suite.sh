#!/bin/bash
# the whole -m gpu suite, smoke(), the default bench line
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/suite.txt 2>&1
tail -15 gpurun_out/suite.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.txt 2>&1; tail -3 gpurun_out/smoke.txt
timeout 600 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 1500 gpurun_out/bench_default.json
