#!/bin/bash
# tools/prof.sh NAME [rocprofv3 args...] -- CMD...
# Runs rocprofv3 with CSV output into gpurun_out/NAME (never blocks on stdin, bounded by timeout).
set -u
NAME=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$NAME
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --output-format csv -d "$OUT" -o p "$@" > "$OUT/run.log" 2>&1 < /dev/null
echo "rocprofv3 rc=$?"
ls "$OUT" | head -20
