#!/bin/bash
# A/B driver, round 5: tools/ktime.py under several environments / variant libraries (same box, one call)
#   tools/r5_ab.sh OUTNAME "TAG=a ENV=1" "TAG=b G1S_LIB=v_name" ...   (G1S_LIB=v_NAME -> grav1synth_amd/libg1s_v_NAME.so)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
OUT=gpurun_out/${1:-r05_ab}.txt
shift
: > $OUT
run() {
  echo "## $*" >> $OUT
  local args=()
  for a in "$@"; do
    case "$a" in G1S_LIB=v_*) a="G1S_LIB=$PWD/grav1synth_amd/libg1s_${a#G1S_LIB=}.so";; esac
    args+=("$a")
  done
  env "${args[@]}" python tools/ktime.py ${NB:-4} 2>/dev/null | tail -1 >> $OUT
}
python tools/ktime.py 2 > /dev/null 2>&1
for spec in "$@"; do run $spec; done
cat $OUT
