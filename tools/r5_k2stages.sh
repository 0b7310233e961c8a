#!/bin/bash
# tools/r5_k2stages.sh -- k2w_select_units cut short after its stages (variant libraries libg1s_v_k2s{1,2,3}.so: -DG1S_K2_STOP=n), timed by
# rocprofv3 --kernel-trace --stats (the job's fold fails on such a build; only the kernel's own duration is read)
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
O=gpurun_out/r05h_select_stages.txt; : > $O
for wl in 4k10 1080p8; do
  B=64; [ $wl = 1080p8 ] && B=128
  for v in k2s1 k2s2 k2s3 full; do
    L=$PWD/grav1synth_amd/libg1s_v_$v.so; [ $v = full ] && L=$PWD/grav1synth_amd/libg1s_diff.so
    WL=$wl BATCH=$B G1S_LIB=$L G1S_ONE_STREAM=1 tools/prof.sh k2_$v_$wl --kernel-trace --stats -- python $PWD/tools/ktime.py 1 > /dev/null 2>&1
    echo "## $wl $v" >> $O
    python tools/kstats.py gpurun_out/k2_$v_$wl 2>/dev/null | grep "k2w_select" >> $O
  done
done
cat $O
