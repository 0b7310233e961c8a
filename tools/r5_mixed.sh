cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
O=gpurun_out/r05_mixed.txt; : > $O
for spec in "TAG=10_10" "TAG=10_8 DEN_BD=8" "TAG=10_8_stream DEN_BD=8 G1S_K3=stream" "TAG=10_12 DEN_BD=12"; do
  env $spec python tools/ktime.py 3 2>/dev/null | tail -1 >> $O
done
cat $O
