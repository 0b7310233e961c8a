mkdir -p gpurun_out
bash tools/r4_quick.sh > gpurun_out/r04f_quick.txt 2>&1
python tools/host_cores.py > gpurun_out/r04_host_cores.txt 2>&1
python tools/half_scaling.py >> gpurun_out/r04_host_cores.txt 2>&1
cat gpurun_out/r04f_quick.txt gpurun_out/r04_host_cores.txt
