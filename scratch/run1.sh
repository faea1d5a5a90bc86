timeout -k 10 300 python scratch/k64_check.py > gpurun_out/s9_check.log 2>&1 || { tail -40 gpurun_out/s9_check.log; exit 1; }; tail -1 gpurun_out/s9_check.log
timeout -k 10 120 python scratch/kbench.py conv 2>&1 | grep "halo=False\|s2\|convT" > gpurun_out/s9_kbench.log 2>&1
cat gpurun_out/s9_kbench.log
