#!/bin/bash
python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/smoke.log
bash profiles/collect.sh r01m > gpurun_out/collect.log 2>&1; echo "collect rc=$?"; tail -5 gpurun_out/collect.log
timeout -k 10 400 python bench.py > gpurun_out/default_bench.json 2> gpurun_out/default_bench.err; echo "default bench rc=$?"; tail -c 600 gpurun_out/default_bench.json
