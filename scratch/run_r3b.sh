#!/bin/bash
# 2-rank rehearsal (one GPU, gloo) of the bench flow with the rounds of a phase merged into one pass: weak at full size, strong, reduced sizes
mkdir -p gpurun_out
export SBG_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
run() {   # tag, bench args...
  tag=$1; shift
  cat > /tmp/b2.py <<PY
import os, runpy, sys
os.environ['LOCAL_RANK'] = '0'
sys.argv = ['bench.py', '--gpus', '2'] + """$*""".split()
runpy.run_path('bench.py', run_name='__main__')
PY
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 /tmp/b2.py > gpurun_out/dist2_$tag.log 2>&1
  echo "$tag exit $?"; grep '^{"metric"' gpurun_out/dist2_$tag.log | cut -c1-700 || tail -5 gpurun_out/dist2_$tag.log
}
run weak_full --steps 2 --warmup 1 --no-cpu-baseline
run strong_full --steps 2 --warmup 1 --scaling strong --no-cpu-baseline
run weak_small --steps 2 --warmup 1 --batch 16 --batch-gpu 8
run ffhq_res64 --steps 2 --warmup 1 --workload ffhq_sg2 --res 64 --batch 8 --batch-gpu 4
