#!/bin/bash
# rehearse the N=2 bench flow on ONE GPU (both ranks on device 0, gloo instead of RCCL)
mkdir -p gpurun_out
export SBG_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
cat > /tmp/b2.py <<'PY'
import os, runpy, sys
os.environ['LOCAL_RANK'] = '0'
sys.argv = ['bench.py', '--gpus', '2', '--steps', '2', '--warmup', '1']
runpy.run_path('bench.py', run_name='__main__')
PY
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 /tmp/b2.py > gpurun_out/dist2.log 2>&1
echo "exit $?" >> gpurun_out/dist2.log
tail -5 gpurun_out/dist2.log
