#!/bin/bash
# Run on the GPU box from the repo root:  bash profiles/collect.sh <tag>
# rocprofv3 passes over the same bench command, each its own run (the TCC counters do not fit one pass, and gpurun refuses --pmc
# together with API tracing):  (1) kernel trace + stats, with bench.py also writing its launch log (shape of every launch, in order);
# (2) FETCH_SIZE;  (3) WRITE_SIZE;  (4) SQ / GRBM counters: matrix-pipe busy cycles, wave cycles, LDS bank conflicts.
# profiles/summarize.py condenses them into the files committed under profiles/.  A failing trace pass fails the script; the outcome of
# every counter pass is recorded (see below).
set -e
set -o pipefail
TAG=${1:-r02}
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
# --single-thread-autograd: every kernel is submitted by ONE host thread (one of the hypotheses tested for the counter-collection aborts
# below; it did not remove them, and is kept so that launch order = dispatch order for the launch-log join of summarize.py).
CMD="python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --single-thread-autograd"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o $TAG --output-format csv -- $CMD --launch-log $OUT/launch_log.jsonl > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
echo "stats pass done"
# Counter-collection passes.  rocprofv3 counter collection on this pool dies now and then in the first seconds of a run -- queue abort
# HSA_STATUS_ERROR_INVALID_PACKET_FORMAT, or SIGSEGV inside the profiler's dispatch interception -- while code objects are being loaded: a
# one-line torch program without this library crashes the same way (deterministically at dlopen with HIP_ENABLE_DEFERRED_LOADING=0; see
# DESIGN.md, "counter collection").  Each pass therefore runs ONCE, is never retried, and its outcome is written to <tag>_pmc_status.txt;
# the summary is built from the passes that completed, and says which did not.
STATUS=$OUT/pmc_status.txt; : > $STATUS
run_pmc() {   # name, counters..., then extra bench flags after --
  local name=$1; shift
  local counters=(); while [ "$1" != "--" ]; do counters+=("$1"); shift; done; shift
  if timeout -k 10 400 rocprofv3 --kernel-trace --pmc "${counters[@]}" -d $OUT/$name -o $TAG --output-format csv -- $CMD "$@" > /dev/null 2> $OUT/$name.log; then
    echo "$name ok" >> $STATUS
  else
    echo "$name FAILED: $(grep -m1 -E 'aborting with error|SIGSEGV|Aborted' $OUT/$name.log | cut -c1-160)" >> $STATUS
  fi
  echo "$name pass done: $(tail -1 $STATUS)"
}
run_pmc fetch FETCH_SIZE --
run_pmc write WRITE_SIZE --
run_pmc sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -- --launch-log $OUT/launch_log_sq.jsonl
cp $STATUS profiles/${TAG}_pmc_status.txt
python3 profiles/summarize.py $OUT $TAG
# keep only the condensed files in gpurun_out (the raw traces exceed what gpurun merges back)
mkdir -p gpurun_out/profiles_out && cp profiles/${TAG}_* gpurun_out/profiles_out/ && cp $OUT/*.log gpurun_out/profiles_out/ && rm -rf $OUT/stats $OUT/fetch $OUT/write $OUT/sq
