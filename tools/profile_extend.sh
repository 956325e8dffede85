#!/bin/bash
# Profiles one bench.py configuration: kernel trace + one rocprofv3 run per counter group.
# Usage (on the GPU box, from the repo root): bash tools/profile_extend.sh <tag> [bench args...]
# Writes gpurun_out/prof_<tag>/ (csv only) and gpurun_out/prof_<tag>.json (tools/extend_pmc_json.py).
set -u
TAG=${1:-run}; shift || true
ARGS=${@:-"--steps 2 --warmup 1 --no-cpu"}
OUT=$PWD/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
echo "$ARGS" > $OUT/bench_args.txt
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py $ARGS > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
grep '^{' $OUT/trace.log | tail -1 > $OUT/bench_line_under_rocprof.json
pass() { # name counters...
  local name=$1; shift
  timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_$name -o pmc -- python3 bench.py $ARGS > $OUT/pmc_$name.log 2>&1 || { echo "pmc $name failed"; tail -5 $OUT/pmc_$name.log; exit 1; }
  echo "pass $name done"
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU
pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
pass tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
pass fetch FETCH_SIZE
# the vector L1 and its address / data paths (cache line accesses, requests to L2, busy and stall cycles summed over the CUs)
pass tcp1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum TD_TC_STALL_sum
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
find $OUT -type f ! -name "*.csv" ! -name "*.log" ! -name "*.txt" ! -name "*.json" -delete; find $OUT -name "*agent_info.csv" -delete
python3 tools/extend_pmc_json.py $OUT gpurun_out/prof_$TAG.json > /dev/null && echo "summary in gpurun_out/prof_$TAG.json"
