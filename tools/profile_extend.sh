#!/bin/bash
# Profiles the closest-hit kernel on the 1M-triangle scene with random rays.
# Usage (on the GPU box, from the repo root): bash tools/profile_extend.sh <tag> [bench args...]
# Writes gpurun_out/prof_<tag>/: kernel-trace stats and one directory per PMC pass.
set -u
TAG=${1:-run}; shift || true
ARGS=${@:-"--workload closest_hit --tris 1000000 --rays 16000000 --steps 3 --warmup 1 --no-cpu"}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py $ARGS > $OUT/trace.log 2>&1 || { echo "trace failed"; tail -5 $OUT/trace.log; exit 1; }
pass() { # name counters...
  local name=$1; shift
  timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_$name -o pmc -- python3 bench.py $ARGS > $OUT/pmc_$name.log 2>&1 || { echo "pmc $name failed"; tail -5 $OUT/pmc_$name.log; exit 1; }
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU
pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM
pass tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
pass fetch FETCH_SIZE
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
find $OUT -type f ! -name "*.csv" ! -name "*.log" ! -name "*.txt" -delete; find $OUT -name "*agent_info.csv" -delete; echo done
