#!/bin/bash
# PMC counter passes for the conv kernels (one rocprofv3 run per pass; --pmc never combined with trace domains other than kernel-trace)
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r01}
POP=${2:-64}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp
run() { # name, counters...
  n=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -o $n -- python $R/bench.py --pop $POP --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-parity --no-supplementary > $OUT/$n.log 2>&1
  echo "pass $n rc=$?"
}
run sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM TCC_HIT_sum TCC_MISS_sum
python $R/scripts/summarize_pmc.py $OUT $OUT/pmc_summary.json $POP > $OUT/summary.txt 2>&1
tail -12 $OUT/summary.txt
