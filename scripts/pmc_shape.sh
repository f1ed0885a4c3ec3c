#!/bin/bash
# PMC counter passes of one supplementary shape (per-kernel averages): scripts/pmc_shape.sh TAG SHAPE -> gpurun_out/pmc_TAG/*.txt
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-x}; SHAPE=${2:-ref160}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp
run() { # name, counters...
  n=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -o $n -- python $R/bench.py --shape $SHAPE --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-parity --no-supplementary > $OUT/$n.log 2>&1
  echo "pass $n rc=$?"
  python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/$n/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for row in csv.DictReader(open(f[0])):
    k = row["Kernel_Name"][:70]
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
    cnt[(k, row["Counter_Name"])] += 1
for k in sorted(agg, key=lambda k: -sum(agg[k].values()))[:8]:
    print(k)
    for c, v in agg[k].items():
        print("    %-28s %16.0f per launch (%d launches)" % (c, v / cnt[(k, c)], cnt[(k, c)]))
PY
}
run sq SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE > $OUT/sq.txt 2>&1
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM TCC_HIT_sum TCC_MISS_sum > $OUT/lds.txt 2>&1
rm -rf $OUT/sq $OUT/lds
head -60 $OUT/sq.txt $OUT/lds.txt
