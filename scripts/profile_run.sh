#!/bin/bash
# One profiling round on the GPU box: scripts/profile_run.sh TAG
#   1. plain bench line (the number the profiles describe)            -> gpurun_out/prof_TAG/bench.json
#   2. rocprofv3 --kernel-trace --stats of the SAME command             -> gpurun_out/prof_TAG/kernel_stats.csv
#   3. PMC passes (scripts/pmc_passes.sh: one rocprofv3 run per group)  -> gpurun_out/pmc_TAG/pmc_summary.json
# Copy what is to be judged into profiles/ (tracked) afterwards.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
python $R/bench.py --steps 3 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-supplementary > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
find $OUT/kt -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
head -8 $OUT/kernel_stats.csv
cd $R
bash $R/scripts/pmc_passes.sh $TAG 256
cp $R/gpurun_out/pmc_$TAG/pmc_summary.json $OUT/pmc_summary.json 2>/dev/null
# the bench line again, now that a PMC summary of THIS build exists (roofline.traffic)
cp $OUT/pmc_summary.json $R/profiles/pmc_summary_latest.json
cd /tmp
python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-supplementary > $OUT/bench_with_traffic.json 2>/dev/null
rm -rf $OUT/kt $R/gpurun_out/pmc_$TAG/*/  # raw traces are large
python - <<PY
import json
d=json.loads(open("$OUT/bench_with_traffic.json").read().strip().splitlines()[-1])
print("value", d["value"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], d["roofline"]["traffic_note"])
PY
