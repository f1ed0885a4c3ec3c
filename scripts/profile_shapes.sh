#!/bin/bash
# rocprofv3 kernel stats of the supplementary shapes: scripts/profile_shapes.sh TAG shape ... -> gpurun_out/prof_TAG/kernel_stats_<shape>.csv
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT
cd /tmp
for sh in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$sh -o kt -- python $R/bench.py --shape $sh --steps 3 --warmup 1 --no-roofline > $OUT/bench_$sh.json 2> $OUT/rocprof_$sh.err
  find $OUT/kt_$sh -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_$sh.csv \;
  rm -rf $OUT/kt_$sh
  echo "== $sh"; head -6 $OUT/kernel_stats_$sh.csv | cut -c1-140
done
