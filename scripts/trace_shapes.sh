#!/bin/bash
# Kernel traces of the small-map shapes (where launches under-fill the chip): scripts/trace_shapes.sh TAG "shape-args" ...
# -> gpurun_out/trace_TAG/<n>_kernel_trace.csv + <n>_kernel_stats.csv + <n>_bench.json; scripts/trace_gaps.py summarises them.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp
i=0
for args in "$@"; do
  i=$((i+1))
  python $R/bench.py $args --steps 5 --warmup 2 --no-cpu-baseline > $OUT/${i}_bench.json 2> $OUT/${i}_bench.err
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt$i -o kt -- python $R/bench.py $args --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $OUT/${i}_under_rocprof.json 2> $OUT/${i}_rocprof.err
  find $OUT/kt$i -name "*kernel_stats.csv" -exec cp {} $OUT/${i}_kernel_stats.csv \;
  find $OUT/kt$i -name "*kernel_trace.csv" -exec cp {} $OUT/${i}_kernel_trace.csv \;
  rm -rf $OUT/kt$i
  echo "== $args"; python -c "import json;d=json.loads(open('$OUT/${i}_bench.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['roofline']['all_conv_kernels'])"
  python $R/scripts/trace_gaps.py $OUT/${i}_kernel_trace.csv | tail -25
done
