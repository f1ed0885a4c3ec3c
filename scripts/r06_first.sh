#!/bin/bash
# round 6, first GPU call: bf16 MFMA order probe, the new hard reference-order test at the headline shape, the default bench line, kernel stats
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
O=gpurun_out/r06_a; mkdir -p $O
( time scripts/_timing/mfma_bf16_order 40000 ) > $O/mfma_bf16_order.txt 2>&1; tail -25 $O/mfma_bf16_order.txt
python -m pytest tests/test_gpu_round2.py -m gpu -x -q -k "element_order_c3 or rccl" -s > $O/pytest_c3.log 2>&1; tail -8 $O/pytest_c3.log
( time python bench.py --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err ) 2>&1 | tail -3
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-supplementary > $GRAFT_REPO_ROOT/$O/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/rocprof.err
cd $GRAFT_REPO_ROOT
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/kt
head -6 $O/kernel_stats.csv
python - <<PY
import json
d=json.loads(open("$O/bench_full.json").read().strip().splitlines()[-1])
print("value", d["value"], "frac", d["roofline"]["frac"], json.dumps(d["parity_check"].get("vs_c_reference_order")))
PY
