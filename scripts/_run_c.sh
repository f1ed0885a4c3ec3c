cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04_c; mkdir -p $O
python -m pytest tests/test_gpu_round2.py -x -q -k "bench_gpus" > $O/pytest_bench.log 2>&1; tail -5 $O/pytest_bench.log
python scripts/winograd_study.py --genomes 256 --out $O/winograd_study.json > $O/winograd.txt 2>&1; tail -12 $O/winograd.txt
for sh in ref160 c2 ref640; do
  python bench.py --shape $sh --steps 6 --warmup 2 2>/dev/null | tail -1 > $O/bench_$sh.json
done
python scripts/perop.py $O/bench_ref160.json $O/bench_c2.json $O/bench_ref640.json
for v in 0 1; do
  (cd /tmp && EIGEN_PIPE2=$v rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/kt$v -o kt -- python $GRAFT_REPO_ROOT/bench.py --shape c2 --steps 3 --warmup 1 --no-roofline > /dev/null 2> $GRAFT_REPO_ROOT/$O/rocprof$v.err)
  f=$(find $O/kt$v -name "*kernel_trace.csv" | head -1)
  echo "== c2 EIGEN_PIPE2=$v"; python scripts/trace_overlap.py $f; python scripts/trace_gaps.py $f | head -8
  cp $f $O/c2_pipe${v}_kernel_trace.csv; rm -rf $O/kt$v
done
