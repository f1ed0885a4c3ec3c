cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04_m; mkdir -p $O
EIGEN_WINO_MODE=8 python -m pytest tests/test_gpu_parity.py -x -q -k "winograd" 2>&1 | tail -3
for m in 4 8 4 8; do
  EIGEN_WINO_MODE=$m python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-supplementary 2>$O/bench.err | tail -1 > $O/bench_m$m.json
  python -c "
import json
d=json.loads(open('$O/bench_m$m.json').read()); r=d['roofline']
print('EIGEN_WINO_MODE=$m: %.2f evals/s conv ms %.1f  ' % (d['value'], r['all_conv_kernels']['total_ms']), [(o['layer'], o['op'][:5], round(o['ms'],1)) for o in r['per_op'] if o['ms']>10 and o['layer']>0])"
done
for sh in "ref160 10" "ref640 4"; do
  set -- $sh
  for m in 4 8; do
  EIGEN_WINO_MODE=$m python bench.py --shape $1 --steps $2 --warmup 2 2>$O/bench.err | tail -1 > $O/bench_$1.json
  python -c "
import json
d=json.loads(open('$O/bench_$1.json').read()); r=d['roofline']
print('$1 mode $m: %.2f evals/s conv ms %.2f' % (d['value'], r['all_conv_kernels']['total_ms']))"
  done
done
