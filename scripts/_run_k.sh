cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04_k; mkdir -p $O
for m in 4 5 6 7; do
  EIGEN_WINO_MODE=$m python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-supplementary 2>$O/bench.err | tail -1 > $O/bench_m$m.json
  python -c "
import json
d=json.loads(open('$O/bench_m$m.json').read()); r=d['roofline']
print('EIGEN_WINO_MODE=$m: %.2f evals/s conv ms %.1f  ConvLSTM layers 1-3:' % (d['value'], r['all_conv_kernels']['total_ms']), [round(o['ms'],1) for o in r['per_op'] if o['op']=='lstm' and o['layer']>0])"
done
