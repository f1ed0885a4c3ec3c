cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04_p; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -k "winograd or rollout_frames" 2>&1 | tail -2
for lib in prev new prev new; do
  if [ $lib = prev ]; then export EIGEN_HIP_LIB=$GRAFT_REPO_ROOT/evolutionary_illusion_generator_amd/libeigen_hip_prev.so; else unset EIGEN_HIP_LIB; fi
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-supplementary 2>$O/bench.err | tail -1 > $O/bench_$lib.json
  python -c "
import json
d=json.loads(open('$O/bench_$lib.json').read()); r=d['roofline']
print('$lib: %.2f evals/s conv ms %.1f' % (d['value'], r['all_conv_kernels']['total_ms']), [(o['layer'], round(o['ms'],1)) for o in r['per_op'] if o['op']=='lstm' and o['layer']>0])"
done
