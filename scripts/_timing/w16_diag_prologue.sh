export TMPDIR=/tmp; R=$(pwd)
run() { name=$1; shift; env "$@" python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-parity --no-supplementary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
ops={(o['layer'],o['op']):o for o in r['per_op']}
l=[ops[(i,'lstm')]['ms']/ops[(i,'lstm')]['launches'] for i in (1,2,3)]
b=[l[0]*1e3/192, l[1]*1e3/96, l[2]*1e3/48]
print('%-26s %8.2f evals/s  lstm ms/launch %s  us/block %s  P=2*b1-b2 %.2f us' % ('$name', d['value'], [round(x,3) for x in l], [round(x,1) for x in b], 2*b[0]-b[1]))
"; }
L=$R/evolutionary_illusion_generator_amd
for rep in 1 2; do
run default X=1
run no_dma_wait EIGEN_HIP_LIB=$L/libeigen_diag4.so
run no_dma_wait_no_transform0 EIGEN_HIP_LIB=$L/libeigen_diag12.so
run K_loops_only EIGEN_HIP_LIB=$L/libeigen_diag14.so
done
