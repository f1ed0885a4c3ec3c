# Measurement only (wrong results): what the prologue / epilogue of a wino16_kernel<4, EPI_LSTM> block cost.  Needs two measurement libraries built from a
# TEMPORARY patch of csrc/conv_wino16.h (never committed: it would change the kernel-source hash of the stamped profiles):
#   EIG_W16_DIAG & 1: the lstm_cell(...) call of the EPI_LSTM epilogue replaced by  cn = (zi + bi) + (zf + bf) + pi * c;  hn = (zc + bc) + (zo + bo) + pf * po;
#   EIG_W16_DIAG & 2: `return` right after the last K-block (behind a store that never happens, which keeps the accumulator chains alive)
#   hipcc ... -DEIG_W16_DIAG=1 -o libeigen_diag1.so ; -DEIG_W16_DIAG=2 -o libeigen_diag2.so     record: profiles/r04_zzzz_w16_prologue_epilogue.txt
export TMPDIR=/tmp; R=$(pwd)
run() { name=$1; shift; env "$@" python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-parity --no-supplementary 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
ops={(o['layer'],o['op']):o for o in r['per_op']}
l=[ops[(i,'lstm')]['ms']/ops[(i,'lstm')]['launches'] for i in (1,2,3)]
b=[l[0]*1e3/192, l[1]*1e3/96, l[2]*1e3/48]
print('%-10s %8.2f evals/s  lstm ms/launch %s  us/block %s  P=2*b1-b2 %.2f us  f=(b3-P)/72 %.3f us' % ('$name', d['value'], [round(x,3) for x in l], [round(x,1) for x in b], 2*b[0]-b[1], (b[2]-(2*b[0]-b[1]))/72))
"; }
L=$R/evolutionary_illusion_generator_amd
for rep in 1 2; do
run default X=1
run no_gates EIGEN_HIP_LIB=$L/libeigen_diag1.so
run no_epilogue EIGEN_HIP_LIB=$L/libeigen_diag2.so
done
