#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
O=gpurun_out/${1:-r06_f}; mkdir -p $O
for l in libeigen_l0b.so libeigen_l0c.so; do EIGEN_HIP_LIB=$GRAFT_REPO_ROOT/evolutionary_illusion_generator_amd/$l timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rollout_frames" > $O/pytest_$l.log 2>&1; echo "$l: $(tail -1 $O/pytest_$l.log)"; done
ARGS="--steps 4" bash scripts/ab_libs.sh $O/ab libeigen_l0b.so libeigen_l0c.so 2>&1 | tee $O/ab_libs.txt
for l in "" libeigen_l0b.so libeigen_l0c.so; do
if [ -n "$l" ]; then export EIGEN_HIP_LIB=$GRAFT_REPO_ROOT/evolutionary_illusion_generator_amd/$l; else unset EIGEN_HIP_LIB; fi
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-supplementary 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('${l:-default}', {(o['layer'],o['op']):o['ms'] for o in d['roofline']['per_op'] if o['layer']<1})"
done 2>&1 | tee $O/perop.txt
