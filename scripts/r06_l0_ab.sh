#!/bin/bash
# round 6: image layer -- lstm0_direct_kernel with the four gate chains side by side + the one-block 2x2 pass not storing unused columns: parity, then same-box A/B
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
O=gpurun_out/${1:-r06_e}; mkdir -p $O
L=$GRAFT_REPO_ROOT/evolutionary_illusion_generator_amd/${2:-libeigen_l0.so}
EIGEN_HIP_LIB=$L timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -x -q -k "rollout or specialised or conv_chain or eval_population or smoke or api" > $O/pytest_l0.log 2>&1; tail -4 $O/pytest_l0.log
ARGS="--steps 4" bash scripts/ab_libs.sh $O/ab ${2:-libeigen_l0.so} 2>&1 | tee $O/ab_libs.txt
for sh in c2 ref160; do for lib in "" $L; do EIGEN_HIP_LIB=$lib python bench.py --shape $sh --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-supplementary --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$sh', '${lib:-default}'[-16:], round(d['value'],1))"; done; done 2>&1 | tee $O/shapes.txt
EIGEN_HIP_LIB=$L python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-supplementary 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print({(o['layer'],o['op']):o['ms'] for o in d['roofline']['per_op'] if o['layer']<2})"
