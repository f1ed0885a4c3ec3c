#!/bin/bash
# round 6: N-split half blocks (twelve waves, two N-tiles each) against the six-wave ones (libeigen_ns0.so = this tree built with -DEIG_W4_NSPLIT=0) -- parity of the
# product library, then same-box A/Bs per shape (recipe kept as the record of how profiles/r06_y_* were taken)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
O=gpurun_out/${1:-r06_y}; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -x -q -k "winograd_operators or specialised or rollout_frames or end_to_end" > $O/pytest_nsplit.log 2>&1; tail -6 $O/pytest_nsplit.log
for rep in 1 2; do for sh in ${SHAPES_AB:-c1 c2 ref160 headline}; do for lib in ${LIBS_AB:-libeigen_ns0.so} ""; do
  [ -n "$lib" ] && export EIGEN_HIP_LIB=$GRAFT_REPO_ROOT/evolutionary_illusion_generator_amd/$lib || unset EIGEN_HIP_LIB
  python bench.py --shape $sh --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-supplementary --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$sh', '${lib:-product}', round(d['value'],1))"
done; done; done 2>&1 | tee $O/shapes.txt
