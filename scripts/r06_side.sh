#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
O=gpurun_out/${1:-r06_q}; mkdir -p $O
export EIGEN_HIP_LIB=$GRAFT_REPO_ROOT/evolutionary_illusion_generator_amd/libeigen_side.so
timeout 900 python -m pytest tests/test_gpu_api.py -m gpu -x -q -k "specialised" > $O/pytest_side.log 2>&1; tail -3 $O/pytest_side.log
for sh in ref160 c2 c1 headline; do for g in 0 1 0 1; do
  EIGEN_SIDE_STREAM=$g python bench.py --shape $sh --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-supplementary --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$sh', 'EIGEN_SIDE_STREAM=$g', round(d['value'],1), 'nonzero', d.get('nonzero_fitness'))"
done; done 2>&1 | tee $O/shapes.txt
