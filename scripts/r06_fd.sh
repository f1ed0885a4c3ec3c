#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
O=gpurun_out/${1:-r06_p}; mkdir -p $O
scripts/_timing/fastdiv_check | tee $O/fastdiv_check.txt
L=$GRAFT_REPO_ROOT/evolutionary_illusion_generator_amd/libeigen_fd.so
EIGEN_HIP_LIB=$L timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -x -q -k "det_math or rollout_frames or winograd_operators_frames_bit_exact and None or 256 or eval_population" > $O/pytest_fd.log 2>&1; tail -4 $O/pytest_fd.log
ARGS="--steps 4" bash scripts/ab_libs.sh $O/ab libeigen_fd.so 2>&1 | tee $O/ab_libs.txt
