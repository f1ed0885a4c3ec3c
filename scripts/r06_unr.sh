#!/bin/bash
# round 6: the walk as a lambda called from the loop (no spills) / as straight-line code for fixed walk lengths -- parity, same-box A/B
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
O=gpurun_out/${1:-r06_m}; mkdir -p $O
L=$GRAFT_REPO_ROOT/evolutionary_illusion_generator_amd/libeigen_unr.so
for u in 0 1; do EIGEN_TRY_UNROLL=$u EIGEN_HIP_LIB=$L timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "winograd_operators_frames_bit_exact or 256" > $O/pytest_u$u.log 2>&1; echo "EIGEN_TRY_UNROLL=$u: $(tail -1 $O/pytest_u$u.log)"; done
ARGS="--steps 4" bash scripts/ab_libs.sh $O/ab libeigen_unr.so 2>&1 | tee $O/ab_libs.txt
export EIGEN_HIP_LIB=$L
REPS=2 bash scripts/ab_env.sh $O/abenv "" "EIGEN_TRY_UNROLL=1" "EIGEN_W4_PARTS=2" "EIGEN_W4_PARTS=1" 2>&1 | tee $O/ab_env.txt
