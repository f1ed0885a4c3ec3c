#!/bin/bash
# round 6: the walking F(4x4) kernel -- bit-exactness first, then the same-box A/B against the round-5 kernel (libeigen_base.so = HEAD of round 5 + the ABI-4 export)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
O=gpurun_out/${1:-r06_b}; mkdir -p $O
( time scripts/_timing/mfma_bf16_order 40000 ) > $O/mfma_bf16_order.txt 2>&1; grep -A28 "^best" $O/mfma_bf16_order.txt | cut -c1-160
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "winograd_operators or specialised or rollout or conv_chain" > $O/pytest_wino.log 2>&1; tail -6 $O/pytest_wino.log
for p in 1 2 99; do EIGEN_W4_PARTS=$p timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "winograd_operators_frames_bit_exact and None" > $O/pytest_parts$p.log 2>&1; echo "EIGEN_W4_PARTS=$p: $(tail -1 $O/pytest_parts$p.log)"; done
ARGS="--steps 4" bash scripts/ab_libs.sh $O/ab libeigen_base.so 2>&1 | tee $O/ab_libs.txt
REPS=1 bash scripts/ab_env.sh $O/abenv "EIGEN_W4_PARTS=99" "EIGEN_W4_PARTS=1" "EIGEN_W4_PARTS=2" "EIGEN_W4_PARTS=3" "" 2>&1 | tee $O/ab_env.txt
