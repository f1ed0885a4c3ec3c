#!/bin/bash
# round 6: walking F(4x4) kernel, layout v2 (two-round exchange outside the prefetched area) -- bit-exactness, same-box A/B against the round-5 kernel, per-wave timeline
# (recipe kept as the record of how profiles/r06_c_* were taken: libeigen_base.so = the round-5 tree + the ABI-4 export, built with `python __graft_entry__.py --lib`;
#  scripts/_timing/libeigen_timing.so = this tree with -DEIG_TIMING=1)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
O=gpurun_out/${1:-r06_c}; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "winograd_operators or specialised" > $O/pytest_wino.log 2>&1; tail -3 $O/pytest_wino.log
for p in 1 99; do EIGEN_W4_PARTS=$p timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "winograd_operators_frames_bit_exact and None" > $O/pytest_parts$p.log 2>&1; echo "EIGEN_W4_PARTS=$p: $(tail -1 $O/pytest_parts$p.log)"; done
ARGS="--steps 4" bash scripts/ab_libs.sh $O/ab libeigen_base.so 2>&1 | tee $O/ab_libs.txt
REPS=1 bash scripts/ab_env.sh $O/abenv "EIGEN_W4_PARTS=99" "EIGEN_W4_PARTS=1" "" "EIGEN_HIP_LIB=$GRAFT_REPO_ROOT/evolutionary_illusion_generator_amd/libeigen_base.so" 2>&1 | tee $O/ab_env.txt
EIG_TL_WAVES=12 EIGEN_TIMELINE_ALL=1 EIGEN_TIMELINE=$O/tl python scripts/timeline_wino4.py > $O/timeline_walk.txt 2>&1; grep -E "^==|MATRIX|per wave" $O/timeline_walk.txt | cut -c1-260
rm -f $O/tl/*.bin
