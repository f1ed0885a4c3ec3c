#!/bin/bash
# round 6: the tall (32 x 16) block shape of the F(4x4) kernel -- parity with the shape forced on / off and as chosen by map size, then same-box A/Bs per shape
# (recipe kept as the record of how profiles/r06_j_* were taken; second argument: a library variant under evolutionary_illusion_generator_amd/, default libeigen_tall.so = this tree)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
O=gpurun_out/${1:-r06_j}; mkdir -p $O
L=$GRAFT_REPO_ROOT/evolutionary_illusion_generator_amd/${2:-libeigen_tall.so}
export EIGEN_HIP_LIB=$L
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -x -q -k "winograd_operators or specialised or rollout_frames" > $O/pytest_tall.log 2>&1; tail -6 $O/pytest_tall.log
EIGEN_W4_TALL=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rollout_frames or 256 or 512 or 640" > $O/pytest_tall1.log 2>&1; echo "EIGEN_W4_TALL=1: $(tail -1 $O/pytest_tall1.log)"
for sh in ref160 c2 c1 ref640; do for t in 0 "" 1; do
  [ -n "$t" ] && export EIGEN_W4_TALL=$t || unset EIGEN_W4_TALL
  python bench.py --shape $sh --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-supplementary --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$sh', 'EIGEN_W4_TALL=${t:-default}', round(d['value'],1))"
done; done 2>&1 | tee $O/shapes.txt
unset EIGEN_W4_TALL
REPS=1 bash scripts/ab_env.sh $O/abenv "EIGEN_W4_TALL=0" "" "EIGEN_W4_TALL=1" 2>&1 | tee $O/ab_env.txt
