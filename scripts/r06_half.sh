#!/bin/bash
# round 6: the half (8 x 32, six-wave) block shape of the F(4x4) kernel -- parity with the shape forced on / off and as chosen by launch size, then same-box A/Bs per shape
# (recipe kept as the record of how profiles/r06_u_* were taken)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
O=gpurun_out/${1:-r06_u}; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -x -q -k "winograd_operators or specialised or rollout_frames" > $O/pytest_half.log 2>&1; tail -6 $O/pytest_half.log
EIGEN_W4_TALL=0 EIGEN_W4_HALF=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rollout_frames or 256 or 512 or 640" > $O/pytest_half1.log 2>&1; echo "EIGEN_W4_HALF=1: $(tail -1 $O/pytest_half1.log)"
for rep in 1 2; do for sh in ${SHAPES_AB:-c1 c2 ref160 ref640 headline}; do for t in 0 ""; do
  [ -n "$t" ] && export EIGEN_W4_HALF=$t || unset EIGEN_W4_HALF
  python bench.py --shape $sh --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-supplementary --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$sh', 'EIGEN_W4_HALF=${t:-default}', round(d['value'],1))"
done; done; done 2>&1 | tee $O/shapes.txt
