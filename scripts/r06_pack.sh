#!/bin/bash
# round 6: packed tiles (half blocks whose sixteen MFMA rows are all real tiles of 20 x 15 / 16 x 16 maps; first as a linear tile list, then as main + edge blocks) -- parity with the
# form forced on / off and as chosen, then same-box A/Bs per shape (recipe kept as the record of how profiles/r06_w_* were taken)
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
O=gpurun_out/${1:-r06_w}; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_api.py -m gpu -x -q -k "winograd_operators or specialised or rollout_frames or end_to_end" > $O/pytest_pack.log 2>&1; tail -6 $O/pytest_pack.log
for rep in 1 2; do for sh in ${SHAPES_AB:-c2 ref160 c1 headline}; do for t in 0 ""; do
  [ -n "$t" ] && export EIGEN_W4_PACK=$t || unset EIGEN_W4_PACK
  python bench.py --shape $sh --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-supplementary --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$sh', 'EIGEN_W4_PACK=${t:-default}', round(d['value'],1))"
done; done; done 2>&1 | tee $O/shapes.txt
