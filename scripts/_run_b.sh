cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q --durations=12 > gpurun_out/r04_b_pytest.log 2>&1; tail -25 gpurun_out/r04_b_pytest.log
bash scripts/ab_gate_order.sh gpurun_out/r04_b_gate > gpurun_out/r04_b_gate.txt 2>&1; cat gpurun_out/r04_b_gate.txt
SHAPES="--steps@4 --pop@32@--steps@8 --shape@ref160@--steps@10 --shape@c2@--steps@20 --shape@ref640@--steps@4 --shape@ref640@--pop@4@--steps@8" bash scripts/ab_shapes.sh gpurun_out/r04_b_pipe "EIGEN_PIPE2=1" "EIGEN_PIPE2=1 EIGEN_PIPE2_SYNC=1" > gpurun_out/r04_b_pipe2.txt 2>&1; cat gpurun_out/r04_b_pipe2.txt
