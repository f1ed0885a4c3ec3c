cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04_o; mkdir -p $O
python -m pytest tests -m gpu -x -q --durations=6 > $O/pytest.log 2>&1; tail -10 $O/pytest.log
( time python bench.py --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err ) 2>&1 | tail -3
bash scripts/profile_run.sh r04_o > $O/profile_run.txt 2>&1; tail -6 $O/profile_run.txt
