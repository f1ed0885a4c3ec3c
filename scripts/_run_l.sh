cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04_l; mkdir -p $O
python -m pytest tests -m gpu -x -q --durations=6 > $O/pytest.log 2>&1; tail -10 $O/pytest.log
bash scripts/profile_run.sh r04_l > $O/profile_run.txt 2>&1; tail -14 $O/profile_run.txt
