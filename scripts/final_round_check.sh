#!/bin/bash
# What the end of a round runs on the GPU box: the GPU suite, the default bench line as the driver runs it, and one profiling round
# (bench + rocprofv3 kernel stats + PMC passes -> gpurun_out/prof_TAG, copy what is to be judged into profiles/).
# usage: scripts/final_round_check.sh TAG
cd ${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
TAG=${1:-final}
O=gpurun_out/$TAG; mkdir -p $O
python -m pytest tests -m gpu -x -q --durations=6 > $O/pytest.log 2>&1; tail -10 $O/pytest.log
( time python bench.py --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err ) 2>&1 | tail -3
bash scripts/profile_run.sh $TAG > $O/profile_run.txt 2>&1; tail -6 $O/profile_run.txt
