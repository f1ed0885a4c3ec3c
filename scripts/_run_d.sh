cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04_d; mkdir -p $O
python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest.log 2>&1; tail -14 $O/pytest.log
( time python bench.py --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err ) 2>&1 | tail -3
python - <<PY
import json
d=json.loads(open("$O/bench_full.json").read().strip().splitlines()[-1])
print("value", d["value"], "lstm", d["roofline"]["frac"], "allconv", d["roofline"]["all_conv_kernels"]["frac"], "parity_fail", d.get("parity_fail"))
print("cpu", {k: d["cpu_baseline"].get(k) for k in ("value","cores","thread_sweep_prednet_s_per_genome","prednet_gflops")}, "whole_host", d["cpu_baseline"].get("whole_host"), d.get("gpu_over_cpu"), d.get("gpu_over_cpu_whole_host"))
print("sup", {k: (v.get("value"), v.get("all_conv_frac"), v.get("error")) for k, v in d["supplementary"].items()})
PY
bash scripts/profile_run.sh r04_d > $O/profile_run.txt 2>&1; tail -15 $O/profile_run.txt
