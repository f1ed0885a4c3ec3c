cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04_i; mkdir -p $O
python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest.log 2>&1; tail -14 $O/pytest.log
( time python bench.py --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err ) 2>&1 | tail -3
python - <<PY
import json
d=json.loads(open("$O/bench_full.json").read().strip().splitlines()[-1])
print("value", d["value"], "lstm", d["roofline"]["frac"], "allconv", d["roofline"]["all_conv_kernels"]["frac"], "parity_fail", d.get("parity_fail"), d.get("parity_fail_reasons"))
c=d["cpu_baseline"]; print("cpu", {k: c.get(k) for k in ("value","value_timed_loop","value_from_thread_sweep","cores","thread_sweep_prednet_s_per_genome","prednet_gflops")}, "whole_host", c.get("whole_host"), d.get("gpu_over_cpu"), d.get("gpu_over_cpu_whole_host"))
print("sup", {k: (v.get("value"), v.get("all_conv_frac"), v.get("error")) for k, v in d["supplementary"].items()})
s=d["parity_check"]["population_vs_reference_order"]
print({k:v for k,v in d["parity_check"].items() if k!='population_vs_reference_order'})
print({k:v for k,v in s.items() if k not in('outside_1e-4_detail','cliff_rels','flips_of_cliff_genomes','control','control_cpu')})
print(s.get("control")); print(s.get("control_cpu"))
PY
