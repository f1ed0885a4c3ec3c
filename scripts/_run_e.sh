cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04_e; mkdir -p $O
cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc; lscpu | grep -i "socket\|thread\|core\|model name" | head -6
python -m pytest tests/test_gpu_parity.py -x -q -k "winograd" > $O/pytest_wino.log 2>&1; tail -30 $O/pytest_wino.log
for v in 0 14 6 2; do
  EIGEN_WINOGRAD=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-supplementary 2>$O/bench_w$v.err | tail -1 > $O/bench_w$v.json
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_w$v.json").read())
    r=d["roofline"]
    print("EIGEN_WINOGRAD=$v: %.2f evals/s  nonzero %d  conv ms %.1f" % (d["value"], d["nonzero_fitness"], r["all_conv_kernels"]["total_ms"]))
    for o in r["per_op"]:
        if o["op"].startswith("lstm") and o["layer"]>0: print("   ", o)
except Exception as e: print("EIGEN_WINOGRAD=$v failed", e); print(open("$O/bench_w$v.err").read()[-1500:])
PY
done
