cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04_f; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -k "winograd" > $O/pytest_wino.log 2>&1; tail -4 $O/pytest_wino.log
EIGEN_WINO_STAGGER=0 python -m pytest tests/test_gpu_parity.py -x -q -k "winograd" > $O/pytest_wino0.log 2>&1; tail -4 $O/pytest_wino0.log
for cfg in "0 1" "14 1" "14 0" "14 1"; do
  set -- $cfg
  EIGEN_WINOGRAD=$1 EIGEN_WINO_STAGGER=$2 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-supplementary 2>$O/bench.err | tail -1 > $O/bench_w$1_s$2.json
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_w$1_s$2.json").read())
    r=d["roofline"]
    print("EIGEN_WINOGRAD=$1 STAGGER=$2: %.2f evals/s  nonzero %d  conv ms %.1f  dominant %.4f of peak (executed), %.1f TF as direct conv" % (d["value"], d["nonzero_fitness"], r["all_conv_kernels"]["total_ms"], r["frac"], r["dominant_kernel_tflops_as_direct_convolution"]))
    for o in r["per_op"]:
        if o["op"].startswith("lstm") and o["layer"]>0: print("   ", o)
except Exception as e: print("EIGEN_WINOGRAD=$1 failed", e); print(open("$O/bench.err").read()[-1500:])
PY
done
