cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04_g; mkdir -p $O
for m in 0 2 3 4; do
  EIGEN_WINO_MODE=$m python -m pytest tests/test_gpu_parity.py -x -q -k "winograd" 2>&1 | tail -1
  EIGEN_WINOGRAD=14 EIGEN_WINO_MODE=$m python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-supplementary 2>$O/bench.err | tail -1 > $O/bench_m$m.json
  python - <<PY
import json
try:
    d=json.loads(open("$O/bench_m$m.json").read())
    r=d["roofline"]
    print("EIGEN_WINO_MODE=$m: %.2f evals/s  nonzero %d  conv ms %.1f  dominant %.4f of peak (executed), %.1f TF as direct conv" % (d["value"], d["nonzero_fitness"], r["all_conv_kernels"]["total_ms"], r["frac"], r["dominant_kernel_tflops_as_direct_convolution"]), [round(o["ms"],1) for o in r["per_op"] if o["op"]=="lstm" and o["layer"]>0])
except Exception as e: print("mode $m failed", e); print(open("$O/bench.err").read()[-1500:])
PY
done
