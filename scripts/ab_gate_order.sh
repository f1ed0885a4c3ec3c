#!/bin/bash
# Same-box A/B of the canonical gate epilogue: EIG_GATE_ORDER=1 (the reference's knowable element-wise order,
# the default build) against EIG_GATE_ORDER=0 (rounds 1-3), each as a matching pair of HIP library + C oracle:
#   python __graft_entry__.py --lib evolutionary_illusion_generator_amd/libeigen_hip_gate0.so -DEIG_GATE_ORDER=0;  make -C oracle gate0
# For each: bench.py's parity leg (genome 0 bit-exact vs ITS oracle; all 256 genomes classified against the reference-order
# implementations; the reference-order A-vs-B control) -> OUT/gate{0,1}.json and a one-line digest.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$1; mkdir -p $OUT
for g in 1 0; do
  if [ $g = 0 ]; then export EIGEN_HIP_LIB=$R/evolutionary_illusion_generator_amd/libeigen_hip_gate0.so EIG_ORACLE_LIB=$R/oracle/libeig_oracle_gate0.so; else unset EIGEN_HIP_LIB EIG_ORACLE_LIB; fi
  python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-supplementary 2>$OUT/gate$g.err | tail -1 > $OUT/gate$g.json
  python - <<PY
import json
d=json.loads(open("$OUT/gate$g.json").read())
p=d["parity_check"]; s=p["population_vs_reference_order"]
print("gate_order %s: %.2f evals/s lstm %.4f | genome0 rel_err %.2g | vs matmul: within %d outside %d unexplained %d flip rate %.3g identical %d | vs MIOpen: %s | control A-B: %s | cpu control: %s" % (
    p["gate_order"], d["value"], d["roofline"]["frac"], p["rel_err"], s["within_1e-4"], s["outside_1e-4"], s["outside_1e-4_unexplained"], s["byte_flip_rate"], s["identical_frames"],
    s.get("vs_second_reference_order_implementation"), {k: v for k, v in s.get("control", {}).items() if k in ("control_within_1e-4", "control_outside_1e-4", "control_byte_flip_rate", "control_identical_frames", "error")},
    {k: v for k, v in s.get("control_cpu", {}).items() if k in ("control_within_1e-4", "control_outside_1e-4", "control_byte_flip_rate", "hip_vs_cpu_reference_order")}))
PY
done
