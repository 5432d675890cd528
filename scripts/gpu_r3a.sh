#!/bin/bash
# Round 3, call A: the new parity tests at full size (reference-CPU goldens, every arm), the reference's Engine over the HIP
# backend, fused-solve tokens / wide head / cfg 4 as named, hipGraph HVP probes, three bench lines.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
python -c "import torch; print(torch.cuda.get_device_name(0))"
timeout 900 python -m pytest tests/test_cfg2_goldens.py -m gpu -q -s 2>&1 | grep -E "cfg2 |passed|failed|Error|assert" > $O/r3a_cfg2_goldens.log; tail -3 $O/r3a_cfg2_goldens.log
timeout 600 python -m pytest tests/test_dropin_reference.py -m gpu -q 2>&1 | tail -3 | tee $O/r3a_dropin.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "token or wide_head or roberta_base_as_named" 2>&1 | grep -E "wide head|Roberta|passed|failed|Error|assert" | tee $O/r3a_new_tests.log
for c in cfg2 cfg3 cfg3prox; do timeout 400 python scripts/hvp_graph_probe.py $c 5 > $O/r3a_graph_$c.log 2>&1; echo "probe $c rc=$?"; tail -4 $O/r3a_graph_$c.log; done
timeout 900 python scripts/hvp_graph_probe.py cfg5 1 > $O/r3a_graph_cfg5.log 2>&1; echo "probe cfg5 rc=$?"; tail -4 $O/r3a_graph_cfg5.log
run() { tag=$1; shift
  timeout 400 python bench.py "$@" 2> $O/r3a_bench_$tag.err > $O/r3a_bench_$tag.json
  python - <<PY
import json
try:
    d=json.loads(open("$O/r3a_bench_$tag.json").read().strip().splitlines()[-1])
    r=d["roofline"] or {}; h=d["hvp_roofline"] or {}
    print("== %-16s value %.1f steps/s ms/step %.3f iter_us %.1f (events %.1f) frac28N %.3f hvp_frac %.3f outside_ms %.3f" % ("$tag", d["value"], d["ms_per_step"], d.get("per_iteration_us") or 0, r.get("avg_launch_us_hip_events") or 0, r.get("frac") or 0, h.get("frac") or 0, d.get("outside_k_loop_ms") or 0))
except Exception as e:
    print("== $tag bench failed:", e); print(open("$O/r3a_bench_$tag.err").read()[-1500:])
PY
}
run default --cpu-steps 0
run autograd_graph --hvp autograd --steps 40 --cpu-steps 0
BHG_HVP_GRAPH=0 run autograd_eager --hvp autograd --steps 40 --cpu-steps 0
run neumann --algo neumann --cg-iters 10 --cpu-steps 0
