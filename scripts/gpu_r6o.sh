#!/bin/bash
# Round 6, call (o): side-stream overlap outside the K loop (the first layer's product beside the packing launch; the once-per-solve Gram
# products S_l / D_l beside the first iteration's chain): same-box A/B on the measurement build (debug key side_overlap), the product line,
# and the tests that would see a race (every-arm goldens, product goldens, solver forms, bit-reproducibility).
set -u
O=gpurun_out/r6o; mkdir -p $O
line() { python - "$1" "$2" <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); r = d["roofline"]
print("== %-28s %.1f steps/s %.4f ms (min-max %s) iter %.2f us outside %.3f ms" % (sys.argv[1], d["value"], d["ms_per_step"], ["%.4f" % v for v in d["regions"]["ms_per_step_min_max"]], r["avg_launch_us"], d["outside_k_loop_ms"]))
PY
}
run() { tag=$1; shift; timeout 400 python bench.py --cpu-steps 0 --no-parity "$@" 2> $O/bench_$tag.err > $O/bench_$tag.json; line $tag $O/bench_$tag.json; }
for rep in 1 2; do
  run ab_overlap_on_$rep --ab-lib
  run ab_overlap_off_$rep --debug side_overlap=0
done
run product_1
run product_2
run neumann_on --algo neumann --cg-iters 10 --ab-lib
run neumann_off --algo neumann --cg-iters 10 --debug side_overlap=0
timeout 1500 python -m pytest tests/test_cfg2_goldens.py tests/test_shapes_goldens.py -m gpu -q -x > $O/pytest_goldens.log 2>&1; echo "goldens rc=$?"; tail -2 $O/pytest_goldens.log
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "hoisted_and_projected or fused_solver_matches or widths_that or right_hand_side or packed_prepare" > $O/pytest_forms.log 2>&1; echo "forms rc=$?"; tail -2 $O/pytest_forms.log
