#!/bin/bash
# A/B of the prefetching k_head_forward (BHG_HEAD_NO_PREFETCH turns it off): parity first, then bench lines + timeline.
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "fused or structured_goldens or cfg2 or power_of_two" 2>&1 | tail -3
run() { tag=$1; shift
  timeout 300 python bench.py --steps 100 --cpu-steps 0 --no-kernel-timing "$@" 2> gpurun_out/bench_$tag.err > gpurun_out/bench_$tag.json
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/bench_$tag.json").read().strip().splitlines()[-1])
    print("== %-28s value %.1f steps/s  ms/step %.3f" % ("$tag", d["value"], d["ms_per_step"]))
except Exception as e:
    print("== $tag bench failed:", e); print(open("gpurun_out/bench_$tag.err").read()[-800:])
PY
}
BHG_HEAD_NO_PREFETCH=1 run nopf
run pf
BHG_HEAD_NO_PREFETCH=1 run nopf_again
run pf_again
rm -rf /tmp/tr_d
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_d -o t -- python $GRAFT_REPO_ROOT/scripts/iter_trace.py 3 cg fused > /tmp/tr_d.log 2>&1)
f=$(find /tmp/tr_d -name '*kernel_trace.csv' | head -1)
if [ -n "$f" ]; then python scripts/print_iter_timeline.py $f k_cg_beta | grep -E "head_forward|iteration span"; else tail -3 /tmp/tr_d.log; fi
