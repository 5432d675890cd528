#!/bin/bash
# Round 5 — an arm that existed for one call, in the measurement build only (strips_first=1: strip tiles leading their problem's blocks; 57.9 vs 57.4 us,
# not kept, removed again: profiles/r05_ab_strips_first_not_kept.txt).  Kept as the record of that call; the key no longer exists.
set -u
O=gpurun_out/r5o; mkdir -p $O; export TMPDIR=/tmp
sha256sum betty_amd/csrc/libbhg.so betty_amd/csrc/libbhg_ab.so | tee $O/lib.sha
run() { tag=$1; shift; timeout 400 python bench.py --cpu-steps 0 --no-parity "$@" 2> $O/bench_$tag.err > $O/bench_$tag.json; python - <<PY
import json
try:
    d=json.loads(open('$O/bench_$tag.json').read().strip().splitlines()[-1]); r=d.get('roofline') or {}
    print('== %-28s %.1f steps/s  %.3f ms/step  iter_us %s (%s) frac %.3f outside %.3f ms  %s' % ('$tag', d['value'], d['ms_per_step'], '%.2f'%r['avg_launch_us'], ['%.2f'%v for v in (r.get('avg_launch_us_min_max') or [])], r['frac'], d['outside_k_loop_ms'] or 0, d['config']['lib'][:12]))
except Exception as e:
    print('== $tag unreadable', e, open('$O/bench_$tag.err').read()[-1500:])
PY
}
run cg_ab_defaults --ab-lib
run cg_strips_first --debug strips_first=1
run cg_ab_defaults_again --ab-lib
run cg_strips_first_again --debug strips_first=1
run neumann_ab_defaults --ab-lib --algo neumann --cg-iters 10
run neumann_strips_first --algo neumann --cg-iters 10 --debug strips_first=1
timeout 200 python -m pytest tests/test_cfg2_goldens.py -m gpu -x -q -k "metric_configuration" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -1 $O/pytest.log
