#!/bin/bash
# rocprofv3 kernel stats of the default bench command + the bench line printed under the profiler.
set -u
export TMPDIR=/tmp; mkdir -p gpurun_out/prof_default
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_default -o bench -- python $GRAFT_REPO_ROOT/bench.py --cpu-steps 0 > /tmp/prof_default.log 2>&1; echo "rocprof stats rc=$?"
cd $GRAFT_REPO_ROOT; cp /tmp/prof_default/*kernel_stats*.csv gpurun_out/prof_default/ 2>/dev/null
grep "^{\"metric\"" /tmp/prof_default.log | tail -1 > gpurun_out/prof_default/bench_line_under_rocprof.json
head -c 300 gpurun_out/prof_default/bench_line_under_rocprof.json
