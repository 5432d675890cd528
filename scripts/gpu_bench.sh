#!/bin/bash
# bench.py (default flags) + rocprofv3 kernel stats of the same command.  Outputs -> gpurun_out/
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
TAG=${1:-auto}
shift || true
timeout 900 python bench.py "$@" 2> gpurun_out/bench_$TAG.err | tee gpurun_out/bench_$TAG.json | tail -2
tail -3 gpurun_out/bench_$TAG.err
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- python $GRAFT_REPO_ROOT/bench.py --cpu-steps 0 "$@" > /tmp/prof_$TAG.log 2>&1; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/prof_$TAG; cp /tmp/prof_$TAG/*stats*.csv gpurun_out/prof_$TAG/ 2>/dev/null
tail -1 /tmp/prof_$TAG.log | cut -c1-300 > gpurun_out/prof_$TAG/bench_line_under_rocprof.json
python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_$TAG/bench_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot/1e6)
for r in rows[:14]:
    print(f'{r["Name"][:90]:90s} calls={r["Calls"]:>6s} avg_us={float(r["AverageNs"])/1e3:9.2f} pct={r["Percentage"]}')
PY
