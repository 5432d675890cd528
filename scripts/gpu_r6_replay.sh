#!/bin/bash
# Round 6: the driver's command with profiles/r06_pmc_traffic.json committed — `roofline.traffic` must be replayed (sha-matched).
set -u
O=gpurun_out/r6replay; mkdir -p $O
sha256sum betty_amd/csrc/libbhg.so | tee $O/lib.sha
for i in a b; do timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 2> $O/bench_$i.err > $O/bench_$i.json; python -c "
import json; d=json.loads(open('$O/bench_$i.json').read().strip().splitlines()[-1]); r=d['roofline']; print('bench', d['value'], r['avg_launch_us'], r['frac'], r['traffic'], r['own']['frac_of_own_floor'], d['parity']['metric_instance']['vs_reference_fp64'], d['ranks_seen'])"; done
