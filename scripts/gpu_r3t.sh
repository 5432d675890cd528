#!/bin/bash
# round 3: k per workgroup of the K-split Gram products
mkdir -p gpurun_out/r3t
cd "$GRAFT_REPO_ROOT"
run() { tag=$1; shift; timeout 400 python bench.py --cpu-steps 0 "$@" 2> gpurun_out/r3t/bench_$tag.err > gpurun_out/r3t/bench_$tag.json; python -c "
import json
d=[json.loads(l) for l in open('gpurun_out/r3t/bench_$tag.json') if l.startswith('{')][-1]; print('== %-14s %.1f steps/s  %.3f ms/step  iter_us %s' % ('$tag', d['value'], d['ms_per_step'], d.get('per_iteration_us')))" 2>&1 | tail -1; }
for c in 256 384 512 768 1024; do BHG_GRAM_KCHUNK=$c run chunk$c; done
run default_again
