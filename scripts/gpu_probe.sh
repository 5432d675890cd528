#!/bin/bash
set -u
export TMPDIR=/tmp; mkdir -p gpurun_out
for arm in "base" "nostore BHG_DEBUG_GEMM=1" "nomfma BHG_DEBUG_GEMM=2"; do
  set -- $arm; tag=$1; shift
  cd /tmp && env "$@" timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/gp_$tag -o t -- python $GRAFT_REPO_ROOT/scripts/gemm_probe.py 768 3072 12288 > /tmp/gp_$tag.log 2>&1; echo "probe $tag rc=$?"
  cd $GRAFT_REPO_ROOT
  f=$(ls /tmp/gp_$tag/*kernel_trace.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then python scripts/print_gemm_probe.py $f 768 3072 12288; else tail -5 /tmp/gp_$tag.log; fi
done
