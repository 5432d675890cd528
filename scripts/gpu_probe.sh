#!/bin/bash
set -u
export TMPDIR=/tmp
for arm in "base" "cap12 BHG_SPLIT_CAP=12" ; do
  set -- $arm; tag=$1; shift
  cd /tmp && env "$@" timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/gp_$tag -o t -- python $GRAFT_REPO_ROOT/scripts/gemm_probe.py 768 1536 3072 6144 12288 > /tmp/gp_$tag.log 2>&1; echo "probe $tag rc=$?"
  cd $GRAFT_REPO_ROOT
  f=$(ls /tmp/gp_$tag/*kernel_trace.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then python scripts/print_gemm_probe.py $f; else tail -5 /tmp/gp_$tag.log; fi
done
