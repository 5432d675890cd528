#!/bin/bash
# round 3: what bounds the G(raw) launch? timelines with the small slices' blocks in a launch of their own
mkdir -p gpurun_out/r3u; export TMPDIR=/tmp
for arm in default small_alone; do
  if [ $arm = small_alone ]; then export BHG_PROJ_SMALL_ALONE=1; else unset BHG_PROJ_SMALL_ALONE; fi
  cd /tmp && rm -rf /tmp/tr_$arm && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$arm -o t -- python $GRAFT_REPO_ROOT/scripts/iter_trace.py 3 cg fused > /tmp/tr_$arm.log 2>&1; echo "trace $arm rc=$?"
  cd $GRAFT_REPO_ROOT
  f=$(ls /tmp/tr_$arm/*kernel_trace.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python scripts/print_iter_timeline.py $f "k_proj_step" | tee gpurun_out/r3u/timeline_$arm.txt
done
