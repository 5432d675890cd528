#!/bin/bash
# Round 5, call 5 — (1) the abort at interpreter exit after smoke() seen in call 4: how often, and where (faulthandler); (2) the GPU
# suite with the printed result lines (-rP), now with cfg 5 as named against its reference-CPU golden and the deeper nets in the
# six-launch form; (3) two ranks on the one GPU over gloo (closed-form upper net + average_over under the DDP wrapper).
set -u
O=gpurun_out/r5e; mkdir -p $O; export TMPDIR=/tmp
sha256sum betty_amd/csrc/libbhg.so betty_amd/csrc/libbhg_ab.so | tee $O/lib.sha
for i in 1 2 3; do timeout 300 python -X faulthandler -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_RETURNED__')" > $O/smoke_$i.log 2>&1; echo "smoke $i rc=$?"; tail -4 $O/smoke_$i.log; done
for v in case_logreg_cg5 case_reweight_neumann10 case_deep_cg6 case_logreg_darts; do timeout 300 python -X faulthandler - > $O/smoke_$v.log 2>&1 <<PY
import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "tests"), os.path.join(os.getcwd(), "oracle")]
import torch, zoo
from betty_amd import Config
from betty_amd import hypergradient as hg
name = "$v"[5:]
case = zoo.CASE_BY_NAME[name]
inputs = zoo.seed_family_inputs(case.family)
c, p, v = zoo.build_case(case, inputs, Config, device="cuda:0")
out = hg.jvp_fn_mapping[case.algo](v, c, p, False)
torch.cuda.synchronize()
print("done", name)
PY
echo "$v rc=$?"; tail -2 $O/smoke_$v.log; done
timeout 1500 python -m pytest tests -m gpu -q -rP --durations=8 > $O/pytest_gpu_full.log 2>&1; echo "pytest rc=$?"; grep -vE "^Extension|Warning|warn" $O/pytest_gpu_full.log | tail -14
grep -E "withheld beta|packed prepare|update inside k_graw|cfg5 as named|supernet neumann20|resnet12 cg20" $O/pytest_gpu_full.log | head -30
BHG_ALL_RANKS_ON_GPU0=1 timeout 300 python bench.py --gpus 2 --dist-backend gloo --steps 40 --cpu-steps 0 2> $O/bench_selflaunch_2ranks_one_gpu_gloo.err > $O/bench_selflaunch_2ranks_one_gpu_gloo.json; echo "self-launch --gpus 2 rc=$?"; tail -c 600 $O/bench_selflaunch_2ranks_one_gpu_gloo.json; tail -5 $O/bench_selflaunch_2ranks_one_gpu_gloo.err
