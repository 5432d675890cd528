#!/bin/bash
# profiles/r04_* after the round's last GPU calls.  Three libraries appear (config.lib_sha256 of every bench line says which):
#   1b038e12  first full pass (scripts/gpu_r4_full.sh): seven launches per iteration, 661.5 steps/s   -> files *_pass1*, PMC, SQ, opaque, probes
#   80b4f0d8 / e861d9e8  k_wskpl + k_wskpu (scripts/gpu_r4f.sh last call, scripts/gpu_r4g.sh): six launches, 701-702 steps/s -> the r04 files
#   d1a20ddf  a later edit that lost one host line (update blocks' arguments): every arm that launches k_wskpl crashed; the arms that do
#             not (lin_first=0, packed_chain=0, rnew_in_graw=0, Neumann, keep-solution) are valid and kept, labelled
# The shipped library (3d11e78e) = e861d9e8 + two host-side conditions in hoist_plan (four-layer nets only for the linear first
# product; projected forms only below 2^20 float4s per batch-sized array); device code byte-identical.
set -u
c() { [ -f "$1" ] && cp "$1" "$2"; }
p=profiles
for f in bench_default.json bench_default_kernel_stats.csv bench_line_under_rocprof.json timeline_fused_fully_projected.txt stamps_k_graw_k_pstep.txt ab_same_box_lines.txt bench_cg_default_again.json; do
  b=${f%.*}; e=${f##*.}; [ -f $p/r04_$f ] && git mv -f $p/r04_$f $p/r04_pass1_$f 2>/dev/null
done
c gpurun_out/r4g/bench_default.json $p/r04_bench_default.json
grep -E "passed|failed|^[0-9.]+s (call|setup)|slowest" gpurun_out/r4g/pytest_gpu_full.log > $p/r04_pytest_gpu.log
c gpurun_out/r4g/smoke.log $p/r04_smoke.log
c gpurun_out/r4f/timeline_default.txt $p/r04_timeline_fused_fully_projected.txt
c gpurun_out/r4/timeline_lin0.txt $p/r04_timeline_kpstep_launch_arm.txt
c gpurun_out/r4/timeline_unpacked.txt $p/r04_timeline_round3_product_same_box.txt
grep -v "k_wskpl tiles\|^   tiles  *n= *2[0-9][0-9]  entry 1[0-9][0-9][0-9]" gpurun_out/r4f/stamps_default.txt > $p/r04_stamps_k_graw_update_blocks.txt
for t in default default_again upd_in_first upd_in_first_again lin0 neumann; do c gpurun_out/r4f/bench_$t.json $p/r04_bench_k_wskpl_$t.json; done
for t in cg_kpstep_launch cg_graw_stores_raw cg_round3_product neumann_fused neumann_fused_again neumann_round3_product cg_keep_solution; do c gpurun_out/r4/bench_$t.json $p/r04_bench_$t.json; done
python - <<'PY'
import json, glob
def line(f, tag):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception:
        return None
    return "%-44s %8.1f steps/s  %7.3f ms/step  iteration %s us  lib %s  arms=%s" % (tag, d["value"], d["ms_per_step"],
            ("%.1f" % d["per_iteration_us"]) if d.get("per_iteration_us") else "-", d["config"].get("lib_sha256"), d["config"].get("debug_arms"))
rows = ["# same-box A/B, scripts/gpu_r4f.sh (last call): k_wskpl default vs its arms"]
for t in ("default", "default_again", "upd_in_first", "upd_in_first_again", "lin0", "neumann"):
    rows.append(line("gpurun_out/r4f/bench_%s.json" % t, t))
rows.append("# scripts/gpu_r4g.sh: the default line with parity object and CPU baseline (the GPU suite ran in the same call)")
rows.append(line("gpurun_out/r4g/bench_default.json", "default (full line)"))
rows.append("# scripts/gpu_r4_final.sh on library d1a20ddf (see the header of scripts/collect_profiles_r04_final.sh): arms that do not launch k_wskpl")
for t in ("cg_kpstep_launch", "cg_graw_stores_raw", "cg_round3_product", "neumann_fused", "neumann_fused_again", "neumann_round3_product", "cg_keep_solution"):
    rows.append(line("gpurun_out/r4/bench_%s.json" % t, t))
open("profiles/r04_ab_same_box_lines.txt", "w").write("\n".join(r for r in rows if r) + "\n")
print("\n".join(r for r in rows if r))
PY
