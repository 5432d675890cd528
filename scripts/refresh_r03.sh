#!/bin/bash
# Local driver of the round-3 refresh: CLEAN build first (BHG_HIP_CHECK embeds __LINE__, so an incrementally built library can
# differ from what build() produces from the same sources — and profiles/r03_pmc_traffic.json is stamped with the sha256 of the
# library that ran), then the GPU run, then the copy into profiles/.
set -eu
cd "$(dirname "$0")/.."
make -C betty_amd/csrc clean >/dev/null
python -c "import __graft_entry__ as g; g.build()"
sha256sum betty_amd/csrc/libbhg.so | cut -c1-16
/usr/local/graft/bin/gpurun --timeout 2400 -- 'bash scripts/gpu_r3_full.sh' | tail -40
bash scripts/collect_profiles_r03.sh
python -c "import bench; print(bench.pmc_traffic('cg_iter_fused', 10034826))"
