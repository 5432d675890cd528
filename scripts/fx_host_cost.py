import os, sys, time, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
import torch.distributed as dist
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
import bench
from betty_amd import hypergradient as hg
import betty_amd.global_hvp as gh
curr, prev, vector = bench.build(dev, 0, K=20, algo="cg")
bench.declare_structure(curr, "hip")
fn = hg.jvp_fn_mapping["cg_global"]
for gather in (False, True):
    gh.FX_ALWAYS_GATHER = gather
    for _ in range(5): fn(vector, curr, prev, False)
    torch.cuda.synchronize()
    host, total = [], []
    for _ in range(20):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fn(vector, curr, prev, False)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        host.append(t1 - t0); total.append(t2 - t0)
    med = lambda v: sorted(v)[len(v)//2]
    print(f"always_gather={gather}: host enqueue {med(host)*1e3:.3f} ms per solve, until the GPU is done {med(total)*1e3:.3f} ms")
dist.destroy_process_group()
