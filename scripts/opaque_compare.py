#!/usr/bin/env python
"""Product vs the reference's algorithm on the SAME GPU for the configurations whose inner loss stays OPAQUE (autograd double
backward; no declared structure): what this backend buys where only the recurrence / vector work is ours.

  cfg 2, opaque   MLP 3072-2048-1536-384-10 (N = 10,034,826, 8 tensors), CG K = 20          bench.build
  cfg 3           ResNet-12 (10.43 M parameters, 122 tensors, 25 x 3 x 84 x 84), CG K = 20   tests/zoo.py
  cfg 4           RobertaForSequenceClassification (124.6 M parameters, 201 tensors, 16 x 50 tokens), darts
  cfg 5           Network(16, 10, 8) as named: scripts/cfg5_oracle_on_gpu.py (round 5: 37 s per step with forward-over-reverse HVP passes
                  against 368 s for the reference's algorithm, profiles/r05_cfg5_product_vs_reference_algorithm_on_gpu.txt)

"reference algorithm" = oracle/hypergrad_oracle.py (the line-for-line restatement of cg.py / darts.py, pinned bit-for-bit to the
reference) on the same device tensors — i.e. what leopard-ai/betty itself launches on this GPU.  Prints steps/s of both."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import hypergrad_oracle as horc
import zoo
import bench
from betty_amd import Config, hypergradient as hg

dev = "cuda:0"


def rate(fn, n, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return n / (time.perf_counter() - t0)


# cfg 2, opaque
curr, prev, vector = bench.build(torch.device(dev), 0, K=20, algo="cg")
ours = rate(lambda: hg.cg(vector, curr, prev, False), 20)
ref = rate(lambda: horc.cg(vector, curr, prev, False), 20)
curr.hypergradient_graph = "persistent"
ours_g = rate(lambda: hg.cg(vector, curr, prev, False), 20, warm=4)
print(f"cfg 2 opaque  CG-20 : betty_amd {ours:7.2f} steps/s (persistent HIP-graph replay, opt-in: {ours_g:.2f}) | reference algorithm on this GPU {ref:7.2f} steps/s | x{ours / ref:.2f}")
del curr, prev, vector

# cfg 3
g = torch.Generator().manual_seed(77); torch.manual_seed(77)
inner, upper = zoo.ResNet12().to(dev), zoo.ResNet12().to(dev)
for p, q in zip(inner.parameters(), upper.parameters()):
    q.data.copy_(p.data + 0.05 * torch.randn(p.shape, generator=g).to(dev))
x = torch.randn(25, 3, 84, 84, generator=g).to(dev); y = torch.arange(5).repeat_interleave(5).to(dev)
vector = [0.01 * torch.randn(p.shape, generator=g).to(dev) for p in inner.parameters()]
prev = zoo.StubProblem("upper", upper, config=Config())
curr = zoo.StubProblem("inner", inner, config=Config(type="cg", cg_iterations=20), loss_fn=zoo.make_imaml_loss(prev, 0.5), batch=(x, y))
ours = rate(lambda: hg.cg(vector, curr, prev, False), 5)
ref = rate(lambda: horc.cg(vector, curr, prev, False), 5)
print(f"cfg 3 ResNet-12 CG-20 : betty_amd {ours:7.2f} steps/s | reference algorithm on this GPU {ref:7.2f} steps/s | x{ours / ref:.2f}")
# the opt-in forward-over-reverse passes (round 5; pays for grouped convolutions, cfg 5 — measured here on a dense-convolution net)
want = torch.cat([t.reshape(-1) for t in hg.cg(vector, curr, prev, False)]).double()
curr.hypergradient_hvp = "forward_over_reverse"
ours_f = rate(lambda: hg.cg(vector, curr, prev, False), 5)
got = torch.cat([t.reshape(-1) for t in hg.cg(vector, curr, prev, False)]).double()
print(f"cfg 3 ResNet-12 CG-20 : betty_amd with forward-over-reverse HVP passes {ours_f:7.2f} steps/s (x{ours_f / ref:.2f} over the reference's algorithm; "
      f"vs the double-backward product {float((got - want).norm() / want.norm()):.2e})")
curr.hypergradient_hvp = None
del inner, upper, curr, prev, vector
torch.cuda.empty_cache()

# cfg 4
try:
    g = torch.Generator().manual_seed(17); torch.manual_seed(17)
    inner, upper = zoo.RobertaInner().to(dev), zoo.MWN(500).to(dev)
    B, S = 16, 50
    batch = (torch.randint(3, 50264, (B, S), generator=g).to(dev), torch.ones(B, S, dtype=torch.long, device=dev),
             torch.zeros(B, S, dtype=torch.long, device=dev), torch.randint(0, 2, (B,), generator=g).to(dev))
    vector = [1e-3 * torch.randn(p.shape, generator=g).to(dev) for p in inner.parameters()]
    prev = zoo.StubProblem("upper", upper, config=Config())
    curr = zoo.StubProblem("inner", inner, config=Config(type="darts", darts_alpha=1.0), loss_fn=zoo.make_roberta_reweight_loss(prev), batch=batch)
    ours = rate(lambda: hg.darts(vector, curr, prev, False), 8)
    ref = rate(lambda: horc.darts(vector, curr, prev, False), 8)
    print(f"cfg 4 RoBERTa-base darts: betty_amd {ours:7.2f} steps/s | reference algorithm on this GPU {ref:7.2f} steps/s | x{ours / ref:.2f}")
except Exception as exc:   # transformers missing
    print("cfg 4 skipped:", repr(exc)[:200])
