"""cfg 5 as named (the reference's Network(16, 10, 8), batch 64): where does an opaque HVP's time go on the MI355X?
One mode per process:  default | nocudnn (torch.backends.cudnn.enabled = False: ATen's own depthwise / im2col convolutions, no MIOpen
solver look-up, no first-use kernel compiles) | benchfast (cudnn.benchmark = True under MIOPEN_FIND_MODE=FAST).
fwdrev: the SAME Hessian-vector product by forward-over-reverse differentiation (torch.autograd.forward_ad dual parameters through
loss + backward: one pass per product, no double-backward graph — ATen's _convolution_double_backward loops over the GROUPS of a grouped
convolution, 16-64 tiny convolutions per depthwise layer of this supernet, which is what makes its double backward host-bound).
Prints the time of the loss + gradient-with-graph and of each of N HVPs, and the norms of the HVP (so modes can be compared)."""
import os
import sys
import time
import types

mode = sys.argv[1] if len(sys.argv) > 1 else "default"
n_hvp = int(sys.argv[2]) if len(sys.argv) > 2 else 2
if mode == "benchfast":
    os.environ["MIOPEN_FIND_MODE"] = "FAST"
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAS = os.path.join(ROOT, "oracle", "_ref", "examples_nas")
stub = types.ModuleType("utils")
stub.accuracy = lambda output, target, topk=(1,): [torch.zeros(()) for _ in topk]
sys.modules["utils"] = stub
sys.path.insert(0, NAS)
import model_search as ms  # noqa: E402

if mode == "nocudnn":
    torch.backends.cudnn.enabled = False
elif mode == "benchfast":
    torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(5)
torch.manual_seed(5)
inner = ms.Network(16, 10, 8, torch.nn.CrossEntropyLoss()).to(dev)
upper = ms.Architecture(4).to(dev)
x = torch.randn(64, 3, 32, 32, generator=g).to(dev)
y = torch.randint(0, 10, (64,), generator=g).to(dev)
params = list(inner.parameters())
vec = [1e-2 * torch.randn(p.shape, generator=g).to(dev) for p in params]
if mode == "fwdrev":
    import torch.autograd.forward_ad as fwAD
    from torch.nn.utils import stateless

    names = [n for n, _ in inner.named_parameters()]
    for i in range(n_hvp + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with fwAD.dual_level():
            duals = [fwAD.make_dual(p.detach().requires_grad_(True), v) for p, v in zip(params, vec)]
            with stateless._reparametrize_module(inner, dict(zip(names, duals))):
                loss = inner.loss(x, upper(), y)
            g = torch.autograd.grad(loss, duals)
            hv = [fwAD.unpack_dual(t).tangent for t in g]
        t_enq = time.perf_counter() - t0
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        nrm = float(torch.sqrt(sum((h.double() ** 2).sum() for h in hv)))
        print(f"[{mode}] loss + grad + HVP {i}: {t:.2f} s (enqueue {t_enq:.2f} s)  |Hv| = {nrm:.9e}", flush=True)
    sys.exit(0)
torch.cuda.synchronize()
t0 = time.perf_counter()
loss = inner.loss(x, upper(), y)
grads = torch.autograd.grad(loss, params, create_graph=True)
torch.cuda.synchronize()
print(f"[{mode}] loss + grad-with-graph: {time.perf_counter() - t0:.2f} s  loss={float(loss):.6f}", flush=True)
for i in range(n_hvp):
    t0 = time.perf_counter()
    hv = torch.autograd.grad(grads, params, grad_outputs=vec, retain_graph=True)
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    nrm = float(torch.sqrt(sum((h.double() ** 2).sum() for h in hv)))
    print(f"[{mode}] HVP {i}: {t:.2f} s (enqueue {t_enq:.2f} s)  |Hv| = {nrm:.9e}", flush=True)
