"""The caller slice of the hot path: a compact ``ImplicitProblem`` with the reference's surface.

Only what sits directly around the hypergradient call is restated here:
``backward`` (direct gradient, then one ``get_grads`` per path with the reference's ``do_sync`` /
``retain_graph`` rules — betty/problems/problem.py:521-581), ``set_grads`` (583-597),
``training_step_exec`` (327-332), ``one_step_descent`` (334-369), the unroll/gradient-accumulation
step schedule (371-415) and ``ImplicitProblem``'s parameter accessors and optimizer step
(betty/problems/implicit_problem.py:40-65,80-84).  Logging, validation, roll-back, LR schedulers'
warm-up, FSDP/accelerate and the iterative-differentiation problems stay in the reference: a Betty
user keeps using ``betty.Engine`` + ``betty_amd.install()``; this module exists so the package
(tests, examples, benchmarks) also runs where the reference is not installed.
"""
from __future__ import annotations

import copy

import torch

from .backend import get_backend
from .configs import Config
from .hypergradient import get_grads


def _to_device(obj, device):
    if torch.is_tensor(obj):
        return obj.to(device)
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_device(o, device) for o in obj)
    if isinstance(obj, dict):
        return {k: _to_device(v, device) for k, v in obj.items()}
    return obj


class Problem:
    """One level of a multilevel optimisation problem.  Subclasses define ``training_step(batch)``
    (and optionally ``on_inner_loop_start``, ``param_callback``, ``grad_callback``)."""

    def __init__(self, name, config=None, module=None, optimizer=None, scheduler=None, train_data_loader=None,
                 device=None):
        self._name = name
        self._config = config if config is not None else Config()
        self.module, self.optimizer, self.scheduler = module, optimizer, scheduler
        self.train_data_loader = train_data_loader
        self._iter = None
        self.device = device
        self.cur_batch = None
        self._parents, self._children, self._paths = [], [], []
        self.ready = []
        self._count = 0
        self._inner_loop_start = True
        self._training = True
        self._strategy = "default"
        self.leaf = False
        self._roll_back = False
        self._world_size = 1
        self._snapshot = None
        # flat asynchronous hypergradient exchange (Engine: strategy "distributed", large upper problems)
        self._flat_exchange = False
        self._pending_exchange = []
        # fp16 dynamic loss scaler (problem.py:165-174): created on first use so that a CPU-only import works
        self.scaler = None
        # forward module seen by other problems (a DDP wrapper under strategy "distributed")
        self.fwd = module

    # ---- identity / wiring (engine.py:232-291, problem.py:808-870) ------------------------------------
    @property
    def name(self):
        return self._name

    @property
    def config(self):
        return self._config

    @property
    def paths(self):
        return self._paths

    @property
    def children(self):
        return self._children

    @property
    def parents(self):
        return self._parents

    @property
    def count(self):
        return self._count

    def add_child(self, problem):
        assert problem not in self._children
        self._children.append(problem)
        self.ready = [False] * len(self._children)

    def add_parent(self, problem):
        assert problem not in self._parents
        self._parents.append(problem)

    def add_paths(self, paths):
        self._paths.extend(paths)

    def clear_dependencies(self):
        self._parents, self._children, self._paths, self.ready = [], [], [], []

    def set_problem_attr(self, problem):
        """Other problems are reachable as attributes by name (``self.inner(x)``), problem.py:792-806."""
        if self.__dict__.get(problem.name) is problem:
            return  # already wired (a second Engine over the same problems)
        if problem.name in self.__dict__ or hasattr(type(self), problem.name):
            raise ValueError(f"problem name {problem.name!r} clashes with an attribute")
        setattr(self, problem.name, problem)

    def __call__(self, *args, **kwargs):
        return self.forward(*args, **kwargs)

    def forward(self, *args, **kwargs):
        return self.fwd(*args, **kwargs)

    # ---- parameters (implicit_problem.py:80-84, problem.py:850-854) --------------------------------------
    def parameters(self):
        raise NotImplementedError

    def trainable_parameters(self):
        raise NotImplementedError

    def meta_trainable_parameters(self):
        return self.trainable_parameters()

    # ---- data (problem.py:456-487) --------------------------------------------------------------------------
    def get_batch(self):
        loader = self.train_data_loader
        if loader is None:
            return None
        if self._iter is None:
            self._iter = iter(loader)
        try:
            batch = next(self._iter)
        except StopIteration:
            self._iter = iter(loader)
            batch = next(self._iter)
        return _to_device(batch, self.device) if self.device is not None else batch

    # ---- loss / step (problem.py:327-369, 489-519) ----------------------------------------------------------
    def training_step(self, batch):
        raise NotImplementedError

    def training_step_exec(self, batch):
        precision = self._config.precision
        if precision in ("fp16", "bf16"):
            dtype = torch.float16 if precision == "fp16" else torch.bfloat16
            with torch.autocast(device_type="cuda", dtype=dtype):
                return self.training_step(batch)
        return self.training_step(batch)

    @property
    def gas(self):
        return self._config.gradient_accumulation

    def gradient_accumulation_boundary(self):
        return bool(self._count % self.gas == 0)

    def _ensure_scaler(self):
        """problem.py:165-174: ``precision="fp16"`` trains under a dynamic loss scaler built from
        ``Config.initial_dynamic_scale`` / ``Config.scale_factor``."""
        if self._config.precision == "fp16" and self.scaler is None:
            assert torch.cuda.is_available(), "fp16 training needs a GPU"
            self.scaler = torch.amp.GradScaler("cuda", init_scale=self._config.initial_dynamic_scale,
                                               growth_factor=self._config.scale_factor)
        return self.scaler

    def get_loss(self, batch):
        out = self.training_step_exec(batch)
        loss = out["loss"] if isinstance(out, dict) else out
        if self._ensure_scaler() is not None:   # problem.py:508-509: the SCALED loss is what gets differentiated
            loss = self.scaler.scale(loss)
        return loss / self.gas

    def one_step_descent(self, batch=None):
        if batch is None:
            self.cur_batch = self.get_batch()
            batch = self.cur_batch
        loss = self.get_loss(batch)
        self.backward(
            loss=loss,
            params=self.trainable_parameters(),
            paths=self._paths,
            create_graph=not self._config.first_order,
            retain_graph=self._config.retain_graph,
            allow_unused=self._config.allow_unused,
        )
        # flat exchange (gas == 1 by construction): the averaged pieces must sit in .grad BEFORE the user's grad_callback,
        # as they do in the reference (problem.py:356-360)
        self.finish_exchanges()
        if hasattr(self, "grad_callback"):
            self.grad_callback()
        if self.gradient_accumulation_boundary():
            self.optimizer_step()
            if hasattr(self, "param_callback"):
                self.param_callback()
            # problem.py:363-364: replicas whose weights are perturbed in place per rank (darts / sama) are
            # re-synchronised every 20 optimizer steps under the non-default strategies
            if self._strategy != "default" and self._count % (self.gas * 20) == 0:
                self.synchronize_params(self.trainable_parameters())
            self.zero_grad()
        return loss

    def backward(self, loss, params, paths, create_graph=False, retain_graph=True, allow_unused=True):
        """problem.py:521-581."""
        if self._flat_exchange:
            # Extension (EngineConfig.flat_exchange_min_params): every gradient piece of this step — the direct one and
            # one per path — is averaged over the ranks by ONE flat asynchronous all-reduce (RCCL over xGMI) issued as
            # soon as the piece exists; the exchange of path i overlaps with the CG solve of path i+1 (which then uses
            # the streaming CG kernels: the resident one needs every CU).  The averaged pieces are accumulated into
            # .grad just before the optimizer step.
            from .distributed import exchange_async  # noqa: PLC0415

            grads = torch.autograd.grad(loss, params, create_graph=create_graph, retain_graph=retain_graph or len(paths) > 0,
                                        allow_unused=allow_unused)
            try:
                self._queue_exchange(params, grads, exchange_async)
                if self._config.first_order:
                    last = len(paths) - 1
                    for idx, path in enumerate(paths):
                        self._queue_exchange(params, get_grads(loss, path, retain_graph=(idx != last), do_sync=False), exchange_async)
            except BaseException:
                # a later path raised: retire the exchanges already in flight (every rank runs the same program, so the
                # collectives themselves still match up) so the collectives-in-flight count returns to zero
                for _, handle in self._pending_exchange:
                    handle.wait()
                self._pending_exchange = []
                raise
            return
        # direct gradient: through autograd.grad + set_grads while a hypergradient (or another
        # accumulation step) is still to come, through backward() (DDP-syncing) otherwise
        if len(paths) > 0 or not self.gradient_accumulation_boundary():
            grads = torch.autograd.grad(loss, params, create_graph=create_graph, retain_graph=retain_graph,
                                        allow_unused=allow_unused)
            self.set_grads(params, grads)
        else:
            torch.autograd.backward(loss, inputs=params, create_graph=create_graph, retain_graph=retain_graph)
        # indirect gradient through the lower levels' best responses
        if self._config.first_order:
            last = len(paths) - 1
            for idx, path in enumerate(paths):
                do_sync = bool(idx == last and self.gradient_accumulation_boundary())
                grads = get_grads(loss, path, retain_graph=(idx != last), do_sync=do_sync)
                if not do_sync:
                    self.set_grads(params, grads)

    def _queue_exchange(self, params, grads, exchange_async):
        # EVERY rank exchanges the FULL parameter layout: a parameter that is unused on this rank (data-dependent
        # branch, sampled op) contributes zeros, exactly what DDP(find_unused_parameters=True) does in the reference —
        # otherwise the ranks would issue all-reduces of different sizes
        params = list(params)
        if params:
            full = [g if g is not None else torch.zeros_like(p) for p, g in zip(params, grads)]
            self._pending_exchange.append((params, exchange_async(full)))

    def finish_exchanges(self):
        """Wait (on the stream, not the host) for the flat exchanges of this step and accumulate the averages."""
        for plist, handle in self._pending_exchange:
            self.set_grads(plist, [t.clone() for t in handle.wait()])
        self._pending_exchange = []

    def set_grads(self, params, grads):
        """problem.py:583-597: accumulate ``grads`` into ``.grad`` (assign where there is none), skip
        None.  When many tensors already hold a gradient (e.g. iMAML, M = N in 122 tensors) the T
        ``param.grad + grad`` launches become ONE multi-tensor accumulate."""
        acc_par, acc_cur, acc_new = [], [], []
        for param, grad in zip(params, grads):
            if grad is None:
                continue
            cur = getattr(param, "grad", None)
            if cur is None:
                param.grad = grad
            elif cur.is_cuda and cur.dtype == torch.float32 and grad.dtype == torch.float32 and cur.shape == grad.shape:
                acc_par.append(param)
                acc_cur.append(cur)
                acc_new.append(grad)
            else:
                param.grad = cur + grad
        if len(acc_par) >= 4:
            # out-of-place like the reference (`param.grad = param.grad + grad`: nobody's tensor is mutated),
            # but as two multi-tensor launches into one fresh flat buffer instead of T `add` launches
            be = get_backend()
            layout = be.layout(acc_cur)
            flat = layout.new_flat()
            be.flatten(layout, acc_cur, flat, 1.0)
            views = layout.views(flat, acc_cur)
            be.axpy_multi(layout, views, acc_new, None, 1.0)
            for param, v in zip(acc_par, views):
                param.grad = v
        else:
            for param, cur, grad in zip(acc_par, acc_cur, acc_new):
                param.grad = cur + grad

    def synchronize_params(self, params, all_reduce=False):
        """problem.py:599-609 with ONE collective instead of one per tensor:
        gather the tensors into a flat buffer, broadcast from rank 0 (or average over ranks), scatter
        back.  Matters for T = 1,399 (DARTS supernet) and 201 (RoBERTa) over RCCL."""
        import torch.distributed as dist

        if self._world_size <= 1 or not dist.is_available() or not dist.is_initialized():
            return
        tensors = [p.data for p in params]
        if not tensors:
            return
        be = get_backend()
        layout = be.layout(tensors)
        flat = layout.new_flat()
        be.flatten(layout, tensors, flat, 1.0)
        if all_reduce:
            flat.div_(self._world_size)
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        else:
            dist.broadcast(flat, 0)
        be.scatter(layout, flat, tensors, 1.0)

    # ---- roll-back snapshot (implicit_problem.py:67-78) ------------------------------------
    @staticmethod
    def _is_bulk(v):
        return torch.is_tensor(v) and v.dtype == torch.float32 and v.numel() > 1 and v.is_contiguous()

    def _snapshot_entries(self):
        """(kind, owner, key, tensor) of every fp32 tensor that defines the problem's state:
        parameters, float buffers, optimizer state."""
        entries = [("param", p, None, p.data) for p in self.module.parameters()]
        entries += [("buffer", b, None, b) for b in self.module.buffers() if self._is_bulk(b)]
        if self.optimizer is not None:
            for p in self.module.parameters():
                for k, v in self.optimizer.state.get(p, {}).items():
                    if self._is_bulk(v):
                        entries.append(("opt", p, k, v))
        return entries

    def cache_states(self):
        """Device-side snapshot: one flat HBM buffer written by one multi-tensor kernel instead of
        ``copy.deepcopy(state_dict())`` of every tensor (the reference does this at every inner-loop
        start when ``roll_back=True``)."""
        entries = self._snapshot_entries()
        tensors = [e[3] for e in entries]
        be = get_backend()
        layout = be.layout(tensors)
        flat = layout.new_flat()
        be.flatten(layout, tensors, flat, 1.0)
        small = None
        if self.optimizer is not None:  # step counters / scalars / non-fp32 entries: tiny, copied as is
            small = {
                id(p): {k: copy.deepcopy(v) for k, v in self.optimizer.state.get(p, {}).items() if not self._is_bulk(v)}
                for p in self.module.parameters()
            }
        other_buffers = [b.clone() for b in self.module.buffers() if not self._is_bulk(b)]
        keys = [(kind, id(owner), key) for kind, owner, key, _ in entries]
        self._snapshot = (layout, flat, keys, small, other_buffers)

    def recover_states(self, clean=True):
        layout, flat, keys, small, other_buffers = self._snapshot
        now = {(kind, id(owner), key): t for kind, owner, key, t in self._snapshot_entries()}
        missing = [k for k in keys if k not in now]
        if missing:
            raise RuntimeError("recover_states: tensors present at cache_states() have disappeared")
        get_backend().scatter(layout, flat, [now[k] for k in keys], 1.0)
        if small is not None:
            saved = set(keys)
            for p in self.module.parameters():
                st = self.optimizer.state.get(p)
                if st is None:
                    continue
                # drop state created after the snapshot (e.g. Adam moments born inside the window) ...
                for k in [k for k, v in st.items() if self._is_bulk(v) and ("opt", id(p), k) not in saved]:
                    del st[k]
                # ... and put the small entries back; an empty saved state means "not initialised yet"
                if not small[id(p)] and not any(kk[0] == "opt" and kk[1] == id(p) for kk in keys):
                    st.clear()
                else:
                    st.update(copy.deepcopy(small[id(p)]))
        for b, saved_b in zip([b for b in self.module.buffers() if not self._is_bulk(b)], other_buffers):
            b.copy_(saved_b)
        if clean:
            self._snapshot = None

    def optimizer_step(self):
        raise NotImplementedError

    def zero_grad(self):
        for p in self.trainable_parameters():
            p.grad = None

    # ---- schedule (problem.py:371-454) ---------------------------------------------------------------------
    def check_ready(self):
        return all(self.ready) if self._children else True

    def step_normal(self, global_step=None):
        if not self.check_ready():
            return
        if self._inner_loop_start:
            if hasattr(self, "on_inner_loop_start"):
                self.on_inner_loop_start()
            self._inner_loop_start = False
            if self._roll_back:  # problem.py:379-381
                self.cache_states()
        if self._training:
            self._count += 1
        self.one_step_descent()
        if self.scheduler is not None and not self._roll_back:
            self.scheduler.step()
        period = self._config.unroll_steps * self.gas
        if self._training and self._count % period == 0 and self._count > self._config.warmup_steps:
            for parent in self._parents:
                parent.ready[parent.children.index(self)] = True
                parent.step_normal(global_step=global_step)
            self._inner_loop_start = True
        self.ready = [False] * len(self._children)

    def step_after_roll_back(self):
        """problem.py:417-436: restore the state cached at the inner-loop start, take ONE step on the
        last batch, and let the parents do the same."""
        if self.check_ready() and self._training:
            if self._roll_back:
                self.recover_states()
                self.one_step_descent(batch=self.cur_batch)
                if self.scheduler is not None:
                    self.scheduler.step()
                for parent in self._parents:
                    parent.ready[parent.children.index(self)] = True
                    parent.step_after_roll_back()
            self.ready = [False] * len(self._children)

    def step(self, global_step=None):
        self.step_normal(global_step=global_step)
        period = self._config.unroll_steps * self.gas
        if self._count % period == 0 and self._count > self._config.warmup_steps:
            self.step_after_roll_back()

    def train(self):
        self._training = True
        if self.module is not None:
            self.module.train()

    def eval(self):
        self._training = False
        if self.module is not None:
            self.module.eval()


class ImplicitProblem(Problem):
    """Problem differentiated by implicit differentiation (cg / neumann / darts), the class every
    BASELINE config uses (betty/problems/implicit_problem.py:13-92)."""

    def parameters(self):
        return list(self.module.parameters())

    def trainable_parameters(self):
        return list(self.module.parameters())

    def optimizer_step(self):
        """implicit_problem.py:40-65: [unscale] -> clip -> step -> record SAMA's last gradient -> [scaler update]."""
        clip = self._config.gradient_clipping
        scaler = self._ensure_scaler()
        if scaler is not None:
            scaler.unscale_(self.optimizer)
        if clip > 0.0:
            torch.nn.utils.clip_grad_norm_(self.trainable_parameters(), max_norm=clip)
        if scaler is not None:
            scaler.step(self.optimizer)   # skipped by the scaler when a gradient overflowed
        else:
            self.optimizer.step()
        if self._config.type == "sama":  # implicit_problem.py:60-65: SAMA's preconditioner needs the last gradient
            for param in self.trainable_parameters():
                state = self.optimizer.state[param]
                if param.grad is not None and len(state) != 0:
                    state["last_grad"] = param.grad.detach().clone()
        if scaler is not None:
            scaler.update()
