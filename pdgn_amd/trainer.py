"""One PDGN training iteration (models/PDGNet_v2.py:171-256) and its batch-axis data parallelism.

Reference parallelism: single-process ``nn.DataParallel`` (:101-105) -- per-replica BatchNorm
statistics, gradients summed onto GPU 0.  Here: one process per GPU, local batch per rank,
BatchNorm statistics stay local (same semantics), and gradients are all-reduced over RCCL/xGMI
as ONE flat fp32 buffer per network (G: 50.8 MB, D1-4: 6.8 MB) -- parameters' ``.grad`` tensors
are views into that buffer, so there is no bucket copy in or out.
"""
import os

import torch
import torch.distributed as dist
import torch.nn as nn

from . import streams as _streams
from .fused import clear_zero_colsum, flush_bn_counters, hold_bn_counters, release_zero_arena, reset_zero_arena
from .generator import PointDiscriminator, PointGenerator
from . import losses
from .losses import LocalPairLoss

PAIRS = ((0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3))   # get_local_pair calls :232-237


class FlatGrads:
    """One contiguous fp32 buffer per network for the gradient all-reduce.

    Single process: autograd simply assigns fresh ``.grad`` tensors (``begin`` clears them, so no
    per-parameter accumulate kernel runs) and the optimiser consumes them.  Data parallel: after the
    backward one multi-tensor copy packs the gradients into the buffer, ONE collective reduces the
    whole network (G: 50.8 MB, D1-4: 6.8 MB) and the parameters' ``.grad`` become views of it."""

    def __init__(self, params, first=None):
        """first: the parameters whose gradients a backward pass completes FIRST (the deepest block's); they are laid
        out in front and form the early bucket of reduce_early()."""
        ps = [p for p in params if p.requires_grad]
        early = {id(p) for p in (first or [])}
        self.params = [p for p in ps if id(p) in early] + [p for p in ps if id(p) not in early]
        self.n_early = sum(1 for p in ps if id(p) in early)
        # every gradient starts on a 16-byte boundary of the buffer (zero padding between them: the optimizer kernels' float4 path
        # -- csrc/adam.hip, torch's fused Adam alike -- needs it once .grad are these views)
        slot = lambda p: (p.numel() + 3) // 4 * 4
        total = sum(slot(p) for p in self.params)
        ref = self.params[0]
        self.buf = torch.zeros(total, dtype=ref.dtype, device=ref.device)
        self.views, off = [], 0
        for i, p in enumerate(self.params):
            if i == self.n_early:
                self.early_numel = off
            self.views.append(self.buf[off:off + p.numel()].view_as(p))
            off += slot(p)
        if self.n_early == len(self.params):
            self.early_numel = off
        self._early_work, self._early_done = None, False
        self._armed, self._arrived, self._hooks, self._group = False, 0, [], None

    def begin(self):
        """Before a backward: drop the old gradients (autograd then writes, never accumulates)."""
        for p in self.params:
            p.grad = None
        self._early_work, self._early_done = None, False
        self._arrived = 0
        clear_zero_colsum()
        if self.params and self.params[0].is_cuda:
            reset_zero_arena(self.params[0].device, self)

    def resume(self):
        """A further backward pass whose gradients will be ADDED to the ones already there (the second half of a split
        discriminator update): nothing is cleared, only the zero arena is re-opened on the current stream."""
        if self.params and self.params[0].is_cuda:
            reset_zero_arena(self.params[0].device, self)

    def zero_(self):                       # kept for callers that accumulate into the views
        self.buf.zero_()

    def pack(self):
        """Gather the fresh .grad tensors into the flat buffer and re-point .grad at its views (the early bucket only
        if reduce_early() has not already taken it: its slice may be inside a running all-reduce)."""
        self._pack(self.n_early if self._early_done else 0, len(self.params))

    def _pack(self, lo, hi):
        views, params = self.views[lo:hi], self.params[lo:hi]
        have = [(v, p) for v, p in zip(views, params) if p.grad is not None]
        missing = [v for v, p in zip(views, params) if p.grad is None]
        if have:
            self._copy([v for v, _ in have], [p.grad for _, p in have], (lo, hi) if len(have) == hi - lo else None)
        for v in missing:
            v.zero_()
        for v, p in zip(views, params):
            p.grad = v

    def _copy(self, dsts, srcs, full=None):
        """views <- fresh gradients: one launch of csrc/adam.hip's multi-tensor copy per 128 tensors (torch._foreach_copy_: 82 us for
        the generator's 160 gradients); anything but contiguous fp32 CUDA tensors of equal sizes goes torch's way."""
        if (os.environ.get("PDGN_OWN_ADAM", "1") == "1" and dsts[0].is_cuda and all(
                d.dtype == torch.float32 and s.dtype == torch.float32 and d.is_contiguous() and s.is_contiguous() and d.numel() == s.numel()
                and s.is_cuda for d, s in zip(dsts, srcs))):
            import ctypes
            from . import _lib
            n = len(dsts)
            # (full = (lo, hi): every view of that range takes part -- the views are static, their pointer and size arrays are built once)
            cache = self.__dict__.setdefault("_copy_cache", {})
            hit = cache.get(full) if full is not None else None
            if hit is None:
                vp = ctypes.c_void_p * n
                hit = (vp, vp(*[d.data_ptr() for d in dsts]), (ctypes.c_longlong * n)(*[d.numel() for d in dsts]))
                if full is not None:
                    cache[full] = hit
            _lib.check(_lib.lib().pdgn_copy_multi(n, hit[1], hit[0](*[s.data_ptr() for s in srcs]), hit[2], _lib.stream_of(dsts[0])),
                       "pdgn_copy_multi")
            return
        torch._foreach_copy_(dsts, srcs)

    def arm_early(self, on=True, group=None):
        """Let the backward itself start the early bucket's all-reduce: a post-accumulate-grad hook on every early
        parameter counts the gradients that HAVE ARRIVED since begin(), and the one that completes the bucket calls
        reduce_early().  Arrival, not a position in the graph, is the trigger: autograd runs a node created before the
        forward (the re-associated operands of PointGenerator.preassemble) after everything of higher sequence number
        that is ready, so "the backward has reached the block's input" does not say that the block's weight gradients
        exist (the early all-reduce then ran on zero-filled slices while AccumulateGrad was still to write
        them).  A parameter that receives no gradient in some backward simply leaves the bucket to all_reduce_mean()."""
        self._armed, self._group = bool(on), group
        if on and not self._hooks and self.n_early:
            for p in self.params[:self.n_early]:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._arrival))

    def _arrival(self, _param):
        if not self._armed or self._early_done:
            return
        self._arrived += 1
        if self._arrived == self.n_early:
            self.reduce_early(self._group)

    def reduce_early(self, group=None):
        """Called from inside the backward once the early bucket's gradients exist (arm_early): pack them and start their
        all-reduce (asynchronous: on RCCL's stream, underneath the rest of the backward).  all_reduce_mean() later
        reduces the rest and joins.  No-op without a process group or without an early bucket, and REFUSED while a
        gradient of the bucket is still missing (its slice would be zero-filled under a running collective)."""
        if self._early_done or self.n_early == 0 or not (dist.is_available() and dist.is_initialized()):
            return
        rec = _recorder(self.buf.device)
        if _capturing(self.buf.device) and rec is None:
            return
        if any(p.grad is None for p in self.params[:self.n_early]):
            return
        self._pack(0, self.n_early)
        self._early_done = True
        if rec is not None:                                      # launch-list capture: the collective becomes a host point of the list

            def start(fg=self, group=group):
                fg._early_work = dist.all_reduce(fg.buf[:fg.early_numel], op=dist.ReduceOp.SUM, group=group, async_op=True)
            rec.point(start)
            return
        self._early_work = dist.all_reduce(self.buf[:self.early_numel], op=dist.ReduceOp.SUM, group=group, async_op=True)

    def all_reduce_mean(self, group=None):
        """Average over ranks (RCCL all-reduce over xGMI when the backend is nccl)."""
        if dist.is_available() and dist.is_initialized():
            rec = _recorder(self.buf.device)
            if rec is not None:                                  # launch-list capture: host points; the division is captured
                early = self._early_done

                def reduce(fg=self, group=group, early=early):
                    if early:
                        if fg.early_numel < fg.buf.numel():
                            dist.all_reduce(fg.buf[fg.early_numel:], op=dist.ReduceOp.SUM, group=group)
                        fg._early_work.wait()
                        fg._early_work = None
                    else:
                        dist.all_reduce(fg.buf, op=dist.ReduceOp.SUM, group=group)
                rec.point(reduce)
                if dist.get_world_size(group) > 1:
                    self.buf.div_(dist.get_world_size(group))
                return
            if self._early_done:
                if self.early_numel < self.buf.numel():
                    dist.all_reduce(self.buf[self.early_numel:], op=dist.ReduceOp.SUM, group=group)
                self._early_work.wait()
                self._early_work = None
            else:
                dist.all_reduce(self.buf, op=dist.ReduceOp.SUM, group=group)
            if dist.get_world_size(group) > 1:
                self.buf.div_(dist.get_world_size(group))


def _capturing(device):
    return device.type == "cuda" and torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


def _recorder(device):
    """The active host-point recorder of a launch-list capture (replay.Recorder), or None."""
    if device.type != "cuda":
        return None
    from . import replay
    rec = replay.recorder()
    return rec if rec is not None and _capturing(device) else None


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


class LeanAdamStep:
    """`optimizer.step()` of a fused, capturable `torch.optim.Adam` without its per-call Python (profile hooks, `_init_group`,
    grouping by device and dtype: ~0.3 ms of host time per call, five calls per training step -- tools/host_prof.py): after the
    first, ordinary step the same two calls torch makes -- `_foreach_add_` on the step counters, `_fused_adam_` on the parameter /
    gradient / moment lists -- are issued on lists cached here.  The state stays the optimizer's own (checkpoints, broadcasts and
    `state_dict()` are untouched); anything unusual (a parameter without gradient, several groups, amsgrad, weight decay, a
    non-fused optimizer, step hooks, replaced state tensors) keeps calling `optimizer.step()`; the invariants are re-checked on
    every call.  `torch._fused_adam_` / `torch._foreach_add_` are private torch entry points (written against torch 2.10): a
    changed signature falls back to `optimizer.step()` for good."""

    def __init__(self, opt):
        self.opt, self.lists, self._table = opt, None, None

    def reset(self):
        """The optimizer's state tensors were replaced (load_state_dict): rebuild the lists after the next ordinary step."""
        self.lists = None
        self._table = None

    _OWN = os.environ.get("PDGN_OWN_ADAM", "1") == "1"           # A/B switch: 0 = torch._fused_adam_

    def _own_adam(self, ps, grads, exp_avgs, exp_avg_sqs, steps, g):
        """The whole list in ceil(n / 72) launches of csrc/adam.hip, one workgroup per 4096 elements (torch's fused kernel: 64 K-element
        chunks, five launches and 230 us for the generator's 12.7 M parameters at the end of every iteration).  The pointers travel in
        the kernel arguments: nothing to upload, and a recorded iteration re-issues them as they were.  False (torch's kernel runs)
        for anything but contiguous fp32 CUDA tensors."""
        if not self._OWN or not ps or not ps[0].is_cuda:
            return False
        from . import _lib
        import ctypes
        n = len(ps)
        tab = self._table
        if tab is None or tab[0] != n or tab[1][0] != ps[0].data_ptr() or tab[2][n - 1] != exp_avgs[n - 1].data_ptr():
            if any(t.dtype != torch.float32 or not t.is_contiguous() for lst in (ps, exp_avgs, exp_avg_sqs) for t in lst) \
                    or steps[0].dtype != torch.float32:
                return False
            vp = ctypes.c_void_p * n
            tab = self._table = (n, vp(*[t.data_ptr() for t in ps]), vp(*[t.data_ptr() for t in exp_avgs]),
                                 vp(*[t.data_ptr() for t in exp_avg_sqs]), (ctypes.c_longlong * n)(*[t.numel() for t in ps]), vp)
        if any(t.dtype != torch.float32 or not t.is_contiguous() for t in grads):
            return False
        _lib.check(_lib.lib().pdgn_adam_multi(n, tab[1], tab[5](*[t.data_ptr() for t in grads]), tab[2], tab[3], tab[4],
                                              ctypes.c_double(g["lr"]), ctypes.c_double(g["betas"][0]), ctypes.c_double(g["betas"][1]),
                                              ctypes.c_double(g["eps"]), _lib.ptr(steps[0]), _lib.stream_of(ps[0])), "pdgn_adam_multi")
        return True

    def step(self):
        opt = self.opt
        if self.lists is None:
            opt.step()
            g = opt.param_groups[0]
            ok = (len(opt.param_groups) == 1 and g.get("fused") and g.get("capturable") and not g.get("amsgrad")
                  and not g.get("maximize") and not g.get("differentiable") and g.get("weight_decay", 0) == 0
                  and not isinstance(g["lr"], torch.Tensor) and all(p.grad is not None for p in g["params"])
                  and os.environ.get("PDGN_LEAN_ADAM", "1") == "1")
            if ok:
                ps, st = list(g["params"]), opt.state
                self.lists = (ps, [st[p]["exp_avg"] for p in ps], [st[p]["exp_avg_sq"] for p in ps], [st[p]["step"] for p in ps])
            else:
                self.lists = False
            return
        if self.lists is False:
            opt.step()
            return
        ps, exp_avgs, exp_avg_sqs, steps = self.lists
        grads = [p.grad for p in ps]
        g = opt.param_groups[0]
        # cheap invariants, every call: one group over the same parameters, no weight decay / amsgrad / maximize, a Python float
        # learning rate, the optimizer's own state tensors still the cached ones, no step hooks, every gradient present
        st0 = opt.state.get(ps[0]) if ps else None
        if (len(opt.param_groups) != 1 or len(g["params"]) != len(ps) or g.get("weight_decay", 0) != 0 or g.get("amsgrad")
                or g.get("maximize") or isinstance(g["lr"], torch.Tensor) or st0 is None or st0.get("exp_avg") is not exp_avgs[0]
                or opt._optimizer_step_pre_hooks or opt._optimizer_step_post_hooks or any(gr is None for gr in grads)):
            self.lists = None                                    # re-validated after the next ordinary step
            opt.step()
            return
        try:
            with torch.no_grad():
                torch._foreach_add_(steps, 1)
                if self._own_adam(ps, grads, exp_avgs, exp_avg_sqs, steps, g):
                    return
                torch._fused_adam_(ps, grads, exp_avgs, exp_avg_sqs, [], steps, amsgrad=False, lr=g["lr"], beta1=g["betas"][0],
                                   beta2=g["betas"][1], weight_decay=0.0, eps=g["eps"], maximize=False, grad_scale=None,
                                   found_inf=None)
        except TypeError:                                        # the private op's signature changed (another torch version):
            torch._foreach_sub_(steps, 1)                        # undo the counter and take the public path from now on
            self.lists = False
            opt.step()


class PDGNTrainer:
    """Generator + D1..D4 + their Adam optimisers (lr 1e-4, betas (0.5, 0.999), :121-125) and the
    op sequence of one iteration.  ``step`` returns the six logged losses (:259-261) as 0-dim
    device tensors (no host sync inside the step)."""

    def __init__(self, device="cuda", lr=1e-4, num_k=20, base_points=128, generator=None,
                 discriminators=None, distributed=None):
        self.device = torch.device(device)
        self.G = (generator or PointGenerator(num_k=num_k, base_points=base_points)).to(self.device)
        self.D = [d.to(self.device) for d in
                  (discriminators or [PointDiscriminator(i, (2 * base_points) << (i - 1)) for i in (1, 2, 3, 4)])]
        self.local_pair = LocalPairLoss(20)
        self.distributed = world_size() > 1 if distributed is None else distributed
        # the deepest block's gradients are complete when the backward reaches that block's input: they form the early
        # bucket whose all-reduce runs underneath the rest of the backward (PDGN_BUCKETS=0: one all-reduce at the end)
        self._buckets = os.environ.get("PDGN_BUCKETS", "1") == "1"
        deepest = (list(self.G.bilateral4.parameters()) + list(self.G.mlp4.parameters())) if self._buckets and hasattr(self.G, "bilateral4") else None
        self.gradG = FlatGrads(self.G.parameters(), first=deepest)
        if self.distributed and self._buckets:
            self.gradG.arm_early()
        self.gradD = [FlatGrads(d.parameters()) for d in self.D]
        cap = self.device.type == "cuda"                     # device-side step counter: graph-capturable
        adam = lambda m: torch.optim.Adam(m.parameters(), lr=lr, betas=(0.5, 0.999), capturable=cap, fused=cap and os.environ.get("PDGN_FUSED_ADAM", "1") == "1")
        self.optG, self.optD = adam(self.G), [adam(d) for d in self.D]
        self._stepG, self._stepD = LeanAdamStep(self.optG), [LeanAdamStep(o) for o in self.optD]
        # stream-overlapped schedule of the eager step (see _step_overlapped); PDGN_OVERLAP=0 turns it off
        self.overlap = cap and os.environ.get("PDGN_OVERLAP", "1") == "1"
        self._side = None
        self._lw = {}
        for ws in {1, world_size() if self.distributed else 1}:
            self._loss_weights(ws)
        self._early_tail = os.environ.get("PDGN_EARLY_TAIL", "1") == "1"
        self._split_d = os.environ.get("PDGN_SPLIT_D", "1") == "1"
        self._defer_d = os.environ.get("PDGN_DEFER_D", "1") == "1"
        self.sync_replicas()

    def sync_replicas(self, src=0):
        """nn.DataParallel replicates module 0's parameters and buffers onto every device each forward
        (models/PDGNet_v2.py:101-105); with one process per GPU the replicas are made identical ONCE -- here, and
        after load() -- by broadcasting rank `src`'s parameters, buffers and Adam state; identical averaged gradients
        keep the parameters identical afterwards (BatchNorm running statistics stay per replica, as they do under
        DataParallel, where only replica 0's survive).  No-op in a single process."""
        if not (self.distributed and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return
        with torch.no_grad():
            for net in [self.G] + self.D:
                flat = [t for t in list(net.parameters()) + list(net.buffers())]
                for dtype in sorted({t.dtype for t in flat}, key=str):   # same order on every rank
                    group = [t for t in flat if t.dtype == dtype]
                    buf = torch.cat([t.reshape(-1) for t in group])
                    dist.broadcast(buf, src)
                    off = 0
                    for t in group:
                        t.copy_(buf[off:off + t.numel()].view_as(t))
                        off += t.numel()
            for opt in [self.optG] + self.optD:
                for st in opt.state.values():
                    for v in st.values():
                        if torch.is_tensor(v):
                            dist.broadcast(v, src)

    def train(self):
        self.G.train()
        for d in self.D:
            d.train()

    # ---- checkpoints in the reference's own two-file format (models/PDGNet_v2.py:384-408 save, :331-382 load)
    @staticmethod
    def _ref_model_state(module):
        # the reference wraps every net in nn.DataParallel (:101-105): its files carry 'module.'-prefixed keys
        return {"module." + k: v.detach().cpu() for k, v in module.state_dict().items()}

    @staticmethod
    def _ref_optim_state(opt):
        sd = opt.state_dict()
        state = {}
        for i, st in sd["state"].items():
            # capturable/fused Adam keeps `step` as a device tensor; the reference's torch 1.7 Adam holds an int
            state[i] = {k: (int(v.item()) if k == "step" else v.detach().cpu()) if torch.is_tensor(v) else v
                        for k, v in st.items()}
        keep = ("lr", "betas", "eps", "weight_decay", "amsgrad", "params")
        groups = [{k: g[k] for k in keep} for g in sd["param_groups"]]
        return {"state": state, "param_groups": groups}

    def save(self, checkpoint_dir, index_epoch, category="chair"):
        """Writes `<epoch>_<category>_G.pth` / `_D.pth` with the reference's keys (:391-407) so that either code
        base can resume from the other's files."""
        os.makedirs(checkpoint_dir, exist_ok=True)
        stem = os.path.join(checkpoint_dir, "%s_%s" % (index_epoch, category))
        held = hold_bn_counters(False)
        flush_bn_counters()
        hold_bn_counters(held)
        torch.save({"G_model": self._ref_model_state(self.G), "G_optimizer": self._ref_optim_state(self.optG),
                    "G_epoch": index_epoch}, stem + "_G.pth")
        dfile = {"D_epoch": index_epoch}
        for i, (d, o) in enumerate(zip(self.D, self.optD), 1):
            dfile["D_model%d" % i] = self._ref_model_state(d)
            dfile["D_optimizer%d" % i] = self._ref_optim_state(o)
        torch.save(dfile, stem + "_D.pth")
        return stem + "_G.pth", stem + "_D.pth"

    @staticmethod
    def _load_optim(opt, sd):
        groups = []
        for mine, theirs in zip(opt.state_dict()["param_groups"], sd["param_groups"]):
            g = dict(mine)                                   # keep this build's capturable/fused/foreach flags
            g.update({k: theirs[k] for k in ("lr", "betas", "eps", "weight_decay", "amsgrad") if k in theirs})
            groups.append(g)
        opt.load_state_dict({"state": sd["state"], "param_groups": groups})

    def load(self, path_G, path_D):
        """Resume from a reference (or own) checkpoint pair; returns the stored epoch (:352, :374).
        A missing file raises FileNotFoundError (the reference calls exit(), :345-347)."""
        from .generator import load_reference_state_dict
        g = torch.load(path_G, map_location="cpu")
        d = torch.load(path_D, map_location="cpu")
        load_reference_state_dict(self.G, g["G_model"])
        self._load_optim(self.optG, g["G_optimizer"])
        for i, (m, o) in enumerate(zip(self.D, self.optD), 1):
            load_reference_state_dict(m, d["D_model%d" % i])
            self._load_optim(o, d["D_optimizer%d" % i])
        for st in [self._stepG] + self._stepD:                   # the optimizers' state tensors are new objects now
            st.reset()
        self.sync_replicas()
        return g["G_epoch"]

    def _loss_weights(self, ws):
        """(1.2, 1.2, 1.2, 1, 0.1*ws): weights of the four adversarial terms and of the shape loss (:254); the shape loss
        is a batch SUM, so under data parallelism it is scaled by the world size before the mean all-reduce."""
        w = self._lw.get(ws)
        if w is None:                                            # created outside any graph capture (see __init__)
            w = self._lw[ws] = torch.tensor([1.2, 1.2, 1.2, 1.0, 0.1 * ws], dtype=torch.float32, device=self.device)
        return w

    def _freeze_D(self, frozen):
        params = self.__dict__.get("_d_params")
        if params is None:                                       # (module.parameters() walks the module tree: 0.4 ms per call)
            params = self.__dict__["_d_params"] = [p for d in self.D for p in d.parameters()]
        for p in params:
            p.requires_grad_(not frozen)

    def similar_terms(self, clouds, pairs, own=None):
        """{(a, b): (like_mu, like_cov)} of get_local_pair (:232-237) for the given resolution pairs.  `own` caches
        the statistics of cloud a around itself across calls (they are shared by all pairs (a, .))."""
        terms, own = {}, ({} if own is None else own)
        for a, b in pairs:
            if a not in own:
                cl = clouds[a].transpose(1, 2).contiguous()
                own[a] = self.local_pair.stats(cl, cl)
            terms[(a, b)] = self.local_pair(clouds[a], clouds[b], self_stats=own[a])
        return terms

    @staticmethod
    def _sum_terms(terms):
        return torch.stack([terms[p][0] for p in PAIRS] + [terms[p][1] for p in PAIRS]).sum()

    def similar_loss(self, clouds):
        """Sum of the 6 like_mu and the 6 like_cov terms, in the reference's order (:251-252)."""
        return self._sum_terms(self.similar_terms(clouds, PAIRS))

    # The iteration is written as six SEGMENTS separated by the five gradient all-reduces, so that
    # each segment can be captured into a hipGraph (no RCCL call inside a capture) and replayed:
    #   0: G(z1) under no_grad (:179) + D1 forward/backward        -> all-reduce grad D1
    #   1..3: Adam D_i (:191) + D_{i+1} forward/backward           -> all-reduce grad D_{i+1}
    #   4: Adam D4 + G(z2), local-pair losses, D(gen), backward    -> all-reduce grad G
    #   5: Adam G (:256)
    def _seg_d(self, st, i):
        D = self.D[i]
        self.gradD[i].begin()
        lossD = losses.mse_const(D(st["reals"][i]), 1.0, 0.5) + losses.mse_const(D(st["fakes"][i]), 0.0, 0.5)
        lossD.backward()
        st["out"]["d_loss%d" % (i + 1)] = lossD.detach()

    # The same update in two halves (overlapped schedule): lossD = mse(D(real), 1) / 2 + mse(D(fake), 0) / 2 (:186-190) is
    # a sum of two terms that share nothing but the parameters, and the REAL term does not depend on the generator at all.
    # Its forward and backward are issued at the start of the iteration and run underneath generator pass #1 (whose first
    # three blocks are small kernels that leave the chip idle); when the fake cloud exists only the fake term's forward
    # and backward are left on the discriminator's chain, which was as long as the generator's own.
    # Same order of D's forward calls (real, then fake: BatchNorm running statistics), gradients summed once.
    def _seg_d_real(self, st, i):
        D = self.D[i]
        self.gradD[i].begin()
        params = self.gradD[i].params
        loss_r = losses.mse_const(D(st["reals"][i]), 1.0, 0.5)
        st["d_half"][i] = (loss_r.detach(), torch.autograd.grad(loss_r, params))

    def _seg_d_fake(self, st, i):
        D = self.D[i]
        fg = self.gradD[i]
        fg.resume()
        loss_f = losses.mse_const(D(st["fakes"][i]), 0.0, 0.5)
        loss_r, g_real = st["d_half"][i]
        st.setdefault("keep", []).append(loss_r)            # (it may have been computed on another stream than this half runs on: it must
                                                            #  not return to that stream's pool before the sum below has RUN)
        g_fake = torch.autograd.grad(loss_f, fg.params)
        torch._foreach_add_(g_real, g_fake)
        for p, g in zip(fg.params, g_real):
            p.grad = g
        st["d_half"][i] = None
        st["out"]["d_loss%d" % (i + 1)] = loss_r + loss_f.detach()

    def _segment(self, st, k):
        if k == 0:
            # generator pass #1 (:179): only its detached outputs are ever used => no graph needed;
            # BatchNorm running statistics update exactly as in the reference.
            with torch.no_grad():
                st["fakes"] = self.G(self._z(st, "z1"))
            self._seg_d(st, 0)
        elif k in (1, 2, 3):
            self._stepD[k - 1].step()
            self._seg_d(st, k)
        elif k == 4:
            self._stepD[3].step()
            self.gradG.begin()
            # The reference lets lossG.backward() also fill the discriminators' .grad and throws
            # that away at the next zero_grad (:183); freezing D skips those weight-gradient GEMMs.
            self._freeze_D(True)
            gen = self.G(self._z(st, "z2"))
            similar = self.similar_loss(gen)
            g_loss = [losses.mse_const(self.D[i](gen[i]), 1.0) for i in range(4)]
            adv = 1.2 * g_loss[0] + 1.2 * g_loss[1] + 1.2 * g_loss[2] + g_loss[3]
            lossG = adv + 0.1 * similar
            # MSE is a batch MEAN, the shape loss a batch SUM (chamfer_loss.py:16-20): to reproduce
            # one reference step at the global batch, scale the sum term by world_size before the
            # mean all-reduce.
            ws = st["ws"]
            (adv + (0.1 * ws) * similar if ws > 1 else lossG).backward()
            self._freeze_D(False)
            st["out"]["g_loss"], st["out"]["similar_loss"] = lossG.detach(), similar.detach()
        else:
            self._stepG.step()

    def _z(self, st, name):
        """Noise of a generator pass: the caller's tensor, or -- when it is None -- drawn on the
        device inside the step (z ~ N(0, 0.2^2), :178/:228), which keeps a captured iteration free
        of host-side inputs."""
        z = st[name]
        return z if z is not None else torch.randn(st["B"], 128, device=self.device) * 0.2

    def _comm(self, k):
        """Collective that follows segment k (RCCL all-reduce of one flat gradient buffer)."""
        if self.distributed and k < 5:
            fg = self.gradD[k] if k < 4 else self.gradG
            fg.pack()
            fg.all_reduce_mean()

    def _state(self, reals, z1, z2):
        B = reals[0].shape[0]
        return {"reals": reals, "z1": z1, "z2": z2, "B": B, "out": {}, "ws": world_size() if self.distributed else 1}

    def step(self, reals, z1, z2):
        """reals: four tensors (B,3,N_k); z1 / z2: noise (B,128) of the two generator passes
        (:178, :228).  Returns dict of 0-dim device tensors (no host sync inside the step)."""
        if self.overlap:
            return self._step_overlapped(reals, z1, z2)
        st = self._state(reals, z1, z2)
        for k in range(6):
            self._segment(st, k)
            self._comm(k)
        release_zero_arena()
        return st["out"]

    def _step_overlapped(self, reals, z1, z2, st=None):
        """Same iteration, scheduled for the GPU: the four discriminator updates are independent of each other
        and of everything the generator does after emitting their resolution, and they are chains of small
        kernels (D1-D3 see 256-1024 points) that leave most CUs idle.  D_k's update (forward real + fake,
        backward, all-reduce, Adam) is enqueued on its own HIP stream the moment G(z1) has produced level k,
        so it runs underneath the deeper levels' GEMMs / gather-sums; D4's runs underneath G(z2).  In the same
        way D_k(G(z2)_k) (behind D_k's update, on D_k's stream) and the loss pairs that end at level k start from
        G(z2)'s stage hook; the default stream joins all of them before the backward.  Results are those of the
        sequential order (the reference's D updates do not read each other)."""
        # num_batches_tracked: the iteration's increments in one launch at its end (fused.hold_bn_counters)
        held_counts = hold_bn_counters(True)
        try:
            return self._step_overlapped_body(reals, z1, z2, st)
        finally:
            hold_bn_counters(held_counts)
            flush_bn_counters()

    def _step_overlapped_body(self, reals, z1, z2, st=None):
        st = st if st is not None else self._state(reals, z1, z2)
        main = torch.cuda.current_stream(self.device)
        pl = _streams.plan(self.device)                     # streams by measured hardware queue (streams.py)
        self._side, self._side_lp = pl.d, pl.lp
        st["fakes"] = [None] * 4
        st["keep"] = []
        mark = st.get("mark") or (lambda name: None)        # tools/phase_events.py: HIP events on the default stream
        mark("start")
        if st.get("tag_streams"):
            # capture for a launch list (capture_list): a captured graph does not say which stream a node was recorded
            # on, so every stream of the schedule opens with a marker node carrying its id (csrc/replay.hip)
            from . import replay as _replay
            _replay.mark(_replay.MAIN, main)
            for sid, s in enumerate(list(pl.d) + [pl.lp, pl.knn], 1):
                s.wait_stream(main)
                _replay.mark(sid, s)

        # D_k's update starts on the DEVICE when level k exists (an event recorded in the stage hook), but its ~150
        # launches are issued by the host only after the whole generator pass has been issued: with a host that runs
        # ahead (steady state) the device sees the same schedule, and right after a synchronise -- the first timed step
        # of a benchmark, the first step after a checkpoint -- the default stream is not left idle while the host
        # works through three discriminator updates between the generator's blocks (G(z1) forward of such a step:
        # 15-25 ms before).
        levels = []

        def d_mark(level, cloud):
            st["fakes"][level] = cloud
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            levels.append((level, ev))
            if level < 3:
                mark("G(z1) level %d" % (level + 1))

        split = self._split_d

        def d_update(level, ev):
            side = self._side[level]
            side.wait_event(ev)
            with torch.cuda.stream(side), torch.enable_grad():
                if split:
                    self._seg_d_fake(st, level)
                else:
                    self._seg_d(st, level)
                self._comm(level)
                self._stepD[level].step()

        # The generator's re-associated GEMM operands (and, on the bf16 matrix cores, their pre-split planes) depend on the
        # parameters only: built once for both passes, with grad enabled -- pass #2's backward reaches the conv weights through
        # them.  (On the issuing stream: built on the idle kNN stream behind an event the iteration faulted at B = 35 --
        # the adjoint of the assembly then runs on that stream too -- and the four launches are ~60 us.)
        # Under data parallelism with the early gradient bucket the DEEPEST block's operands are pre-assembled for pass #1
        # only (no graph): pass #2 assembles them inside its forward, so that the adjoint of that assembly -- the conv
        # weights' gradients, 69 % of the early bucket -- runs right behind the block's backward instead of at the very end
        # (autograd orders ready nodes by creation: a node made before the forward goes last).
        pre_ok = hasattr(self.G, "preassemble") and os.environ.get("PDGN_PREASM", "1") == "1"
        if pre_ok:
            with torch.enable_grad():
                self.G.preassemble(deepest_without_graph=self.distributed and self._buckets)
        # (Rule for anything added to this schedule: a side stream may wait for events of the ISSUING stream only.  hipStreamEndCapture
        # (ROCm 7.2) recurses without bound -- a segmentation fault inside hip::Stream::EndCapture -- as soon as one captured side stream
        # waits for an event recorded on ANOTHER side stream; round 6 ran pass #1 on D4's stream beside pass #2 that way and routed
        # every dependency through the issuing stream.  It measured slower -- the iteration is bound by the sum of its kernels -- and
        # is in git at 5c84567 (trainer.py, PDGN_PASS1_SIDE))
        if split:
            st["d_half"] = [None] * 4
            for level, side in enumerate(self._side):
                side.wait_stream(main)                      # (the real batch / targets may have been produced on `main`)
                with torch.cuda.stream(side), torch.enable_grad():
                    self._seg_d_real(st, level)

        def d_now(level, cloud):
            d_mark(level, cloud)
            d_update(*levels.pop())

        hook1 = d_mark if self._defer_d else d_now
        with torch.no_grad():
            self.G(self._z(st, "z1"), stage_hook=hook1)
        for level, ev in levels:
            d_update(level, ev)
        mark("G(z1) level 4")
        self.gradG.begin()
        self._freeze_D(True)
        # The shape-preserving loss (12 kNN + Chamfer terms) and the four D(gen) passes read the clouds and nothing of
        # each other: they run on their own streams, forward and (autograd keeps an op's backward on its forward's
        # stream) backward.  They also do not need the whole forward: pair (a, k) and D_k(gen_k) only need level k, so
        # they are enqueued from the generator's stage hook and run underneath the deeper blocks; after the forward
        # only D4(gen) and the three pairs with the 2048-point cloud are left to wait for.  Being created last, those
        # are also the first adjoints autograd issues -- the ones block 4's backward waits for.
        terms, own, g_loss, gen_so_far = {}, {}, [None] * 4, []
        early = self._early_tail

        def d_gen(level, cloud):                            # D_k(G(z2)_k) behind D_k's update, on D_k's stream
            side = self._side[level]
            side.wait_stream(main)
            with torch.cuda.stream(side):
                g_loss[level] = losses.mse_const(self.D[level](cloud), 1.0)

        def tail(level, cloud):
            gen_so_far.append(cloud)
            if level >= 1:
                self._side_lp.wait_stream(main)
                with torch.cuda.stream(self._side_lp):
                    terms.update(self.similar_terms(gen_so_far, [(a, level) for a in range(level)], own))
            d_gen(level, cloud)

        gen = self.G(self._z(st, "z2"), stage_hook=tail if early else None)
        mark("G(z2) forward")
        if not early:
            self._side_lp.wait_stream(main)
            with torch.cuda.stream(self._side_lp):
                terms = self.similar_terms(gen, PAIRS)
            for i, side in enumerate(self._side):
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    g_loss[i] = losses.mse_const(self.D[i](gen[i]), 1.0)
        for side in self._side:
            main.wait_stream(side)
        main.wait_stream(self._side_lp)
        mark("wait: D(gen) + local-pair forward")
        similar = self._sum_terms(terms)
        # 1.2*(g1+g2+g3) + g4 + 0.1*similar (:254) as one weighted sum: this sits between the joins and the backward,
        # where every launch is on the critical path (three launches here and one in backward instead of eight and five)
        ws = st["ws"]
        terms = torch.stack(g_loss + [similar])
        lossG = (terms * self._loss_weights(1)).sum()
        (lossG if ws == 1 else (terms * self._loss_weights(ws)).sum()).backward()
        mark("backward")
        self._freeze_D(False)
        st["out"]["g_loss"], st["out"]["similar_loss"] = lossG.detach(), similar.detach()
        self._comm(4)
        self._stepG.step()
        if pre_ok:
            self.G.drop_preassembled()
        release_zero_arena()                                # no later backward may receive slices of this step's arena
        mark("all-reduce + Adam G")
        return st["out"]

    # ---------------------------------------------------------------- hipGraph replay
    def capture(self, reals, z1=None, z2=None, warmup=3):
        """Capture the iteration into hipGraphs (static input buffers, one shared memory pool):
        ~2500 kernel launches become one graph launch (six when gradients are all-reduced
        between segments).  (Drawing the noise on the device inside the capture is not used: a
        captured torch.randn faulted on this ROCm build.)

        ROCm 7.2 note (measured on MI355X): an eager kernel enqueued between two graph launches
        on the same stream is NOT reliably ordered against them (replays with `copy_` of the
        inputs in between fault, the same replays with a stream synchronise around the copies,
        or with no eager work in between, are clean) -- hence `_sync()` around every eager
        operation that sits between replays."""
        self._static = self._state([r.clone() for r in reals], z1.clone() if z1 is not None else None,
                                   z2.clone() if z2 is not None else None)
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(warmup):
                for k in range(6):
                    self._segment(self._static, k)
                    self._comm(k)
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        # One graph per run of segments between collectives: the whole iteration when there is
        # nothing to all-reduce, six graphs (replayed with the all-reduces between them) otherwise.
        groups = [[k] for k in range(6)] if self.distributed else [list(range(6))]
        self._graphs, pool = [], None
        if self.overlap and not self.distributed:
            # the stream-overlapped schedule as ONE graph: the side streams fork from / join the capturing stream
            g = torch.cuda.CUDAGraph()
            defer, self._defer_d = self._defer_d, False       # capture order = node order of the replay: keep D_k's
            try:                                              # nodes next to the level that feeds them
                with torch.cuda.graph(g):
                    self._step_overlapped(None, None, None, st=self._static)
            finally:
                self._defer_d = defer
            self._graphs.append((g, 5))
            release_zero_arena()                            # the arena of the capture lives in the graph's private pool
            return self
        for group in groups:
            g = torch.cuda.CUDAGraph()
            # thread_local: with a communicator alive its watchdog thread polls events (hipEventQuery) at any time; under
            # the default "global" capture mode such a call from ANOTHER thread invalidates the capture (and surfaced as an
            # abort when the process group was destroyed, one run in four)
            with torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):
                for k in group:
                    self._segment(self._static, k)
            pool = g.pool()
            self._graphs.append((g, group[-1]))
            self._comm(group[-1])
        release_zero_arena()
        return self

    # ---------------------------------------------------------------- launch-list replay (csrc/replay.hip)
    def capture_list(self, reals, z1, z2, warmup=2):
        """Record the stream-overlapped iteration ONCE (stream capture into a hipGraph that is kept but never instantiated)
        and turn it into a launch list: `step_list` then re-issues the same ~1300 launches with plain HIP calls on the
        streams the eager schedule uses -- ~4 us of host time per launch instead of ~20 (autograd nodes, allocations,
        ctypes), and none of hipGraphLaunch's own scheduling (slower than eager here).
        Data parallel: a collective cannot be captured, so every gradient all-reduce -- D_k's on D_k's stream, the generator's
        early bucket from inside the backward, the rest behind it -- is recorded as a HOST POINT (a marker node, replay.Recorder):
        `step_list` issues the list in ranges and makes the RCCL calls between them, on the streams and at the places of the
        eager schedule (packing, the division by the world size and the optimizer steps are ordinary captured launches).
        The `warmup` iterations are real optimizer updates."""
        from . import replay
        if not self.overlap:
            raise RuntimeError("capture_list: stream-overlapped schedule only")
        self._static = self._state([r.clone() for r in reals], z1.clone(), z2.clone())
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(warmup):                          # allocator pools, the optimizers' lean lists, arena sizes
                self._step_overlapped(None, None, None, st=self._state(self._static["reals"], self._static["z1"], self._static["z2"]))
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        if self.distributed and dist.is_available() and dist.is_initialized():
            # The process group's watchdog thread polls the completion events of the collectives it still lists (every 100 ms).
            # Such a poll while this thread captures ends the process (measured on ROCm 7.2: "operation not permitted on an
            # event last recorded in a capturing stream", one capture in three): the warm-up's collectives are complete after
            # the synchronise above -- give the watchdog its next rounds to drop them, so that it holds nothing during the capture.
            #   What is deterministic is done first: every Work handle this trainer still holds has been waited for and is
            # dropped (a step ends with `_early_work = None`: FlatGrads.all_reduce_mean), the device is idle (the synchronise above),
            # so every collective the watchdog lists is COMPLETE and it removes them on its next pass -- which nothing in the public
            # API lets this thread observe or force.  The wait below therefore depends on the watchdog's poll period (100 ms in
            # torch 2.10): five periods by default, PDGN_CAPTURE_QUIESCE_S to change it.  The failure mode is an abort inside the
            # watchdog thread, not an exception: there is no fallback to fall back to, which is why the margin is 5x.
            for fg in (self.gradG, *self.gradD):
                if getattr(fg, "_early_work", None) is not None:
                    fg._early_work.wait()
                    fg._early_work = None
            import time
            time.sleep(float(os.environ.get("PDGN_CAPTURE_QUIESCE_S", "0.5")))
        g = torch.cuda.CUDAGraph(keep_graph=True)            # the recorded graph is read back, never launched
        defer, self._defer_d = self._defer_d, False           # capture order = issue order of the replay: D_k's launches
        self._static["tag_streams"] = True                   # next to the level that feeds them
        rec = replay.Recorder()
        try:
            with rec, torch.cuda.graph(g, capture_error_mode="thread_local"):
                self._step_overlapped(None, None, None, st=self._static)
        finally:
            self._defer_d = defer
            self._static.pop("tag_streams", None)
        release_zero_arena()                                # the capture's arena lives in the graph's private pool
        self._list = replay.LaunchList(g)
        if len(self._list.points) != len(rec.fns):
            raise RuntimeError("capture_list: %d host points recorded, %d found in the list" % (len(rec.fns), len(self._list.points)))
        self._list_points = [(pos, rec.fns[i], label) for pos, i, label in self._list.points]
        # the iteration's last launch on the issuing stream must be behind everything else (the next iteration's input copies and
        # the caller's reads of the losses are ordered against THAT stream only): every chain's last node must reach it
        info = self._list.info
        if info["chains"] != info["labelled"] or not self._list.joined:
            raise RuntimeError("capture_list: %d chains, %d tagged, joined=%s: a launch on an untagged stream, or a side stream whose "
                               "last launch the issuing stream never waits for" % (info["chains"], info["labelled"], self._list.joined))
        self._list_spare = [torch.cuda.Stream(device=self.device) for _ in range(2)]
        self._list_done = None
        self._list_pace = float(os.environ.get("PDGN_LIST_PACE", "1.0"))     # 1.0: the previous iteration's end
        return self

    def step_list(self, reals=None, z1=None, z2=None):
        """One iteration from the launch list of `capture_list`.  Inputs, when given, are copied into the static buffers
        first (ordinary launches on the current stream: plain stream order, no synchronisation needed)."""
        from . import replay
        st = self._static
        if reals is not None:
            for d, s in zip(st["reals"], reals):
                if d.data_ptr() != s.data_ptr():
                    d.copy_(s)
        if z1 is not None:
            st["z1"].copy_(z1)
            st["z2"].copy_(z2)
        pl = _streams.plan(self.device)                      # the eager schedule's streams for the current stream
        streams = {replay.MAIN: torch.cuda.current_stream(self.device), replay.LP: pl.lp, replay.KNN: pl.knn}
        for i, s in enumerate(pl.d):
            streams[replay.D0 + i] = s
        self._list.bind(streams, self._list_spare)
        # One iteration in flight: the list is issued in ~5 ms, the device needs ~6x that.  A host that runs several
        # iterations ahead fills the runtime's queues, blocks inside a launch for ONE stream and starves the others
        # (measured: 30.5 ms/step unbounded, 30.7 with two iterations in flight, 29.2 with one; the eager host's 25 ms
        # per iteration paced it by accident).  The wait is on an event recorded after `_list_pace` of the issuing stream's
        # launches of the previous iteration: 1.0 (the default, the measured best) = its end.
        if self._list_done is not None:
            self._list_done.synchronize()
        cut = self._list.position(replay.MAIN, self._list_pace)
        # the list in ranges: [.. cut) | pacing event | .. and, under data parallelism, every host point's collective behind the
        # range that ends with its marker, on the stream the marker was recorded on
        stops = sorted([(cut, 0, None, None)] + [(pos + 1, 1, fn, label) for pos, fn, label in self._list_points], key=lambda t: t[:2])
        lo = 0
        for at, kind, fn, label in stops:
            if at > lo:
                self._list.launch(lo, at)
                lo = at
            if kind == 0:
                if self._list_done is None:
                    self._list_done = torch.cuda.Event()
                self._list_done.record(streams[replay.MAIN])
            else:
                with torch.cuda.stream(streams[label]):
                    fn()
        self._list.launch(lo)
        return st["out"]

    def _sync(self):
        torch.cuda.current_stream(self.device).synchronize()

    def step_graphed(self, reals=None, z1=None, z2=None):
        """Replay of `capture`.  Inputs, when given, are copied into the static buffers first."""
        st = self._static
        copies = []
        if reals is not None:
            copies += [(d, s) for d, s in zip(st["reals"], reals) if d.data_ptr() != s.data_ptr()]
        if z1 is not None and st["z1"] is not None:
            copies += [(st["z1"], z1), (st["z2"], z2)]
        if copies:
            self._sync()
            for d, s in copies:
                d.copy_(s)
            self._sync()
        for g, last in self._graphs:
            g.replay()
            if self.distributed and last < 5:
                self._sync()
                fg = self.gradD[last] if last < 4 else self.gradG
                fg._early_work, fg._early_done = None, False        # begin() is captured Python: it does not re-run
                self._comm(last)
                self._sync()
        return st["out"]


def synthetic_batch(B, device, seed=9999, n_points=2048, resolutions=(256, 512, 1024, 2048)):
    """ShapeNet-shaped synthetic real clouds: (B,2048,3) ~ N(0,1),
    per-shape `shape_unit` normalisation (datasets_4point.py:335-337, :353), sub-resolutions
    drawn WITH replacement (:374-379), transposed to (B,3,N) (:184)."""
    g = torch.Generator().manual_seed(seed)
    pts = torch.randn(B, n_points, 3, generator=g)
    pts = (pts - pts.mean(dim=1, keepdim=True)) / pts.reshape(B, -1).std(dim=1).view(B, 1, 1)
    reals = []
    for r in resolutions:
        if r == n_points:
            sub = pts
        else:
            sel = torch.randint(0, n_points, (B, r), generator=g)
            sub = torch.gather(pts, 1, sel.unsqueeze(2).expand(B, r, 3))
        reals.append(sub.transpose(1, 2).contiguous().to(device))
    return reals


def noise(B, device, generator=None):
    """z ~ N(0, 0.2^2), (B,128) (:178, :228)."""
    return (torch.randn(B, 128, generator=generator) * 0.2).to(device)
