"""Data-parallel training step over rays: one process per GPU, ``torch.distributed`` ("nccl" == RCCL over xGMI on ROCm;
"gloo" in the CPU tests).

EMAP itself is single-GPU (SURVEY.md par. 2a); this is the sharding the hot path admits (par. 8e): rays are independent
through sampling, MLP and compositing, so every rank renders its slice of ONE global batch with replicated weights.  The
step arithmetic restates reference src/runner/runner_udf.py:90-168 (mask of ones, MSE * edge_weight + igr_ns_weight *
ge_ns + igr_weight * ge, zero_grad / backward / step).

``Trainer`` is the native step (no autograd graph, no per-parameter copies):

  forward     emap_render_fwd on the rank's rays                                   (HIP, no collective)
  stats       [sum(relax), sum(near), sum((edge-gt)^2), sum(relax*err), sum(near*err)]   5 floats
  backward    emap_render_bwd writes dL/dtheta of the rank's share of the GLOBAL loss straight into one flat fp32 buffer laid
              out in parameters() order - the parameters themselves are views of one flat buffer as well
  all-reduce  ONE collective over that buffer (about 1.85 MB for d8 w256; the stats ride in its tail)
  Adam        one launch over the flat buffers (emap_adam_step: torch.optim.Adam's arithmetic with the reference's two parameter
              groups, runner_base.py:110-117); on the CPU (tests) torch.optim.Adam itself
The statistics, dL/d(edge) and the loss scalars are one small kernel each on the GPU (emap_train_stats / emap_train_loss) instead
of ~25 element-wise torch launches - at 2 ms per step their launch latencies were 7 % of it.

The eikonal terms are masked means over ALL rays of the batch (udf_renderer_blending.py:618-625): their denominators must be
the global mask sums for the step to equal the single-GPU step.  ``eikonal_sync="exact"`` (default) therefore all-reduces the 5
stats before the backward (20 bytes, the only other collective of the step); ``eikonal_sync="local"`` uses the rank's own
denominators (a mean of per-rank means - exact whenever the masks cover every sample, as they do for scenes inside the unit
sphere) and the step has exactly one collective.
"""
from __future__ import annotations

from typing import Callable, Dict, Iterable, List, Optional

import torch
import torch.distributed as dist


def shard(t: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """Rank's contiguous slice [rank*N/world, (rank+1)*N/world) of a global per-ray tensor."""
    n = t.shape[0]
    assert n % world == 0, "global ray batch must divide evenly over ranks"
    per = n // world
    return t[rank * per:(rank + 1) * per].contiguous()


def _world(group=None) -> int:
    return dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1


class OneShotAllReduce:
    """SUM all-reduce of a small fp32 bucket by direct peer reads over xGMI (csrc/allreduce.hip; SURVEY.md par. 8e "xGMI note"): every rank
    reads every other rank's staging buffer through ``hipIpcMemHandle`` mappings and sums in rank order - one launch, one synchronisation,
    bit-identical results on all ranks; no host work per call (graph-capturable).  The alternative to RCCL's ring for the step's
    latency-bound 1.85 MB gradient bucket (``Trainer(..., allreduce="oneshot")``, ``bench.py --allreduce oneshot``).

    Construction is collective: the ranks exchange their 64-byte IPC handles through ``torch.distributed`` (any backend, gloo included -
    the ranks may even share one GPU, which is how the one-GPU tests run it).  One node only (IPC handles do not cross hosts)."""

    def __init__(self, n_floats: int, device, group=None):
        import ctypes as C
        from . import _lib
        assert dist.is_available() and dist.is_initialized(), "OneShotAllReduce needs an initialised process group (for the handle exchange)"
        self.group, self.device = group, torch.device(device)
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.n_floats = int(n_floats)
        L = _lib.lib()
        nb = C.c_size_t()
        _lib.check(L.emap_ar_local_bytes(self.n_floats, C.byref(nb)), "ar_local_bytes")
        self.region_bytes = nb.value
        self._own = C.c_void_p()
        handle = (C.c_ubyte * 64)()
        with torch.cuda.device(self.device):
            _lib.check(L.emap_ar_alloc(self.region_bytes, C.byref(self._own), handle), "ar_alloc")
        handles = [None] * self.world
        dist.all_gather_object(handles, bytes(handle), group=group)
        self._peers = []
        regions = (C.c_void_p * self.world)()
        with torch.cuda.device(self.device):
            for r, h in enumerate(handles):
                if r == self.rank:
                    regions[r] = self._own.value
                    continue
                p = C.c_void_p()
                _lib.check(L.emap_ar_open((C.c_ubyte * 64).from_buffer_copy(h), C.byref(p)), "ar_open")
                self._peers.append(p)
                regions[r] = p.value
        self._regions = regions
        dist.barrier(group=group)               # every rank has mapped every region before the first launch touches one

    def __call__(self, t: torch.Tensor) -> torch.Tensor:
        """In place: t <- sum over the ranks of t.  fp32, contiguous, on this device, at most n_floats elements; every rank must call it
        with the same number of elements, in the same order."""
        from . import _lib
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() <= self.n_floats
        with torch.cuda.device(t.device):
            _lib.check(_lib.lib().emap_ar_allreduce_sum(_lib.ptr(t), t.numel(), self.rank, self.world, self._regions, self.region_bytes,
                                                        _lib.stream_ptr(t.device)), "ar_allreduce_sum")
        return t

    @staticmethod
    def set_timeout_ms(ms: int):
        """Bound of one peer wait inside the kernel (default 10 s; x6 for a region's first two launches).  Process-wide."""
        from . import _lib
        _lib.check(_lib.lib().emap_ar_set_timeout_ms(int(ms)), "ar_set_timeout_ms")

    def check(self):
        """Raise if a launch gave up waiting for a peer (host read: synchronises).  Such a launch has also written NaN into its bucket
        (csrc/allreduce.hip step 4), so the loss and the parameters of this rank are NaN from that step on - the time-out cannot pass
        unnoticed even where nobody calls this."""
        import ctypes as C
        from . import _lib
        e = C.c_int()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().emap_ar_error(self._own, C.byref(e)), "ar_error")
        if e.value:
            raise RuntimeError("OneShotAllReduce: a peer's buffer did not arrive within the kernel's time-out (a rank died or the ranks do "
                               "not call the collective in lock step)")

    def close(self):
        from . import _lib
        L = _lib.lib()
        if getattr(self, "_own", None) is None:
            return
        try:
            torch.cuda.synchronize(self.device)
            dist.barrier(group=self.group)       # nobody unmaps while a peer may still read
        except Exception:
            pass
        with torch.cuda.device(self.device):
            for p in self._peers:
                L.emap_ar_close(p)
            L.emap_ar_free(self._own)
        self._peers, self._own = [], None


class FlatParams:
    """Re-homes `params` as views of ONE flat fp32 buffer (and their .grad as views of one flat gradient buffer), so that a
    gradient all-reduce and an optimizer step touch one tensor instead of one per parameter."""

    def __init__(self, params: Iterable[torch.nn.Parameter], extra: int = 0):
        self.params: List[torch.nn.Parameter] = list(params)
        dev = self.params[0].device
        self.numel = sum(p.numel() for p in self.params)
        self.data = torch.empty(self.numel, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(self.numel + extra, dtype=torch.float32, device=dev)   # tail: step statistics
        off = 0
        self.offsets = {}
        with torch.no_grad():
            for p in self.params:
                n = p.numel()
                self.data[off:off + n].copy_(p.detach().reshape(-1))
                p.data = self.data[off:off + n].view(p.shape)
                p.grad = self.grad[off:off + n].view(p.shape)
                self.offsets[id(p)] = off
                off += n

    def span(self, params: Iterable[torch.nn.Parameter]):
        """(start, end) of a run of parameters that are adjacent in the flat buffer."""
        ps = list(params)
        s = self.offsets[id(ps[0])]
        e = self.offsets[id(ps[-1])] + ps[-1].numel()
        assert e - s == sum(p.numel() for p in ps), "parameters are not contiguous in the flat buffer"
        return s, e


class _AdamGroups:
    """What the step reads from `Trainer.optimizer` when the update runs in emap_adam_step: the two parameter groups of the
    reference (runner_base.py:110-117) with the keys its schedulers touch (`g["lr"] = ...`, runner_base.py:140-141,159-160)."""

    def __init__(self, trainer, lr_geo, lr, betas=(0.9, 0.999), eps=1e-8):
        self._t = trainer
        self.param_groups = [{"params": [trainer.p_geo], "lr": lr_geo, "betas": betas, "eps": eps},
                             {"params": [trainer.p_sc], "lr": lr, "betas": betas, "eps": eps}]

    def zero_grad(self, set_to_none: bool = False):       # the backward overwrites the flat gradient buffer
        pass

    def step(self):
        self._t._native_adam()


class Trainer:
    """Native data-parallel training step of the render hot path (see the module docstring)."""

    N_STATS = 8          # tail of the gradient bucket: [0, 5) the step statistics ("local"); then 2 range maxima per rank ("exact_lagged")
    MAX_RANKS = 16

    def __init__(self, renderer, lr_geo: float = 1e-4, lr: float = 5e-4, edge_weight: float = 1.0, igr_weight: float = 0.1,
                 igr_ns_weight: float = 0.0, group=None, eikonal_sync: str = "exact", fused_adam: Optional[bool] = None,
                 native_tail: Optional[bool] = None, allreduce: str = "rccl"):
        assert eikonal_sync in ("exact", "exact_lagged", "local")
        # only "exact_lagged" uses the rank-indexed maxima slots of the bucket's tail: "exact" / "local" run on any number of ranks
        assert eikonal_sync != "exact_lagged" or _world(group) <= self.MAX_RANKS, \
            f"eikonal_sync='exact_lagged' keeps two range maxima per rank in the gradient bucket's tail: at most {self.MAX_RANKS} ranks"
        self.r = renderer
        self.group = group
        self.eikonal_sync = eikonal_sync
        self.edge_weight, self.igr_weight, self.igr_ns_weight = float(edge_weight), float(igr_weight), float(igr_ns_weight)
        net = renderer.udf_network
        self.geo = list(net.parameters())
        self.scalars = [renderer.deviation_network.variance, renderer.beta_network.beta, renderer.beta_network.gamma]
        # tail of the bucket: the step statistics; "exact_lagged" only: + two range maxima per rank
        self.flat = FlatParams(self.geo + self.scalars, extra=self.N_STATS + (2 * self.MAX_RANKS if eikonal_sync == "exact_lagged" else 0))
        renderer._lay = None                      # the layout caches tensor identities / pointers: rebuild on the flat views
        lay = renderer._layout()
        assert lay.numel == self.flat.numel and all(lay.offsets[id(p)] == self.flat.offsets[id(p)] for p in self.flat.params)
        g0, g1 = self.flat.span(self.geo)
        s0, s1 = self.flat.span(self.scalars)
        # two flat "parameters" = the reference's two Adam groups (runner_base.py:110-117)
        self.p_geo = torch.nn.Parameter(self.flat.data[g0:g1])
        self.p_sc = torch.nn.Parameter(self.flat.data[s0:s1])
        self.p_geo.grad = self.flat.grad[g0:g1]
        self.p_sc.grad = self.flat.grad[s0:s1]
        dev = self.flat.data.device
        if fused_adam is None:
            fused_adam = dev.type == "cuda"
        # native tail (GPU default): statistics / loss scalars / Adam as three small HIP kernels (csrc/train.hip)
        if native_tail is None:      # native_tail=False: torch element-wise ops + torch.optim.Adam (the CPU tests; A/B 2.12 vs 2.05 ms)
            native_tail = dev.type == "cuda"
        self.native_tail = bool(native_tail)
        if self.native_tail:
            assert g0 == 0 and s0 == g1 and s1 == self.flat.numel, "flat layout: geometry parameters first, then the scalars"
            self._n_geo = g1
            self._m = torch.zeros(self.flat.numel, device=dev)
            self._v = torch.zeros(self.flat.numel, device=dev)
            self._adam_t = torch.zeros(1, device=dev)          # step counter on the device: the update is graph-capturable
            # the scalars' trainable mask and their own step counts live on the device too (emap_adam_step_masked): no host index tensors
            # in the step, nothing a captured graph could bake in.  The mask is refreshed from requires_grad at every step() / replay().
            self._tail_mask = torch.ones(self.flat.numel - g1, device=dev)
            self._tail_step = torch.zeros(self.flat.numel - g1, device=dev)
            self._tail_flags = None
            self._mask_pinned = torch.ones(self.flat.numel - g1).pin_memory() if dev.type == "cuda" else None
            self._stats = torch.zeros(5, device=dev)
            self.optimizer = _AdamGroups(self, lr_geo, lr)
        else:
            # capturable: the step counters live on the device, so a whole step can be captured in a hipGraph (capture())
            self.optimizer = torch.optim.Adam([{"params": [self.p_geo], "lr": lr_geo}, {"params": [self.p_sc]}], lr=lr,
                                              **({"fused": True, "capturable": True} if fused_adam else {}))
        # steady state: "exact" 3 (statistics SUM, range maxima MAX, gradients), "exact_lagged" 2 (the maxima ride in the gradient bucket
        # and are used one step late; the very first step takes the exact path), "local" 1
        self.collectives_per_step = 0 if _world(group) == 1 else {"exact": 3, "exact_lagged": 2, "local": 1}[eikonal_sync]
        self._lag = torch.zeros(2, device=dev)      # exact_lagged: the GLOBAL range maxima of the previous step
        self._lag_valid = False
        # "local": every rank normalises by its own mask sums, so the sum over ranks needs the 1/world of a mean of means
        k = 1.0 / _world(group) if eikonal_sync == "local" else 1.0
        self._igr = torch.tensor([self.igr_weight * k], device=dev)
        self._igr_ns = torch.tensor([self.igr_ns_weight * k], device=dev)
        self._idx = torch.tensor([4, 6, 3, 5], device=dev)
        self.last_stats = None
        # the gradient bucket's collective: "rccl" = torch.distributed.all_reduce on the group's backend (RCCL / gloo);
        # "oneshot" = direct peer reads over hipIpc mappings (OneShotAllReduce; one node, GPU only).  The 20-byte statistics / 8-byte
        # maxima exchanges of the exact modes stay on torch.distributed either way.
        assert allreduce in ("rccl", "oneshot")
        self.allreduce = allreduce
        self._oneshot = None
        if allreduce == "oneshot" and _world(group) > 1:
            self._oneshot = OneShotAllReduce(self.flat.grad.numel(), dev, group)

    # ---- the three device stages; the CPU tests substitute oracle implementations for the two HIP ones ----
    def _forward(self, rays):
        r = self.r
        call = r._prepare(rays["rays_o"], rays["rays_d"], rays["near"], rays["far"], rays.get("depth_scale"), rays.get("cos_anneal_ratio"),
                          rays.get("perturb_overwrite", -1), rays.get("background_rgb"), rays.get("flip_saturation", 0.0), rays.get("t_rand"))
        v = r._render_hip(call)
        return call, v, v["edge"], v["scalars"]

    def _backward(self, call, v, d_edge, scalars_glob, flat_grad):
        self.r.backward_into(call, v, d_edge, None, self._igr, self._igr_ns if self.igr_ns_weight != 0.0 else None,
                             flat=flat_grad, scalars=scalars_glob)

    def _scalar_flags(self):
        for p in self.geo:
            if not p.requires_grad:
                raise NotImplementedError("Trainer: a UDF network parameter with requires_grad=False is not supported")
        return tuple(bool(p.requires_grad) for p in self.scalars)

    def check_errors(self):
        """Host-side health check (synchronises): the renderer's device error word and - ``allreduce="oneshot"`` - the collective's
        time-out word.  Call it wherever the loop reads the loss anyway (bench.py and scripts/train_synthetic.py do, every report)."""
        self.r.check_errors()
        if self._oneshot is not None:
            self._oneshot.check()

    def close(self):
        """Check for a timed-out collective one last time and unmap the peers' regions (collective: every rank calls it)."""
        one, self._oneshot = self._oneshot, None
        if one is not None:
            try:
                one.check()
            finally:
                one.close()

    def refresh_trainable_mask(self, capturing: bool = False):
        """Mirror `requires_grad` of variance / beta / gamma (runner_udf.py:141-154 flips them during training) into the device mask
        of the fused Adam.  Eager: called by every step.  Under a captured graph the mask BUFFER is what the graph reads, so a
        refresh between replays takes effect without re-capturing."""
        flags = self._scalar_flags()
        if flags == self._tail_flags:
            return False
        if capturing:
            raise RuntimeError("Trainer: requires_grad of variance / beta / gamma changed inside a graph capture")
        vals = []
        for p, f in zip(self.scalars, flags):
            vals += [1.0 if f else 0.0] * p.numel()
        if self._mask_pinned is not None:
            self._mask_pinned.copy_(torch.tensor(vals))
            self._tail_mask.copy_(self._mask_pinned, non_blocking=True)
        else:
            self._tail_mask.copy_(torch.tensor(vals))
        self._tail_flags = flags
        return True

    def _native_adam(self):
        from . import _lib
        g0, g1 = self.optimizer.param_groups
        dev = self.flat.data.device
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().emap_adam_step_masked(_lib.ptr(self.flat.data), _lib.ptr(self.flat.grad), _lib.ptr(self._m), _lib.ptr(self._v),
                                                        _lib.ptr(self._adam_t), self.flat.numel, self._n_geo, float(g0["lr"]), float(g1["lr"]),
                                                        float(g0["betas"][0]), float(g0["betas"][1]), float(g0["eps"]),
                                                        _lib.ptr(self._tail_mask), _lib.ptr(self._tail_step), _lib.stream_ptr(dev)),
                       "adam_step")

    # The native step is written as four device phases separated by the (at most three) collectives, so that it can run eagerly, be
    # captured whole in one hipGraph (one rank) or be captured phase by phase with the collectives launched between the replays
    # (any backend, also gloo whose collectives are host code): capture(segmented=True).
    def _ph_forward(self, S, rays, true_edge, n_rays_global):
        from . import _lib
        L = _lib.lib()
        world = _world(self.group)
        call, v, edge, scalars = self._forward(rays)
        dev = edge.device
        n_local = edge.numel()
        S.update(call=call, v=v, scalars=scalars, dev=dev, world=world, n_local=n_local,
                 n_glob=n_rays_global if n_rays_global is not None else n_local * world)
        te = true_edge.reshape(-1).to(torch.float32).contiguous()
        assert te.numel() == n_local
        S["d_edge"] = torch.empty(n_local, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.emap_train_stats(_lib.ptr(edge), _lib.ptr(te), _lib.ptr(scalars), n_local, 2.0 * self.edge_weight / S["n_glob"],
                                          _lib.ptr(S["d_edge"]), _lib.ptr(self._stats), _lib.stream_ptr(dev)), "train_stats")
        S["stats"] = self._stats

    def _ph_composite_bwd(self, S):
        sc_glob = S["scalars"]
        exactish = self.eikonal_sync in ("exact", "exact_lagged")
        if S["world"] > 1 and exactish:      # self._stats now holds the GLOBAL sums
            sc_glob = S["scalars"].clone()
            sc_glob[4], sc_glob[6] = self._stats[0], self._stats[1]
        S["sc_glob"] = sc_glob
        g = self.flat.grad
        both = not (S["world"] > 1 and exactish)
        self.r.backward_into(S["call"], S["v"], S["d_edge"], None, self._igr, self._igr_ns if self.igr_ns_weight != 0.0 else None,
                             flat=g[:self.flat.numel], scalars=sc_glob, stages=3 if both else 1)
        S["staged"] = not both
        S["lagged"] = S["staged"] and self.eikonal_sync == "exact_lagged"
        if S["lagged"]:
            # The two range maxima of this rank go into its slots of the bucket's tail (the SUM all-reduce of the bucket then works as an
            # all-gather of them: every other slot is zero) and become next step's GLOBAL maxima.  THIS step's sweep uses the previous
            # step's global maxima x 4 (the kernel rounds to a power of two) instead of waiting for a MAX all-reduce - or the rank's own
            # maxima where they exceed that, so that fp16 can never overflow: a rank whose gradients grew more than 4x in one step then
            # differs from the others in rounding for that step only.  The first step (no history) takes the exact path.
            self._lag_publish(self.r.bwd_absmax(S["call"]), dist.get_rank(self.group))

    def _lag_publish(self, cur: torch.Tensor, rank: int):
        """exact_lagged, before the sweep: write this rank's two range maxima `cur` into ITS two slots of the bucket's tail (all other
        slots zero: the bucket's SUM all-reduce is then an all-gather of them) and turn `cur` IN PLACE into the scale this step's sweep
        uses - 4 x last step's GLOBAL maxima, clamped to [own maxima, 16 x own maxima]: the lower clamp keeps fp16 from overflowing
        when the gradients GREW more than 4 x in one step, the upper one from flushing small adjoints when they SHRANK more than 4 x
        (a stale scale 1000 x too large leaves fp16 three binades).  Inside the clamp every rank uses the same number; at a
        clamp the ranks differ in rounding for that step only.  Without history (first step) `cur` is left alone."""
        tail = self._maxima_tail()
        tail.zero_()
        tail[2 * rank:2 * rank + 2] = cur
        if self._lag_valid:
            cur.copy_(torch.minimum(torch.maximum(self._lag * 4.0, cur), cur * 16.0))

    def _lag_collect(self):
        """exact_lagged, after the bucket's all-reduce: the gathered per-rank maxima of THIS step become next step's global maxima."""
        if self._lag_valid:
            self._lag.copy_(self._maxima_tail().view(self.MAX_RANKS, 2).max(dim=0).values)
        self._lag_valid = True

    def _maxima_tail(self):
        n = self.flat.numel + self.N_STATS
        return self.flat.grad[n:n + 2 * self.MAX_RANKS]

    def _ph_mlp_bwd(self, S):
        g = self.flat.grad
        if S["lagged"] and not self._lag_valid:
            self._lag.copy_(self.r.bwd_absmax(S["call"]))       # first step: the MAX all-reduce just ran; it seeds the history
        if S["staged"]:
            self.r.backward_into(S["call"], S["v"], S["d_edge"], None, self._igr, self._igr_ns if self.igr_ns_weight != 0.0 else None,
                                 flat=g[:self.flat.numel], scalars=S["sc_glob"], stages=2)
        if S["world"] > 1 and self.eikonal_sync == "local":
            g[self.flat.numel:self.flat.numel + 5] = self._stats      # the statistics ride in the bucket's tail

    def _ph_update(self, S):
        from . import _lib
        L = _lib.lib()
        dev = S["dev"]
        stats = self._stats
        if S["world"] > 1 and self.eikonal_sync == "local":
            stats = self.flat.grad[self.flat.numel:self.flat.numel + 5]
        if S["lagged"]:
            self._lag_collect()
        self.optimizer.step()      # frozen scalars are skipped inside the kernel (device mask, refresh_trainable_mask)
        self.r.udf_network.invalidate_packed()   # the flat update does not bump the per-parameter version counters
        out = torch.empty(2, device=dev)         # a fresh tensor per step: callers keep what step() returned
        with torch.cuda.device(dev):
            _lib.check(L.emap_train_loss(_lib.ptr(stats), self.edge_weight / S["n_glob"], self.igr_weight, self.igr_ns_weight, _lib.ptr(out),
                                         _lib.stream_ptr(dev)), "train_loss")
        self.last_stats = out
        return out

    def _collectives(self):
        """The collectives of one step, in order, as (after_phase, callable): exact = [global eikonal mask sums + MSE sum (20 B),
        the two range maxima of the MLP backward (8 B, MAX), the flat gradient]; local = [the flat gradient with the statistics in
        its tail].  The maxima exchange makes every rank's backward use the SAME fp16 range scale, so the step does not depend on
        how the rays are sharded (beyond the order of the floating-point sums)."""
        if _world(self.group) == 1:
            return []
        ar = lambda t, op: (lambda S: dist.all_reduce(t(S), op=op, group=self.group))
        grad = ar(lambda S: self.flat.grad, dist.ReduceOp.SUM) if self._oneshot is None else (lambda S: self._oneshot(self.flat.grad))
        if self.eikonal_sync == "local":
            return [(2, grad)]
        stats = (0, ar(lambda S: self._stats, dist.ReduceOp.SUM))
        if self.eikonal_sync == "exact_lagged" and self._lag_valid:      # the maxima ride in the gradient bucket's tail
            return [stats, (2, grad)]
        return [stats, (1, ar(lambda S: self.r.bwd_absmax(S["call"]), dist.ReduceOp.MAX)), (2, grad)]

    def _step_native(self, rays: Dict, true_edge: torch.Tensor, n_rays_global: Optional[int]):
        S = {}
        coll = self._collectives()
        phases = [lambda: self._ph_forward(S, rays, true_edge, n_rays_global), lambda: self._ph_composite_bwd(S),
                  lambda: self._ph_mlp_bwd(S), lambda: self._ph_update(S)]
        out = None
        for i, ph in enumerate(phases):
            out = ph()
            for after, fn in coll:
                if after == i:
                    fn(S)
        return out

    def step(self, rays: Dict, true_edge: torch.Tensor, n_rays_global: Optional[int] = None):
        """One optimizer step on this rank's rays.  Returns the device tensor [loss, edge_loss] of the GLOBAL batch (no host
        synchronisation happens here)."""
        if self.native_tail:
            self.refresh_trainable_mask(capturing=torch.cuda.is_current_stream_capturing())
            return self._step_native(rays, true_edge, n_rays_global)
        world = _world(self.group)
        call, v, edge, scalars = self._forward(rays)
        n_local = edge.numel()
        n_glob = n_rays_global if n_rays_global is not None else n_local * world
        te = true_edge.reshape(-1).to(edge.dtype)
        diff = edge.reshape(-1) - te
        # stats: [sum(relax), sum(near), sum(relax*err), sum(near*err), sum(diff^2)] - scalars[3:7] = e_rel, c_rel, e_ns, c_ns
        stats = torch.cat([scalars[self._idx], (diff * diff).sum().reshape(1)])
        sc_glob = scalars
        if world > 1 and self.eikonal_sync in ("exact", "exact_lagged"):      # (no fp16 range scale on this path: lagged = exact)
            dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=self.group)
            sc_glob = scalars.clone()
            sc_glob[4], sc_glob[6] = stats[0], stats[1]
        d_edge = diff * (2.0 * self.edge_weight / n_glob)
        g = self.flat.grad
        self._backward(call, v, d_edge, sc_glob, g[:self.flat.numel])
        if world > 1:
            if self.eikonal_sync == "local":
                g[self.flat.numel:self.flat.numel + 5] = stats      # the statistics ride in the bucket's tail
            dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group)
            if self.eikonal_sync == "local":
                stats = g[self.flat.numel:self.flat.numel + 5].clone()
        # torch.optim.Adam on the two flat parameters cannot skip single elements: a frozen scalar (requires_grad = False,
        # runner_udf.py:144-154) is put back and its moments cleared (CPU tests only; the native path masks inside the kernel)
        frozen = [i for p in self.scalars if not p.requires_grad
                  for i in range(self.flat.offsets[id(p)] - self.flat.offsets[id(self.scalars[0])], self.flat.offsets[id(p)] - self.flat.offsets[id(self.scalars[0])] + p.numel())]
        keep = self.p_sc.data[frozen].clone() if frozen else None
        self.optimizer.step()
        if frozen:
            self.p_sc.data[frozen] = keep
            st = self.optimizer.state.get(self.p_sc, {})
            for k in ("exp_avg", "exp_avg_sq"):
                if k in st:
                    st[k][frozen] = 0.0
        self.r.udf_network.invalidate_packed()   # the flat update does not bump the per-parameter version counters
        edge_loss = stats[4] / n_glob * self.edge_weight
        loss = edge_loss + self.igr_weight * stats[2] / (stats[0] + 1e-5) + self.igr_ns_weight * stats[3] / (stats[1] + 1e-5)
        self.last_stats = torch.stack([loss, edge_loss])
        return self.last_stats


    def capture(self, rays: Dict, true_edge: torch.Tensor, n_rays_global: Optional[int] = None, warmup: int = 3,
                segmented: Optional[bool] = None):
        """Capture one whole step (pack -> render forward -> statistics -> HIP backward -> [all-reduce] -> Adam) for this batch
        shape and return ``replay(rays=None, true_edge=None) -> [loss, edge_loss]`` (device tensor, static).  New rays / targets
        are copied into the static buffers before the replay.  `warmup` real steps are taken first (workspaces, function
        attributes, Adam state must exist before capturing).

        One rank: ONE hipGraph.  Several ranks (`segmented`, default then): one hipGraph per device phase and the collectives
        launched between the replays - nothing about the transport is assumed (works with RCCL and with gloo, whose collectives
        are host code and cannot be captured); `segmented=False` puts the collectives inside one graph (RCCL only)."""
        if not self.native_tail:
            if segmented:
                raise ValueError("Trainer.capture(segmented=True) needs the native tail (native_tail=True): the per-phase graphs are the "
                                 "native step's four device phases")
            segmented = False
        world = _world(self.group)
        if segmented is None:
            segmented = world > 1
        if rays.get("t_rand") is None and (rays.get("perturb_overwrite", -1) != 0) and self.r.perturb > 0:
            raise ValueError("Trainer.capture: pass rays['t_rand'] explicitly (the reference's CPU-generator jitter draw plus its "
                             "host-to-device copy cannot be part of a device graph)")
        dev = self.flat.data.device
        static = {k: (v.detach().clone() if isinstance(v, torch.Tensor) else v) for k, v in rays.items()}
        te = true_edge.detach().clone()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(warmup, 1)):
                self.step(static, te, n_rays_global)
        torch.cuda.current_stream(dev).wait_stream(side)
        if not segmented:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = self.step(static, te, n_rays_global)
            graphs, coll, S = [graph], [], None
        else:
            S = {}
            coll = self._collectives()
            phases = [lambda: self._ph_forward(S, static, te, n_rays_global), lambda: self._ph_composite_bwd(S),
                      lambda: self._ph_mlp_bwd(S), lambda: self._ph_update(S)]
            graphs, out = [], None
            pool = None
            for i, ph in enumerate(phases):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool):
                    out = ph()
                pool = g.pool()
                graphs.append(g)
                g.replay()                  # capture does not execute: run the phase now, so that this pass is one real step and
                for after, fn in coll:      # the next phase is captured on real data
                    if after == i:
                        fn(S)
        keep = self.r.live_buffers()

        def replay(rays: Optional[Dict] = None, true_edge: Optional[torch.Tensor] = None):
            if self.native_tail:
                self.refresh_trainable_mask()     # a set_trainable() between replays reaches the captured Adam through its mask buffer
            if rays is not None:
                for k, v in rays.items():
                    if isinstance(v, torch.Tensor):
                        static[k].copy_(v)
            if true_edge is not None:
                te.copy_(true_edge)
            for i, g in enumerate(graphs):
                g.replay()
                for after, fn in coll:
                    if after == i:
                        fn(S)
            return out

        replay.graph = graphs[0] if len(graphs) == 1 else None
        replay.graphs = graphs
        replay.segmented = bool(segmented)
        replay._keep = keep
        return replay


class FusedAdam(torch.optim.Optimizer):
    """``torch.optim.Adam`` for the drop-in's training loop (runner_base.py:106-117) as ONE kernel launch per step.

    Same constructor shape as ``torch.optim.Adam([{"params": geo, "lr": lr_geo}, {"params": rest}, ...], lr=lr)``: the parameters of
    the first non-empty group become the "geo" range, all later ones the tail.  On the first ``step()`` every parameter is re-homed as a
    view of one flat fp32 buffer (their values, ``requires_grad`` flags and identities are kept - the modules never notice); a
    step then gathers the gradients into one flat buffer (one ``torch.cat``) and calls ``emap_adam_step_masked``: a parameter whose
    ``.grad`` is None (frozen: ``variance`` before ``set_trainable()``) is skipped exactly as torch's Adam skips it and its step
    count starts when it first gets a gradient.  ``param_groups[i]["lr"]`` is read at every step, so the runner's schedulers work
    unchanged.  Replaces 32 tensors x ~6 element-wise launches of the stock optimizer."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        if not self._live_groups():
            raise ValueError("FusedAdam: no parameters")
        if len({g["lr"] for g in self._tail_groups}) > 1:
            raise ValueError("FusedAdam: the groups behind the first one must share one learning rate (the runner's do)")
        self._flat = None

    # The groups are looked up in ``self.param_groups`` at every use: ``Optimizer.load_state_dict`` / ``__setstate__`` REPLACE the group
    # dicts, and the runner's schedulers write ``lr`` into whatever dicts ``optimizer.param_groups`` holds at that time
    # (runner_base.py:128-150) - a reference bound in __init__ would keep reading the construction-time learning rate after a resume.
    def _live_groups(self):
        return [g for g in self.param_groups if len(g["params"])]

    @property
    def _geo_group(self):
        return self._live_groups()[0]

    @property
    def _tail_groups(self):
        return self._live_groups()[1:]

    @classmethod
    def from_adam(cls, adam: torch.optim.Optimizer) -> "FusedAdam":
        """A FusedAdam over the SAME parameter groups as a ``torch.optim.Adam`` (runner_base.py:106-117), taking over its learning rates,
        betas, eps and - if it has stepped already - its state (checkpoint layout, load_state_dict)."""
        groups = []
        for g in adam.param_groups:
            if g.get("amsgrad") or g.get("weight_decay") or g.get("maximize"):
                raise NotImplementedError("FusedAdam.from_adam: amsgrad / weight_decay / maximize are not implemented")
            groups.append({"params": list(g["params"]), "lr": g["lr"], "betas": tuple(g["betas"]), "eps": g["eps"]})
        g0 = next(g for g in groups if g["params"])
        opt = cls(groups, lr=g0["lr"], betas=g0["betas"], eps=g0["eps"])
        if len(adam.state):
            opt.load_state_dict(adam.state_dict())
        return opt

    def _build(self):
        geo = list(self._geo_group["params"])
        tail = [p for g in self._tail_groups for p in g["params"]]
        grads = [p.grad for p in geo + tail]    # FlatParams points .grad at its own flat buffer; this optimizer takes the gradients autograd
        self._flat = FlatParams(geo + tail)     # (or the caller) put there, so they are put back
        for p, g_ in zip(geo + tail, grads):
            p.grad = g_
        dev = self._flat.data.device
        self._n_geo = sum(p.numel() for p in geo)
        self._geo, self._tail = geo, tail
        n = self._flat.numel
        self._m, self._v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        self._t = torch.zeros(1, device=dev)
        self._tail_mask = torch.ones(max(n - self._n_geo, 1), device=dev)
        self._tail_step = torch.zeros(max(n - self._n_geo, 1), device=dev)
        self._flags = None
        self._zeros = {}

    # ---- checkpointing (runner_udf.py:260 saves optimizer.state_dict(), :273 loads it): torch.optim.Adam's per-parameter layout ----
    def _span(self, p):
        o = self._flat.offsets[id(p)]
        return o, o + p.numel()

    def _mirror_state(self):
        """Expose the flat moments as ``self.state[p] = {step, exp_avg, exp_avg_sq}`` (views of the flat buffers, torch.optim.Adam's
        keys), so that ``Optimizer.state_dict()`` writes a checkpoint a stock Adam can load and vice versa.  A parameter that has
        never been stepped (frozen so far) has no entry, exactly like torch's Adam.  Reads the step counters: a host sync."""
        t_geo = float(self._t.item())
        tail_steps = self._tail_step.tolist()
        self.state.clear()
        for p in self._geo + self._tail:
            a, b = self._span(p)
            step = t_geo if a < self._n_geo else float(tail_steps[a - self._n_geo])
            if step <= 0:
                continue
            self.state[p] = {"step": torch.tensor(step, dtype=torch.float32), "exp_avg": self._m[a:b].view(p.shape),
                             "exp_avg_sq": self._v[a:b].view(p.shape)}

    def state_dict(self):
        if self._flat is not None:
            self._mirror_state()
        return super().state_dict()

    @torch.no_grad()
    def load_state_dict(self, state_dict):
        """Accepts a checkpoint of this class or of ``torch.optim.Adam`` over the same parameter groups: moments and step counts go into
        the flat buffers (the geometry range has ONE step counter - its parameters always step together; every tail element its own)."""
        super().load_state_dict(state_dict)           # validates the groups, casts the tensors to the parameters' device
        if self._flat is None:
            self._build()
        self._m.zero_(); self._v.zero_(); self._t.zero_(); self._tail_step.zero_()
        t_geo = 0.0
        for p in self._geo + self._tail:
            st = self.state.get(p)
            if not st:
                continue
            a, b = self._span(p)
            self._m[a:b].copy_(st["exp_avg"].reshape(-1))
            self._v[a:b].copy_(st["exp_avg_sq"].reshape(-1))
            step = float(st["step"])
            if a < self._n_geo:
                t_geo = max(t_geo, step)
            else:
                self._tail_step[a - self._n_geo:b - self._n_geo] = step
        self._t.fill_(t_geo)
        self._flags = None
        self._mirror_state()

    @torch.no_grad()
    def step(self, closure=None):
        from . import _lib
        loss = closure() if closure is not None else None
        if self._flat is None:
            self._build()
        dev = self._flat.data.device
        parts = []
        # the geometry gradients may already BE one flat buffer (RenderFn.backward with direct_param_grads, or a Trainer-style caller):
        # consecutive fp32 views in this optimizer's order -> no gather
        geo_ptr, off = None, 0
        for p in self._geo:
            g_ = p.grad
            if g_ is None:
                raise NotImplementedError("FusedAdam: a parameter of the first group without gradient")
            if off == 0:
                geo_ptr = g_.data_ptr()
            if geo_ptr is not None and (g_.dtype != torch.float32 or not g_.is_contiguous() or g_.data_ptr() != geo_ptr + 4 * off):
                geo_ptr = None
            off += p.numel()
        if geo_ptr is None:
            parts = [p.grad.reshape(-1) for p in self._geo]
        flags = []
        for p in self._tail:
            has = p.grad is not None
            flags += [1.0 if has else 0.0] * p.numel()
            if has:
                parts.append(p.grad.reshape(-1))
            else:
                z = self._zeros.get(p.numel())
                if z is None:
                    z = self._zeros[p.numel()] = torch.zeros(p.numel(), device=dev)
                parts.append(z)
        if flags != self._flags:                # rare: a set_trainable() of the runner
            self._tail_mask[:len(flags)].copy_(torch.tensor(flags))
            self._flags = flags
        live = self._live_groups()              # the CURRENT group dicts (see _live_groups): lr is read per step
        geo_group, tail_groups = live[0], live[1:]
        lr_tail = tail_groups[0]["lr"] if tail_groups else geo_group["lr"]
        b1, b2 = geo_group["betas"]
        L, st = _lib.lib(), _lib.stream_ptr(dev)
        with torch.cuda.device(dev):
            if geo_ptr is None:
                grad = torch.cat(parts)
                _lib.check(L.emap_adam_step_masked(_lib.ptr(self._flat.data), _lib.ptr(grad), _lib.ptr(self._m), _lib.ptr(self._v),
                                                   _lib.ptr(self._t), self._flat.numel, self._n_geo, float(geo_group["lr"]),
                                                   float(lr_tail), float(b1), float(b2), float(geo_group["eps"]),
                                                   _lib.ptr(self._tail_mask), _lib.ptr(self._tail_step), st), "adam_step")
            else:
                # two launches on disjoint ranges, same arithmetic: the geometry range straight from the caller's flat gradient buffer
                # (its step counter self._t is bumped by this call), the few tail scalars gathered as before (their own step counts)
                import ctypes as C
                ng, nt = self._n_geo, self._flat.numel - self._n_geo
                _lib.check(L.emap_adam_step(_lib.ptr(self._flat.data), C.c_void_p(geo_ptr), _lib.ptr(self._m), _lib.ptr(self._v),
                                            _lib.ptr(self._t), ng, ng, float(geo_group["lr"]), float(lr_tail), float(b1), float(b2),
                                            float(geo_group["eps"]), st), "adam_step")
                if nt > 0:
                    gt = torch.cat(parts)
                    if getattr(self, "_t_tail_dummy", None) is None:
                        self._t_tail_dummy = torch.zeros(1, device=dev)
                    _lib.check(L.emap_adam_step_masked(_lib.ptr(self._flat.data[ng:]), _lib.ptr(gt), _lib.ptr(self._m[ng:]), _lib.ptr(self._v[ng:]),
                                                       _lib.ptr(self._t_tail_dummy), nt, 0, float(geo_group["lr"]), float(lr_tail),
                                                       float(b1), float(b2), float(geo_group["eps"]), _lib.ptr(self._tail_mask),
                                                       _lib.ptr(self._tail_step), st), "adam_step")
        # the in-place flat update is invisible to the per-tensor version counters UDFNetwork.packed() keys its fragment cache on
        inc = getattr(torch.autograd.graph, "increment_version", None)
        for p in self._geo:
            if inc is not None:
                inc(p)
            else:
                p.add_(0.0)
        return loss


# ---------------------------------------------------------------------------------------------
# generic autograd-based step (any differentiable render_fn; used with the drop-in classes and by the CPU tests)
# ---------------------------------------------------------------------------------------------
def training_step(render_fn: Callable[[], Dict[str, torch.Tensor]], true_edge: torch.Tensor, flat: FlatParams, optimizer,
                  edge_weight: float = 1.0, igr_weight: float = 0.1, igr_ns_weight: float = 0.0, group=None,
                  n_rays_global: Optional[int] = None):
    """One optimizer step on this rank's ray shard through autograd; equals the single-process step on the global batch.

    render_fn() -> render dict for this rank's rays (must contain "edge", "gradient_error", "gradient_error_near_surface" and
    "eikonal_sums" = [sum(relax*err), sum(relax), sum(near*err), sum(near)]).  `flat` holds the parameters (FlatParams): the
    gradients accumulate into its flat buffer and ONE all-reduce covers them.  Returns (loss_global, edge_loss_global)."""
    world = _world(group)
    out = render_fn()
    edge = out["edge"]
    n_local = edge.shape[0]
    n_glob = n_rays_global if n_rays_global is not None else n_local * world
    sums = out["eikonal_sums"].detach()
    mse_sum = ((edge - true_edge) ** 2).sum()
    stats = torch.stack([sums[1], sums[3], mse_sum.detach(), sums[0], sums[2]]).to(torch.float32)
    if world > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)   # global mask counts before backward (+ the loss statistics)
    # local numerators with gradient: ge_local * (c_local + 1e-5)
    e_rel = out["gradient_error"] * (sums[1] + 1e-5)
    e_ns = out["gradient_error_near_surface"] * (sums[3] + 1e-5)
    n_elem_glob = n_glob * (edge.numel() // n_local)
    loss_local = mse_sum / n_elem_glob * edge_weight + e_ns / (stats[1] + 1e-5) * igr_ns_weight + e_rel / (stats[0] + 1e-5) * igr_weight
    flat.grad.zero_()
    for p in flat.params:
        p.grad = flat.grad[flat.offsets[id(p)]:flat.offsets[id(p)] + p.numel()].view(p.shape)   # autograd accumulates in place
    loss_local.backward()
    if world > 1:
        dist.all_reduce(flat.grad, op=dist.ReduceOp.SUM, group=group)
    optimizer.step()
    edge_loss = stats[2] / n_elem_glob * edge_weight
    loss = edge_loss + igr_weight * stats[3] / (stats[0] + 1e-5) + igr_ns_weight * stats[4] / (stats[1] + 1e-5)
    return loss, edge_loss
