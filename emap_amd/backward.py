"""Training backward of the render hot path, evaluated by libemap_hip (SURVEY.md par. 8 f1).

What the reference gets from ``loss.backward()`` (reference src/runner/runner_udf.py:166-167) is autograd through
``render_core`` and ``UDFNetwork.gradient(create_graph=True)`` (src/models/udf_renderer_blending.py:457-625,
src/models/udf_model.py:121-135).  Here two ``torch.autograd.Function`` objects hand that job to the HIP kernels:

``RenderFn``   forward = ``emap_render_fwd`` (the whole render), backward = ``emap_render_bwd``
               (composite_bwd -> udf_mlp_vjp sweep -> weight-gradient GEMMs -> reduction + weight-norm VJP).
``UdfFn``      forward = ``emap_udf_fwd[_grad]``, backward = ``emap_udf_vjp`` - for direct calls of
               ``UDFNetwork.forward/udf/gradient`` with trainable parameters.

Parameter gradients are written by the kernels into ONE flat fp32 buffer laid out in ``parameters()`` order and returned
to autograd as views of it (a data-parallel caller all-reduces that buffer as it is, emap_amd/parallel.py).
There is no PyTorch fallback: a loss that depends on a render output whose gradient the kernels do not provide raises.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import torch

from . import _lib


class ParamLayout:
    """Flat layout of the trainable tensors of (UDFNetwork, variance, beta, gamma) and the pointer tables the C ABI takes."""

    def __init__(self, net, deviation_network=None, beta_network=None):
        self.net = net
        gs, vs, bs = net._gvb()
        self.n_lin = len(vs)
        self.weight_norm = bool(net.weight_norm)
        order = list(net.parameters())
        self.extra = []
        if deviation_network is not None:
            self.extra = [deviation_network.variance, beta_network.beta, beta_network.gamma]
        self.tensors: List[torch.Tensor] = order + self.extra
        self.offsets = {}
        off = 0
        for p in self.tensors:
            self.offsets[id(p)] = off
            off += p.numel()
        self.numel = off
        self.gs, self.vs, self.bs = gs, vs, bs

    def tables(self, flat: torch.Tensor):
        """(ParamGrads struct, keep-alive list) with the d* tables pointing into `flat`."""
        ck = (flat.data_ptr(), tuple(t.data_ptr() for t in self.vs + (self.gs if self.weight_norm else [])))
        hit = getattr(self, "_tables_hit", None)
        if hit is not None and hit[0] == ck:            # a persistent gradient buffer (Trainer, direct_param_grads): same tables every step
            pg, keep = hit[1]
            pg.accumulate, pg.grad_scale = 0, 1.0
            return pg, keep
        pg, keep = self._tables(flat)
        self._tables_hit = (ck, (pg, keep))
        return pg, keep

    def _tables(self, flat: torch.Tensor):
        n = self.n_lin
        es = flat.element_size()
        base = flat.data_ptr()

        def arr(ts, grad):
            if grad:
                return (C.c_void_p * n)(*[base + es * self.offsets[id(t)] for t in ts])
            return (C.c_void_p * n)(*[t.data_ptr() for t in ts])

        pg = _lib.ParamGrads()
        keep = []
        v_a, dv_a, db_a = arr(self.vs, False), arr(self.vs, True), arr(self.bs, True)
        keep += [v_a, dv_a, db_a]
        pg.v_host, pg.dv_host, pg.db_host = v_a, dv_a, db_a
        if self.weight_norm:
            g_a, dg_a = arr(self.gs, False), arr(self.gs, True)
            keep += [g_a, dg_a]
            pg.g_host, pg.dg_host = g_a, dg_a
        pg.weight_norm = int(self.weight_norm)
        pg.accumulate = 0
        pg.grad_scale = 1.0
        return pg, keep

    def check(self):
        for t in self.vs + self.bs + (self.gs if self.weight_norm else []):
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise RuntimeError("emap_amd backward: UDFNetwork parameters must be contiguous fp32")

    def views(self, flat: torch.Tensor, need: List[bool]):
        out = []
        for p, nd in zip(self.tensors, need):
            o = self.offsets[id(p)]
            out.append(flat[o:o + p.numel()].view(p.shape) if nd else None)
        return out


_WS_MAX_ENTRIES = 8            # distinct launch shapes per cache (training batch, validation chunk, ...): far fewer in practice
_WS_BUDGET_BYTES = 24 << 30    # ... and their total size: the backward's preferred workspace is ~17 KiB per point (8.8 GB at 4096 x 128)


def _workspace(cache: dict, key, nbytes: int, dev) -> torch.Tensor:
    """Cached scratch buffer for one launch shape.  The cache keeps one buffer PER KEY and never moves one that is still big enough:
    a captured hipGraph (UDFRendererBlending.capture, Trainer.capture) has the device pointers of the buffers of ITS shape baked in,
    so a render of another shape in between (validation.render_image between replays of a training graph) must not evict them.
    Entries are dropped least-recently-used first once the cache holds more than _WS_MAX_ENTRIES shapes OR more than
    _WS_BUDGET_BYTES (a caller with a varying point count - UdfFn on ragged extraction batches - would otherwise pin one multi-GB
    backward workspace per size).  Dropping an entry only drops the CACHE's reference: a captured graph holds its own
    (``live_buffers``), so its buffers stay where they are.  ``UDFNetwork.backward_workspace_limit`` caps a single workspace."""
    key = (key, str(dev))
    ws = cache.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        cache.pop(key, None)
        while cache and (len(cache) >= _WS_MAX_ENTRIES or sum(b.numel() for b in cache.values()) + nbytes > _WS_BUDGET_BYTES):
            cache.pop(next(iter(cache)))
    else:
        cache.pop(key)          # re-insert: dict order = recency
    cache[key] = ws
    return ws


class UdfFn(torch.autograd.Function):
    """(udf (P,1), grad (P,3) | None) of UDFNetwork at x with HIP forward and HIP parameter gradients."""

    @staticmethod
    def forward(ctx, net, x, with_grad, *params):
        udf, grad = net.hip_udf(x, with_grad=with_grad or x.requires_grad)
        ctx.net = net
        ctx.with_grad = with_grad
        ctx.x_needs = x.requires_grad
        ctx.save_for_backward(x.detach(), grad if x.requires_grad else None)
        ctx.set_materialize_grads(False)
        if with_grad:
            return udf, grad
        return udf, None

    @staticmethod
    def backward(ctx, d_udf, d_grad):
        net = ctx.net
        x, grad = ctx.saved_tensors
        lay = ParamLayout(net)
        lay.check()
        xs = _lib.f32c(x.reshape(-1, 3))
        P = xs.shape[0]
        dev = xs.device
        du = _lib.f32c(d_udf.reshape(-1)) if d_udf is not None else torch.zeros(P, device=dev)
        dg = _lib.f32c(d_grad.reshape(-1, 3)) if d_grad is not None else torch.zeros(P, 3, device=dev)
        if ctx.x_needs and d_grad is not None:
            raise NotImplementedError("emap_amd: d/dx of grad_x udf (a second-order input gradient) is not provided by the HIP backward")
        flat = torch.empty(lay.numel, dtype=torch.float32, device=dev)
        pg, keep = lay.tables(flat)
        L = _lib.lib()
        prec = _lib.PRECISIONS[net.precision]
        cfg = net.net_config()
        nb = C.c_size_t()
        with _lib.on_device(xs):
            _lib.check(L.emap_udf_vjp_workspace_bytes(C.byref(cfg), prec, P, C.byref(nb)), "udf_vjp_workspace_bytes")
            nbytes = nb.value if net.backward_workspace_limit is None else min(nb.value, int(net.backward_workspace_limit))
            ws = _workspace(net._vjp_ws, ("udf", P), nbytes, dev)
            _lib.check(L.emap_udf_vjp(C.byref(cfg), _lib.ptr(net.packed()), prec, _lib.ptr(xs), P, _lib.ptr(du), _lib.ptr(dg),
                                      C.byref(pg), _lib.ptr(ws), min(ws.numel(), nbytes), _lib.ptr(net.err_word(dev)), _lib.stream_ptr(dev)),
                       "udf_vjp")
        need = [p.requires_grad for p in lay.tensors]
        dx = None
        if ctx.x_needs:
            dx = (du.view(-1, 1) * grad).view(x.shape)   # d udf / dx = grad_x udf
        return (None, dx, None) + tuple(lay.views(flat, need))


class RenderFn(torch.autograd.Function):
    """render(): forward = emap_render_fwd, backward = emap_render_bwd."""

    DIFF = ("edge", "depth", "gradient_error", "gradient_error_near_surface")
    # returned through the Function so that a loss which touches them fails loudly instead of silently dropping a term
    GUARDED = ("udf", "weights", "normals", "gradients", "gradients_flip", "gradient_mag", "weight_sum", "alpha", "sparse_error")

    @staticmethod
    def forward(ctx, renderer, call, *params):
        v = renderer._render_hip(call)
        ctx.renderer = renderer
        ctx.call = call
        ctx.v = v
        call["_v"] = v   # the non-differentiable entries of the dict (masks, z_vals, ...) for render()
        ctx.set_materialize_grads(False)
        N, S = call["N"], call["S"]
        outs = (v["edge"].view(N, 1), v["depth"].view(N, 1), v["scalars"][0], v["scalars"][1],
                v["udf"].view(N, S), v["weights"].view(N, S), v["normals"].view(N, 3), v["gradients"].view(N, S, 3),
                v["gradients_flip"].view(N, S, 3), v["gradient_mag"].view(N, S), v["weight_sum"].view(N, 1), v["alpha"].view(N, S),
                v["scalars"][2])
        # what the backward reads later must be what this forward used: the sample distance lives in a workspace the next render()
        # of this shape overwrites (other near / far), the packed weights are re-packed after an optimizer step
        v["_sd"] = v["_ws"][:4].clone()
        ctx.packed_key = renderer.udf_network._pack_cache[_lib.PRECISIONS[call["prec_name"]]][0]
        return outs

    @staticmethod
    def backward(ctx, d_edge, d_depth, d_ge, d_ge_ns, *guarded):
        for k, g in zip(RenderFn.GUARDED, guarded):
            if g is not None:
                raise NotImplementedError(
                    f"emap_amd: the loss depends on render()['{k}'], whose gradient the HIP backward does not provide "
                    "(differentiable outputs: edge, depth, gradient_error, gradient_error_near_surface, variance, beta, gamma)")
        r, call, v = ctx.renderer, ctx.call, ctx.v
        packed = r.udf_network.packed(call["prec_name"])
        if r.udf_network._pack_cache[_lib.PRECISIONS[call["prec_name"]]][0] != ctx.packed_key:
            raise RuntimeError("emap_amd: the UDF network's parameters changed between render() and backward() (an optimizer step or an "
                               "in-place edit in between): the gradient would be taken at other weights than the forward used")
        lay = r._layout()
        need = [p.requires_grad for p in lay.tensors]
        ctx.v = None
        if getattr(r, "direct_param_grads", False) and all(p.grad is None for p, nd in zip(lay.tensors, need) if nd):
            # Fast path of the drop-in training step (dropin.patch_runner(train=True)): the kernels write into ONE persistent flat buffer
            # and its views become the parameters' .grad here; autograd gets None for them.  Handing the 32 views to autograd instead makes
            # every AccumulateGrad node clone its view (32 small copy kernels + their launches per step) and the optimizer re-gather them.
            # Only when no parameter holds a gradient yet (zero_grad(set_to_none=True), the default): accumulation into existing
            # gradients takes the ordinary path below.  The buffer is overwritten by the next backward of this renderer.
            G = getattr(r, "_grad_flat", None)
            if G is None or G.numel() != lay.numel or G.device != call["dev"]:
                G = r._grad_flat = torch.empty(lay.numel, dtype=torch.float32, device=call["dev"])
            r.backward_into(call, v, d_edge, d_depth, d_ge, d_ge_ns, flat=G, packed=packed)
            vk = (G.data_ptr(), tuple(need))
            views = getattr(r, "_grad_views", None)
            if views is None or views[0] != vk:              # the buffer is persistent: so are its views
                views = r._grad_views = (vk, lay.views(G, need))
            for p, g_ in zip(lay.tensors, views[1]):
                if g_ is not None:
                    p.grad = g_
            return (None, None) + (None,) * len(lay.tensors)
        flat = r.backward_into(call, v, d_edge, d_depth, d_ge, d_ge_ns, packed=packed)
        return (None, None) + tuple(lay.views(flat, need))
