"""Fields with the reference interface (reference src/models/udf_model.py), evaluated by libemap_hip.

``UDFNetwork``, ``SingleVarianceNetwork``, ``BetaNetwork`` and ``RenderingNetwork`` keep the
reference's constructor signatures, parameter names/order (so ``state_dict`` keys and the runner's Adam
param groups are unchanged: ``lin{l}.bias``, ``lin{l}.parametrizations.weight.original0/1``,
``variance``, ``second_variance``, ``beta``, ``gamma``, ``zeta``) and method signatures.

What differs is *how* ``forward`` / ``udf`` / ``gradient`` are evaluated: one fused HIP kernel
(positional encoding + all Linear/Softplus layers [+ forward-mode spatial gradient]) reading a packed
copy of the weight-normed weights that is rebuilt only when a parameter changes.

Autograd: with grad mode on and trainable parameters, ``forward`` / ``udf`` / ``gradient`` run inside
``emap_amd.backward.UdfFn`` - HIP forward, HIP parameter gradients (``emap_udf_vjp``: the double backward of
udf_model.py:121-135 as kernels).  There is no PyTorch evaluation of the network anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .embedder import get_embedder, embed_torch

DEFAULT_PRECISION = os.environ.get("EMAP_PRECISION", "f16x3")


class UDFNetwork(nn.Module):
    """Reference: src/models/udf_model.py:7-135."""

    def __init__(self, d_in, d_out, d_hidden, n_layers, skip_in=(4,), multires=0, scale=1, bias=0.5,
                 geometric_init=True, weight_norm=True, udf_type="abs", precision=None):
        super().__init__()
        dims = [d_in] + [d_hidden for _ in range(n_layers)] + [d_out]
        self.embed_fn_fine = None
        if multires > 0:
            embed_fn, input_ch = get_embedder(multires, input_dims=d_in)
            self.embed_fn_fine = embed_fn
            dims[0] = input_ch
        self.num_layers = len(dims)
        self.skip_in = tuple(skip_in)
        self.scale = scale
        self.geometric_init = geometric_init
        self.multires = multires
        self.d_in, self.d_out, self.d_hidden = d_in, d_out, d_hidden
        self.weight_norm = weight_norm
        self.udf_type = udf_type
        self.precision = precision or DEFAULT_PRECISION

        # same construction and init-call order as udf_model.py:39-76, so a seeded reference run and a
        # seeded emap_amd run start from identical parameters
        for l in range(self.num_layers - 1):
            out_dim = dims[l + 1] - dims[0] if (l + 1) in self.skip_in else dims[l + 1]
            lin = nn.Linear(dims[l], out_dim)
            if geometric_init:
                if l == self.num_layers - 2:
                    nn.init.normal_(lin.weight, mean=np.sqrt(np.pi) / np.sqrt(dims[l]), std=0.0001)
                    nn.init.constant_(lin.bias, -bias)
                elif multires > 0 and l == 0:
                    nn.init.constant_(lin.bias, 0.0)
                    nn.init.constant_(lin.weight[:, 3:], 0.0)
                    nn.init.normal_(lin.weight[:, :3], 0.0, np.sqrt(2) / np.sqrt(out_dim))
                elif multires > 0 and l in self.skip_in:
                    nn.init.constant_(lin.bias, 0.0)
                    nn.init.normal_(lin.weight, 0.0, np.sqrt(2) / np.sqrt(out_dim))
                    nn.init.constant_(lin.weight[:, -(dims[0] - 3):], 0.0)
                else:
                    nn.init.constant_(lin.bias, 0.0)
                    nn.init.normal_(lin.weight, 0.0, np.sqrt(2) / np.sqrt(out_dim))
            if weight_norm:
                lin = nn.utils.parametrizations.weight_norm(lin)
            setattr(self, "lin" + str(l), lin)

        self.activation = nn.Softplus(beta=100)
        self.relu = nn.ReLU()
        self._pack_cache = {}
        self._vjp_ws = {}
        # upper bound in bytes on the backward's workspace (None: the library's preferred size, ~17 KiB per point up to 8.8 GB); with less
        # the backward runs in more, smaller chunks (include/emap_hip.h)
        self.backward_workspace_limit = None
        self._scratch = {}
        self._err = None

    # ---- packed weights -----------------------------------------------------------------------
    def _gvb(self):
        """(g, v, b) per layer; without weight_norm g = ||v|| so that g*v/||v|| = v."""
        # cached: the walk through nine parametrized modules costs ~0.1 ms of host time and sits on the critical path of the drop-in
        # training step (twice per backward).  Valid while EVERY layer still holds the module and parameter objects it recorded
        # (a re-parametrised or replaced middle layer must not be packed from stale tensors) - checked through the modules'
        # own dicts: ~50 dict lookups, a few microseconds.
        c = getattr(self, "_gvb_cache", None)
        if c is not None and self.weight_norm:
            mods = self._modules
            for name, lin, pc, pl, g, v, b in c[3]:
                if (mods.get(name) is not lin or lin._modules.get("parametrizations") is not pc or pc._modules.get("weight") is not pl
                        or pl._parameters.get("original0") is not g or pl._parameters.get("original1") is not v
                        or lin._parameters.get("bias") is not b):
                    break
            else:
                return c[:3]
        gs, vs, bs, recs = [], [], [], []
        for l in range(self.num_layers - 1):
            lin = getattr(self, "lin" + str(l))
            if self.weight_norm:
                g = lin.parametrizations.weight.original0
                v = lin.parametrizations.weight.original1
                recs.append(("lin" + str(l), lin, lin.parametrizations, lin.parametrizations.weight, g, v, lin.bias))
            else:
                v = lin.weight
                g = torch.linalg.norm(v.detach(), dim=1, keepdim=True)
            gs.append(g); vs.append(v); bs.append(lin.bias)
        if self.weight_norm:
            self._gvb_cache = (gs, vs, bs, recs)
        return gs, vs, bs

    def net_config(self) -> _lib.NetConfig:
        if self.d_in != 3 or self.multires < 0:
            raise NotImplementedError("the HIP UDF MLP needs d_in=3 (all EMAP configs)")
        if self.d_out != 1:
            raise NotImplementedError("the HIP UDF MLP supports d_out=1 (all EMAP configs); feature outputs are unused")
        skips = [s for s in self.skip_in if 0 < s < self.num_layers - 1]
        if len(skips) > 1:
            raise NotImplementedError("at most one skip connection is supported")
        return _lib.NetConfig(self.d_hidden, self.num_layers - 1, skips[0] if skips else -1, self.multires, self.d_out,
                              _lib.UDF_TYPES[self.udf_type], float(self.scale))

    def packed(self, precision=None):
        """Device buffer with the folded/permuted weights for `precision`; rebuilt when parameters change."""
        prec = _lib.PRECISIONS[precision or self.precision]
        gs, vs, bs = self._gvb()
        _lib.require_cuda(vs[0], "UDFNetwork parameters")
        # version counters catch optimizer steps / in-place ops; the data pointers catch re-assigned .data; edits made through
        # `.data` in place bump neither - call invalidate_packed() after those
        # (without weight_norm the g's are temporaries synthesised by _gvb: only v and b identify the state)
        ident = (gs if self.weight_norm else []) + vs + bs
        key = (prec, vs[0].device, tuple(int(t._version) for t in ident), tuple(t.data_ptr() for t in ident))
        hit = self._pack_cache.get(prec)
        if hit is not None and hit[0] == key:
            return hit[1]
        L = _lib.lib()
        cfg = self.net_config()
        nbytes = C.c_size_t()
        _lib.check(L.emap_packed_bytes(C.byref(cfg), prec, C.byref(nbytes)), "packed_bytes")
        buf = hit[1] if (hit is not None and hit[1].numel() == nbytes.value and hit[1].device == vs[0].device) else \
            torch.empty(nbytes.value, dtype=torch.uint8, device=vs[0].device)
        n = len(vs)
        # pointer tables: cached while every source tensor is the contiguous fp32 tensor at the same address it was (the usual case: an
        # optimizer updates in place) - the re-pack after every step then costs one key comparison and one launch on the host
        ptrs = key[3]
        hit_t = getattr(self, "_pack_tables", None)
        if (self.weight_norm and hit_t is not None and hit_t[0] == ptrs
                and all(t.dtype == torch.float32 and t.is_contiguous() for t in ident)):
            ga, va, ba = hit_t[1]
        else:
            keep = [[_lib.f32c(t.detach()) for t in ts] for ts in (gs, vs, bs)]  # keep contiguous copies alive
            ga = (C.c_void_p * n)(*[t.data_ptr() for t in keep[0]])
            va = (C.c_void_p * n)(*[t.data_ptr() for t in keep[1]])
            ba = (C.c_void_p * n)(*[t.data_ptr() for t in keep[2]])
            same = all(k_.data_ptr() == t.data_ptr() for ks_, ts in zip(keep, (gs, vs, bs)) for k_, t in zip(ks_, ts))
            self._pack_tables = (ptrs, (ga, va, ba)) if (self.weight_norm and same) else None
        with _lib.on_device(buf):
            _lib.check(L.emap_pack_weights(C.byref(cfg), ga, va, ba, _lib.ptr(buf), prec, _lib.stream_ptr(buf.device)), "pack_weights")
        self._pack_cache[prec] = (key, buf)
        return buf

    def invalidate_packed(self):
        """Force a re-pack on the next call (after editing parameters through ``.data`` in place).  The packed buffers are KEPT and
        re-packed in place: a captured hipGraph (RenderGraph, Trainer.capture) has their addresses baked in."""
        self._pack_cache = {prec: (None, buf) for prec, (_, buf) in self._pack_cache.items()}

    def err_word(self, dev):
        if self._err is None or self._err.device != dev:
            self._err = torch.zeros(1, dtype=torch.int32, device=dev)
        return self._err

    # ---- HIP evaluation -----------------------------------------------------------------------
    def _needs_autograd(self, x):
        return torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))

    def hip_udf(self, x, with_grad=False, precision=None):
        """(udf (P,1), grad (P,3) | None) from the fused kernel; no autograd."""
        _lib.require_cuda(x, "x")
        prec = _lib.PRECISIONS[precision or self.precision]
        xs = _lib.f32c(x.detach().reshape(-1, 3))
        P = xs.shape[0]
        cfg = self.net_config()
        buf = self.packed(precision)
        udf = torch.empty(P, 1, device=xs.device, dtype=torch.float32)
        L = _lib.lib()
        if P == 0:
            return udf, (torch.empty(0, 3, device=xs.device, dtype=torch.float32) if with_grad else None)
        with _lib.on_device(xs):
            if with_grad:
                grad = torch.empty(P, 3, device=xs.device, dtype=torch.float32)
                nb = C.c_size_t()
                _lib.check(L.emap_udf_scratch_bytes(C.byref(cfg), prec, P, C.byref(nb)), "udf_scratch_bytes")
                scr = None
                if nb.value:
                    scr = self._scratch.get(xs.device)
                    if scr is None or scr.numel() < nb.value:
                        scr = torch.empty(nb.value, dtype=torch.uint8, device=xs.device)
                        self._scratch = {xs.device: scr}
                _lib.check(L.emap_udf_fwd_grad(C.byref(cfg), _lib.ptr(buf), prec, _lib.ptr(xs), P, _lib.ptr(udf), _lib.ptr(grad),
                                              _lib.ptr(scr), nb.value, _lib.stream_ptr(xs.device)), "udf_fwd_grad")
                return udf, grad
            _lib.check(L.emap_udf_fwd(C.byref(cfg), _lib.ptr(buf), prec, _lib.ptr(xs), P, _lib.ptr(udf),
                                     _lib.stream_ptr(xs.device)), "udf_fwd")
        return udf, None

    # ---- reference interface ------------------------------------------------------------------
    def udf_out(self, x):
        if self.udf_type == "abs":
            return torch.abs(x)
        if self.udf_type == "square":
            return x ** 2
        return x

    def forward(self, inputs):
        """-> (out (P,d_out), PE (P,d0))   (udf_model.py:90-110)."""
        _lib.require_cuda(inputs, "inputs")
        _lib.lib()
        if self._needs_autograd(inputs):
            from .backward import UdfFn
            udf, _ = UdfFn.apply(self, inputs, False, *self.parameters())
        else:
            udf, _ = self.hip_udf(inputs)
        xs = inputs.detach() * self.scale
        pe = self.embed_fn_fine(xs) if self.embed_fn_fine is not None else xs       # multires = 0: no embedding (udf_model.py:92-93)
        return udf, pe

    def udf(self, x):
        feature_out, pe = self.forward(x)
        return feature_out[:, :1], feature_out[:, 1:], pe

    def udf_hidden_appearance(self, x):
        return self.forward(x)

    def gradient(self, x):
        """-> (P,1,3) grad_x udf   (udf_model.py:121-135; differentiable when grads are enabled)."""
        _lib.require_cuda(x, "x")
        _lib.lib()
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from .backward import UdfFn
            _, g = UdfFn.apply(self, x.detach(), True, *self.parameters())   # the reference re-leafs x too (:122)
        else:
            _, g = self.hip_udf(x, with_grad=True)
        return g.unsqueeze(1)


class RenderingNetwork(nn.Module):
    """Reference: src/models/udf_model.py:138-209.  Never instantiated by the reference runner
    (SURVEY.md par. 2 #4); kept as a plain PyTorch module so the name and signature exist."""

    def __init__(self, d_feature, mode, d_in, d_out, d_hidden, n_layers, weight_norm=True, multires_view=0,
                 squeeze_out=True):
        super().__init__()
        self.mode, self.squeeze_out, self.d_out = mode, squeeze_out, d_out
        dims = [d_in + d_feature] + [d_hidden for _ in range(n_layers)] + [d_out]
        self.multires_view = multires_view if mode != "no_view_dir" else 0
        if self.multires_view > 0:
            dims[0] += 6 * multires_view
        self.num_layers = len(dims)
        for l in range(self.num_layers - 1):
            lin = nn.Linear(dims[l], dims[l + 1])
            if weight_norm:
                lin = nn.utils.parametrizations.weight_norm(lin)
            setattr(self, "lin" + str(l), lin)
        self.relu = nn.ReLU()

    def forward(self, points, normals, view_dirs, feature_vectors):
        if self.multires_view > 0:
            view_dirs = embed_torch(view_dirs, self.multires_view)
        normals = normals.detach()
        if self.mode == "idr":
            x = torch.cat([points, view_dirs, normals, -1 * normals, feature_vectors], dim=-1)
        elif self.mode == "no_view_dir":
            x = torch.cat([points, normals, -1 * normals, feature_vectors], dim=-1)
        else:
            x = torch.cat([points, view_dirs, feature_vectors], dim=-1)
        for l in range(self.num_layers - 1):
            x = getattr(self, "lin" + str(l))(x)
            if l < self.num_layers - 2:
                x = self.relu(x)
        return torch.sigmoid(x[:, : self.d_out]) if self.squeeze_out else x[:, : self.d_out]


class SingleVarianceNetwork(nn.Module):
    """Reference: src/models/udf_model.py:212-232."""

    def __init__(self, init_val, requires_grad=True):
        super().__init__()
        self.variance = nn.Parameter(torch.Tensor([init_val]), requires_grad=requires_grad)
        self.second_variance = nn.Parameter(torch.Tensor([init_val]), requires_grad=requires_grad)

    def set_trainable(self):
        self.variance.requires_grad = True
        self.second_variance.requires_grad = True

    def forward(self, x):
        return torch.ones([len(x), 1], device=x.device) * torch.exp(self.variance * 10.0)

    def get_secondvariance(self, x):
        return torch.ones([len(x), 1], device=x.device) * torch.exp(self.second_variance * 10.0)


class BetaNetwork(nn.Module):
    """Reference: src/models/udf_model.py:235-286."""

    def __init__(self, init_var_beta=0.1, init_var_gamma=0.1, init_var_zeta=0.05, beta_min=0.00005,
                 requires_grad_beta=True, requires_grad_gamma=True, requires_grad_zeta=True):
        super().__init__()
        self.beta = nn.Parameter(torch.Tensor([init_var_beta]), requires_grad=requires_grad_beta)
        self.gamma = nn.Parameter(torch.Tensor([init_var_gamma]), requires_grad=requires_grad_gamma)
        self.zeta = nn.Parameter(torch.Tensor([init_var_zeta]), requires_grad=requires_grad_zeta)
        self.beta_min = beta_min

    def get_beta(self):
        return torch.exp(self.beta * 10).clip(0, 1.0 / self.beta_min)

    def get_gamma(self):
        return torch.exp(self.gamma * 10)

    def get_zeta(self):
        return self.zeta.abs()

    def set_beta_trainable(self):
        self.beta.requires_grad = True

    @torch.no_grad()
    def set_gamma(self, x):
        self.gamma = nn.Parameter(torch.Tensor([x]), requires_grad=self.gamma.requires_grad).to(self.gamma.device)

    def forward(self):
        return self.get_beta(), self.get_gamma(), self.get_zeta()
