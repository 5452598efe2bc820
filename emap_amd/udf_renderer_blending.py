"""Renderer with the reference interface (reference src/models/udf_renderer_blending.py:112-975),
evaluated by libemap_hip: ``render()`` enqueues the coarse sampler, the MLP passes, the K
occlusion-aware up-sampling steps, the final MLP value+gradient pass and the compositing kernel on the
current stream in ONE C call with no host synchronisation.  With trainable parameters and grad mode on,
the same forward runs inside ``emap_amd.backward.RenderFn`` and ``loss.backward()`` is ONE more C call
(``emap_render_bwd``: composite_bwd + the MLP double-backward kernels); there is no PyTorch fallback.

Supported configuration = what every EMAP conf selects (SURVEY.md par. 2 #5-#7):
``sdf2alpha_type="numerical"``, ``upsampling_type="classical"``, ``use_unbias_render=True``,
``use_norm_grad_for_cosine=False``, ``n_outside=0``.  Anything else raises at construction.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .backward import ParamLayout, RenderFn, _workspace

_PER_SAMPLE = ("weights", "alpha", "mid_z", "dists", "inside_sphere", "gradient_mag")


def sample_pdf(bins, weights, n_samples, det=False):
    """reference udf_renderer_blending.py:69-109.  det=True (what the render path calls): the deterministic u grid; det=False: u drawn as
    the reference draws it - torch.rand(N, n_samples) on the CPU generator, then moved to the device (:84-85)."""
    _lib.require_cuda(bins, "bins")
    b, w = _lib.f32c(bins), _lib.f32c(weights)
    N, n = b.shape
    out = torch.empty(N, n_samples, device=b.device, dtype=torch.float32)
    with _lib.on_device(b):
        if det:
            _lib.check(_lib.lib().emap_sample_pdf(_lib.ptr(b), _lib.ptr(w), N, n, n_samples, _lib.ptr(out), None, None,
                                                  _lib.stream_ptr(b.device)), "sample_pdf")
        else:
            u = torch.rand([N, n_samples]).to(b.device).contiguous()
            _lib.check(_lib.lib().emap_sample_pdf_u(_lib.ptr(b), _lib.ptr(w), _lib.ptr(u), N, n, n_samples, _lib.ptr(out), None, None,
                                                    _lib.stream_ptr(b.device)), "sample_pdf_u")
    return out


class _ReducedOperand:
    """Stand-in for a per-sample render() entry in the reduced-output launch mode (UDFRendererBlending._render_reduced_compat):
    no per-sample tensor exists there.  Supports exactly the validation loop's expression (runner_udf.py:375-405)

        (out["gradients_flip"] * out["weights"][:, :S, None]).sum(dim=1)      ->  the rendered normals (N, 3)

    and ``is not None`` tests; everything else raises."""

    def __init__(self, name, normals, S, stage="entry"):
        self._name, self._normals, self._S, self._stage = name, normals, S, stage

    def _no(self, what):
        raise RuntimeError(f"emap_amd: render() ran in the reduced-output mode (inference_reduced): out['{self._name}'] has no per-sample values "
                           f"({what}); only (gradients_flip * weights[:, :S, None]).sum(dim=1) is defined - render without inference_reduced")

    def __getitem__(self, key):
        ok = (self._name == "weights" and self._stage == "entry" and isinstance(key, tuple) and len(key) == 3 and key[0] == slice(None)
              and isinstance(key[1], slice) and key[1].start in (None, 0) and key[1].step in (None, 1)
              and (key[1].stop is None or key[1].stop >= self._S) and key[2] is None)
        if not ok:
            self._no(f"indexing with {key!r}")
        return _ReducedOperand("weights", None, self._S, "sliced")

    def __mul__(self, other):
        if not (self._name == "gradients_flip" and self._stage == "entry" and isinstance(other, _ReducedOperand) and other._stage == "sliced"):
            self._no("multiplication by anything but weights[:, :S, None]")
        return _ReducedOperand("gradients_flip * weights", self._normals, self._S, "product")

    def sum(self, dim=None, **kw):
        if not (self._stage == "product" and dim == 1 and not kw):
            self._no(f"sum(dim={dim!r})")
        return self._normals

    def __getattr__(self, item):          # .shape, .cpu(), .detach(), ...: there is nothing to look at
        if item.startswith("__") and item.endswith("__"):
            raise AttributeError(item)
        self._no(f"attribute .{item}")

    __rmul__ = __add__ = __radd__ = __sub__ = __truediv__ = __matmul__ = lambda self, other: self._no("arithmetic")

    def __iter__(self):
        self._no("iteration")

    def __len__(self):
        self._no("len()")


class UDFRendererBlending:
    def __init__(self, nerf, udf_network, deviation_network, beta_network, n_samples, n_importance, n_outside,
                 up_sample_steps, perturb, sdf2alpha_type="numerical", upsampling_type="classical",
                 sparse_scale_factor=25000, use_norm_grad_for_cosine=False, use_unbias_render=True, near_surface=0.05,
                 device="cuda", precision=None):
        if n_outside != 0 or nerf is not None:
            raise NotImplementedError("n_outside>0 / background NeRF is dead code in the reference (SURVEY par. 2 #7)")
        if sdf2alpha_type != "numerical" or upsampling_type != "classical" or not use_unbias_render or use_norm_grad_for_cosine:
            raise NotImplementedError("only sdf2alpha_type='numerical', upsampling_type='classical', "
                                      "use_unbias_render=True, use_norm_grad_for_cosine=False are on the hot path")
        self.nerf = nerf
        self.udf_network = udf_network
        self.deviation_network = deviation_network
        self.beta_network = beta_network
        self.n_samples = n_samples
        self.n_importance = n_importance
        self.n_outside = n_outside
        self.perturb = perturb
        self.up_sample_steps = up_sample_steps
        self.use_unbias_render = use_unbias_render
        self.sdf2alpha_type = sdf2alpha_type
        self.upsampling_type = upsampling_type
        self.sparse_scale_factor = sparse_scale_factor
        self.use_norm_grad_for_cosine = use_norm_grad_for_cosine
        self.near_surface = near_surface
        self.device = device
        self.precision = precision  # None -> udf_network.precision
        self.inference_reduced = False   # True: render() without autograd returns per-ray results only (_render_reduced_compat)
        # drop-in training step (emap_amd.dropin.patch_runner(train=True) sets both; host_scalars.py / backward.py:RenderFn):
        self.host_mirror_scalars = False  # variance / beta / gamma of the render dict answer host reads from a pinned copy made BEFORE the forward
        self.direct_param_grads = False   # RenderFn.backward installs its flat gradient buffer's views as .grad itself (no 32 AccumulateGrad clones)
        self._mirror = None
        self._ws = {}
        self._ws_pool = {}
        self._bws = {}
        self._bws_bytes = {}
        self._err = None
        self._lay = None
        self._const = {}

    # ---- helpers ------------------------------------------------------------------------------
    @property
    def samples_per_ray(self):
        m = self.n_importance // self.up_sample_steps if self.n_importance > 0 else 0
        return self.n_samples + m * (self.up_sample_steps if m > 0 else 0)

    def _params(self, N, cos_anneal_ratio, flip_saturation, background_rgb):
        p = _lib.RenderParams()
        p.n_rays, p.n_samples, p.n_importance, p.up_sample_steps = N, self.n_samples, self.n_importance, self.up_sample_steps
        p.inv_s = p.beta = p.gamma = 0.0
        p.cos_anneal_ratio = float(cos_anneal_ratio) if cos_anneal_ratio is not None else 0.0
        p.has_cos_anneal = int(cos_anneal_ratio is not None)
        p.flip_saturation = float(flip_saturation)
        p.near_surface = float(self.near_surface)
        p.sparse_scale = float(self.sparse_scale_factor)
        p.has_background = int(background_rgb is not None)
        p.background = 0.0
        if background_rgb is not None:
            bg = torch.as_tensor(background_rgb).reshape(-1)
            if bg.numel() != 1 and not bool((bg == bg[0]).all()):
                raise NotImplementedError("edge is single-channel: background_rgb must be a scalar / constant")
            p.background = float(bg[0])
        p.variance_dev = self.deviation_network.variance.data_ptr()
        p.beta_dev = self.beta_network.beta.data_ptr()
        p.gamma_dev = self.beta_network.gamma.data_ptr()
        p.beta_min = float(self.beta_network.beta_min)
        return p

    def error_flags(self) -> int:
        """Device error word (NaN in sample_pdf / gradient_error). Reading it synchronises: call it lazily."""
        return 0 if self._err is None else int(self._err.item())

    def check_errors(self):
        f = self.error_flags()
        if f:
            self._err.zero_()
            raise RuntimeError(f"emap_amd render: non-finite values detected on device (flags={f}: "
                               f"{'z_samples ' if f & _lib.F_NAN_SAMPLES else ''}{'gradient_error ' if f & _lib.F_NAN_GRADERR else ''}"
                               f"{'MLP output (fp16 range exceeded? use precision=bf16x3) ' if f & _lib.F_MLP_NONFINITE else ''})")

    # ---- reference interface ------------------------------------------------------------------
    def _layout(self) -> ParamLayout:
        if self._lay is None:
            self._lay = ParamLayout(self.udf_network, self.deviation_network, self.beta_network)
        return self._lay

    def _prepare(self, rays_o, rays_d, near, far, depth_scale, cos_anneal_ratio, perturb_overwrite, background_rgb,
                 flip_saturation, t_rand, reduced=False):
        _lib.require_cuda(rays_o, "rays_o")
        dev = rays_o.device
        N = len(rays_o)
        net = self.udf_network
        prec_name = self.precision or net.precision
        ro, rd = _lib.f32c(rays_o.detach()), _lib.f32c(rays_d.detach())
        if not isinstance(near, torch.Tensor):
            key = ("nf", N, float(near), float(far), dev)
            nf = self._const.get(key)
            if nf is None:
                nf = (torch.full((N,), float(near), device=dev, dtype=torch.float32),
                      torch.full((N,), float(far), device=dev, dtype=torch.float32))
                if len(self._const) > 16:       # bounded; never replaces a live entry in place (a captured graph may point at it)
                    self._const.pop(next(k for k in self._const if k not in ("_scr", "_jit")))
                self._const[key] = nf
            near_t, far_t = nf
        else:
            near_t = _lib.f32c(near.detach().to(dev)).reshape(-1).expand(N).contiguous()
            far_t = _lib.f32c(far.detach().to(dev)).reshape(-1).expand(N).contiguous()
        perturb = self.perturb if perturb_overwrite < 0 else perturb_overwrite
        tr = None
        if t_rand is not None:
            tr = _lib.f32c(t_rand.to(dev)).reshape(-1)
        elif perturb > 0:
            tr = self._jitter_draw(N, dev)                        # the reference's CPU-generator draw (:719)
        ds = _lib.f32c(depth_scale.detach().to(dev)).reshape(-1) if depth_scale is not None else None
        return {"N": N, "S": self.samples_per_ray, "dev": dev, "prec_name": prec_name, "ro": ro, "rd": rd, "near": near_t,
                "far": far_t, "t_rand": tr, "ds": ds, "reduced": reduced,
                "p": self._params(N, cos_anneal_ratio, flip_saturation, background_rgb)}

    def _jitter_draw(self, N, dev):
        """(torch.rand([N, 1]) - 0.5).to(dev) - the same values from the same CPU generator - through a small ring of pinned staging
        buffers, so that the copy is asynchronous (a pageable host-to-device copy costs ~0.12 ms of host time per step).  A slot is
        reused only after its copy has completed (event)."""
        ring = self._const.get("_jit")
        if ring is None or ring["N"] != N or ring["dev"] != dev:
            ring = {"N": N, "dev": dev, "i": 0, "slots": [(torch.empty(N, 1).pin_memory(), torch.cuda.Event()) for _ in range(4)]}
            self._const["_jit"] = ring
        buf, ev = ring["slots"][ring["i"] % 4]
        if ring["i"] >= 4:
            ev.synchronize()
        ring["i"] += 1
        torch.rand([N, 1], out=buf)
        buf.sub_(0.5)
        tr = buf.to(dev, non_blocking=True).reshape(-1)
        ev.record(torch.cuda.current_stream(dev))
        return tr

    def _render_hip(self, call):
        """One emap_render_fwd call.  Returns the dict of flat output views (full: every per-sample entry of the
        reference's dict; reduced: only edge / depth / normals / weight_sum / scalars are written - 28 B per ray instead
        of 48 B per sample, the mode of the full-image path, SURVEY par. 8 f4)."""
        N, S, dev, reduced = call["N"], call["S"], call["dev"], call["reduced"]
        net = self.udf_network
        prec = _lib.PRECISIONS[call["prec_name"]]
        per_ray = {"edge": N, "depth": N, "weight_sum": N, "normals": N * 3, "scalars": 16}
        mlp_out = {"z_vals": N * S, "udf": N * S, "gradients": N * S * 3}
        sizes = dict(per_ray)
        if not reduced:
            sizes.update(mlp_out)
            sizes["gradients_flip"] = N * S * 3
            for k in _PER_SAMPLE:
                sizes[k] = N * S
        flat = torch.empty(sum(sizes.values()), device=dev, dtype=torch.float32)
        v, off = {}, 0
        for k, n in sizes.items():
            v[k] = flat[off:off + n]
            off += n
        if reduced:   # the MLP's own outputs are intermediates here: a cached scratch, not a fresh allocation
            sc = _workspace(self._const.setdefault("_scr", {}), (N, S), 4 * sum(mlp_out.values()), dev).view(torch.float32)
            off = 0
            for k, n in mlp_out.items():
                v[k] = sc[off:off + n]
                off += n
        p = call["p"]
        L = _lib.lib()
        cfg = net.net_config()
        with _lib.on_device(call["ro"]):
            key = (N, str(dev), prec)
            ws = self._ws.get(key)
            if ws is None:
                nb = C.c_size_t()
                _lib.check(L.emap_render_workspace_bytes(C.byref(cfg), prec, C.byref(p), C.byref(nb)), "render_workspace_bytes")
                ws = _workspace(self._ws_pool, key, nb.value, dev)   # one buffer per launch shape, never evicted by another shape
                if len(self._ws) >= 8:
                    self._ws.pop(next(iter(self._ws)))
                self._ws[key] = ws
            if self._err is None or self._err.device != dev:
                self._err = torch.zeros(1, dtype=torch.int32, device=dev)
            co = _lib.CompositeOut()
            names = ("edge", "depth", "weight_sum", "normals", "scalars") if reduced else \
                _PER_SAMPLE + ("gradients_flip", "edge", "depth", "weight_sum", "normals", "scalars")
            for k in names:
                setattr(co, k, v[k].data_ptr())
            packed = net.packed(call["prec_name"])
            _lib.check(L.emap_render_fwd(C.byref(cfg), _lib.ptr(packed), prec, C.byref(p), _lib.ptr(call["ro"]), _lib.ptr(call["rd"]),
                                         _lib.ptr(call["near"]), _lib.ptr(call["far"]), _lib.ptr(call["t_rand"]), _lib.ptr(call["ds"]),
                                         _lib.ptr(v["z_vals"]), _lib.ptr(v["udf"]), _lib.ptr(v["gradients"]), C.byref(co), _lib.ptr(ws),
                                         ws.numel(), _lib.ptr(self._err), _lib.stream_ptr(dev)), "render_fwd")
        v["_ws"] = ws
        return v

    def backward_into(self, call, v, d_edge, d_depth=None, d_ge=None, d_ge_ns=None, flat=None, scalars=None, grad_scale=1.0,
                      stages=3, packed=None):
        """One emap_render_bwd call: parameter gradients of  sum(d_edge*edge) + sum(d_depth*depth) + d_ge*gradient_error +
        d_ge_ns*gradient_error_near_surface  into `flat` (parameters() order of the UDF network, then variance, beta,
        gamma; allocated when None).  `scalars`: the forward's scalars, or a copy with GLOBAL eikonal mask sums in [4],[6]
        (data-parallel).  `stages`: 1 = compositing adjoint only, 2 = MLP backward only (after a stages=1 call with the same
        arguments), 3 = both; between 1 and 2 a data-parallel step max-reduces `bwd_absmax(call)` over the ranks.  Returns flat."""
        N, S, dev = call["N"], call["S"], call["dev"]
        net = self.udf_network
        lay = self._layout()
        lay.check()
        prec = _lib.PRECISIONS[call["prec_name"]]
        if flat is None:
            flat = torch.empty(lay.numel, dtype=torch.float32, device=dev)
        f = lambda t, n: None if t is None else _lib.f32c(t.detach().reshape(-1).expand(n))
        de, dd, dge, dns = f(d_edge, N), f(d_depth, N), f(d_ge, 1), f(d_ge_ns, 1)
        cg = _lib.CompositeGrads()
        cg.d_edge, cg.d_depth = (None if de is None else de.data_ptr()), (None if dd is None else dd.data_ptr())
        cg.d_gradient_error = None if dge is None else dge.data_ptr()
        cg.d_gradient_error_near_surface = None if dns is None else dns.data_ptr()
        sc = v["scalars"] if scalars is None else scalars
        cg.scalars = sc.data_ptr()
        es = flat.element_size()
        extra = lay.extra
        cg.d_variance = flat.data_ptr() + es * lay.offsets[id(extra[0])]
        cg.d_beta = flat.data_ptr() + es * lay.offsets[id(extra[1])]
        cg.d_gamma = flat.data_ptr() + es * lay.offsets[id(extra[2])]
        cg.grad_scale = float(grad_scale)
        cg.accumulate = 0
        # second_variance / zeta / unused scalar slots (tiny): cleared by the compositing adjoint's reduce kernel, not by a launch of their own
        o0 = lay.offsets[id(extra[0])]
        cg.zero_tail, cg.n_zero_tail = flat.data_ptr() + es * o0, lay.numel - o0
        pg, keep = lay.tables(flat)
        pg.grad_scale = float(grad_scale)
        p = call["p"]
        L = _lib.lib()
        cfg = net.net_config()
        if packed is None:
            packed = net.packed(call["prec_name"])
        with _lib.on_device(call["ro"]):
            nbk = (N, S, prec, self.n_samples, self.n_importance, self.up_sample_steps)
            nbv = self._bws_bytes.get(nbk)
            if nbv is None:
                nb = C.c_size_t()
                _lib.check(L.emap_render_bwd_workspace_bytes(C.byref(cfg), prec, C.byref(p), C.byref(nb)), "render_bwd_workspace_bytes")
                nbv = self._bws_bytes[nbk] = nb.value
            lim = net.backward_workspace_limit
            ws = _workspace(self._bws, (N, S, prec), nbv if lim is None else min(nbv, int(lim)), dev)
            _lib.check(L.emap_render_bwd_staged(C.byref(cfg), _lib.ptr(packed), prec, C.byref(p),
                                                _lib.ptr(call["ro"]), _lib.ptr(call["rd"]), _lib.ptr(call["ds"]), _lib.ptr(v["z_vals"]),
                                                _lib.ptr(v["udf"]), _lib.ptr(v["gradients"]), _lib.ptr(v.get("_sd", v["_ws"])), C.byref(cg), C.byref(pg),
                                                _lib.ptr(ws), ws.numel() if lim is None else min(ws.numel(), int(lim)), _lib.ptr(self._err),
                                                _lib.stream_ptr(dev), int(stages)),
                       "render_bwd")
        return flat

    def bwd_absmax(self, call):
        """The two floats [max|dL/dudf|, max|dL/dgrad|] a stages=1 backward_into() left in the backward workspace, as a float32
        view of that workspace: what a data-parallel step max-reduces over its ranks before the stages=2 call, so that every rank's
        MLP backward uses the same fp16 range scale (emap_hip.h: emap_render_bwd_staged)."""
        N, S, dev = call["N"], call["S"], call["dev"]
        prec = _lib.PRECISIONS[call["prec_name"]]
        cfg = self.udf_network.net_config()
        p = call["p"]
        L = _lib.lib()
        nb, off = C.c_size_t(), C.c_size_t()
        _lib.check(L.emap_render_bwd_workspace_bytes(C.byref(cfg), prec, C.byref(p), C.byref(nb)), "render_bwd_workspace_bytes")
        _lib.check(L.emap_render_bwd_absmax_offset(C.byref(cfg), prec, C.byref(p), C.byref(off)), "render_bwd_absmax_offset")
        lim = self.udf_network.backward_workspace_limit
        ws = _workspace(self._bws, (N, S, prec), nb.value if lim is None else min(nb.value, int(lim)), dev)
        return ws[off.value:off.value + 8].view(torch.float32)

    def _trainable(self):
        if not torch.is_grad_enabled():
            return False
        lay = self._layout()       # its tensor list is checked against the modules' current parameters (ParamLayout.check) where it matters
        return any(q.requires_grad for q in lay.tensors)

    def render(self, rays_o, rays_d, near, far, depth_scale, cos_anneal_ratio=None, perturb_overwrite=-1,
               background_rgb=None, flip_saturation=0, color_maps=None, pose=None, fx=None, fy=None, img_index=None,
               rays_uv=None, t_rand=None):
        """reference udf_renderer_blending.py:679-800.  Extra keyword `t_rand` ((N,1) in [-0.5,0.5)) injects
        the jitter draw (tests); otherwise it is drawn exactly like the reference: torch.rand([N,1]) on the
        CPU generator (:719)."""
        if self.inference_reduced and not self._trainable():
            return self._render_reduced_compat(rays_o, rays_d, near, far, depth_scale, cos_anneal_ratio, perturb_overwrite,
                                               background_rgb, flip_saturation, t_rand)
        call = self._prepare(rays_o, rays_d, near, far, depth_scale, cos_anneal_ratio, perturb_overwrite, background_rgb,
                             flip_saturation, t_rand)
        N, S, dev = call["N"], call["S"], call["dev"]
        train = self._trainable()
        extras_train = train and any(q.requires_grad for q in self._layout().extra)
        pre = None
        if extras_train or (train and self.host_mirror_scalars):
            # variance / beta / gamma as ordinary torch expressions (:466-472,656-658): differentiable when the scalars are trainable.
            # Evaluated BEFORE the forward is enqueued: they depend on the parameters only, and with host_mirror_scalars their values start
            # travelling to a pinned buffer now, so the runner's host reads of them (runner_udf.py:141-148) do not wait for the render.
            with torch.set_grad_enabled(extras_train):
                inv_s = self.deviation_network(torch.zeros([1, 3], device=dev))[:, :1].clip(1e-6, 1e6)
                pre = (1.0 / inv_s, 1.0 / self.beta_network.get_beta().clip(1e-6, 1e6), self.beta_network.get_gamma().clip(1e-6, 1e6))
            if self.host_mirror_scalars:
                if self._mirror is None or self._mirror.dev != dev:
                    from .host_scalars import ScalarMirror
                    self._mirror = ScalarMirror(dev)
                mirror = self._mirror.push(torch.cat([q.detach().reshape(1) for q in pre]))
        if train:
            lay = self._layout()
            res = RenderFn.apply(self, call, *lay.tensors)
            v = call["_v"]
        else:
            v = self._render_hip(call)
        out = {
            "udf": v["udf"].view(N, S), "edge": v["edge"].view(N, 1), "weights": v["weights"].view(N, S),
            "depth": v["depth"].view(N, 1), "gradient_error": v["scalars"][0], "gradient_error_near_surface": v["scalars"][1],
            "normals": v["normals"].view(N, 3), "gradients": v["gradients"].view(N, S, 3),
            "gradients_flip": v["gradients_flip"].view(N, S, 3), "gradient_mag": v["gradient_mag"].view(N, S),
            "weight_sum": v["weight_sum"].view(N, 1),
        }
        if train:
            out.update(dict(zip(RenderFn.DIFF + RenderFn.GUARDED, res)))
        sc = v["scalars"]
        if extras_train:
            s_val, beta_out, gamma_out = pre[0].expand(N * S, 1), pre[1], pre[2]
        else:
            # the same three numbers, written by the compositing kernel (no extra launches)
            s_val = sc[8:9].view(1, 1).expand(N * S, 1)
            beta_out = sc[9:10]
            gamma_out = sc[10:11]
        if pre is not None and self.host_mirror_scalars:
            from .host_scalars import HostScalar, LazyMaskable
            s_val, beta_out, gamma_out = (HostScalar.wrap(q, mirror, i) for i, q in enumerate((s_val, beta_out, gamma_out)))
            out["udf"] = out["udf"].as_subclass(LazyMaskable)     # runner_udf.py:126 indexes a reduction of it with a boolean mask: no sync for that
        return {
            "udf": out["udf"], "edge": out["edge"], "weight_sum": out["weight_sum"], "weight_sum_fg_bg": out["weight_sum"],
            "depth": out["depth"], "variance": s_val, "beta": beta_out, "gamma": gamma_out,
            "normals": out["normals"], "gradients": out["gradients"], "gradients_flip": out["gradients_flip"],
            "weights": out["weights"], "gradient_error": out["gradient_error"],
            "gradient_error_near_surface": out["gradient_error_near_surface"],
            "inside_sphere": v["inside_sphere"].view(N, S), "gradient_mag": out["gradient_mag"],
            "mid_z_vals": v["mid_z"].view(N, S), "dists": v["dists"].view(N, S),
            # extras (not in the reference dict)
            "z_vals": v["z_vals"].view(N, S), "alpha": out.get("alpha", v["alpha"].view(N, S)),
            "sparse_error": out.get("sparse_error", v["scalars"][2]),
            "eikonal_sums": v["scalars"][3:7],
        }

    def _render_reduced_compat(self, rays_o, rays_d, near, far, depth_scale, cos_anneal_ratio, perturb_overwrite, background_rgb,
                               flip_saturation, t_rand):
        """render() for a caller that consumes per-ray results only - the validation loop, runner_udf.py:333-407: ``edge``, ``depth``
        and  sum_s gradients_flip[:, s] * weights[:, s]  - served by the reduced-output launch mode (no per-sample tensor leaves the
        GPU kernels).  The per-sample entries the loop touches are GUARDS, not tensors (``_ReducedOperand``): the one expression of
        runner_udf.py:375-405, ``(gradients_flip * weights[:, :S, None]).sum(dim=1)``, evaluates to the rendered normals; any other use
        of them raises instead of handing out made-up per-sample values.  Selected by ``self.inference_reduced``
        (emap_amd.dropin.validate_wrapper sets it around Runner_UDF.validate); only without autograd."""
        o = self.render_reduced(rays_o, rays_d, near, far, depth_scale, cos_anneal_ratio, perturb_overwrite, background_rgb,
                                flip_saturation, t_rand)
        S = self.samples_per_ray
        flip = _ReducedOperand("gradients_flip", o["normals"], S)
        return {"edge": o["edge"], "depth": o["depth"], "weight_sum": o["weight_sum"], "weight_sum_fg_bg": o["weight_sum"],
                "normals": o["normals"], "gradient_error": o["gradient_error"], "sparse_error": o["sparse_error"],
                "gradients_flip": flip, "gradients": None,
                "weights": _ReducedOperand("weights", None, S), "inside_sphere": _ReducedOperand("inside_sphere", None, S), "reduced": True}

    def capture(self, rays_o, rays_d, near, far, depth_scale, cos_anneal_ratio=None, background_rgb=None, flip_saturation=0,
                t_rand=None, reduced=False):
        """Capture one inference render of this batch shape in a hipGraph (SURVEY par. 7.1 step 8 / H3: the 15 dependent launches
        of emap_render_fwd replay as ONE graph launch).  Returns a RenderGraph; see its docstring."""
        return RenderGraph(self, rays_o, rays_d, near, far, depth_scale, cos_anneal_ratio, background_rgb, flip_saturation, t_rand,
                           reduced)

    def live_buffers(self):
        """Strong references to every cached device buffer the launch chain may currently point at (workspaces, near/far constants,
        reduced-mode scratch, error word, packed weights, scratch of the UDF network): a captured graph keeps this list."""
        net = self.udf_network
        keep = [list(self._ws.values()), list(self._ws_pool.values()), list(self._bws.values()), self._err,
                [v for k, v in self._const.items() if k not in ("_scr", "_jit")], list(self._const.get("_scr", {}).values()),
                [b for _, b in net._pack_cache.values()], list(net._vjp_ws.values()), list(net._scratch.values()), net._err]
        return keep

    def render_reduced(self, rays_o, rays_d, near, far, depth_scale, cos_anneal_ratio=None, perturb_overwrite=-1,
                       background_rgb=None, flip_saturation=0, t_rand=None):
        """Inference-only render that writes just the per-ray results (edge, depth, normals, weight_sum): the launch mode of
        the full-image path (reference runner_udf.py:297-407 consumes only per-ray sums of the per-sample entries)."""
        call = self._prepare(rays_o, rays_d, near, far, depth_scale, cos_anneal_ratio, perturb_overwrite, background_rgb,
                             flip_saturation, t_rand, reduced=True)
        v = self._render_hip(call)
        N = call["N"]
        return {"edge": v["edge"].view(N, 1), "depth": v["depth"].view(N, 1), "normals": v["normals"].view(N, 3),
                "weight_sum": v["weight_sum"].view(N, 1), "gradient_error": v["scalars"][0], "sparse_error": v["scalars"][2]}


class RenderGraph:
    """A captured ``render()`` (inference): static input buffers + one hipGraph of the whole launch chain.

        g = renderer.capture(rays_o, rays_d, near, far, depth_scale, cos_anneal_ratio=1.0, t_rand=t)   # shapes are fixed here
        out = g(rays_o2, rays_d2, near2, far2, depth_scale2, t_rand=t2)     # copies into the static buffers, replays, returns

    The returned dict aliases the graph's static output buffers: consume (or clone) it before the next replay.  Scalars that
    the kernels read from device memory (variance / beta / gamma, the packed weights) are read at replay time, so parameter
    updates are seen as long as ``UDFNetwork.packed()`` is refreshed by the caller after an optimizer step (it is a kernel
    launch outside the graph).  cos_anneal_ratio / flip_saturation / background are baked in at capture.
    The jitter must be passed explicitly (`t_rand`, (N,1) in [-0.5, 0.5)): the reference's CPU-generator draw plus its
    host-to-device copy (udf_renderer_blending.py:719) cannot be part of a device graph."""

    def __init__(self, r, rays_o, rays_d, near, far, depth_scale, cos_anneal_ratio, background_rgb, flip_saturation, t_rand, reduced):
        _lib.require_cuda(rays_o, "rays_o")
        if r.perturb > 0 and t_rand is None:
            raise ValueError("RenderGraph: pass t_rand explicitly (or construct the renderer with perturb=0)")
        self.r = r
        dev = rays_o.device
        N = len(rays_o)
        st = lambda x: None if x is None else _lib.f32c(x.detach().to(dev)).clone()
        self.ro, self.rd, self.ds, self.tr = st(rays_o), st(rays_d), st(depth_scale), st(t_rand)
        if isinstance(near, torch.Tensor):
            self.near, self.far = st(near.reshape(-1).expand(N)), st(far.reshape(-1).expand(N))
        else:
            self.near, self.far = float(near), float(far)
        self.kw = dict(cos_anneal_ratio=cos_anneal_ratio, background_rgb=background_rgb, flip_saturation=flip_saturation)
        self.reduced = reduced
        r.udf_network.packed(r.precision)                        # pack outside the capture
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(2):                                    # allocate workspaces / set function attributes before capturing
                self._run()
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.out = self._run()
        self._keep = r.live_buffers()   # the graph has these device pointers baked in: they live as long as the graph does

    def _run(self):
        r = self.r
        f = r.render_reduced if self.reduced else r.render
        return f(self.ro, self.rd, self.near, self.far, self.ds, perturb_overwrite=(-1 if self.tr is not None else 0), t_rand=self.tr,
                 **self.kw)

    def __call__(self, rays_o=None, rays_d=None, near=None, far=None, depth_scale=None, t_rand=None):
        for dst, src in ((self.ro, rays_o), (self.rd, rays_d), (self.ds, depth_scale), (self.tr, t_rand)):
            if src is not None:
                dst.copy_(src.reshape(dst.shape))
        if near is not None and isinstance(self.near, torch.Tensor):
            self.near.copy_(near.reshape(-1).expand_as(self.near)); self.far.copy_(far.reshape(-1).expand_as(self.far))
        self.graph.replay()
        return self.out
