"""Algorithm mirrors of the training-backward kernels.  TEST INFRASTRUCTURE ONLY (see emap_oracle.py).

The HIP backward (SURVEY.md par. 8 f1: ``emap_composite_bwd`` + ``emap_udf_vjp``) is hand-derived calculus, not
autograd.  This file restates the *same derivation* with plain torch CPU ops so that ``tests/test_vjp_math.py`` can
check it against ``torch.autograd`` through the oracle (which is itself pinned to the reference by the goldens)
before / independently of the GPU: a wrong formula fails here, on the CPU, in seconds.

What the reference differentiates (udf_renderer_blending.py:457-625 under autograd, udf_model.py:121-135 with
``create_graph=True``, runner_udf.py:124-168):   loss(edge, gradient_error, gradient_error_near_surface), where
every quantity depends on theta only through  u = udf(x; theta)  and  g = grad_x udf(x; theta)  at the (detached)
sample points.  So the backward splits into

  composite_bwd :  dL/d{edge, depth, ge, ge_ns}  ->  dL/du (N,S), dL/dg (N,S,3), dL/d{inv_s, beta, gamma}
  mlp_vjp       :  (dL/du, dL/dg)                ->  dL/dW_l, dL/db_l    (then the weight-norm VJP -> dL/dg_l, dL/dv_l)

and the second one needs no second-order autograd: per point  phi = du*u + dg.g = du*U(h)/scale + U'(h) * D_v h  with
v = dg, i.e. ONE forward-mode tangent column along v next to the value column, followed by one reverse sweep over
both columns (6F MFMA work per point instead of the 12F of a 3-tangent formulation).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import emap_oracle as O


# ---------------------------------------------------------------------------------------------
# MLP: value + one tangent column forward, two adjoint columns backward
# ---------------------------------------------------------------------------------------------
def pe_and_tangent(xs: torch.Tensor, v: torch.Tensor, multires: int):
    """PE(xs) and its directional derivative along v (embedder.py:10-35)."""
    pe = [xs]
    dpe = [v]
    for k in range(multires):
        f = float(2 ** k)
        pe += [torch.sin(xs * f), torch.cos(xs * f)]
        dpe += [f * torch.cos(xs * f) * v, -f * torch.sin(xs * f) * v]
    return torch.cat(pe, -1), torch.cat(dpe, -1)


def mlp_vjp(state: dict, cfg: O.UDFConfig, x: torch.Tensor, du: torch.Tensor, dg: torch.Tensor):
    """d/dtheta of  sum_p du[p]*udf(x_p) + dg[p].grad_x udf(x_p)   (x detached).

    Returns ({"lin{l}.weight": dW_l (grad w.r.t. the folded weight g*v/||v||), "lin{l}.bias": db_l}, extras).
    Mirrors the sweep kernel: per layer  z = W a + b, z' = W a', a+ = softplus(z), a'+ = s z'  (s = sigmoid(100 z));
    backward  zb = s ab + 100 (1-s) a'+ ab',  zb' = s ab'   [since s'' z' = 100 s (1-s) z' = 100 (1-s) a'+],
    dW += zb a^T + zb' a'^T,  db += zb,  ab = W^T zb,  ab' = W^T zb'."""
    dt = x.dtype
    Ws, bs = O._weights(state, cfg, dt)
    n_lin = cfg.n_lin
    xs = x * cfg.scale
    v = dg.to(dt)                      # tangent direction in scaled coordinates (d xs/dx = scale cancels udf/scale)
    pe, dpe = pe_and_tangent(xs, v, cfg.multires)
    a, ap = pe, dpe
    ins, acts = [], []
    for l in range(n_lin):
        if l in cfg.skip_in:
            a = torch.cat([a, pe], 1) / np.sqrt(2)
            ap = torch.cat([ap, dpe], 1) / np.sqrt(2)
        ins.append((a, ap))
        z = F.linear(a, Ws[l], bs[l])
        zp = F.linear(ap, Ws[l])
        if l < n_lin - 1:
            s = torch.sigmoid(100.0 * z)
            a, ap = F.softplus(z, beta=100), s * zp
            acts.append((s, ap))
        else:
            h, hp = z[:, :1], zp[:, :1]
    if cfg.udf_type == "abs":
        U1, U2 = torch.sign(h), torch.zeros_like(h)
    elif cfg.udf_type == "square":
        U1, U2 = 2 * h, torch.full_like(h, 2.0)
    else:
        U1, U2 = torch.ones_like(h), torch.zeros_like(h)
    hb = du.reshape(-1, 1).to(dt) * U1 / cfg.scale + U2 * hp      # adjoint of h   (value column)
    hbp = U1                                                       # adjoint of h'  (tangent column)
    grads = {}
    zb = torch.zeros(x.shape[0], Ws[-1].shape[0], dtype=dt); zb[:, :1] = hb
    zbp = torch.zeros_like(zb); zbp[:, :1] = hbp
    for l in range(n_lin - 1, -1, -1):
        a, ap = ins[l]
        grads[f"lin{l}.weight"] = zb.t() @ a + zbp.t() @ ap
        grads[f"lin{l}.bias"] = zb.sum(0)
        if l == 0:
            break
        ab, abp = zb @ Ws[l], zbp @ Ws[l]
        if l in cfg.skip_in:
            n_prev = Ws[l].shape[1] - pe.shape[1]
            ab, abp = ab[:, :n_prev] / np.sqrt(2), abp[:, :n_prev] / np.sqrt(2)
        s, apl = acts[l - 1]
        zb = s * ab + 100.0 * (1.0 - s) * apl * abp
        zbp = s * abp
    return grads, {"h": h, "hp": hp}


def weight_norm_vjp(g: torch.Tensor, v: torch.Tensor, dW: torch.Tensor):
    """W = g v / ||v||_row  ->  (dg [out,1], dv [out,in])   (nn.utils.parametrizations.weight_norm, udf_model.py:73-74)."""
    n = torch.linalg.norm(v, dim=1, keepdim=True)
    dot = (dW * v).sum(1, keepdim=True)
    return dot / n, g / n * (dW - dot * v / (n * n))


# ---------------------------------------------------------------------------------------------
# compositing: reverse of render_core's tail (udf_renderer_blending.py:463-625)
# ---------------------------------------------------------------------------------------------
def _sdf2alpha_bwd(sdf, tabs, dists, inv_s, car, dval):
    """Backward of sdf2alpha(sdf, -tabs, dists, inv_s, car) (udf_renderer_blending.py:379-411, numerical branch) for an
    upstream gradient `dval` on its (clipped) output.  tabs = |true_cos| >= 0.  Returns (d_sdf, d_tabs, d_inv_s)."""
    if car is not None:
        ic = -((0.5 * tabs + 0.5) * (1.0 - car) + tabs * car)
        dic_dt = torch.where(tabs > 0, -(0.5 * (1.0 - car) + car) * torch.ones_like(tabs), -(0.5 * (1.0 - car)) * torch.ones_like(tabs))
    else:
        ic = -tabs
        dic_dt = -torch.ones_like(tabs)
    hh = ic * dists * 0.5
    en, ep = sdf + hh, sdf - hh
    pc, nc = torch.sigmoid(ep * inv_s), torch.sigmoid(en * inv_s)
    val = (pc - nc + 1e-5) / (pc + 1e-5)
    live = ((val >= 0) & (val <= 1)).to(sdf.dtype)
    dv = dval * live
    dpc = dv * nc / (pc + 1e-5) ** 2
    dnc = -dv / (pc + 1e-5)
    dep = dpc * pc * (1 - pc) * inv_s
    den = dnc * nc * (1 - nc) * inv_s
    d_inv_s = dpc * pc * (1 - pc) * ep + dnc * nc * (1 - nc) * en
    d_sdf = dep + den
    d_hh = den - dep
    d_tabs = d_hh * dists * 0.5 * dic_dt
    return d_sdf, d_tabs, d_inv_s


def composite_bwd(rays_o, rays_d, z_vals, sample_dist, udf, grads, inv_s, beta, gamma, cos_anneal_ratio,
                  flip_saturation, near_surface, background, d_edge, d_depth, depth_scale, c_ge, c_ns):
    """Reverse of render_core's tail.  udf (N,S), grads (N,S,3); d_edge, d_depth (N,1) or None;
    c_ge = dL/d(gradient_error) / (sum(relax)+1e-5), c_ns likewise for the near-surface term (scalars).
    Returns d_udf (N,S), d_grad (N,S,3), d_inv_s, d_beta, d_gamma (0-d)."""
    N, S = z_vals.shape
    dt = z_vals.dtype
    one = torch.ones(N, 1, dtype=dt)
    dists = torch.cat([z_vals[:, 1:] - z_vals[:, :-1], torch.full((N, 1), float(sample_dist), dtype=dt)], -1)
    mid = z_vals + dists * 0.5
    pts = rays_o[:, None, :] + rays_d[:, None, :] * mid[..., None]
    tc = (rays_d[:, None, :] * grads).sum(-1)
    tabs = tc.abs()
    # ---- forward recompute ----
    E = torch.exp(-beta * udf)
    raw = beta * E / (1 + E) ** 2
    q = F.relu(raw) * gamma * dists
    occ = 1.0 - torch.exp(-q)
    vm = torch.cat([(tc[:, 1:] < 0.01).to(dt), one], -1)
    a_in = 1.0 - occ + flip_saturation * vm
    a = a_in.clip(0, 1) + 1e-7
    vp_raw = torch.cumprod(torch.cat([one, a], -1), -1)[:, :-1]
    vp = vp_raw.clip(0, 1)
    ap = O.sdf2alpha(udf.reshape(-1, 1), -tabs.reshape(-1, 1), dists.reshape(-1, 1), inv_s, cos_anneal_ratio).reshape(N, S)
    am = O.sdf2alpha(-udf.reshape(-1, 1), -tabs.reshape(-1, 1), dists.reshape(-1, 1), inv_s, cos_anneal_ratio).reshape(N, S)
    alpha = ap * vp + am * (1 - vp)
    om = 1.0 - alpha + 1e-7
    T = torch.cumprod(torch.cat([one, om], -1), -1)[:, :-1]
    w = alpha * T
    # ---- backward ----
    dw = torch.zeros(N, S, dtype=dt)
    if d_edge is not None:
        dw = dw + d_edge.reshape(N, 1) * (1.0 - (background if background is not None else 0.0))
    if d_depth is not None:
        dw = dw + d_depth.reshape(N, 1) * depth_scale.reshape(N, 1) * mid
    ww = dw * w
    suffix = torch.flip(torch.cumsum(torch.flip(ww, [1]), 1), [1]) - ww          # sum_{k>e} dw_k w_k
    dalpha = dw * T - suffix / om
    dap, dam = dalpha * vp, dalpha * (1 - vp)
    dvp = dalpha * (ap - am) * ((vp_raw >= 0) & (vp_raw <= 1)).to(dt)
    vv = dvp * vp_raw
    da = (torch.flip(torch.cumsum(torch.flip(vv, [1]), 1), [1]) - vv) / a        # sum_{e>i} dvp_e vp_raw_e / a_i
    docc = -da * ((a_in >= 0) & (a_in <= 1)).to(dt)
    dq = docc * (1.0 - occ)
    draw = dq * gamma * dists * (raw > 0).to(dt)
    d_gamma = (dq * F.relu(raw) * dists).sum()
    fE = (1 - E) / (1 + E) ** 3
    d_udf = draw * (-beta * beta * E * fE)
    d_beta = (draw * (E / (1 + E) ** 2 - beta * udf * E * fE)).sum()
    s1, t1, i1 = _sdf2alpha_bwd(udf, tabs, dists, inv_s, cos_anneal_ratio, dap)
    s2, t2, i2 = _sdf2alpha_bwd(-udf, tabs, dists, inv_s, cos_anneal_ratio, dam)
    d_udf = d_udf + s1 - s2
    d_tabs = t1 + t2
    d_inv_s = (i1 + i2).sum()
    d_tc = d_tabs * torch.sign(tc)
    d_grad = d_tc[..., None] * rays_d[:, None, :]
    # eikonal terms (:612-625): masks are detached
    gm = torch.linalg.norm(grads, dim=-1)
    relax = (torch.linalg.norm(pts, dim=-1) < 2.4).to(dt)
    ns = (udf < near_surface).to(dt)
    coef = (c_ge * relax + c_ns * ns) * 2.0 * (gm - 1.0) / gm.clamp_min(1e-30)
    d_grad = d_grad + coef[..., None] * grads
    return d_udf, d_grad, d_inv_s, d_beta, d_gamma
