"""INTERIM training backward (SURVEY.md par. 8 row f1) - PyTorch-ROCm recomputation on the GPU.

The forward render is HIP (libemap_hip).  Until the ``composite_bwd`` / ``udf_mlp_vjp`` kernels land
(next-round item f1), parameter gradients are obtained by re-evaluating ``render_core`` with
differentiable torch ops *on the device* from the z_vals the HIP sampler produced (the reference's
importance_sample is ``@torch.no_grad`` and detaches z_samples, udf_renderer_blending.py:344,802, so
this is exactly the graph the reference differentiates) and calling autograd on it.  Nothing here
runs on the CPU in the product, and nothing here is used by the forward/inference path.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from .embedder import embed_torch


def _folded_weights(net):
    Ws, bs = [], []
    for l in range(net.num_layers - 1):
        lin = getattr(net, "lin" + str(l))
        Ws.append(lin.weight)  # the parametrization evaluates g * v / ||v||
        bs.append(lin.bias)
    return Ws, bs


def udf_forward_torch(net, inputs):
    """Differentiable UDFNetwork.forward (reference udf_model.py:90-110)."""
    Ws, bs = _folded_weights(net)
    xs = inputs * net.scale
    pe = embed_torch(xs, net.multires) if net.multires > 0 else xs
    x = pe
    n_lin = net.num_layers - 1
    for l in range(n_lin):
        if l in net.skip_in:
            x = torch.cat([x, pe], 1) / np.sqrt(2)
        x = F.linear(x, Ws[l], bs[l])
        if l < n_lin - 1:
            x = F.softplus(x, beta=100)
    out = torch.cat([net.udf_out(x[:, :1]) / net.scale, x[:, 1:]], dim=-1)
    return out, pe


def udf_gradient_torch(net, x):
    """Differentiable UDFNetwork.gradient (reference udf_model.py:121-135)."""
    x.requires_grad_(True)
    with torch.enable_grad():
        y = udf_forward_torch(net, x)[0][:, :1]
        g = torch.autograd.grad(y, x, torch.ones_like(y), create_graph=True, retain_graph=True, only_inputs=True)[0]
    return g.unsqueeze(1)


def sdf2alpha_torch(sdf, true_cos, dists, inv_s, cos_anneal_ratio=None):
    """reference udf_renderer_blending.py:379-411 ('numerical')."""
    if cos_anneal_ratio is not None:
        iter_cos = -(F.relu(-true_cos * 0.5 + 0.5) * (1.0 - cos_anneal_ratio) + F.relu(-true_cos) * cos_anneal_ratio)
    else:
        iter_cos = true_cos
    est_next = sdf + iter_cos * dists * 0.5
    est_prev = sdf - iter_cos * dists * 0.5
    prev_cdf = torch.sigmoid(est_prev * inv_s)
    next_cdf = torch.sigmoid(est_next * inv_s)
    return ((prev_cdf - next_cdf + 1e-5) / (prev_cdf + 1e-5)).clip(0.0, 1.0)


def render_core_torch(renderer, rays_o, rays_d, z_vals, sample_dist, cos_anneal_ratio, background_rgb,
                      flip_saturation):
    """Differentiable render_core (reference udf_renderer_blending.py:418-677) on z_vals (no grad)."""
    net, dev_net, beta_net = renderer.udf_network, renderer.deviation_network, renderer.beta_network
    N, S = z_vals.shape
    dev = z_vals.device
    one = torch.ones([N, 1], device=dev)
    dists = z_vals[..., 1:] - z_vals[..., :-1]
    dists = torch.cat([dists, sample_dist.reshape(1, 1).expand(N, 1)], -1)
    mid_z = z_vals + dists * 0.5
    pts = (rays_o[:, None, :] + rays_d[:, None, :] * mid_z[..., :, None]).reshape(-1, 3)
    dirs = rays_d[:, None, :].expand(N, S, 3).reshape(-1, 3)

    p = pts.detach().clone().requires_grad_(True)
    udf = udf_forward_torch(net, p)[0][:, :1]
    gradients = torch.autograd.grad(udf, p, torch.ones_like(udf), create_graph=True, retain_graph=True)[0]

    gmag = torch.linalg.norm(gradients, ord=2, dim=-1, keepdim=True)
    gnorm = gradients / (gmag + 1e-5)
    inv_s = dev_net(torch.zeros([1, 3], device=dev))[:, :1].clip(1e-6, 1e6).expand(N * S, 1)
    beta = beta_net.get_beta().clip(1e-6, 1e6)
    gamma = beta_net.get_gamma().clip(1e-6, 1e6)
    true_cos = (dirs * gradients).sum(-1, keepdim=True)
    with torch.no_grad():
        cos = (dirs * gnorm).sum(-1, keepdim=True)
        flip_sign = torch.sign(cos) * -1
        flip_sign[flip_sign == 0] = 1
    e = torch.exp(-beta * udf)
    raw_occ = (beta * e / (1 + e) ** 2).reshape(N, S)
    alpha_occ = 1.0 - torch.exp(-F.relu(raw_occ) * gamma * dists)
    vis_mask = (true_cos < 0.01).float().reshape(N, S)
    vis_mask = torch.cat([vis_mask[:, 1:], one], dim=-1)
    vis_prob = torch.cumprod(torch.cat([one, (1.0 - alpha_occ + flip_saturation * vis_mask).clip(0, 1) + 1e-7], -1), -1)[:, :-1]
    vis_prob = vis_prob.clip(0, 1)
    ap = sdf2alpha_torch(udf, -1 * torch.abs(true_cos), dists.view(-1, 1), inv_s, cos_anneal_ratio).reshape(N, S)
    am = sdf2alpha_torch(-udf, -1 * torch.abs(true_cos), dists.view(-1, 1), inv_s, cos_anneal_ratio).reshape(N, S)
    alpha = ap * vis_prob + am * (1 - vis_prob)
    udf = udf.reshape(N, S)
    pts_norm = torch.linalg.norm(pts, ord=2, dim=-1, keepdim=True).reshape(N, S)
    relax = (pts_norm < 2.4).float().detach()
    near_surface = (udf < renderer.near_surface).float().detach()
    weights = alpha * torch.cumprod(torch.cat([one, 1.0 - alpha + 1e-7], -1), -1)[:, :-1]
    wsum = weights.sum(dim=-1, keepdim=True)
    edge = wsum
    if background_rgb is not None:
        edge = edge + background_rgb * (1.0 - wsum)
    depth = (mid_z * weights).sum(dim=1, keepdim=True)
    gerr = (torch.linalg.norm(gradients.reshape(N, S, 3), ord=2, dim=-1) - 1.0) ** 2
    sums = {"e_rel": (relax * gerr).sum(), "c_rel": relax.sum(), "e_ns": (near_surface * gerr).sum(),
            "c_ns": near_surface.sum()}
    gradients = gradients.reshape(N, S, 3)
    gflip = flip_sign.reshape(N, S, 1) * gradients
    return {"udf": udf, "edge": edge, "weights": weights, "depth": depth,
            "gradient_error": sums["e_rel"] / (sums["c_rel"] + 1e-5),
            "gradient_error_near_surface": sums["e_ns"] / (sums["c_ns"] + 1e-5),
            "normals": (gflip * weights[:, :, None]).sum(dim=1), "gradients": gradients, "gradients_flip": gflip,
            "gradient_mag": gmag.reshape(N, S), "weight_sum": wsum, "_sums": sums}


DIFF_KEYS = ("udf", "edge", "weights", "depth", "gradient_error", "gradient_error_near_surface", "normals",
             "gradients", "gradients_flip", "gradient_mag", "weight_sum")


class RenderCoreInterim(torch.autograd.Function):
    """forward: passes the HIP results through; backward: autograd through render_core_torch."""

    @staticmethod
    def forward(ctx, renderer, rays_o, rays_d, z_vals, sample_dist, cos_anneal_ratio, background_rgb, flip_saturation,
                depth_scale, n_params, *tensors):
        params = tensors[:n_params]
        hip_vals = tensors[n_params:]
        ctx.renderer = renderer
        ctx.args = (rays_o, rays_d, z_vals, sample_dist, cos_anneal_ratio, background_rgb, flip_saturation, depth_scale)
        ctx.params = params
        return tuple(v.clone() for v in hip_vals)  # clone: outputs must not alias inputs

    @staticmethod
    def backward(ctx, *grads):
        rays_o, rays_d, z_vals, sample_dist, car, bg, fs, depth_scale = ctx.args
        with torch.enable_grad():
            out = render_core_torch(ctx.renderer, rays_o, rays_d, z_vals, sample_dist, car, bg, fs)
            out["depth"] = out["depth"] * depth_scale
            outs, gos = [], []
            for k, g in zip(DIFF_KEYS, grads):
                if g is not None and out[k].requires_grad:
                    outs.append(out[k]); gos.append(g.reshape(out[k].shape))
            ps = [p for p in ctx.params if p.requires_grad]
            pg = torch.autograd.grad(outs, ps, gos, allow_unused=True) if outs and ps else []
        it = iter(pg)
        param_grads = [next(it) if p.requires_grad else None for p in ctx.params]
        return (None,) * 10 + tuple(param_grads) + (None,) * len(DIFF_KEYS)
