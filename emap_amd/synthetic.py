"""Deterministic synthetic inputs for the EMAP render hot path.

There is no dataset or checkpoint in the build/benchmark environment, so every
test, golden vector and benchmark uses the generators in this file:

* ``udf_layer_dims`` / ``make_udf_state`` - a seeded (NumPy PCG64) UDF-MLP
  state dict with the reference's geometric-init statistics
  (reference ``src/models/udf_model.py:24-71``) plus an N(0, pert^2)
  perturbation, in the reference's own state-dict key layout
  (``lin{l}.bias``, ``lin{l}.parametrizations.weight.original0`` = g,
  ``...original1`` = v), so the same dict loads into the reference
  ``UDFNetwork`` (golden generation) and into ``emap_amd.UDFNetwork``.
* ``make_rays`` - cameras on a radius-3 sphere looking at the origin, 60 degree
  frustum, ``near``/``far`` as (N,1) tensors (reference ``render`` :700 branch).

Only NumPy is used for the random streams so results do not depend on the
torch version.
"""
from __future__ import annotations

import numpy as np
import torch


def pe_dim(multires: int, d_in: int = 3) -> int:
    """Embedding width 3 + 6*multires (reference embedder.py:14-29)."""
    return d_in + 2 * d_in * multires if multires > 0 else d_in


def udf_layer_dims(d_in=3, d_out=1, d_hidden=256, n_layers=8, skip_in=(4,), multires=10):
    """[(out, in)] for lin0..lin{n_layers} (reference udf_model.py:24-45)."""
    dims = [d_in] + [d_hidden] * n_layers + [d_out]
    dims[0] = pe_dim(multires, d_in)
    shapes = []
    for l in range(len(dims) - 1):
        out_dim = dims[l + 1] - dims[0] if (l + 1) in skip_in else dims[l + 1]
        shapes.append((out_dim, dims[l]))
    return shapes


def make_udf_state(
    d_in=3,
    d_out=1,
    d_hidden=256,
    n_layers=8,
    skip_in=(4,),
    multires=10,
    bias=0.5,
    seed=42,
    pert=0.02,
    dtype=torch.float32,
):
    """Seeded state dict with geometric-init statistics + perturbation."""
    rng = np.random.Generator(np.random.PCG64(seed))
    shapes = udf_layer_dims(d_in, d_out, d_hidden, n_layers, skip_in, multires)
    d0 = shapes[0][1]
    n_lin = len(shapes)
    state = {}
    for l, (out_dim, in_dim) in enumerate(shapes):
        if l == n_lin - 1:
            v = rng.normal(np.sqrt(np.pi) / np.sqrt(in_dim), 1e-4, size=(out_dim, in_dim))
            b = np.full((out_dim,), -bias)
        elif multires > 0 and l == 0:
            v = np.zeros((out_dim, in_dim))
            v[:, :3] = rng.normal(0.0, np.sqrt(2) / np.sqrt(out_dim), size=(out_dim, 3))
            b = np.zeros((out_dim,))
        elif multires > 0 and l in skip_in:
            v = rng.normal(0.0, np.sqrt(2) / np.sqrt(out_dim), size=(out_dim, in_dim))
            v[:, -(d0 - 3):] = 0.0
            b = np.zeros((out_dim,))
        else:
            v = rng.normal(0.0, np.sqrt(2) / np.sqrt(out_dim), size=(out_dim, in_dim))
            b = np.zeros((out_dim,))
        g = np.linalg.norm(v, axis=1, keepdims=True)
        if pert > 0:
            v = v + pert * rng.normal(size=v.shape)
            b = b + pert * rng.normal(size=b.shape)
            g = g * (1.0 + pert * rng.normal(size=g.shape))
        state[f"lin{l}.bias"] = torch.tensor(b, dtype=dtype)
        state[f"lin{l}.parametrizations.weight.original0"] = torch.tensor(g, dtype=dtype)
        state[f"lin{l}.parametrizations.weight.original1"] = torch.tensor(v, dtype=dtype)
    return state


def make_rays(n_rays: int, seed: int = 1, near: float = 0.05, far: float = 6.0, radius: float = 3.0,
              fov_deg: float = 60.0, n_cams: int = 4, dtype=torch.float32):
    """Synthetic rays: (rays_o, rays_d, near(N,1), far(N,1), depth_scale(N,1))."""
    rng = np.random.Generator(np.random.PCG64(seed))
    cam_id = rng.integers(0, n_cams, size=n_rays)
    # camera centres on the sphere
    phi = rng.uniform(0, 2 * np.pi, size=n_cams)
    cos_t = rng.uniform(-0.6, 0.6, size=n_cams)
    sin_t = np.sqrt(1 - cos_t ** 2)
    centres = radius * np.stack([sin_t * np.cos(phi), sin_t * np.sin(phi), cos_t], -1)
    rays_o = centres[cam_id]
    # camera frame: z looks at origin
    fwd = -rays_o / np.linalg.norm(rays_o, axis=-1, keepdims=True)
    up = np.array([0.0, 0.0, 1.0])
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right, axis=-1, keepdims=True)
    upv = np.cross(right, fwd)
    t = np.tan(np.deg2rad(fov_deg) / 2)
    u = rng.uniform(-t, t, size=(n_rays, 1))
    v = rng.uniform(-t, t, size=(n_rays, 1))
    d_cam = np.concatenate([u, v, np.ones_like(u)], -1)
    d_cam /= np.linalg.norm(d_cam, axis=-1, keepdims=True)
    rays_d = d_cam[:, 0:1] * right + d_cam[:, 1:2] * upv + d_cam[:, 2:3] * fwd
    depth_scale = d_cam[:, 2:3]
    to = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=dtype)
    near_t = torch.full((n_rays, 1), near, dtype=dtype)
    far_t = torch.full((n_rays, 1), far, dtype=dtype)
    return to(rays_o), to(rays_d), near_t, far_t, to(depth_scale)


def make_t_rand(n_rays: int, seed: int = 7, dtype=torch.float32):
    """Per-ray jitter U(-0.5, 0.5) (stands in for the CPU torch.rand draw of render() :719)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return torch.tensor(rng.uniform(-0.5, 0.5, size=(n_rays, 1)), dtype=dtype)


def make_true_edge(n_rays: int, seed: int = 11, dtype=torch.float32):
    """Targets: Bernoulli(0.1) * U(0.5, 1)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    on = rng.uniform(size=(n_rays, 1)) < 0.1
    val = rng.uniform(0.5, 1.0, size=(n_rays, 1))
    return torch.tensor(on * val, dtype=dtype)


def make_scene(n_images: int = 8, H: int = 100, W: int = 120, seed: int = 3, radius: float = 3.0, fov_deg: float = 50.0):
    """Synthetic stand-in for an EMAP dataset directory (no datasets travel): `meta` in the wire format of meta_data.json
    (reference src/dataset/dataset.py:66-104: scene_box{near,far,radius,aabb}, height, width, frames[{intrinsics 4x4,
    camtoworld 4x4, rgb_path}]) and edge maps (n_images,H,W,1) in [0,1] as cv.imread(...,0)/255 would give them (:133-135):
    a few anti-aliased line segments per image on a dark background."""
    g = np.random.Generator(np.random.PCG64(seed))
    f = 0.5 * W / np.tan(np.deg2rad(fov_deg) / 2)
    K = np.array([[f, 0, (W - 1) / 2, 0], [0, f, (H - 1) / 2, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float64)
    frames, edges = [], np.zeros((n_images, H, W, 1), dtype=np.float32)
    ys, xs = np.mgrid[0:H, 0:W]
    for i in range(n_images):
        th, ph = 2 * np.pi * i / n_images, 0.3 * np.sin(1.7 * i)
        c = radius * np.array([np.cos(th) * np.cos(ph), np.sin(ph), np.sin(th) * np.cos(ph)])
        zax = -c / np.linalg.norm(c)
        xax = np.cross([0.0, 1.0, 0.0], zax); xax /= np.linalg.norm(xax)
        yax = np.cross(zax, xax)
        c2w = np.eye(4); c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = xax, yax, zax, c
        frames.append({"intrinsics": K.tolist(), "camtoworld": c2w.tolist(), "rgb_path": f"{i:04d}.png"})
        img = np.zeros((H, W))
        for _ in range(4):
            x0, y0, x1, y1 = g.uniform(0, W), g.uniform(0, H), g.uniform(0, W), g.uniform(0, H)
            t = np.clip(((xs - x0) * (x1 - x0) + (ys - y0) * (y1 - y0)) / ((x1 - x0) ** 2 + (y1 - y0) ** 2 + 1e-9), 0, 1)
            d = np.hypot(xs - (x0 + t * (x1 - x0)), ys - (y0 + t * (y1 - y0)))
            img = np.maximum(img, np.clip(1.5 - d, 0, 1) * g.uniform(0.5, 1.0))
        edges[i, :, :, 0] = np.round(img * 255) / 255
    meta = {"scene_box": {"near": 0.05, "far": 6.0, "radius": 1.0, "aabb": [[-1, -1, -1], [1, 1, 1]]}, "height": H, "width": W,
            "frames": frames}
    return meta, edges


def wireframe_segments(half: float = 0.45):
    """The 12 edges of an axis-aligned cube of half-size `half` plus one face diagonal: (13, 2, 3) end points."""
    c = np.array([[x, y, z] for x in (-half, half) for y in (-half, half) for z in (-half, half)], dtype=np.float64)
    segs = [(c[i], c[j]) for i in range(8) for j in range(i + 1, 8) if np.sum(np.abs(c[i] - c[j]) > 1e-9) == 1]
    segs.append((c[0], c[3]))
    return np.array(segs)


def make_wireframe_scene(n_images: int = 16, H: int = 200, W: int = 200, radius: float = 3.0, fov_deg: float = 30.0, half: float = 0.45):
    """A MULTI-VIEW CONSISTENT synthetic dataset in the same wire format as make_scene: the edge maps are the anti-aliased
    projections of one 3D wire frame (wireframe_segments) into cameras on a ring around it - something a UDF can actually fit,
    for the convergence run of scripts/train_synthetic.py (BASELINE config C5 in miniature)."""
    f = 0.5 * W / np.tan(np.deg2rad(fov_deg) / 2)
    K = np.array([[f, 0, (W - 1) / 2, 0], [0, f, (H - 1) / 2, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float64)
    segs = wireframe_segments(half)
    frames, edges = [], np.zeros((n_images, H, W, 1), dtype=np.float32)
    ys, xs = np.mgrid[0:H, 0:W]
    for i in range(n_images):
        th, ph = 2 * np.pi * i / n_images + 0.2, 0.45 * np.sin(2.3 * i + 0.5)
        c = radius * np.array([np.cos(th) * np.cos(ph), np.sin(ph), np.sin(th) * np.cos(ph)])
        zax = -c / np.linalg.norm(c)
        xax = np.cross([0.0, 1.0, 0.0], zax); xax /= np.linalg.norm(xax)
        yax = np.cross(zax, xax)
        c2w = np.eye(4); c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = xax, yax, zax, c
        frames.append({"intrinsics": K.tolist(), "camtoworld": c2w.tolist(), "rgb_path": f"{i:04d}.png"})
        R = c2w[:3, :3]
        img = np.zeros((H, W))
        for a, b in segs:
            pa, pb = R.T @ (a - c), R.T @ (b - c)                     # camera coordinates (z forward)
            x0, y0 = f * pa[0] / pa[2] + (W - 1) / 2, f * pa[1] / pa[2] + (H - 1) / 2
            x1, y1 = f * pb[0] / pb[2] + (W - 1) / 2, f * pb[1] / pb[2] + (H - 1) / 2
            t = np.clip(((xs - x0) * (x1 - x0) + (ys - y0) * (y1 - y0)) / ((x1 - x0) ** 2 + (y1 - y0) ** 2 + 1e-9), 0, 1)
            d = np.hypot(xs - (x0 + t * (x1 - x0)), ys - (y0 + t * (y1 - y0)))
            img = np.maximum(img, np.clip(1.5 - d, 0, 1))
        edges[i, :, :, 0] = np.round(img * 255) / 255
    meta = {"scene_box": {"near": 0.05, "far": 6.0, "radius": 1.0, "aabb": [[-1, -1, -1], [1, 1, 1]]}, "height": H, "width": W,
            "frames": frames}
    return meta, edges


def scene_rays(meta, edges, img_idx: int, px, py, dtype=torch.float32):
    """Rays of given pixels of one view, as ``Dataset.gen_random_rays_patches_at`` builds them (reference src/dataset/dataset.py:244-287:
    p = K^-1 [x, y, 1], rays_v = R (p / |p|), rays_o = camera centre, depth_scale = the z component of p / |p|), in fp32 torch ops on
    the CPU.  tests/golden/make_goldens.py:g15 checks it against the reference method itself before it records the training run that
    uses it.  -> rays_o (n,3), rays_v (n,3), depth_scale (n,1), edge (n,1)."""
    K = torch.tensor(meta["frames"][img_idx]["intrinsics"], dtype=dtype)
    P = torch.tensor(meta["frames"][img_idx]["camtoworld"], dtype=dtype)[:4, :4]
    px = torch.as_tensor(np.asarray(px)).long()
    py = torch.as_tensor(np.asarray(py)).long()
    e = torch.as_tensor(np.asarray(edges), dtype=dtype)
    if e.dim() == 4:
        e = e[..., 0]
    edge = e[img_idx][(py, px)].reshape(-1, 1)
    p = torch.stack([px, py, torch.ones_like(py)], dim=-1).to(dtype)
    p = torch.matmul(torch.inverse(K)[None, :3, :3], p[:, :, None]).squeeze(-1)
    p_norm = torch.linalg.norm(p, ord=2, dim=-1, keepdim=True)
    pn = p / p_norm
    rays_v = torch.matmul(P[None, :3, :3], pn[:, :, None]).squeeze(-1)
    rays_o = P[None, :3, 3].expand(rays_v.shape)
    return rays_o.contiguous(), rays_v.contiguous(), pn[:, 2:3].contiguous(), edge


def convergence_batch(meta, edges, n_rays: int, seed: int, held_out: int = 0):
    """One training batch of the recorded convergence run (golden g15): a view != held_out, half of the pixels uniform, half uniform over the
    view's edge pixels (edge > 0.1) - a seeded (NumPy PCG64) stand-in for the host draws of dataset.py:228-243.  -> (img_idx, px, py)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n_img = len(meta["frames"])
    img = int(rng.choice([i for i in range(n_img) if i != held_out]))
    H, W = int(meta["height"]), int(meta["width"])
    e = np.asarray(edges)[img].reshape(H, W)
    half = n_rays // 2
    px, py = rng.integers(0, W, size=half), rng.integers(0, H, size=half)
    ys, xs = np.nonzero(e > 0.1)
    k = rng.integers(0, len(ys), size=n_rays - half)
    return img, np.concatenate([px, xs[k]]), np.concatenate([py, ys[k]])
