"""Full-image rendering (SURVEY par. 8 f4): the render loop of ``Runner_UDF.validate`` (src/runner/runner_udf.py:297-407).

The reference splits the H*W rays of an image into ``batch_size`` chunks, calls ``renderer.render`` per chunk and copies
``edge``, ``depth`` and the weighted normal ``sum_s gradients_flip * weights`` of every chunk to the host (three
``.detach().cpu().numpy()`` round trips per chunk).  Here the same quantities are produced by ONE ``emap_render_fwd`` call over all
H*W rays (round 6; ``launch_rays=None``: up to 2**18 rays per call - every kernel of the path strides over its ray / point tiles
with a resident grid, so a 400 x 400 image is one chain of 10 launches instead of 20 chains of 5; rounds 1-5: 8192 rays per call)
in the renderer's REDUCED output mode - the compositing kernel is handed NULL
for every per-sample output and writes only edge, depth, the weighted normal and weight_sum: 28 B per ray instead of the
48 B per sample of the training dict - stay on the device and are copied to the host once.  The per-chunk jitter
draws of the reference (``torch.rand([chunk, 1])`` on the CPU generator, udf_renderer_blending.py:719) are reproduced in the
same order, so with the same seed the image is the reference's image.  Rays are independent, so the result does not depend
on how they are grouped into launches (tests/test_gpu_parity.py::test_image_render_is_chunk_invariant).
"""
import numpy as np
import torch


def render_image(renderer, rays_o, rays_d, near, far, depth_scale, batch_size, cos_anneal_ratio=None, background_rgb=None,
                 launch_rays=None, to_numpy=True):
    """rays_o, rays_d (H,W,3) or (n,3); depth_scale (H,W,1) or (n,1).  Returns {"edge": (n,1), "depth": (n,1),
    "normals": (n,3)} as numpy arrays (the lists ``out_edge_fine / out_depth / out_normal_fine`` of the reference, concatenated)."""
    ro = rays_o.reshape(-1, 3)
    rd = rays_d.reshape(-1, 3)
    ds = depth_scale.reshape(-1, 1)
    n = ro.shape[0]
    if launch_rays is None:
        launch_rays = 1 << 18      # one call for any image up to 512 x 512; the workspace of a call is ~2 KiB per ray
    # jitter: one draw per reference chunk, in the reference's order (render() :718-720 with perturb_overwrite = -1)
    t_rand = None
    if renderer.perturb > 0:
        t_rand = torch.cat([torch.rand([min(batch_size, n - h), 1]) - 0.5 for h in range(0, n, batch_size)]) if n else torch.zeros(0, 1)
    edge, depth, normals = [], [], []
    with torch.no_grad():
        for h in range(0, n, launch_rays):
            t = slice(h, min(h + launch_rays, n))
            nr = near[t] if isinstance(near, torch.Tensor) and near.numel() > 1 else near
            fr = far[t] if isinstance(far, torch.Tensor) and far.numel() > 1 else far
            out = renderer.render_reduced(ro[t], rd[t], nr, fr, depth_scale=ds[t], cos_anneal_ratio=cos_anneal_ratio,
                                          background_rgb=background_rgb, perturb_overwrite=-1 if t_rand is not None else 0,
                                          t_rand=None if t_rand is None else t_rand[t])
            edge.append(out["edge"])
            depth.append(out["depth"])
            normals.append(out["normals"])        # = (gradients_flip * weights[:, :S, None]).sum(1), render_core :662
    cat = lambda xs, w: torch.cat(xs) if xs else torch.zeros(0, w, device=ro.device)
    res = {"edge": cat(edge, 1), "depth": cat(depth, 1), "normals": cat(normals, 3)}
    if to_numpy:
        res = {k: v.detach().cpu().numpy() for k, v in res.items()}
    return res


def to_images(res, H, W):
    """The reference's post-processing of the concatenated lists (runner_udf.py:409-440): edge*255 clipped to uint8 (H,W),
    depth (H,W), normals (H,W,3)."""
    edge = (np.asarray(res["edge"]).reshape(H, W) * 255).clip(0, 255).astype(np.uint8)
    return edge, np.asarray(res["depth"]).reshape(H, W), np.asarray(res["normals"]).reshape(H, W, 3)
