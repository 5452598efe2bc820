"""Dense-grid extraction queries (SURVEY par. 8 f2): the second consumer of the UDF field kernels.

Mirror of ``get_udf_normals_grid`` (src/edge_extraction/extract_pointcloud.py:5-95): same arguments, same return tuple,
same jitter draws (one ``torch.randn((n, sampling_N, 3), device=device)`` per ``max_batch`` chunk, in the same order), but
  * the grid, the thresholded subset and all jittered neighbourhoods are evaluated in a few launches of up to 2^20 points
    instead of the reference's N^3/4096 + 51 x n/4096 calls of 4096 points - for ANY ``func`` / ``func_grad``, in particular the
    closure ``Runner_UDF.extract_edge`` passes (runner_udf.py:520-527: ``udf_network.gradient`` followed by a normalisation).
    Both callables act point by point, so the chunk size cannot change a result; it only decides whether the field kernels
    see 4096 points (a latency-bound launch) or a million (large gradient launches run the reverse-sweep kernel).  When the
    callables are this package's bound ``UDFNetwork.udf`` / ``.gradient`` the wrappers are skipped as well,
  * the line direction ``F.normalize(torch.linalg.svd(grad_ld)[2][:, -1, :])`` (:86-88) is one HIP kernel over the
    3x3 matrices G^T G (``emap_null_direction``): no 50x3 SVD batch.
There is no CPU fallback: tensors must live on the GPU and libemap_hip.so must be loadable.
"""
import ctypes as C

import torch

from . import _lib

_BIG = 1 << 20          # points per launch of the fast path (bounds the temporaries: 12 MB in, 16 MB out)


def null_direction(grads: torch.Tensor) -> torch.Tensor:
    """(n, k, 3) gradient samples -> (n, 3) unit direction of least variation (sign arbitrary), extract_pointcloud.py:86-88."""
    _lib.require_cuda(grads, "grads")
    n, k = int(grads.shape[0]), int(grads.shape[1])
    g = _lib.f32c(grads)
    out = torch.empty(n, 3, device=g.device, dtype=torch.float32)
    if n:
        with _lib.on_device(g):
            _lib.check(_lib.lib().emap_null_direction(_lib.ptr(g), C.c_int64(n), k, _lib.ptr(out), _lib.stream_ptr(g.device)),
                       "null_direction")
    return out


def _fast_net(func, func_grad):
    from .udf_model import UDFNetwork
    net = getattr(func, "__self__", None)
    if isinstance(net, UDFNetwork) and getattr(func_grad, "__self__", None) is net \
            and func.__name__ == "udf" and func_grad.__name__ == "gradient":
        return net
    return None


def _uses_package_net(f) -> bool:
    """True for a callable that evaluates this package's UDFNetwork point by point: the network's own bound methods, or a closure
    over the network / over an object that owns one (the runner's normalising closure, runner_udf.py:520-527).  Such callables are
    driven with 2^20-point launches; any OTHER callable keeps the caller's ``max_batch`` chunks (it may not act point by point, and
    ``max_batch`` may be its memory bound)."""
    from .udf_model import UDFNetwork
    owns = lambda o: isinstance(o, UDFNetwork) or isinstance(getattr(o, "udf_network", None), UDFNetwork)
    if owns(getattr(f, "__self__", None)):
        return True
    cells = [c.cell_contents for c in (getattr(f, "__closure__", None) or []) if c is not None]
    return any(owns(o) for o in cells) or any(owns(o) for o in (getattr(f, "__defaults__", None) or ()))


def _chunk(func, func_grad, max_batch, mult=1):
    return _BIG if (_uses_package_net(func) and _uses_package_net(func_grad)) else max(int(max_batch) * mult, 1)


def _eval(fn, pts, chunk):
    return torch.cat([fn(pts[h:h + chunk]) for h in range(0, pts.shape[0], chunk)]) if pts.shape[0] else fn(pts)


def get_udf_normals_grid(func, func_grad, N, udf_threshold, is_linedirection=False, sampling_N=50, sampling_delta=0.005,
                         max_batch=int(2 ** 12), device="cuda", noise=None):
    """See the module docstring.  ``noise`` (n_below_threshold, sampling_N, 3) replaces the jitter draws (parity tests)."""
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("emap_amd.extraction runs on the GPU only (no CPU fallback)")
    net = _fast_net(func, func_grad)
    # the N^3 lattice on [-1, 1]^3, first coordinate slowest (the reference's point order and fp32 arithmetic, :36-54: index * 2/(N-1) - 1)
    voxel_size = 2.0 / (N - 1)
    axis = torch.arange(N, device=device, dtype=torch.float32) * voxel_size + (-1)
    samples = torch.zeros(N ** 3, 12, device=device)
    samples[:, :3] = torch.stack(torch.meshgrid(axis, axis, axis, indexing="ij"), dim=-1).reshape(-1, 3)

    with torch.no_grad():
        pts = samples[:, :3].contiguous()
        if net is not None:
            df = _eval(lambda p: net.hip_udf(p, with_grad=False)[0], pts, _BIG)
        else:
            df = _eval(lambda p: func(p)[0].detach(), pts, _chunk(func, func_grad, max_batch))
        samples[:, 3:4] = df

        norm_idx = torch.where(samples[:, 3] < udf_threshold)[0]            # :64-65
        sub = samples[norm_idx, :3].contiguous()
        if net is not None:
            grad = _eval(lambda p: net.hip_udf(p, with_grad=True)[1], sub, _BIG).reshape(-1, 1, 3) if len(norm_idx) else sub.reshape(-1, 1, 3)
        else:
            grad = _eval(lambda p: func_grad(p).detach(), sub, _chunk(func, func_grad, max_batch)) if len(norm_idx) else sub.reshape(-1, 1, 3)
        # the reference normalises the (P,1,3) gradient along dim=1 - the singleton - i.e. per component (:71)
        samples[norm_idx, 4:7] = -torch.nn.functional.normalize(grad, dim=1)[:, 0]

        if is_linedirection and len(norm_idx):
            if noise is None:
                # same draw sequence as the reference: one randn per max_batch chunk of the thresholded points (:75-79)
                noise = torch.cat([torch.randn((min(max_batch, len(norm_idx) - h), sampling_N, 3), device=device)
                                   for h in range(0, len(norm_idx), max_batch)])
            ld_pts = (sub.unsqueeze(1) + sampling_delta * noise.to(device)).reshape(-1, 3)
            if net is not None:
                grad_ld = _eval(lambda p: net.hip_udf(p, with_grad=True)[1], ld_pts, _BIG)
            else:
                grad_ld = _eval(lambda p: func_grad(p).detach().reshape(-1, 3), ld_pts, _chunk(func, func_grad, max_batch, sampling_N))
            samples[norm_idx, 8:11] = null_direction(grad_ld.reshape(len(norm_idx), sampling_N, 3))

    df_values = samples[:, 3].reshape(N, N, N)
    vecs = samples[:, 4:7].reshape(N, N, N, 3)
    ld = samples[:, 8:11].reshape(N, N, N, 3)
    return df_values, ld, vecs, samples, torch.tensor(voxel_size)


def get_udf_normals_slow(func, func_grad, voxel_size, xyz, is_linedirection, sampling_N=50, sampling_delta=0.005,
                         max_batch=int(2 ** 12), device="cuda", noise=None):
    """Mirror of ``get_udf_normals_slow`` (extract_pointcloud.py:98-193): values, normals (-grad/|grad|) and optional line
    directions at arbitrary points ``xyz`` (n,3).  Same return tuple ``(df_values, normals, ld, samples)`` with the
    reference's 13-column ``samples``; ``voxel_size`` is unused there as well."""
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("emap_amd.extraction runs on the GPU only (no CPU fallback)")
    net = _fast_net(func, func_grad)
    xyz = xyz.to(device).float()
    n = xyz.shape[0]
    samples = torch.cat([xyz, torch.zeros(n, 10, device=device)], dim=-1)
    with torch.no_grad():
        pts = samples[:, 0:3].contiguous()
        if net is not None:
            res = [net.hip_udf(pts[h:h + _BIG], with_grad=True) for h in range(0, n, _BIG)] if n else []
            df = torch.cat([r[0] for r in res]) if n else pts[:, :1]
            grad = torch.cat([r[1] for r in res]) if n else pts
        else:
            df = _eval(lambda p: func(p)[0].detach(), pts, _chunk(func, func_grad, max_batch))
            grad = _eval(lambda p: func_grad(p).detach()[:, 0], pts, _chunk(func, func_grad, max_batch))
        samples[:, 3] = df.squeeze(-1)
        samples[:, 4:7] = -torch.nn.functional.normalize(grad, dim=1)                     # :158-160
        if is_linedirection and n:
            if noise is None:
                noise = torch.cat([torch.randn((min(max_batch, n - h), sampling_N, 3), device=device)
                                   for h in range(0, n, max_batch)])                           # :163-167, per chunk
            ld_pts = (pts.unsqueeze(1) + sampling_delta * noise.to(device)).reshape(-1, 3)
            if net is not None:
                grad_ld = _eval(lambda p: net.hip_udf(p, with_grad=True)[1], ld_pts, _BIG)
            else:
                grad_ld = _eval(lambda p: func_grad(p.float()).detach()[:, 0], ld_pts, _chunk(func, func_grad, max_batch, sampling_N))
            samples[:, 7:10] = null_direction(grad_ld.reshape(n, sampling_N, 3))
    return samples[:, 3], samples[:, 4:7], samples[:, 7:10], samples
