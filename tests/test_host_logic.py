"""Host-side helpers that need no GPU: the synthetic multi-view scene used by the convergence run."""


def test_wireframe_scene_is_multi_view_consistent():
    """The edge maps of synthetic.make_wireframe_scene are projections of ONE 3D wire frame: a 3D point on a segment projects onto an
    edge pixel in every view (what makes the convergence run of scripts/train_synthetic.py meaningful)."""
    import numpy as np
    from emap_amd import synthetic
    meta, edges = synthetic.make_wireframe_scene(n_images=5, H=80, W=80)
    segs = synthetic.wireframe_segments()
    assert segs.shape == (13, 2, 3) and edges.shape == (5, 80, 80, 1) and 0.0 <= edges.min() and edges.max() <= 1.0
    pts = np.concatenate([a[None] * (1 - t) + b[None] * t for a, b in segs for t in (0.25, 0.5, 0.75)])
    for i, fr in enumerate(meta["frames"]):
        K, c2w = np.array(fr["intrinsics"]), np.array(fr["camtoworld"])
        pc = (pts - c2w[:3, 3]) @ c2w[:3, :3]
        u, v = K[0, 0] * pc[:, 0] / pc[:, 2] + K[0, 2], K[1, 1] * pc[:, 1] / pc[:, 2] + K[1, 2]
        inside = (u >= 1) & (u < 78) & (v >= 1) & (v < 78)
        assert inside.sum() >= 30
        vals = edges[i, np.round(v[inside]).astype(int), np.round(u[inside]).astype(int), 0]
        assert (vals > 0.5).all()
