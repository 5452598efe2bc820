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


def test_fused_adam_reads_the_live_param_groups_after_load_state_dict():
    """``Optimizer.load_state_dict`` replaces the group dicts in ``param_groups``; the runner's schedulers
    (runner_base.py:128-150) write ``lr`` into the NEW dicts.  FusedAdam must read those, not the ones it saw in __init__."""
    import copy
    import torch
    from emap_amd.parallel import FusedAdam
    geo = [torch.nn.Parameter(torch.randn(4, 3)), torch.nn.Parameter(torch.randn(4))]
    tail = [torch.nn.Parameter(torch.randn(1))]
    opt = FusedAdam([{"params": geo, "lr": 1e-4}, {"params": []}, {"params": tail}], lr=5e-4)
    assert opt._geo_group is opt.param_groups[0] and opt._tail_groups[0] is opt.param_groups[2]
    ref = torch.optim.Adam([{"params": [p.detach().clone().requires_grad_() for p in geo], "lr": 1e-4}, {"params": []},
                            {"params": [p.detach().clone().requires_grad_() for p in tail]}], lr=5e-4)
    for p in [q for g in ref.param_groups for q in g["params"]]:
        p.grad = torch.ones_like(p)
    ref.step()
    opt.load_state_dict(copy.deepcopy(ref.state_dict()))
    assert opt._geo_group is opt.param_groups[0] and opt._tail_groups[0] is opt.param_groups[2]
    opt.param_groups[0]["lr"] = 123.0
    opt.param_groups[2]["lr"] = 7.0
    assert opt._geo_group["lr"] == 123.0 and opt._tail_groups[0]["lr"] == 7.0
    # pickling round trip (__setstate__ replaces the dicts as well)
    import pickle
    o2 = pickle.loads(pickle.dumps(opt))
    o2.param_groups[0]["lr"] = 9.0
    assert o2._geo_group["lr"] == 9.0


def test_gvb_cache_notices_a_replaced_middle_layer_parameter():
    """UDFNetwork._gvb() caches the (g, v, b) walk; replacing ANY layer's parameter (not only the first / last one)
    must invalidate it - packed() would otherwise fold stale tensors."""
    import torch
    import emap_amd
    net = emap_amd.UDFNetwork(3, 1, 128, 4, skip_in=(4,), multires=6, bias=0.5, scale=1.0, geometric_init=True, weight_norm=True)
    gs, vs, bs = net._gvb()
    assert net._gvb()[1][2] is vs[2]                                   # served from the cache
    net.lin2.bias = torch.nn.Parameter(torch.zeros_like(net.lin2.bias))
    assert net._gvb()[2][2] is net.lin2.bias and net._gvb()[2][2] is not bs[2]
    new_v = torch.nn.Parameter(net.lin1.parametrizations.weight.original1.detach().clone())
    net.lin1.parametrizations.weight.original1 = new_v
    assert net._gvb()[1][1] is new_v
    old = net.lin3
    net.lin3 = torch.nn.utils.parametrizations.weight_norm(torch.nn.Linear(old.in_features, old.out_features))
    assert net._gvb()[1][3] is net.lin3.parametrizations.weight.original1
