"""GPU tests added in round 5 (-m gpu), all through the C ABI / the drop-in classes:

  * emap_amd.parallel.FusedAdam checkpoints in torch.optim.Adam's per-parameter layout (runner_udf.py:260,273 save / load
    `optimizer.state_dict()`): save -> load -> step equals an uninterrupted torch.optim.Adam run, in both directions.
"""
import copy

import numpy as np
import pytest
import torch

import emap_amd
from emap_amd import _lib, synthetic
from emap_amd.parallel import Trainer, FusedAdam

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _groups(gen, frozen_idx=2):
    shapes_geo = [(16, 7), (16,), (16, 1), (8, 16), (8,)]
    geo = [torch.nn.Parameter(torch.randn(*s, generator=gen).to(DEV)) for s in shapes_geo]
    tail = [torch.nn.Parameter(torch.randn(1, generator=gen).to(DEV)) for _ in range(4)]
    tail[frozen_idx].requires_grad_(False)
    return geo, tail


def _clone(ps):
    out = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    for a, b in zip(out, ps):
        a.requires_grad_(b.requires_grad)
    return out


def _mk(cls, geo, tail):
    return cls([{"params": geo, "lr": 1e-3}, {"params": tail[:2]}, {"params": tail[2:]}, {"params": []}], lr=5e-3)


def _steps(opts_params, gen, n, unfreeze_at=None, s0=0):
    for s in range(s0, s0 + n):
        gs = None
        for opt, ps in opts_params:
            if unfreeze_at is not None and s == unfreeze_at:
                ps[-2].requires_grad_(True)
            opt.zero_grad()
        shapes = [p.shape for p in opts_params[0][1]]
        gs = [torch.randn(sh, generator=gen).to(DEV) for sh in shapes]
        for opt, ps in opts_params:
            for p, g_ in zip(ps, gs):
                if p.requires_grad:
                    p.grad = g_.clone()
            opt.step()


def test_fused_adam_checkpoint_round_trip_against_torch_adam():
    """FusedAdam kept its moments in private tensors and saved an empty `state`.  Now: 3 steps, state_dict(),
    a FRESH optimizer over fresh parameter copies loads it, 3 more steps (a frozen scalar is un-frozen after the reload) -
    equal to torch.optim.Adam running the 6 steps without interruption; and the checkpoint has torch.optim.Adam's keys."""
    gen = torch.Generator().manual_seed(11)
    geo_a, tail_a = _groups(gen)
    geo_b, tail_b = _clone(geo_a), _clone(tail_a)
    oa, ob = _mk(FusedAdam, geo_a, tail_a), _mk(torch.optim.Adam, geo_b, tail_b)
    g1 = torch.Generator().manual_seed(12)
    _steps([(oa, geo_a + tail_a), (ob, geo_b + tail_b)], g1, 3)
    sd = copy.deepcopy(oa.state_dict())
    ref_sd = ob.state_dict()
    assert set(sd["state"].keys()) == set(ref_sd["state"].keys())          # the frozen scalar has no entry in either
    for k, st in sd["state"].items():
        assert set(st.keys()) >= {"step", "exp_avg", "exp_avg_sq"}
        assert float(st["step"]) == float(ref_sd["state"][k]["step"]) == 3.0
        assert torch.allclose(st["exp_avg"], ref_sd["state"][k]["exp_avg"], rtol=3e-6, atol=1e-8)
        assert torch.allclose(st["exp_avg_sq"], ref_sd["state"][k]["exp_avg_sq"], rtol=3e-6, atol=1e-10)
    # resume in a fresh process-like state: new parameter objects, new optimizer
    geo_c, tail_c = _clone(geo_a), _clone(tail_a)
    oc = _mk(FusedAdam, geo_c, tail_c)
    oc.load_state_dict(sd)
    g2 = torch.Generator().manual_seed(13)
    _steps([(oc, geo_c + tail_c), (ob, geo_b + tail_b)], g2, 3, unfreeze_at=4, s0=3)
    for pc, pb in zip(geo_c + tail_c, geo_b + tail_b):
        assert torch.allclose(pc, pb, rtol=3e-6, atol=1e-7), (pc, pb)
    # the late-unfrozen scalar took Adam's FIRST steps after the reload (its own step count, not the global one)
    st = oc.state_dict()["state"]
    idx = len(geo_c) + 2
    assert float(st[idx]["step"]) == 2.0 and float(st[0]["step"]) == 6.0


def test_fused_adam_loads_a_torch_adam_checkpoint_and_vice_versa():
    gen = torch.Generator().manual_seed(21)
    geo_a, tail_a = _groups(gen)
    geo_b, tail_b = _clone(geo_a), _clone(tail_a)
    ob = _mk(torch.optim.Adam, geo_b, tail_b)
    g1 = torch.Generator().manual_seed(22)
    _steps([(ob, geo_b + tail_b)], g1, 4)
    # torch.optim.Adam checkpoint -> FusedAdam
    geo_c, tail_c = _clone(geo_b), _clone(tail_b)
    oc = _mk(FusedAdam, geo_c, tail_c)
    oc.load_state_dict(copy.deepcopy(ob.state_dict()))
    g2 = torch.Generator().manual_seed(23)
    _steps([(oc, geo_c + tail_c), (ob, geo_b + tail_b)], g2, 2, s0=4)
    for pc, pb in zip(geo_c + tail_c, geo_b + tail_b):
        assert torch.allclose(pc, pb, rtol=3e-6, atol=1e-7)
    # FusedAdam checkpoint -> torch.optim.Adam
    geo_d, tail_d = _clone(geo_c), _clone(tail_c)
    od = _mk(torch.optim.Adam, geo_d, tail_d)
    od.load_state_dict(copy.deepcopy(oc.state_dict()))
    g3 = torch.Generator().manual_seed(24)
    _steps([(od, geo_d + tail_d), (ob, geo_b + tail_b)], g3, 2, s0=6)
    for pd_, pb in zip(geo_d + tail_d, geo_b + tail_b):
        assert torch.allclose(pd_, pb, rtol=3e-6, atol=1e-7)
