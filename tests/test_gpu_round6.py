"""GPU tests added in round 6 (-m gpu), all through the C ABI / the drop-in classes."""
import copy

import numpy as np
import pytest
import torch

import emap_amd
from emap_amd import _lib, synthetic
from emap_amd.parallel import Trainer, FusedAdam

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


# ------------------------------------------------------------------------------------------------ FusedAdam after a resume
def test_fused_adam_follows_lr_changes_after_load_state_dict():
    """ADVICE r5 (high).  The resume path of the drop-in (runner_udf.py:260-273: load_checkpoint -> optimizer.load_state_dict, then
    update_learning_rate writes param_groups[i]['lr'] every step): load a checkpoint, CHANGE the learning rates, step - equal to
    torch.optim.Adam doing the same."""
    from test_gpu_round5 import _groups, _clone, _mk, _steps
    gen = torch.Generator().manual_seed(31)
    geo_a, tail_a = _groups(gen)
    geo_b, tail_b = _clone(geo_a), _clone(tail_a)
    oa, ob = _mk(FusedAdam, geo_a, tail_a), _mk(torch.optim.Adam, geo_b, tail_b)
    _steps([(oa, geo_a + tail_a), (ob, geo_b + tail_b)], torch.Generator().manual_seed(32), 3)
    sd = copy.deepcopy(oa.state_dict())
    geo_c, tail_c = _clone(geo_a), _clone(tail_a)
    oc = _mk(FusedAdam, geo_c, tail_c)
    oc.load_state_dict(sd)
    for s in range(3):
        for opt in (oc, ob):                       # the runner's scheduler: geo group its own lr, every other group the common one
            for i, g in enumerate(opt.param_groups):
                g["lr"] = (2e-3 if i == 0 else 7e-3) * (0.5 ** s)
        _steps([(oc, geo_c + tail_c), (ob, geo_b + tail_b)], torch.Generator().manual_seed(40 + s), 1, s0=3 + s)
    for pc, pb in zip(geo_c + tail_c, geo_b + tail_b):
        assert torch.allclose(pc, pb, rtol=3e-6, atol=1e-7), (pc, pb)
    # and from_adam on an Adam that has already stepped (it goes through load_state_dict)
    geo_d, tail_d = _clone(geo_b), _clone(tail_b)
    od_ref = _mk(torch.optim.Adam, geo_d, tail_d)
    od_ref.load_state_dict(copy.deepcopy(ob.state_dict()))
    od = FusedAdam.from_adam(od_ref)
    for opt in (od, ob):
        for i, g in enumerate(opt.param_groups):
            g["lr"] = 3e-3 if i == 0 else 4e-3
    _steps([(od, geo_d + tail_d), (ob, geo_b + tail_b)], torch.Generator().manual_seed(50), 2, s0=6)
    for pd_, pb in zip(geo_d + tail_d, geo_b + tail_b):
        assert torch.allclose(pd_, pb, rtol=3e-6, atol=1e-7)
