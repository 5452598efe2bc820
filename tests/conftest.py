import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # gpu-marked tests are skipped (not failed) when no device is visible and -m gpu was not forced
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def load_golden(name):
    d = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    return {k: d[k] for k in d.files}


def t(a, dtype=torch.float32):
    a = np.asarray(a)
    if a.dtype.kind in "iu":
        return torch.from_numpy(a.astype(np.int64))
    return torch.from_numpy(a.astype(np.float32)).to(dtype)


NETS = {
    # must match tests/golden/make_goldens.py:NETS
    "d8w256L10": (dict(d_in=3, d_out=1, d_hidden=256, n_layers=8, skip_in=(4,), multires=10, bias=0.5), 42, 0.02),
    "d8w256L6": (dict(d_in=3, d_out=1, d_hidden=256, n_layers=8, skip_in=(4,), multires=6, bias=0.5), 43, 0.02),
    "d4w128L10": (dict(d_in=3, d_out=1, d_hidden=128, n_layers=4, skip_in=(4,), multires=10, bias=0.5), 44, 0.02),
    "d8w256L10_init": (dict(d_in=3, d_out=1, d_hidden=256, n_layers=8, skip_in=(4,), multires=10, bias=0.5), 45, 0.0),
}


def net_state(name):
    from emap_amd import synthetic
    kw, seed, pert = NETS[name]
    return kw, synthetic.make_udf_state(seed=seed, pert=pert, **kw)
