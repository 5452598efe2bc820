"""Multi-GPU checks of the product path over RCCL ("nccl" backend).  They need >= 2 visible GPUs and skip cleanly otherwise
(the build's gpurun boxes have one GPU; the driver's 8-GPU node runs them).  The same arithmetic is covered on the CPU with
gloo in tests/test_dist_cpu.py."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _need(n):
    if not torch.cuda.is_available() or torch.cuda.device_count() < n:
        pytest.skip(f"needs {n} GPUs, {torch.cuda.device_count() if torch.cuda.is_available() else 0} visible")


def _rank_step(rank, world, port, n_global, sync, q, backend="nccl"):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import emap_amd
    from emap_amd import synthetic
    from emap_amd.parallel import Trainer, shard
    from conftest import net_state
    # backend "gloo": every rank on GPU 0 (gloo reduces device tensors through the host) - the whole product path, HIP forward and
    # backward included, data-parallel on a one-GPU box; only the transport differs from the RCCL run
    di = rank if backend == "nccl" else 0
    torch.cuda.set_device(di)
    dev = torch.device("cuda", di)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    kw, state = net_state("d8w256L10")
    net = emap_amd.UDFNetwork(precision="f16x3", **kw)
    net.load_state_dict(state)
    net = net.to(dev)
    devn = emap_amd.SingleVarianceNetwork(0.3).to(dev)
    bet = emap_amd.BetaNetwork(0.5, 0.3, 0.3, 5e-5, True, True, False).to(dev)
    r = emap_amd.UDFRendererBlending(None, net, devn, bet, 64, 64, 0, 4, 1.0, device=dev)
    rays = [shard(t_, rank, world).to(dev) for t_ in synthetic.make_rays(n_global, seed=77)]
    te = shard(synthetic.make_true_edge(n_global, seed=78), rank, world).to(dev)
    tr = shard(synthetic.make_t_rand(n_global, seed=79), rank, world).to(dev)
    t = Trainer(r, lr_geo=1e-3, lr=5e-3, igr_weight=0.1, igr_ns_weight=0.05, eikonal_sync=sync)
    batch = dict(zip(("rays_o", "rays_d", "near", "far", "depth_scale"), rays))
    batch.update(cos_anneal_ratio=1.0, flip_saturation=0.9, t_rand=tr)
    stats = None
    p0 = t.flat.data.detach().cpu().numpy().copy()
    for _ in range(2):
        stats = t.step(batch, te, n_rays_global=n_global)
    torch.cuda.synchronize()
    r.check_errors()
    q.put((rank, stats.cpu().numpy(), t.flat.data.cpu().numpy(), t.flat.grad[:t.flat.numel].cpu().numpy(), p0))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _launch(world, n_global, sync, backend="nccl"):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_step, args=(r, world, port, n_global, sync, q, backend)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return sorted(res, key=lambda x: x[0])


def _same_steps(one, two):
    """2 ranks vs 1 process after two optimizer steps.  Not bit-equal by construction: each rank's backward sweep scales its fp16
    adjoints by a power of two taken from ITS shard's maxima and the weight-gradient GEMMs run on 11-bit hi parts, so the two
    partial gradients round differently from the single launch - inside the 1e-3 parity bound of the training gradients; and Adam
    moves an entry whose gradient is rounding noise by ~lr either way, so parameters are compared by their displacement vector."""
    assert np.array_equal(two[0][2], two[1][2])                            # replicas stay identical
    assert np.allclose(two[0][1], one[1], rtol=1e-4)                       # global loss statistics
    g1, g2 = one[3], two[0][3]
    assert np.abs(g1 - g2).max() <= 1e-3 * np.abs(g1).max()
    d1, d2 = one[2] - one[4], two[0][2] - two[0][4]
    cos = float((d1 * d2).sum() / (np.linalg.norm(d1) * np.linalg.norm(d2)))
    assert cos > 0.99 and abs(np.linalg.norm(d2) / np.linalg.norm(d1) - 1.0) < 0.02, cos


@pytest.mark.timeout(900)
def test_two_rank_rccl_training_steps_equal_single_gpu_steps():
    """Two optimizer steps of emap_amd.parallel.Trainer on 2 ranks (RCCL all-reduce of the flat gradient buffer, global eikonal
    denominators) == the same steps on one GPU with the whole batch."""
    _need(2)
    one = _launch(1, 128, "exact")[0]
    two = _launch(2, 128, "exact")
    _same_steps(one, two)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("sync", ["exact", "local"])
def test_two_ranks_on_one_gpu_equal_the_single_process_steps(sync):
    """The same equality on a ONE-GPU box: two ranks share GPU 0 and exchange through gloo (RCCL refuses two ranks on one device).
    Everything but the transport is the product path: HIP forward, HIP backward into the flat gradient buffer, the 5-float
    statistics exchange, one gradient all-reduce, fused Adam.  "local": rank-local eikonal denominators, ONE collective per step -
    differs from the exact step only by the mean-of-means bias."""
    _need(1)
    one = _launch(1, 128, "exact", "gloo")[0]
    two = _launch(2, 128, sync, "gloo")
    if sync == "exact":
        _same_steps(one, two)
    else:
        assert np.array_equal(two[0][2], two[1][2])
        g1, g2 = one[3], two[0][3]
        cos = float((g1 * g2).sum() / (np.linalg.norm(g1) * np.linalg.norm(g2)))
        assert cos > 0.999, cos


@pytest.mark.timeout(900)
@pytest.mark.parametrize("mode", ["render", "train"])
def test_bench_spawns_its_ranks_and_reports_them(mode):
    _need(2)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--mode", mode, "--steps", "5", "--warmup", "2",
                          "--no-cpu-baseline", "--no-other-modes", "--no-parity"], capture_output=True, text=True, timeout=800)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["ranks"] == 2 and line["value"] > 0
    if mode == "train":
        assert "2 collective(s) per step" in line["config"]["parallelism"]


def test_bench_refuses_more_gpus_than_visible():
    n = torch.cuda.device_count() + 1
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1"], capture_output=True, text=True,
                         timeout=300)
    assert out.returncode != 0 and "GPU(s) visible" in (out.stderr + out.stdout)
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True,
                         timeout=300, env=env)
    assert out.returncode != 0 and "refusing" in (out.stderr + out.stdout)   # launched with fewer ranks than it would report
