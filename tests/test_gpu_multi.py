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


def _rank_step(rank, world, port, n_global, sync, q, backend="nccl", capture=False, allreduce="rccl", lag_factor=1.0):
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
    t = Trainer(r, lr_geo=1e-3, lr=5e-3, igr_weight=0.1, igr_ns_weight=0.05, eikonal_sync=sync, allreduce=allreduce)
    batch = dict(zip(("rays_o", "rays_d", "near", "far", "depth_scale"), rays))
    batch.update(cos_anneal_ratio=1.0, flip_saturation=0.9, t_rand=tr)
    p0 = t.flat.data.detach().cpu().numpy().copy()
    step = t.step

    def reset():
        t.flat.data.copy_(torch.from_numpy(p0).to(dev))
        t._m.zero_(); t._v.zero_(); t._adam_t.zero_(); t._tail_step.zero_()
        net.invalidate_packed()

    if capture:      # one hipGraph per device phase, the collectives between the replays (Trainer.capture(segmented=True))
        rp = t.capture(batch, te, n_rays_global=n_global, warmup=1, segmented=True)
        assert rp.segmented and len(rp.graphs) == 4
        reset()                                              # undo the warm-up / capture steps: same start as the eager run
        step = lambda b, e, n_rays_global=None: rp(b, e)
    if sync == "exact_lagged" and world > 1:
        # prime the history: one step from p0 leaves the GLOBAL range maxima of p0 in the trainer (they do not depend on the scale that
        # step itself used), then start over - so the two recorded steps BOTH run with lagged maxima (x 4, no MAX all-reduce)
        step(batch, te, n_rays_global=n_global)
        assert t._lag_valid and len(t._collectives()) == 2
        reset()
        t._lag.mul_(lag_factor)      # != 1: as if the gradients had jumped / collapsed by that factor since the step the history is from
    stats1 = step(batch, te, n_rays_global=n_global).cpu().numpy().copy()
    grad1 = t.flat.grad[:t.flat.numel].cpu().numpy().copy()
    stats = step(batch, te, n_rays_global=n_global)
    torch.cuda.synchronize()
    r.check_errors()
    t.check_errors()                 # renderer error word + the one-shot collective's time-out word
    q.put((rank, stats.cpu().numpy(), t.flat.data.cpu().numpy(), t.flat.grad[:t.flat.numel].cpu().numpy(), p0, stats1, grad1))
    if world > 1:
        t.close()
        dist.barrier()
        dist.destroy_process_group()


def _launch(world, n_global, sync, backend="nccl", capture=False, allreduce="rccl", lag_factor=1.0):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_step, args=(r, world, port, n_global, sync, q, backend, capture, allreduce, lag_factor)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return sorted(res, key=lambda x: x[0])


def _same_steps(one, two, tol=1e-4):
    """2 ranks vs 1 process.  The FIRST step is compared directly: its loss statistics and its gradient (every rank's MLP backward
    uses the same fp16 range scale - the two maxima are max-reduced before the sweep - so what is left is the order of the
    floating-point sums: bounded far inside the 1e-3 single-GPU parity bound of the training gradients).  The second step starts
    from parameters that Adam's first update moved by +-lr wherever the gradient is rounding noise, so it is compared through its
    loss and its displacement vector."""
    assert np.array_equal(two[0][2], two[1][2])                            # replicas stay identical
    assert np.allclose(two[0][5], one[5], rtol=2e-5), (two[0][5], one[5])  # step 1: global loss statistics
    g1, g2 = one[6], two[0][6]
    err = np.abs(g1 - g2).max() / np.abs(g1).max()
    print("step-1 gradient, 2 ranks vs 1 process: max abs diff / max |g| =", err)
    assert err <= tol, err
    assert np.allclose(two[0][1], one[1], rtol=2e-3), (two[0][1], one[1])  # step 2: loss
    d1, d2 = one[2] - one[4], two[0][2] - two[0][4]
    cos = float((d1 * d2).sum() / (np.linalg.norm(d1) * np.linalg.norm(d2)))
    assert cos > 0.99 and abs(np.linalg.norm(d2) / np.linalg.norm(d1) - 1.0) < 0.02, cos


@pytest.mark.timeout(900)
def test_two_rank_rccl_training_steps_equal_single_gpu_steps():
    """Two optimizer steps of emap_amd.parallel.Trainer on 2 ranks (RCCL all-reduce of the flat gradient buffer, global eikonal
    denominators) == the same steps on one GPU with the whole batch."""
    _need(2)
    one = _launch(1, 256, "exact")[0]
    two = _launch(2, 256, "exact")
    _same_steps(one, two)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("sync", ["exact", "exact_lagged", "local"])
def test_two_ranks_on_one_gpu_equal_the_single_process_steps(sync):
    """The same equality on a ONE-GPU box: two ranks share GPU 0 and exchange through gloo (RCCL refuses two ranks on one device).
    Everything but the transport is the product path: HIP forward, HIP backward into the flat gradient buffer, the 5-float
    statistics exchange, one gradient all-reduce, fused Adam.  "local": rank-local eikonal denominators, ONE collective per step -
    differs from the exact step only by the mean-of-means bias."""
    _need(1)
    one = _launch(1, 256, "exact", "gloo")[0]          # 256 rays: every shard is large enough for the reverse-sweep value+grad kernel
    two = _launch(2, 256, sync, "gloo")
    if sync in ("exact", "exact_lagged"):     # exact_lagged: both recorded steps use lagged maxima x 4 (see _rank_step)
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
        assert "2 collective(s) per step" in line["config"]["parallelism"] and line["config"]["collectives_per_step"] == 2


def test_bench_refuses_more_gpus_than_visible():
    n = torch.cuda.device_count() + 1
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1"], capture_output=True, text=True,
                         timeout=300)
    assert out.returncode != 0 and "GPU(s) visible" in (out.stderr + out.stdout)
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True,
                         timeout=300, env=env)
    assert out.returncode != 0 and "refusing" in (out.stderr + out.stdout)   # launched with fewer ranks than it would report


@pytest.mark.timeout(900)
@pytest.mark.parametrize("sync", ["exact", "exact_lagged"])
def test_segmented_graph_replay_of_two_ranks_equals_their_eager_steps(sync):
    """Trainer.capture(segmented=True) on 2 ranks sharing GPU 0 over gloo: four hipGraphs (forward+statistics | compositing adjoint |
    MLP backward | Adam + loss) with the three (exact_lagged: two) collectives launched between the replays reproduce the eager 2-rank
    steps bit for bit."""
    _need(1)
    eager = _launch(2, 256, sync, "gloo")
    graph = _launch(2, 256, sync, "gloo", capture=True)
    for e, g in zip(eager, graph):
        assert np.array_equal(e[5], g[5]) and np.array_equal(e[6], g[6])      # step 1: statistics, gradient
        assert np.array_equal(e[1], g[1]) and np.array_equal(e[2], g[2])      # step 2: statistics, parameters


@pytest.mark.timeout(900)
@pytest.mark.parametrize("args", [["--mode", "render"], ["--mode", "render", "--global-rays", "4096"], ["--mode", "train"],
                                  ["--mode", "train", "--graph", "on", "--eikonal-sync", "local"]],
                         ids=["render", "render-C4-4096-rays", "train", "train-graph-local"])
def test_bench_world_2_branch_on_one_gpu_over_gloo(args):
    """`python bench.py --gpus 2 --backend gloo`: every line of the N > 1 path (re-exec under torch.distributed.run, rank slicing, parity
    on rank 0 while rank 1 waits, barriers, MAX-reduced timings, strong-scaling batch of config C4) runs on a ONE-GPU box."""
    _need(1)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "5", "--warmup", "2",
                          "--settle-steps", "3", "--no-cpu-baseline", "--no-other-modes"] + args, capture_output=True, text=True, timeout=800)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    cfg = line["config"]
    assert line["n_gpus"] == 2 and cfg["ranks"] == 2 and cfg["backend"] == "gloo" and cfg["rccl_ranks"] == 0 and line["value"] > 0
    assert cfg["settle_steps"] == 3
    if "--global-rays" in args:
        assert line["scaling"] == "strong" and cfg["rays_global"] == 4096 and cfg["rays_per_gpu"] == 2048
    else:
        assert line["scaling"] == "weak" and cfg["rays_global"] == 1024
    assert line["parity"]["meets_1e-4"]
    if "train" in args:
        n = 1 if "local" in args else 2          # default --eikonal-sync exact_lagged: 20 B SUM + the gradient bucket
        assert f"{n} collective(s) per step" in cfg["parallelism"]
        if "--graph" in args:
            assert "per phase" in cfg["launch"]


def test_default_bench_line_carries_the_training_step():
    """The driver's command (python bench.py) times the optimizer step too: key `train` of the one JSON line."""
    _need(1)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--settle-steps", "3",
                          "--no-cpu-baseline", "--no-other-modes", "--no-parity"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    tr = line["train"]
    assert "error" not in tr and tr["ms_per_step"] > 0 and 0 < tr["whole_step_frac"] < 1 and tr["value"] > 0
    assert line["config"]["settle_steps"] == 3 and line["roofline"]["kernel"].startswith("udf_mlp_rev32_kernel")
    # round 5: roofline.traffic is MEASURED in the run (two rocprofv3 --pmc child passes after the timing; static figure as the fallback):
    # the sigma' stash round trip of the dominant kernel, 0.65 GB per launch of 65 536 points
    rf = line["roofline"]
    assert rf["traffic"] is not None and 3e8 < rf["traffic"] < 1.5e9, rf
    assert rf["traffic_source"].startswith("MEASURED") or rf["traffic_source"].startswith("STATIC"), rf["traffic_source"]
    print("roofline.traffic:", rf["traffic"], rf["traffic_source"][:60])


def test_dry_run_nccl_self_check_reports():
    """`python bench.py --dry-run-nccl`: on a box with >= 2 GPUs it constructs the nccl group and steps eagerly and from per-phase graphs; with one GPU
    it says so and exits 0 (never a traceback): the first multi-GPU lease starts from a known state."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run-nccl"], capture_output=True, text=True, timeout=1500)
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    if torch.cuda.device_count() < 2:
        assert out.returncode == 0 and line["dry_run_nccl"] == "skipped"
    else:
        assert out.returncode == 0 and line["dry_run_nccl"] == "ok", (line, out.stderr[-2000:])


# ---------------------------------------------------------------------------------------- one-shot peer-to-peer all-reduce (round 5)
def _rank_oneshot(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    from emap_amd.parallel import OneShotAllReduce
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    n = 462985 + 8                                        # the d8 w256 gradient bucket with its statistics tail: NOT a multiple of 4
    ar = OneShotAllReduce(n, dev)
    gen = torch.Generator().manual_seed(100 + rank)
    ok, worst = True, 0.0
    for it in range(25):                                  # back to back: the double-buffered staging, the step counter
        m = n if it % 3 else n - 5 - it                   # shorter messages in between (all ranks the same length)
        x = torch.randn(m, generator=gen).to(dev)
        ref = x.clone()
        dist.all_reduce(ref, op=dist.ReduceOp.SUM)        # gloo, through the host
        y = ar(x.clone())
        torch.cuda.synchronize()
        if world == 2:
            ok = ok and bool(torch.equal(y, ref))         # a + b is the same number in either order
        worst = max(worst, float((y - ref).abs().max() / ref.abs().max()))
        g = [torch.empty_like(y) for _ in range(world)]
        dist.all_gather(g, y)
        ok = ok and all(bool(torch.equal(g[0], t_)) for t_ in g)      # every rank holds the SAME bits (sum in rank order)
    ar.check()
    # a captured launch replays (the step counter lives in the region, nothing of the call is host state)
    xs = torch.randn(n, generator=gen).to(dev)
    buf = xs.clone()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        ar(buf)
    torch.cuda.current_stream(dev).wait_stream(side)
    dist.barrier()
    graph = torch.cuda.CUDAGraph()
    buf.copy_(xs)
    with torch.cuda.graph(graph):
        ar(buf)
    for _ in range(3):
        buf.copy_(xs)
        graph.replay()
        torch.cuda.synchronize()
        ref = xs.clone()
        dist.all_reduce(ref, op=dist.ReduceOp.SUM)
        worst = max(worst, float((buf - ref).abs().max() / ref.abs().max()))
    ar.check()
    q.put((rank, ok, worst))
    ar.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 3])
def test_oneshot_allreduce_between_ranks_sharing_one_gpu(world):
    """csrc/allreduce.hip through OneShotAllReduce: `world` processes on GPU 0 map each other's staging regions through hipIpcMemHandle
    and sum the 1.85 MB gradient bucket by direct reads - against gloo's all-reduce of the same tensors (bit-equal for 2 ranks, 1e-6 for
    3: another summation order), bit-identical across the ranks, ragged lengths, 25 launches back to back, and from a captured graph."""
    _need(1)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_oneshot, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, ok, worst in res:
        assert ok and worst <= 1e-6, (rank, ok, worst)


def _rank_oneshot_timeout(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    from emap_amd.parallel import OneShotAllReduce
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    n = 4096 + 3
    ar = OneShotAllReduce(n, dev)
    OneShotAllReduce.set_timeout_ms(100)                  # x6 for the region's first two launches
    x = torch.ones(n, device=dev)
    for _ in range(3):                                    # three healthy launches (past the start-up allowance)
        y = ar(x.clone())
    torch.cuda.synchronize()
    ar.check()
    healthy = bool((y == world).all())
    dist.barrier()
    raised, all_nan = False, None
    if rank == 0:                                         # rank 1 never shows up for launch 4
        y = ar(x.clone())
        torch.cuda.synchronize()
        all_nan = bool(torch.isnan(y).all())
        try:
            ar.check()
        except RuntimeError:
            raised = True
    q.put((rank, healthy, all_nan, raised))
    dist.barrier()
    # no ar.close(): its barrier/unmap order is for healthy groups; the processes end here
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_oneshot_allreduce_timeout_is_loud():
    """a peer that does not arrive within the time-out must not leave a partial sum in the bucket.  Rank 1 skips a
    launch: rank 0's whole bucket is NaN (every consumer sees it) and check() raises; the launches before it were healthy."""
    _need(1)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank_oneshot_timeout, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict((r[0], r[1:]) for r in (q.get(timeout=300) for _ in range(2)))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res[0] == (True, True, True), res
    assert res[1][0] is True


@pytest.mark.timeout(900)
@pytest.mark.parametrize("sync", ["local", "exact"])
def test_trainer_with_the_oneshot_allreduce_takes_the_gloo_steps(sync):
    """Trainer(allreduce="oneshot") on 2 ranks sharing GPU 0: the gradient bucket travels through the peer-to-peer kernel, everything else as
    before - same statistics, gradients and parameters as the run whose bucket goes through gloo (2 ranks: a + b in either order), bit for bit."""
    _need(1)
    ref = _launch(2, 256, sync, "gloo")
    one = _launch(2, 256, sync, "gloo", allreduce="oneshot")
    for a, b in zip(ref, one):
        assert np.array_equal(a[5], b[5]) and np.array_equal(a[6], b[6])      # step 1: statistics, gradient
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])      # step 2: statistics, parameters


@pytest.mark.timeout(900)
@pytest.mark.parametrize("lag_factor", [1e-3, 1e3], ids=["gradients-jumped-1000x", "gradients-collapsed-1000x"])
def test_exact_lagged_survives_a_stale_range_history(lag_factor):
    """`exact_lagged` scales the fp16 sweep with 4 x LAST step's global maxima.  History 1000 x too small (the gradients jumped):
    every rank falls back to its own maxima - no fp16 overflow; 1000 x too large (they collapsed): the scale is capped at 16 x the rank's
    own maxima instead of pushing the adjoints into fp16's flush range.  Either way the step stays the exact step up to rounding (the
    ranks use different power-of-two scales for that one step), replicas stay identical, nothing is non-finite."""
    _need(1)
    one = _launch(1, 256, "exact", "gloo")[0]
    two = _launch(2, 256, "exact_lagged", "gloo", lag_factor=lag_factor)
    assert np.isfinite(two[0][6]).all() and np.isfinite(two[0][2]).all()
    _same_steps(one, two, tol=3e-4)
