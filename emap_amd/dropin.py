"""Make EMAP's unmodified ``src/runner`` + ``main.py`` use emap_amd.

    import emap_amd.dropin; emap_amd.dropin.install()      # before `from src.runner... import`

registers ``src.models.udf_model``, ``src.models.udf_renderer_blending``, ``src.models.embedder`` and
``src.models.loss`` in ``sys.modules`` as aliases of the emap_amd modules of the same names, so
``runner_base.py:9-13``'s imports resolve to the HIP-backed classes.  Where the reference's other modules can be imported
it also re-routes the two callers either side of the path: ``Dataset.gen_random_rays_patches_at`` (on-device ray sampler,
SURVEY f3) and - once ``src.runner.runner_udf`` is imported, call ``install()`` again or ``patch_runner()`` - the
validation loop (reduced-output renders).  See INTEGRATION.md.
"""
import importlib
import sys
import types

_ALIASES = {"src.models.udf_model": "emap_amd.udf_model",
            "src.models.udf_renderer_blending": "emap_amd.udf_renderer_blending",
            "src.models.embedder": "emap_amd.embedder",
            "src.models.loss": "emap_amd.loss"}


def _default_seed():
    """Seed of the device pixel draw when the dataset carries no ``emap_seed``: torch's seed (so ``torch.manual_seed`` in main.py
    still selects the ray stream) plus the rank of a data-parallel run (every rank its own stream)."""
    import torch
    seed = int(torch.initial_seed()) & 0x7fffffff
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            seed = (seed + 7919 * dist.get_rank()) & 0x7fffffff
    except Exception:
        pass
    return seed


def dataset_method(sampler_cls=None, original=None):
    """Replacement for ``Dataset.gen_random_rays_patches_at`` (src/dataset/dataset.py:222-307): the first call uploads the dataset's
    edge maps / intrinsics / poses once (``DeviceRaySampler``), every call is then ONE kernel launch and no host->device copy.
    Returns the reference's dict (rays{rays_o, rays_v, edge}, pose, intrinsics, rays_ndc_uv, rays_norm_XYZ_cam, depth_scale).
    The pixel stream is a device Philox draw, NOT the reference's host RNG streams (torch.randint / random.choices): same
    distribution, other pixels.  Seeded by ``dataset.emap_seed`` if set, else by ``torch.initial_seed()`` + rank.
    A dataset on a non-CUDA device keeps the reference's own method (`original`)."""
    def gen_random_rays_patches_at(self, img_idx, batch_size, importance_sample=False):
        import torch
        if original is not None and torch.device(self.device).type != "cuda":
            return original(self, img_idx, batch_size, importance_sample)
        s = getattr(self, "_emap_sampler", None)
        if s is None:
            cls = sampler_cls
            if cls is None:
                from .ray_sampler import DeviceRaySampler as cls
            seed = getattr(self, "emap_seed", None)
            s = cls(self.edges, self.intrinsics_all, self.pose_all, device=self.device, seed=_default_seed() if seed is None else seed)
            self._emap_sampler = s
        # the reference takes the importance branch only when the dataset has masks (:236-238)
        smp = s.gen_random_rays_patches_at(int(img_idx), batch_size, importance_sample=bool(importance_sample and self.masks is not None))
        return {"rays": {k: smp["rays"][k] for k in ("rays_o", "rays_v", "edge")}, "pose": self.pose_all[int(img_idx)],
                "intrinsics": self.intrinsics_all[int(img_idx)], "rays_ndc_uv": smp["rays_ndc_uv"],
                "rays_norm_XYZ_cam": smp["rays_norm_XYZ_cam"], "depth_scale": smp["depth_scale"]}
    return gen_random_rays_patches_at


def validate_wrapper(orig_validate):
    """``Runner_UDF.validate`` (src/runner/runner_udf.py:287-484) consumes per-ray results only - ``edge``, ``depth`` and
    sum_s gradients_flip * weights (:333-407).  The wrapper runs it without autograd and with the renderer in its reduced-output
    launch mode (28 B per ray instead of 48 B per sample, ``UDFRendererBlending.inference_reduced``)."""
    import torch

    def validate(self, *a, **k):
        r = self.renderer
        old = getattr(r, "inference_reduced", False)
        r.inference_reduced = True
        try:
            with torch.no_grad():
                return orig_validate(self, *a, **k)
        finally:
            r.inference_reduced = old
    validate.__wrapped__ = orig_validate
    return validate


def _patch_dataset():
    try:   # needs cv2 & co: present where the reference runs, absent in the build image
        ds = importlib.import_module("src.dataset.dataset")
    except Exception:
        return False
    if getattr(ds.Dataset.gen_random_rays_patches_at, "_emap_patched", False):
        return True
    fn = dataset_method(original=ds.Dataset.gen_random_rays_patches_at)
    fn._emap_patched = True
    ds.Dataset.gen_random_rays_patches_at = fn
    return True


class DeferredScalarWriter:
    """Stands in for the runner's ``SummaryWriter`` (runner_udf.py:47): ``add_scalar(tag, value, step)`` with a DEVICE tensor does
    not read it (tensorboard's ``make_np`` is ``value.cpu()``: a stream synchronisation per call, seven per step at runner_udf.py:172-186)
    but keeps the 0-dim tensor; every ``flush_every`` steps - and on ``flush()`` / ``close()`` - the pending values travel in ONE copy
    and are handed to the real writer with their original tags and steps, in order.  Python numbers and host tensors pass through
    unchanged, as does every other method."""

    def __init__(self, writer, flush_every: int = 100, max_pending: int = 4096):
        self._w, self._every, self._max = writer, max(int(flush_every), 1), int(max_pending)
        self._pending, self._last_flush_step = [], None

    def add_scalar(self, tag, scalar_value, global_step=None, *a, **k):
        import torch
        if isinstance(scalar_value, torch.Tensor) and scalar_value.is_cuda:
            if getattr(scalar_value, "_emap_host", None) is not None:     # a host-mirrored scalar (host_scalars.py): its value is on the host already
                scalar_value = float(scalar_value)
            else:
                self._pending.append((tag, scalar_value.detach().reshape(-1)[:1], global_step, a, k))
                if self._last_flush_step is None:
                    self._last_flush_step = global_step if isinstance(global_step, int) else 0
                due = isinstance(global_step, int) and global_step - self._last_flush_step >= self._every
                if due or len(self._pending) >= self._max:
                    self.flush_pending()
                    self._last_flush_step = global_step if isinstance(global_step, int) else 0
                return None
        if self._pending:                  # rows reach the real writer in the order the runner issued them
            self._pending.append((tag, scalar_value, global_step, a, k))
            return None
        return self._w.add_scalar(tag, scalar_value, global_step, *a, **k)

    def flush_pending(self):
        import torch
        if not self._pending:
            return
        dev = [p[1] for p in self._pending if isinstance(p[1], torch.Tensor) and p[1].is_cuda]
        vals = iter(torch.cat(dev).float().cpu().tolist()) if dev else iter(())      # one device-to-host copy
        for tag, v, step, a, k in self._pending:
            self._w.add_scalar(tag, next(vals) if (isinstance(v, torch.Tensor) and v.is_cuda) else v, step, *a, **k)
        self._pending = []

    def flush(self):
        self.flush_pending()
        return self._w.flush() if hasattr(self._w, "flush") else None

    def close(self):
        self.flush_pending()
        return self._w.close() if hasattr(self._w, "close") else None

    def __getattr__(self, name):
        return getattr(self._w, name)


def train_wrapper(orig_train, mod=None, fused_adam: bool = True, defer_scalars: bool = True, single_thread_autograd: bool = True):
    """``Runner_UDF.train_udf`` (src/runner/runner_udf.py:35-250) unmodified, with the host-side cost of its step removed where
    that is possible without touching its code:
      * the renderer hands out host-mirrored ``variance / beta / gamma`` (host_scalars.py): the reads of runner_udf.py:141-148,185 do not
        wait for the forward render, so the runner's own small loss kernels queue up behind it instead of after a synchronisation;
      * ``RenderFn.backward`` installs the parameter gradients itself (``direct_param_grads``) and ``FusedAdam`` (swapped in for the
        runner's ``torch.optim.Adam`` over the same groups: one launch instead of 32 x 6) reads them in place;
      * the tensorboard writer the method creates defers device scalars (``DeferredScalarWriter``, flushed every ``report_freq`` steps);
      * ``loss.backward()`` runs on the calling thread (``torch.autograd.set_multithreading_enabled(False)`` for the duration of the loop).
    What stays: the progress bar's ``loss.item()`` / ``format(psnr)`` (runner_udf.py:164) - one wait for the forward per step.
    Side effects that outlive the call: ``self.optimizer`` STAYS the FusedAdam - the runner's later ``save_checkpoint`` /
    a second ``train_udf`` must see the moments of the steps taken here, and its ``state_dict()`` is in ``torch.optim.Adam``'s layout
    (a stock Adam over the same groups loads it).  Everything else is restored in the ``finally`` block, pending tensorboard rows are
    flushed there too - also when the loop raises."""
    def train_udf(self, *a, **k):
        import torch
        from .parallel import FusedAdam
        r = self.renderer
        old = (getattr(r, "host_mirror_scalars", False), getattr(r, "direct_param_grads", False))
        r.host_mirror_scalars, r.direct_param_grads = True, True
        if fused_adam and isinstance(self.optimizer, torch.optim.Adam) and torch.device(getattr(r, "device", "cuda")).type == "cuda":
            self.optimizer = FusedAdam.from_adam(self.optimizer)
        m = mod if mod is not None else sys.modules.get(type(self).__module__)
        sw = getattr(m, "SummaryWriter", None) if m is not None else None
        if defer_scalars and sw is not None:
            every = int(getattr(self, "report_freq", 100) or 100)
            m.SummaryWriter = lambda *wa, **wk: DeferredScalarWriter(sw(*wa, **wk), flush_every=every)
        # autograd on the calling thread: the engine's hand-off to its per-device worker thread and back costs ~0.1 ms per backward() - with the
        # GPU idle, between the forward and the backward of every step (the graph here is one RenderFn node and six small ones)
        mt = torch.autograd.is_multithreading_enabled()
        if single_thread_autograd:
            torch.autograd.set_multithreading_enabled(False)
        try:
            return orig_train(self, *a, **k)
        finally:
            torch.autograd.set_multithreading_enabled(mt)
            if defer_scalars and sw is not None:
                m.SummaryWriter = sw
            w = getattr(self, "writer", None)
            if isinstance(w, DeferredScalarWriter):
                w.flush_pending()
            r.host_mirror_scalars, r.direct_param_grads = old
    train_udf.__wrapped__ = orig_train
    return train_udf


def _patch_runner(train: bool = False):
    mod = sys.modules.get("src.runner.runner_udf")   # patched only if the caller has it imported (it imports src.models.* itself)
    if mod is None or not hasattr(mod, "Runner_UDF"):
        return False
    done = False
    if not hasattr(mod.Runner_UDF.validate, "__wrapped__"):
        mod.Runner_UDF.validate = validate_wrapper(mod.Runner_UDF.validate)
        done = True
    if train and hasattr(mod.Runner_UDF, "train_udf") and not hasattr(mod.Runner_UDF.train_udf, "__wrapped__"):
        mod.Runner_UDF.train_udf = train_wrapper(mod.Runner_UDF.train_udf, mod)
        done = True
    return done


def install(force: bool = True, patch_dataset: bool = True):
    """Alias ``src.models.*`` to this package; ``patch_dataset=False`` keeps the reference's host-side ray sampler (its RNG streams)."""
    for pkg in ("src", "src.models"):
        if pkg not in sys.modules:
            try:
                importlib.import_module(pkg)
            except ImportError:
                m = types.ModuleType(pkg)
                m.__path__ = []
                sys.modules[pkg] = m
    for alias, real in _ALIASES.items():
        if force or alias not in sys.modules:
            mod = importlib.import_module(real)
            sys.modules[alias] = mod
            setattr(sys.modules["src.models"], alias.rsplit(".", 1)[1], mod)
    if patch_dataset:
        _patch_dataset()
    _patch_runner()
    # extraction queries (SURVEY par. 8 f2): the reference module keeps its other functions; only the two query routines
    # are replaced, and only if the module can be imported at all (it needs nothing but torch)
    try:
        ep = importlib.import_module("src.edge_extraction.extract_pointcloud")
    except Exception:
        ep = None
    if ep is not None:
        from . import extraction
        ep.get_udf_normals_grid = extraction.get_udf_normals_grid
        ep.get_udf_normals_slow = extraction.get_udf_normals_slow
    return sorted(_ALIASES)


def patch_runner(train: bool = False):
    """Call after ``from src.runner.runner_udf import Runner_UDF``: wraps ``Runner_UDF.validate`` (see validate_wrapper) and, with
    ``train=True``, ``Runner_UDF.train_udf`` (see train_wrapper)."""
    return _patch_runner(train)
