"""On-device ray / pixel sampler with the interface of ``Dataset.gen_random_rays_patches_at``
(reference src/dataset/dataset.py:222-307) - SURVEY.md par. 8 f3.

The reference draws the pixels of a training batch on the host (``torch.randint``; python ``random.choices`` over all H*W
pixel probabilities when ``importance_sample=True``), builds the rays with small CPU ops and copies six tensors to the GPU
every step.  ``DeviceRaySampler`` uploads the dataset ONCE (edge maps, inverse intrinsics, poses, per-image pixel lists for
the edge-weighted draw) and produces every batch with one kernel launch (``emap_sample_rays``): zero host->device copies
per step, no host synchronisation, graph-capturable (the step counter is a device word).

Wire format respected: ``meta_data.json`` {scene_box{near,far,radius,aabb}, height, width, frames[{intrinsics 4x4,
camtoworld 4x4, rgb_path}]} (dataset.py:66-104) via ``from_meta``; edge maps are whatever ``cv.imread(path, 0) / 255`` gave the
caller (dataset.py:133-135) - image decoding is outside the hot path and stays with the caller (no cv2 in this image).

The host RNG streams (torch CPU generator, python ``random``) cannot be reproduced on a device: the draw is a
Philox4x32-10 stream keyed by ``seed``; the deterministic part (rays of given pixels) and the sampling distribution are what
the parity tests pin.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib


class DeviceRaySampler:
    def __init__(self, edges, intrinsics_all, pose_all, device="cuda", seed=0, near=None, far=None):
        """edges: (n_images,H,W[,1]) float in [0,1]; intrinsics_all, pose_all: (n_images,4,4) (dataset.py:86-87,117-121)."""
        dev = torch.device(device)
        edges = torch.as_tensor(np.asarray(edges), dtype=torch.float32) if not isinstance(edges, torch.Tensor) else edges.float()
        if edges.dim() == 4:
            edges = edges[..., 0]
        self.n_images, self.H, self.W = [int(v) for v in edges.shape]
        self.image_pixels = self.H * self.W
        K = torch.as_tensor(intrinsics_all, dtype=torch.float32).reshape(self.n_images, 4, 4)
        P = torch.as_tensor(pose_all, dtype=torch.float32).reshape(self.n_images, 4, 4)
        self.intrinsics_all, self.pose_all = K.to(dev), P.to(dev)
        self.intrinsics_all_inv = torch.inverse(K)                  # dataset.py:119 (CPU, once)
        self.focal = K[0][0, 0]
        self.near, self.far = near, far
        self.device = dev
        self.seed = int(seed)
        flat = edges.reshape(self.n_images, -1)
        is_edge = flat > 0.1                                         # dataset.py:240
        # stable partition of the row-major pixel ids: edge pixels first
        order = torch.argsort((~is_edge).to(torch.uint8), dim=1, stable=True).to(torch.int32)
        self._edges = edges.contiguous().to(dev)
        self._order = order.contiguous().to(dev)
        self._n_edge = is_edge.sum(1).to(torch.int32).to(dev)
        self._density = flat.mean(1).float().to(dev)                 # edge_density = np.mean(img_np), :238
        self._kinv = self.intrinsics_all_inv[:, :3, :3].contiguous().to(dev)
        self._pose = P.contiguous().to(dev)
        self._perm = None
        self._counter = torch.zeros(1, dtype=torch.int64, device=dev)   # device-resident step counter (uint64 bits)
        self._ds = _lib.RayDataset(self._edges.data_ptr(), self._order.data_ptr(), self._n_edge.data_ptr(), self._density.data_ptr(),
                                   self._kinv.data_ptr(), self._pose.data_ptr(), None, self.n_images, self.H, self.W, 0)

    @classmethod
    def from_meta(cls, meta: dict, edges, device="cuda", seed=0):
        """meta: the parsed ``meta_data.json`` (dataset.py:66-104); edges: the decoded edge maps in frame order."""
        assert int(meta["height"]) == np.asarray(edges).shape[1] and int(meta["width"]) == np.asarray(edges).shape[2]
        K = torch.stack([torch.tensor(f["intrinsics"], dtype=torch.float32) for f in meta["frames"]])
        P = torch.stack([torch.tensor(f["camtoworld"], dtype=torch.float32)[:4, :4] for f in meta["frames"]])
        box = meta["scene_box"]
        return cls(edges, K, P, device=device, seed=seed, near=box["near"], far=box["far"])

    def set_image_perm(self, perm):
        """The runner's ``image_perm`` (runner_udf.py:79-82): with it ``img_idx=None`` walks the permutation on the device."""
        self._perm = torch.as_tensor(perm, dtype=torch.int32).contiguous().to(self.device)
        self._ds.image_perm = self._perm.data_ptr()

    def gen_random_rays_patches_at(self, img_idx, batch_size, importance_sample=False, pixels=None):
        """-> the reference's ``sample`` dict (dataset.py:288-305), every tensor on the device.  img_idx=None: the image is
        chosen on the device from the step counter (graph-capturable).  pixels ((N,2) int64 x,y): bypass the random draw."""
        dev = self.device
        if dev.type != "cuda":
            raise RuntimeError("emap_amd.DeviceRaySampler: the sampler kernel needs an MI355X (cuda) device; there is no CPU fallback "
                               "(the reference's own host sampler is Dataset.gen_random_rays_patches_at)")
        N = int(batch_size)
        f = torch.empty(N * 14, dtype=torch.float32, device=dev)
        rays_o, rays_v, edge, ds, uv, pc, tr = f[:3 * N].view(N, 3), f[3 * N:6 * N].view(N, 3), f[6 * N:7 * N].view(N, 1), \
            f[7 * N:8 * N].view(N, 1), f[8 * N:10 * N].view(N, 2), f[10 * N:13 * N].view(N, 3), f[13 * N:].view(N, 1)
        pix = torch.empty(N, 2, dtype=torch.int64, device=dev)
        img = torch.empty(1, dtype=torch.int32, device=dev)
        out = _lib.RayBatch(rays_o.data_ptr(), rays_v.data_ptr(), edge.data_ptr(), ds.data_ptr(), uv.data_ptr(), pc.data_ptr(),
                            pix.data_ptr(), img.data_ptr(), tr.data_ptr())
        pin = None
        if pixels is not None:
            px = torch.as_tensor(pixels)
            assert px.shape == (N, 2)
            if px.device.type == "cpu":      # given pixels index the edge maps in the kernel: reject out-of-range ones where it is free
                if bool(((px[:, 0] < 0) | (px[:, 0] >= self.W) | (px[:, 1] < 0) | (px[:, 1] >= self.H)).any()):
                    raise ValueError(f"DeviceRaySampler: pixels outside the {self.W}x{self.H} image")
                pin = px.to(torch.int64).to(dev).contiguous()
            else:                            # device pixels: clamped on the device (no host synchronisation)
                pin = torch.stack([px[:, 0].clamp(0, self.W - 1), px[:, 1].clamp(0, self.H - 1)], -1).to(torch.int64).contiguous()
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().emap_sample_rays(C.byref(self._ds), -1 if img_idx is None else int(img_idx), N, int(bool(importance_sample)),
                                                   self.seed, 0, _lib.ptr(self._counter), _lib.ptr(pin), C.byref(out),
                                                   _lib.stream_ptr(dev)), "sample_rays")
        rays = {"rays_o": rays_o, "rays_v": rays_v, "edge": edge}
        # "t_rand" is not in the reference's dict: render()'s per-ray jitter (udf_renderer_blending.py:719 draws torch.rand([N,1]) - 0.5 on the
        # host generator), from the same device draw - pass it as render(..., t_rand=sample["t_rand"]) and the step has no host draw at all
        sample = {"rays": rays, "rays_ndc_uv": uv, "rays_norm_XYZ_cam": pc, "depth_scale": ds, "pixels": pix, "img_idx": img, "t_rand": tr}
        if img_idx is not None:
            sample["pose"] = self.pose_all[int(img_idx)]
            sample["intrinsics"] = self.intrinsics_all[int(img_idx)]
        return sample
