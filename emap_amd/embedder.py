"""Positional encoding with the reference interface (reference src/models/embedder.py:5-53).

``get_embedder(multires, input_dims=3) -> (embed_fn, out_dim)``; ``embed_fn(x)`` returns
``[x, sin(2^k x), cos(2^k x)]_{k<multires}`` in the reference column order.  On a GPU tensor with
``input_dims == 3`` it runs the HIP ``emap_embed`` kernel; inside the MLP kernels the encoding is fused
and this function is not on the render path.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


class Embedder:
    def __init__(self, **kwargs):
        self.kwargs = kwargs
        d = kwargs["input_dims"]
        n = kwargs["num_freqs"]
        self.input_dims = d
        self.num_freqs = n
        self.include_input = kwargs.get("include_input", True)
        self.out_dim = (d if self.include_input else 0) + 2 * d * n

    def embed(self, inputs: torch.Tensor) -> torch.Tensor:
        if inputs.requires_grad and torch.is_grad_enabled():
            return embed_torch(inputs, self.num_freqs)  # a caller that wants d(PE)/d(inputs) from autograd (not on the render path)
        if self.input_dims != 3 or not self.include_input:
            raise NotImplementedError("the HIP embedder handles input_dims=3 with include_input=True")
        _lib.require_cuda(inputs, "inputs")
        x = _lib.f32c(inputs.reshape(-1, 3))
        out = torch.empty(x.shape[0], self.out_dim, device=x.device, dtype=torch.float32)
        with _lib.on_device(x):
            _lib.check(_lib.lib().emap_embed(_lib.ptr(x), x.shape[0], self.num_freqs, _lib.ptr(out), _lib.stream_ptr(x.device)), "embed")
        return out.reshape(*inputs.shape[:-1], self.out_dim)


def embed_torch(x: torch.Tensor, multires: int) -> torch.Tensor:
    """Differentiable torch formulation of the same encoding (RenderingNetwork view directions; inputs that require grad)."""
    outs = [x]
    for k in range(multires):
        f = float(2 ** k)
        outs.append(torch.sin(x * f))
        outs.append(torch.cos(x * f))
    return torch.cat(outs, -1)


def get_embedder(multires, input_dims=3):
    embed_kwargs = {"include_input": True, "input_dims": input_dims, "max_freq_log2": multires - 1,
                    "num_freqs": multires, "log_sampling": True, "periodic_fns": [torch.sin, torch.cos]}
    eo = Embedder(**embed_kwargs)
    return (lambda x, eo=eo: eo.embed(x)), eo.out_dim
