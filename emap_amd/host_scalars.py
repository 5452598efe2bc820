"""Host-mirrored scalars of the render dict (round 5; the drop-in training step).

The reference's training loop reads three PARAMETER-derived numbers on the host in every step (src/runner/runner_udf.py:141-148:
``variance.mean() < 2 * beta.item()``, ``variance.mean() < 0.01``; :185 ``beta.item()``).  On plain device tensors each read is a
stream synchronisation that waits for the whole forward render enqueued just before - although none of the three depends on it
(``exp(10 variance)``, ``exp(10 beta)``, ``exp(10 gamma)`` depend on the previous optimizer step only).  With
``UDFRendererBlending.host_mirror_scalars`` the renderer computes them BEFORE it enqueues the forward, starts an asynchronous copy into
a pinned buffer and hands out ``HostScalar`` tensors: ordinary device tensors (same storage, attached to autograd, every torch op
works as before) whose HOST reads - ``item()``, ``float()``, ``format()``, comparisons with python numbers, ``mean()`` of the
constant-expanded ``variance`` - are answered from the pinned copy after waiting for THAT copy only.  With trainable scalars the
pinned value is a copy of the very device value; with frozen ones (``requires_grad = False``) the device tensor comes out of the
compositing kernel's scalar block while the mirror is a separate torch expression of the same parameter - equal up to an ulp of
``exp``.  The first host read copies the number out of the ring slot (``_Slot.value``), so a render dict kept for longer than the
ring is deep still answers with ITS step's value; a dict first read only after its slot was recycled falls back to the ordinary
synchronising device read.
"""
from __future__ import annotations

import torch


class HostScalar(torch.Tensor):
    """A device tensor all of whose elements equal ONE number that also exists in pinned host memory (see the module docstring)."""

    __torch_function__ = torch._C._disabled_torch_function_impl      # torch ops see (and return) plain tensors: no dispatch overhead

    @staticmethod
    def wrap(t: torch.Tensor, slot: "_Slot", idx: int) -> "HostScalar":
        r = t.as_subclass(HostScalar)
        r._emap_host = (slot, idx)
        return r

    def _host_value(self) -> float:
        slot, idx = self._emap_host
        v = slot.read(idx)
        if v is None:                          # the ring slot was reused before anybody read this push: the device value it is
            return float(torch.Tensor.item(self.as_subclass(torch.Tensor).reshape(-1)[0]))
        return v

    def _mirrored(self) -> bool:
        return getattr(self, "_emap_host", None) is not None

    # ---- host reads ----
    def item(self):
        return self._host_value() if self._mirrored() else torch.Tensor.item(self)

    def __float__(self):
        return self._host_value() if self._mirrored() else torch.Tensor.__float__(self)

    def __format__(self, spec):
        return format(self._host_value(), spec) if (self._mirrored() and spec) else torch.Tensor.__format__(self, spec)

    def tolist(self):
        if self._mirrored() and self.numel() == 1:
            v = self._host_value()
            for _ in range(self.dim()):
                v = [v]
            return v
        return torch.Tensor.tolist(self)

    def mean(self, *a, **k):
        """mean() over ALL elements of a constant tensor = that constant: a 0-dim device tensor (attached to autograd as usual) that
        keeps the host mirror.  Any other reduction signature is torch's."""
        m = torch.Tensor.mean(self, *a, **k)
        if self._mirrored() and not a and not k:
            return HostScalar.wrap(m, *self._emap_host)
        return m

    def _cmp(self, other, op, fallback):
        if self._mirrored() and isinstance(other, (int, float)) and not isinstance(other, bool):
            return torch.tensor(op(self._host_value(), other))          # a CPU bool tensor: bool() of it does not touch the device
        return fallback(self, other)

    def __lt__(self, o):
        return self._cmp(o, lambda a, b: a < b, torch.Tensor.__lt__)

    def __le__(self, o):
        return self._cmp(o, lambda a, b: a <= b, torch.Tensor.__le__)

    def __gt__(self, o):
        return self._cmp(o, lambda a, b: a > b, torch.Tensor.__gt__)

    def __ge__(self, o):
        return self._cmp(o, lambda a, b: a >= b, torch.Tensor.__ge__)


class _Slot:
    """One push of a ``ScalarMirror``: the pinned buffer + event it travels through and the generation of that buffer it belongs to.
    ``read(idx)`` waits for the copy, takes ALL values out of the pinned buffer once (``value``) and answers from that copy afterwards;
    None if the buffer has been handed to a later push before the first read."""

    __slots__ = ("host", "event", "gen_ref", "gen", "n", "value")

    def __init__(self, host, event, gen_ref, gen, n):
        self.host, self.event, self.gen_ref, self.gen, self.n, self.value = host, event, gen_ref, gen, n, None

    def read(self, idx: int):
        if self.value is None:
            if self.gen_ref[0] != self.gen:
                return None
            self.event.synchronize()           # the small copy only - not the stream
            self.value = self.host[:self.n].tolist()
        return self.value[idx]


class ScalarMirror:
    """Ring of pinned 4-float buffers + events: ``push(values_dev)`` starts the copy on the current stream and returns its ``_Slot``."""

    def __init__(self, dev, depth: int = 8):
        self.dev = dev
        self.slots = [(torch.zeros(4).pin_memory(), torch.cuda.Event(), [0]) for _ in range(depth)]
        self.i = 0

    def push(self, values_dev: torch.Tensor) -> _Slot:
        host, ev, gen = self.slots[self.i % len(self.slots)]
        if self.i >= len(self.slots):
            ev.synchronize()                   # the buffer's previous copy (depth steps ago) - done long since
        self.i += 1
        gen[0] += 1                            # slots of earlier pushes through this buffer stop reading it
        host[:values_dev.numel()].copy_(values_dev, non_blocking=True)
        ev.record(torch.cuda.current_stream(self.dev))
        return _Slot(host, ev, gen, gen[0], values_dev.numel())


class MaskedSelection:
    """``x[bool_mask]`` without its device synchronisation.  Boolean-mask indexing has a data-dependent output shape: torch runs
    ``nonzero`` and WAITS for the count.  The reference's loop does it once per step for a value it only logs
    (runner_udf.py:126: ``udf.min(dim=1)[0][mask[:, 0] > 0.5].mean()``) - on the drop-in that wait sat in the middle of the step, before
    the loop's own loss kernels were even enqueued.  This object stands for the selection; the reductions a selection is normally
    followed by are evaluated WITHOUT materialising it (``mean`` = sum(where(mask, x, 0)) / count(mask): same value up to the order of
    the fp32 sum, NaN for an empty selection like torch's); anything else materialises it the ordinary way (and synchronises)."""

    def __init__(self, x: torch.Tensor, mask: torch.Tensor):
        self._x, self._m, self._t = x, mask, None

    def _expand(self):
        m = self._m
        while m.dim() < self._x.dim():
            m = m.unsqueeze(-1)
        return m

    def materialize(self) -> torch.Tensor:
        if self._t is None:
            self._t = torch.Tensor.__getitem__(self._x.as_subclass(torch.Tensor), self._m)
        return self._t

    def sum(self, *a, **k):
        if a or k:
            return self.materialize().sum(*a, **k)
        x = self._x.as_subclass(torch.Tensor)
        return torch.where(self._expand(), x, torch.zeros((), dtype=x.dtype, device=x.device)).sum()

    def mean(self, *a, **k):
        if a or k:
            return self.materialize().mean(*a, **k)
        n = self._m.sum() * (self._x.numel() // max(self._m.numel(), 1))
        return self.sum() / n

    def __getattr__(self, name):           # everything else: the real tensor
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return getattr(self.materialize(), name)

    def __len__(self):
        return len(self.materialize())

    def __iter__(self):
        return iter(self.materialize())

    def __repr__(self):
        return "MaskedSelection(" + repr(self.materialize()) + ")"


class LazyMaskable(torch.Tensor):
    """A tensor whose boolean-mask indexing returns a ``MaskedSelection`` (see there); ``min(dim=...)`` / ``max(dim=...)`` hand the
    property on to their values.  Everything else - including autograd - is the plain tensor."""

    __torch_function__ = torch._C._disabled_torch_function_impl

    def __getitem__(self, key):
        if isinstance(key, torch.Tensor) and key.dtype == torch.bool and key.dim() >= 1 and key.shape == self.shape[:key.dim()]:
            return MaskedSelection(self, key)
        return torch.Tensor.__getitem__(self, key)

    def _reduce(self, fn, *a, **k):
        r = fn(self, *a, **k)
        if isinstance(r, tuple) and len(r) == 2 and isinstance(r[0], torch.Tensor):
            return (r[0].as_subclass(LazyMaskable), r[1])
        return r

    def min(self, *a, **k):
        return self._reduce(torch.Tensor.min, *a, **k)

    def max(self, *a, **k):
        return self._reduce(torch.Tensor.max, *a, **k)
