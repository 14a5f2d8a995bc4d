"""Adam for the mapper's two optimizers, stepped by one librfx launch (rfx_adam_step).

Reference: ``optim.Adam(trainable_parameters, betas=(0.9, 0.99))`` / ``optim.Adam(rba_parameter, ...)`` in
mp_slam/slam.py:271-286, stepped in mp_slam/mapper.py:416-418 and :497-499.  This is a ``torch.optim.Adam`` (same
constructor, param groups, ``state_dict`` layout: ``step`` / ``exp_avg`` / ``exp_avg_sq`` per parameter); only
``step()`` differs: fp32 device parameters are updated by librfx, everything else (CPU tensors, other dtypes, amsgrad,
maximize, closures that need grad mode) takes torch's own implementation.
"""
import ctypes as C
import math

import torch

from . import _lib


class Adam(torch.optim.Adam):
    def __init__(self, params, **kw):
        kw.pop("fused", None)
        kw.pop("foreach", None)
        super().__init__(params, **kw)

    def _native(self) -> bool:
        for g in self.param_groups:
            if g.get("amsgrad") or g.get("maximize") or g.get("capturable") or g.get("differentiable"):
                return False
            if isinstance(g["lr"], torch.Tensor):
                return False
            for p in g["params"]:
                if p.grad is None:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_cuda
                        and p.grad.dtype == torch.float32 and p.grad.is_contiguous() and not p.grad.is_sparse):
                    return False
        return True

    @torch.no_grad()
    def step(self, closure=None):
        if not self._native():
            return super().step(closure)
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        per_device = {}
        for g in self.param_groups:
            b1, b2 = g["betas"]
            for p in g["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)         # host counter, like torch's default Adam
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if st["step"].is_cuda:                                           # state loaded from a fused-Adam checkpoint
                    st["step"] = st["step"].cpu()
                st["step"] += 1
                k = float(st["step"])
                bc1, bc2 = 1.0 - b1 ** k, 1.0 - b2 ** k
                t = _lib.AdamTensor(p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(),
                                    b1, b2, 1.0 - b1, 1.0 - b2, g["eps"], g["weight_decay"], -(g["lr"] / bc1), math.sqrt(bc2))
                per_device.setdefault(p.device, []).append(t)
        for dev, ts in per_device.items():
            stream = _lib.stream_ptr(dev)
            for i in range(0, len(ts), _lib.ADAM_MAX_TENSORS):
                part = ts[i:i + _lib.ADAM_MAX_TENSORS]
                arr = (_lib.AdamTensor * len(part))(*part)
                with torch.cuda.device(dev):
                    _lib.check(lib.rfx_adam_step(arr, len(part), stream), "rfx_adam_step")
        return loss
