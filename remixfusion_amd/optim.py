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
        self._views = {}                # parameter -> (its state's `step` tensor, numpy view of it)
        # parameter -> (lo, hi): only the elements [lo, hi) of it are stepped (state tensors stay full-size).  A hash table
        # partitioned by level over several GPUs: each rank steps the levels it keeps (mp_slam/sharded.py).
        self.slices = {}
        super().__init__(params, **kw)

    def _patch_step_function(self) -> None:
        # torch wraps Optimizer.step in a hook dispatcher + profiler range (~25 us per call, a quarter of a pose iteration's
        # host time here).  step() below runs un-wrapped and hands over to the wrapped torch path itself whenever a step
        # hook is registered, so hooks keep firing.
        self._zero_grad_profile_name = f"Optimizer.zero_grad#{self.__class__.__name__}.zero_grad"

    def _hooked(self) -> bool:
        from torch.optim import optimizer as _o
        return bool(self._optimizer_step_pre_hooks or self._optimizer_step_post_hooks
                    or getattr(_o, "_global_optimizer_pre_hooks", None) or getattr(_o, "_global_optimizer_post_hooks", None))

    def _native(self) -> bool:
        if self.slices and not all(p.is_cuda for p in self.slices):
            raise _lib.RfxError("a sliced Adam step needs device parameters (there is no CPU path)")
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

    def prepare(self):
        """create the state of every parameter now (zeros, step 0: what the first step() would create on its way) -- a stream's
        first mapper step then issues its optimizer steps as fast as every later one.  Parameters the native step cannot take
        (CPU tensors, other dtypes) are left to torch's own lazy initialisation."""
        for g in self.param_groups:
            if g.get("amsgrad") or g.get("capturable") or g.get("differentiable") or isinstance(g["lr"], torch.Tensor):
                continue
            for p in g["params"]:
                if not (p.is_cuda and p.dtype == torch.float32 and p.requires_grad):
                    continue
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                stp = st["step"]
                if not stp.is_cuda and stp.dtype == torch.float32 and p not in self._views:
                    self._views[p] = (stp, stp.numpy())

    def step(self, closure=None):
        if self._hooked():
            return torch.optim.Optimizer.profile_hook_step(type(self)._step_impl)(self, closure)
        out = self._step_impl(closure)
        self._optimizer_step_code()
        return out

    @torch.no_grad()
    def _step_impl(self, closure=None):
        loss = None
        if closure is not None:                 # first: the gradients it produces decide which implementation steps them
            with torch.enable_grad():
                loss = closure()
        if not self._native():
            if self.slices:
                # torch's Adam would step the WHOLE tensor: outside a rank's own range the gradient buffer is never written
                # (nor zeroed), so the other ranks' levels and their moments would be corrupted silently
                raise _lib.RfxError("Adam.slices is set (a level-partitioned table) but the native step is unavailable for this "
                                    "configuration (amsgrad / capturable / tensor lr / non-contiguous gradient): refusing to "
                                    "fall back to a step over the whole tensor")
            torch_step = torch.optim.Adam.step             # un-wrapped: step() above has dealt with the hooks already
            getattr(torch_step, "__wrapped__", torch_step)(self, None)
            return loss
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
                # `step` stays the host tensor torch's Adam keeps, but is bumped through a numpy view of it: a CPU-tensor
                # `+= 1` costs ~7 us per parameter, which added up to more host time than the launch itself
                stp = st["step"]
                ent = self._views.get(p)
                if ent is None or ent[0] is not stp:                             # fresh state, or one loaded from a checkpoint
                    if stp.is_cuda or stp.dtype != torch.float32:                # (e.g. saved by a fused / capturable Adam)
                        stp = st["step"] = stp.detach().to("cpu", torch.float32)
                    ent = self._views[p] = (stp, stp.numpy())
                k = int(ent[1].item()) + 1
                ent[1][...] = k
                bc1, bc2 = 1.0 - b1 ** k, 1.0 - b2 ** k
                lo, hi = self.slices.get(p, (0, p.numel()))
                o = 4 * lo
                t = _lib.AdamTensor(p.data_ptr() + o, p.grad.data_ptr() + o, st["exp_avg"].data_ptr() + o, st["exp_avg_sq"].data_ptr() + o,
                                    hi - lo, b1, b2, 1.0 - b1, 1.0 - b2, g["eps"], g["weight_decay"], -(g["lr"] / bc1), math.sqrt(bc2))
                per_device.setdefault(p.device, []).append(t)
        for dev, ts in per_device.items():
            stream = _lib.stream_ptr(dev)
            for i in range(0, len(ts), _lib.ADAM_MAX_TENSORS):
                part = ts[i:i + _lib.ADAM_MAX_TENSORS]
                arr = (_lib.AdamTensor * len(part))(*part)
                if torch.cuda.current_device() == dev.index:
                    _lib.check(lib.rfx_adam_step(arr, len(part), stream), "rfx_adam_step")
                else:
                    with torch.cuda.device(dev):
                        _lib.check(lib.rfx_adam_step(arr, len(part), stream), "rfx_adam_step")
        return loss
