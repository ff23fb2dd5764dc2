"""`torch.optim.Adam` as the reference's train.py configures it (train.py:425-435: parameter groups with their own learning
rates, StepLR on top), with the update of ALL parameters in one HIP launch (gfs3d_adam_multi_f32).

A `torch.optim.Optimizer` subclass: `param_groups`, `state_dict()`, learning-rate schedulers and `zero_grad()` work as with
torch's own class; the per-parameter state uses torch's names (`step`, `exp_avg`, `exp_avg_sq`), so a checkpoint's optimizer
state loads into either.  HIP fp32 parameters only - anything else raises (the hot path has no CPU fallback; use
torch.optim.Adam on a CPU model).
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0.0 or eps < 0.0 or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or weight_decay < 0.0:
            raise ValueError("gfs3d.optim.FusedAdam: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        # launches are grouped by (betas, eps, step count): one launch per <= 64 tensors of a group
        batches = {}
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and p.grad.dtype == torch.float32 and not p.grad.is_sparse):
                    raise RuntimeError("gfs3d.optim.FusedAdam: HIP (cuda) fp32 parameters with dense fp32 gradients only")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] = int(st["step"]) + 1
                if not p.is_contiguous():
                    raise RuntimeError("gfs3d.optim.FusedAdam: parameters must be contiguous")
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                key = (float(b1), float(b2), float(group["eps"]), st["step"], p.device.index)
                batches.setdefault(key, []).append((p, g, st, float(group["lr"]), float(group["weight_decay"])))
        L = _lib.lib()
        for (b1, b2, eps, step, dev_index), items in batches.items():
            with torch.cuda.device(dev_index):
                stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
                for s in range(0, len(items), _lib.ADAM_MAX_TENSORS):
                    chunk = items[s:s + _lib.ADAM_MAX_TENSORS]
                    a = _lib.AdamArgs()
                    a.n_tensors = len(chunk)
                    for i, (p, g, st, lr, wd) in enumerate(chunk):
                        a.param[i], a.grad[i] = p.data_ptr(), g.data_ptr()
                        a.exp_avg[i], a.exp_avg_sq[i] = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
                        a.numel[i], a.lr[i], a.weight_decay[i] = p.numel(), lr, wd
                    a.beta1, a.beta2, a.eps = b1, b2, eps
                    a.one_minus_beta1, a.one_minus_beta2 = 1.0 - b1, 1.0 - b2          # formed in double, like torch's scalars
                    a.bias_correction1, a.bias_correction2 = 1.0 - b1 ** step, 1.0 - b2 ** step
                    _lib.check(L.gfs3d_adam_multi_f32(ctypes.byref(a), stream), "gfs3d_adam_multi_f32")
        return loss
