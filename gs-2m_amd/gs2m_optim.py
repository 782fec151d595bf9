"""Fused optimizer step for the Gaussian parameter groups.

`Adam` is a `torch.optim.Optimizer` with `torch.optim.Adam`'s constructor, param-group keys and per-parameter state
layout (`step`, `exp_avg`, `exp_avg_sq`), so it drops into the reference's
    self.optimizer = torch.optim.Adam(l, lr=0.0, eps=1e-15)             scene/gaussian_model.py:245
and the reference code that edits the optimizer afterwards keeps working unchanged: `update_learning_rate`
(GM:251-258), `replace_tensor_to_optimizer` / `_prune_optimizer` / `cat_tensors_to_optimizer` (GM:372-455), and the
checkpoint's `optimizer.state_dict()` / `load_state_dict` (train.py:55-63, GM:98-110) -- state dicts are
interchangeable with torch.optim.Adam's.

`step()` is ONE kernel launch for the whole model through the C ABI (`include/gs2m_optim.h`, csrc/optim.hip):
28 B of HBM traffic per element instead of ~100 B for torch's eight-pass foreach implementation, same arithmetic
order, bit-identical results on the device.  There is no CPU path: parameters must be fp32 CUDA tensors.
"""
import ctypes as C

import torch

import gs2m_native as _native


class _AdamTensor(C.Structure):  # struct gs2m_adam_tensor
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("numel", C.c_uint64), ("lr", C.c_double), ("step", C.c_int64)]


def _step_value(s):
    return int(s.item()) if torch.is_tensor(s) else int(s)


class Adam(torch.optim.Optimizer):
    """torch.optim.Adam(params, lr, betas, eps) with weight_decay = 0, amsgrad = False, maximize = False -- what the
    reference uses -- as a single fused HIP launch."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, *, maximize=False,
                 foreach=None, capturable=False, differentiable=False, fused=None, decoupled_weight_decay=False):
        if weight_decay != 0 or amsgrad or maximize or capturable or differentiable:
            raise NotImplementedError("gs2m_optim.Adam implements the configuration the reference trains with: "
                                      "weight_decay=0, amsgrad=False, maximize=False")
        if not 0.0 <= lr:
            raise ValueError(f"Invalid learning rate: {lr}")
        if not 0.0 <= eps:
            raise ValueError(f"Invalid epsilon value: {eps}")
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError(f"Invalid beta parameter at index 0: {betas[0]}")
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError(f"Invalid beta parameter at index 1: {betas[1]}")
        # the same defaults dict torch.optim.Adam stores, so state_dict()s are interchangeable
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, maximize=False, foreach=None,
                        capturable=False, differentiable=False, fused=None, decoupled_weight_decay=False)
        super().__init__(params, defaults)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        # launches are grouped by (device, betas, eps); the reference has one such group: everything
        batches = {}
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                g = p.grad
                if g.is_sparse:
                    raise RuntimeError("gs2m_optim.Adam does not support sparse gradients")
                if not (p.is_cuda and p.dtype == torch.float32 and g.dtype == torch.float32):
                    raise RuntimeError("gs2m_optim.Adam: parameters and gradients must be fp32 CUDA tensors "
                                       "(the fused step is a HIP kernel; there is no CPU path)")
                if not p.is_contiguous():
                    raise RuntimeError("gs2m_optim.Adam: parameters must be contiguous")
                state = self.state[p]
                if len(state) == 0:  # torch.optim.Adam's lazy initialisation
                    state["step"] = torch.tensor(0.0, dtype=torch.float32)
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                m, v = state["exp_avg"], state["exp_avg_sq"]
                if not (m.is_contiguous() and v.is_contiguous() and m.shape == p.shape and v.shape == p.shape
                        and m.dtype == torch.float32 and v.dtype == torch.float32 and m.device == p.device and v.device == p.device):
                    raise RuntimeError("gs2m_optim.Adam: exp_avg / exp_avg_sq must be contiguous fp32 tensors of the parameter's shape and device")
                if torch.is_tensor(state["step"]):
                    state["step"] += 1
                else:
                    state["step"] = state["step"] + 1
                g = g if g.is_contiguous() else g.contiguous()
                key = (p.device, float(beta1), float(beta2), float(group["eps"]))
                batches.setdefault(key, []).append((p, g, m, v, float(group["lr"]), _step_value(state["step"])))
        for (device, beta1, beta2, eps), items in batches.items():
            arr = (_AdamTensor * len(items))()
            for a, (p, g, m, v, lr, step) in zip(arr, items):
                a.param, a.grad, a.exp_avg, a.exp_avg_sq = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr()
                a.numel, a.lr, a.step = p.numel(), lr, step
            with _native.device_guard(device):
                _native.check(_native.lib().gs2m_adam_step(len(items), arr, beta1, beta2, eps,
                                                           C.c_void_p(_native.stream_ptr(device))),
                              "gs2m_adam_step")
        return loss
