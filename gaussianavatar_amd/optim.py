"""`Adam` — torch.optim.Adam (the reference's optimiser, /root/reference/model/avatar_model.py:152-161)
whose `step()` is ONE HIP launch over every parameter tensor of every group (ganet_adam_step,
gaussianavatar_amd/csrc/ganet_optim.hip) instead of torch's per-group multi-tensor launches.

Same hyper-parameters, same per-parameter state keys (`step`, `exp_avg`, `exp_avg_sq`) and therefore
the same `state_dict()` layout: checkpoints written by either load into the other (`step` is kept as
torch's default flavour keeps it, a float32 CPU scalar tensor; a device-side `step` from a fused-Adam
checkpoint is accepted). Supported configuration = the one the reference uses: amsgrad, weight decay,
maximize, capturable and differentiable off; float32 CUDA parameters with dense gradients.

Overflow safety (no host sync): the rasterizer's backward pass raises a device-side flag when a forward pass had
overflowed its pair buffer — that frame's gradients are zeros (rasterizer.overflow_flag). `step()` hands the flag
to the kernel, which then changes nothing: a step computed from truncated tile lists is never applied — and lowers the
flag again behind its last launch (stream-ordered), so the flag describes ONE step whatever the caller's loop does in
between (module.zero_grad(), `p.grad = None`, two steps per zero_grad: ADVICE r04); `zero_grad()` lowers it as well, so a
backward pass that is never followed by a step cannot make the next valid step skip itself. (A skipped step still advances the `step` counter of the bias
corrections; at the reference's betas that shifts the step size of the following updates by < 1e-3 relative after
a few hundred steps.)
"""
from __future__ import annotations

import torch

from . import _native


def _same_layout(a: torch.Tensor, b: torch.Tensor) -> bool:
    """Same shape and the same memory order (strides of size-1 dimensions do not matter)."""
    return a.shape == b.shape and all(sa == sb for sa, sb, n in zip(a.stride(), b.stride(), a.shape) if n > 1)


class Adam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, foreach=False,
                         fused=False)
        self._table = (_native.GanetAdamTensor * 64)()
        self.skip_on_overflow = True      # read the rasterizer's overflow flag (see the module docstring)

    def _skip_flag(self, device):
        if not self.skip_on_overflow:
            return None
        from . import rasterizer
        return rasterizer.overflow_flag(device).data_ptr()

    def zero_grad(self, set_to_none: bool = True):
        """Also lowers the overflow flag: a backward pass that was never followed by a step (a gradient probe, an
        exception between backward and step) must not make the NEXT step skip itself."""
        super().zero_grad(set_to_none=set_to_none)
        if self.skip_on_overflow:
            for group in self.param_groups:
                for p in group["params"]:
                    if p.is_cuda:
                        _native.ganet_check(_native.ganet().ganet_flag_clear(self._skip_flag(p.device),
                                                                             _native.raw_stream(p.device)))
                        return

    def _group_step(self, group) -> int:
        """Advance the group's step counter: one shared CPU tensor referenced from every parameter's
        state (torch keeps one per parameter; they always agree)."""
        shared = group.get("_shared_step")
        if shared is None:
            first = next((self.state[p]["step"] for p in group["params"] if "step" in self.state[p]), None)
            shared = torch.tensor(float(first) if first is not None else 0.0, dtype=torch.float32)
            group["_shared_step"] = shared
        shared += 1
        return int(shared)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _native.ganet()
        table, n, stream, skip = self._table, 0, None, None
        flush_args = None
        for group in self.param_groups:
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            if group["amsgrad"] or group["weight_decay"] != 0 or group["maximize"]:
                raise RuntimeError("gaussianavatar_amd.optim.Adam: amsgrad / weight_decay / maximize are not supported")
            beta1, beta2 = group["betas"]
            args = (float(beta1), float(beta2), float(group["eps"]))
            if flush_args is not None and args != flush_args and n:
                _native.ganet_check(lib.ganet_adam_step(n, table, *flush_args, skip, stream))
                n = 0
            flush_args = args
            t = self._group_step(group)
            shared = group["_shared_step"]
            bc1, bc2 = 1.0 - beta1 ** t, 1.0 - beta2 ** t
            lr = float(group["lr"])
            for p in params:
                g = p.grad
                st = self.state[p]
                fixed = st.get("_row")          # (param ptr, moment ptrs, numel): verified once, valid while the tensors stay put
                if (fixed is None or fixed[0] != p.data_ptr() or fixed[1] != st["exp_avg"].data_ptr()
                        or fixed[2] != st["exp_avg_sq"].data_ptr()):
                    fixed = self._verify(p, st)
                st["step"] = shared
                # the update is element-wise over memory: parameter, gradient and moments must be dense and share
                # ONE layout (row-major or channels-last; the moments are created with the parameter's)
                if g.is_sparse:
                    raise RuntimeError("gaussianavatar_amd.optim.Adam: dense float32 CUDA parameters only")
                if g.stride() != p.stride() and not _same_layout(g, p):      # (tuple compare first: this runs per tensor and step)
                    g = torch.empty_like(p, memory_format=torch.preserve_format).copy_(g)
                    p.grad = g
                if stream is None:
                    stream = _native.raw_stream(p.device)
                    skip = self._skip_flag(p.device)
                row = table[n]
                row.param, row.exp_avg, row.exp_avg_sq, row.n = fixed
                row.grad = g.data_ptr()
                row.lr, row.bias_correction1, row.bias_correction2 = lr, bc1, bc2
                n += 1
                if n == 64:
                    _native.ganet_check(lib.ganet_adam_step(n, table, *args, skip, stream))
                    n = 0
        if n:
            _native.ganet_check(lib.ganet_adam_step(n, table, *flush_args, skip, stream))
        if skip is not None:          # the flag belonged to this step
            _native.ganet_check(lib.ganet_flag_clear(skip, stream))
        return loss

    def _verify(self, p, st):
        """First step of a parameter (or after it / its state moved): type and layout checks, moments created with the
        parameter's layout; returns the row's fixed part."""
        if not p.is_cuda or p.dtype != torch.float32:
            raise RuntimeError("gaussianavatar_amd.optim.Adam: dense float32 CUDA parameters only")
        if "exp_avg" not in st:
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        if not (p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))):
            raise RuntimeError("gaussianavatar_amd.optim.Adam: parameters must be dense (contiguous or channels-last)")
        for key in ("exp_avg", "exp_avg_sq"):          # (a checkpoint written with another layout)
            if not _same_layout(st[key], p):
                st[key] = torch.empty_like(p, memory_format=torch.preserve_format).copy_(st[key])
        st["_row"] = (p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel())
        return st["_row"]

    def state_dict(self):
        sd = super().state_dict()
        for g in sd["param_groups"]:
            g.pop("_shared_step", None)
        # one independent `step` tensor per parameter, as torch writes them
        for st in sd["state"].values():
            st.pop("_row", None)
            if "step" in st:
                st["step"] = st["step"].detach().clone()
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        for st in self.state.values():
            st.pop("_row", None)
        for group in self.param_groups:
            group.pop("_shared_step", None)
        for st in self.state.values():
            if "step" in st and torch.is_tensor(st["step"]):
                st["step"] = st["step"].detach().to("cpu", torch.float32)
