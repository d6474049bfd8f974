"""Fused Adam (K9) over the model's flat parameter / gradient buffers: one launch per step.
Numerically torch.optim.Adam(lr) with its defaults (train.py:44)."""
import ctypes

import torch

from . import _lib as L


class FusedAdam:
    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8):
        self.model, self.lr, self.betas, self.eps = model, lr, betas, eps
        self.step_count = 0
        self._state_for = None

    def _state(self):
        flat = self.model.flat_parameters()
        if self._state_for != flat.data_ptr():
            # (re)flattened parameters (first use, model.to(), load_state_dict(assign=True)): the moments follow the buffer -- copied when
            # the layout is unchanged, so that step_count and the bias correction stay consistent with them
            old_m, old_v = getattr(self, "exp_avg", None), getattr(self, "exp_avg_sq", None)
            self.exp_avg = torch.zeros_like(flat)
            self.exp_avg_sq = torch.zeros_like(flat)
            if old_m is not None and old_m.numel() == flat.numel():
                self.exp_avg.copy_(old_m.to(flat.device))
                self.exp_avg_sq.copy_(old_v.to(flat.device))
            elif old_m is not None:
                self.step_count = 0
            self._state_for = flat.data_ptr()
        return flat

    def zero_grad(self, set_to_none=False):
        self.model.flat_gradients().zero_()

    def step(self):
        flat = self._state()
        self.model._bind_grads()
        self.step_count += 1
        L.check(L.lib().spair_adam(L.ptr(flat), L.ptr(self.model.flat_gradients()), L.ptr(self.exp_avg), L.ptr(self.exp_avg_sq),
                                   ctypes.c_int64(flat.numel()), ctypes.c_float(self.lr), ctypes.c_float(self.betas[0]),
                                   ctypes.c_float(self.betas[1]), ctypes.c_float(self.eps), int(self.step_count), L.stream()),
                "spair_adam")

    def state_dict(self):
        self._state()
        return dict(step=self.step_count, exp_avg=self.exp_avg, exp_avg_sq=self.exp_avg_sq, lr=self.lr, betas=self.betas, eps=self.eps)

    def load_state_dict(self, sd):
        self._state()
        self.step_count = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.lr = float(sd.get("lr", self.lr))
        self.betas = tuple(sd.get("betas", self.betas))
        self.eps = float(sd.get("eps", self.eps))
