"""Fused Adam (K9) over the model's flat parameter / gradient buffers: one launch per step.
Numerically torch.optim.Adam(lr) with its defaults (train.py:44).

Guarded (spair_adam_guarded): a step whose forward flagged a non-finite loss is left out whole -- parameters and moments untouched -- and
an element whose gradient is NaN / inf on its own; ``lr * NaN`` never reaches a parameter.  ``skipped()`` reports both (synchronises),
``model.forward`` raises on the flag by itself once the failed step has completed (models.py)."""
import ctypes

import torch

from . import _lib as L


class FusedAdam:
    def __init__(self, model, lr=1e-4, betas=(0.9, 0.999), eps=1e-8):
        self.model, self.lr, self.betas, self.eps = model, lr, betas, eps
        self.step_count = 0
        self._state_for = None
        self._counters = None       # device ints: [whole steps left out (non-finite loss), 1 if any element was ever left out (non-finite gradient)]

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
        if self._counters is None or self._counters.device != flat.device:
            self._counters = torch.zeros(2, dtype=torch.int32, device=flat.device)
        return flat

    def zero_grad(self, set_to_none=False):
        self.model.flat_gradients().zero_()

    def step(self):
        flat = self._state()
        self.model._bind_grads()
        self.step_count += 1
        status = getattr(self.model, "_status_dev", None)
        skip = ctypes.c_void_p(status.data_ptr() + 4) if status is not None else ctypes.c_void_p(0)       # this step's bits
        L.check(L.lib().spair_adam_guarded(L.ptr(flat), L.ptr(self.model.flat_gradients()), L.ptr(self.exp_avg), L.ptr(self.exp_avg_sq),
                                           ctypes.c_int64(flat.numel()), ctypes.c_float(self.lr), ctypes.c_float(self.betas[0]),
                                           ctypes.c_float(self.betas[1]), ctypes.c_float(self.eps), int(self.step_count), skip,
                                           L.ptr(self._counters), L.stream()),
                "spair_adam_guarded")

    def skipped(self):
        """(steps left out because their loss was non-finite, whether any single element was ever left out for a non-finite gradient).
        SYNCHRONISES.  A left-out step still advanced ``step_count`` (the bias correction runs one step ahead per skip: 1e-3 relative
        on the update after a thousand steps)."""
        self._state()
        c = self._counters.tolist()
        return int(c[0]), bool(c[1])

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
