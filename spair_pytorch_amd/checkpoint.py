"""Checkpoints.

* Model weights are the reference's ``state_dict`` (same keys and shapes, train.py:85-90), so a file written by the reference's
  ``torch.save(spair_net.state_dict(), path)`` loads here and the ``model`` entry written here loads there.
* The reference never saves optimizer state (a resumed run restarts Adam's moments and its step count); here the fused optimizer's
  moments and step are saved as well, and can be exchanged with ``torch.optim.Adam`` (per-parameter views of the flat buffers in
  ``model.parameters()`` order).
"""
import torch


def save_checkpoint(path, model, optimizer=None, iteration=0, extra=None):
    ck = {"model": {k: v.detach().cpu() for k, v in model.state_dict().items()}, "iteration": int(iteration)}
    if optimizer is not None:
        sd = optimizer.state_dict()
        ck["optimizer"] = {k: (v.detach().cpu() if torch.is_tensor(v) else v) for k, v in sd.items()}
    if extra:
        ck["extra"] = extra
    torch.save(ck, path)


def load_checkpoint(path, model, optimizer=None, map_location="cpu"):
    """Returns the iteration to resume from (0 for a bare reference ``state_dict`` file)."""
    ck = torch.load(path, map_location=map_location)
    if "model" not in ck or not isinstance(ck["model"], dict):      # a bare state_dict, as the reference saves it
        model.load_state_dict(ck)
        return 0
    model.load_state_dict(ck["model"])
    if optimizer is not None and "optimizer" in ck:
        optimizer.load_state_dict(ck["optimizer"])
    return int(ck.get("iteration", 0))


def adam_state_to_torch(model, optimizer):
    """The fused optimizer's state in ``torch.optim.Adam(model.parameters()).state_dict()`` form."""
    flat = model.flat_parameters()
    state, idx = {}, []
    base = flat.data_ptr()
    for i, p in enumerate(model.parameters()):
        off = (p.data_ptr() - base) // 4
        n = p.numel()
        state[i] = {"step": torch.tensor(float(optimizer.step_count)),
                    "exp_avg": optimizer.exp_avg[off:off + n].view_as(p).clone(),
                    "exp_avg_sq": optimizer.exp_avg_sq[off:off + n].view_as(p).clone()}
        idx.append(i)
    group = {"lr": optimizer.lr, "betas": tuple(optimizer.betas), "eps": optimizer.eps, "weight_decay": 0, "amsgrad": False,
             "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
             "decoupled_weight_decay": False, "params": idx}
    return {"state": state, "param_groups": [group]}


def adam_state_from_torch(model, optimizer, sd):
    """Load a ``torch.optim.Adam`` state_dict (parameters in ``model.parameters()`` order) into the fused optimizer."""
    flat = optimizer._state()
    base = flat.data_ptr()
    steps = set()
    for i, p in enumerate(model.parameters()):
        st = sd["state"].get(i)
        if st is None:
            continue
        off = (p.data_ptr() - base) // 4
        n = p.numel()
        optimizer.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1))
        optimizer.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
        steps.add(int(float(st["step"])))
    if steps:       # parameters that never received a gradient (the reference's dead attention block) lag behind in torch.optim.Adam
        optimizer.step_count = max(steps)
    g = sd["param_groups"][0]
    optimizer.lr, optimizer.betas, optimizer.eps = g["lr"], tuple(g["betas"]), g["eps"]
