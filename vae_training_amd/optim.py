"""flax.optim.Adam stand-in (vae.py:113, networks.py:100): Adam(learning_rate).create(model) ->
Optimizer with .target, .state and .apply_gradient(grad); the update itself is libvaek's
fused Adam kernel (beta1 0.9, beta2 0.999, eps 1e-8, no weight decay)."""
from __future__ import annotations

import torch


class _AdamState:
    def __init__(self, n_params, device):
        self.step = 0                                   # host mirror of step_dev (no sync needed)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=device)
        self.m = torch.zeros(n_params, dtype=torch.float32, device=device)          # grad_ema
        self.v = torch.zeros(n_params, dtype=torch.float32, device=device)          # grad_sq_ema
        self.grads = torch.zeros(n_params + 4, dtype=torch.float32, device=device)  # + loss, Dkl, mse, 0


class Adam:
    def __init__(self, learning_rate=None, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
        if (beta1, beta2, eps, weight_decay) != (0.9, 0.999, 1e-8, 0.0):
            raise NotImplementedError("libvaek's Adam kernel implements flax.optim.Adam's defaults only")
        self.learning_rate = float(learning_rate)

    def create(self, target, exchange=None, global_batch=0):
        return Optimizer(self, target, _AdamState(target.module.n_params, target.flat.device), exchange, global_batch)


class Optimizer:
    def __init__(self, optimizer_def, target, state, exchange=None, global_batch=0):
        self.optimizer_def, self.target, self.state = optimizer_def, target, state
        self.exchange, self.global_batch = exchange, global_batch

    def _rebound(self, target):
        return Optimizer(self.optimizer_def, target, self.state, self.exchange, self.global_batch)

    def apply_gradient(self, grad):
        """grad: flat tensor (P floats) or a nested dict shaped like target.params."""
        from . import layout
        eng = self.target.module.engine(1)
        if isinstance(grad, dict):
            flat = torch.zeros_like(self.target.flat)
            from .networks import _copy_tree
            _copy_tree(grad, layout.views(flat, self.target.module.leaves))
            grad = flat
        self.state.step += 1
        self.state.step_dev.fill_(self.state.step)
        eng.adam_step(self.target.flat, grad, self.state.m, self.state.v, self.optimizer_def.learning_rate,
                      step_dev=self.state.step_dev)
        from .networks import Model
        return self._rebound(Model(self.target.module, None, _flat=self.target.flat))

    # ---- flax.serialization.to_state_dict / from_state_dict twins (model.py:85-89, :37-43) ----
    def state_dict(self):
        """Layout ASSUMED-FROM-API of pre-Linen flax.serialization.to_state_dict(optimizer):
        {'target': {'params': tree}, 'state': {'step', 'param_states': tree of {grad_ema, grad_sq_ema}}}."""
        from . import layout
        lv = self.target.module.leaves
        to_np = lambda tree: {k: (to_np(v) if isinstance(v, dict) else v.detach().cpu().numpy()) for k, v in tree.items()}
        pm, pv = to_np(layout.views(self.state.m, lv)), to_np(layout.views(self.state.v, lv))
        zip_tree = lambda a, b: {k: (zip_tree(a[k], b[k]) if isinstance(a[k], dict) else {"grad_ema": a[k], "grad_sq_ema": b[k]}) for k in a}
        return {"target": {"params": to_np(self.target.params)},
                "state": {"step": int(self.state.step), "param_states": zip_tree(pm, pv)}}

    def load_state_dict(self, sd):
        from . import layout
        from .networks import _copy_tree
        lv = self.target.module.leaves
        _validate_checkpoint(sd, self.target.params)
        _copy_tree(sd["target"]["params"], self.target.params)
        ps = sd["state"]["param_states"]
        pick = lambda tree, key: {k: (pick(v, key) if "grad_ema" not in v else v[key]) for k, v in tree.items()}
        _copy_tree(pick(ps, "grad_ema"), layout.views(self.state.m, lv))
        _copy_tree(pick(ps, "grad_sq_ema"), layout.views(self.state.v, lv))
        self.state.step = int(sd["state"]["step"])
        self.state.step_dev.fill_(self.state.step)
        return self


def _validate_checkpoint(sd, params):
    """Key set and leaf shapes of a checkpoint against this model's parameter tree, with a readable error (a mismatched
    architecture would otherwise fail deep inside the copy into the flat views)."""
    import numpy as np
    try:
        tree, ps, step = sd["target"]["params"], sd["state"]["param_states"], sd["state"]["step"]
    except (KeyError, TypeError) as e:
        raise ValueError("not a checkpoint written by vae_training_amd (expected {'target': {'params'}, 'state': {'step', "
                         "'param_states'}})") from e
    if not isinstance(step, (int, np.integer)) or step < 0:
        raise ValueError(f"checkpoint step {step!r} is not a non-negative integer")

    def walk(want, got, moments, path):
        if set(want) != set(got) or set(want) != set(moments):
            raise ValueError(f"checkpoint does not match this architecture at {path or '/'}: has {sorted(got)}, model has {sorted(want)}")
        for k, v in want.items():
            if isinstance(v, dict):
                walk(v, got[k], moments[k], f"{path}/{k}")
                continue
            leaves = [("params", got[k])] + [(n, moments[k].get(n) if isinstance(moments[k], dict) else None)
                                            for n in ("grad_ema", "grad_sq_ema")]
            for what, a in leaves:
                shape = tuple(np.shape(a)) if a is not None else None
                if shape != tuple(v.shape):
                    raise ValueError(f"checkpoint leaf {path}/{k} ({what}) has shape {shape}, model expects {tuple(v.shape)}")
    walk(params, tree, ps, "")
