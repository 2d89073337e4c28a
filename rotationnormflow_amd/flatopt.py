"""Optimizer-state bridge of a flattened flow (``Flow.flatten_parameters``): checkpoints keep the reference's per-tensor layout.

The reference's drivers build ``optim.Adam(self.flow.parameters(), lr)`` (agent.py:23), write ``optimizer_flow.state_dict()`` into every
checkpoint (agent.py:143) and -- in ``Agent.load_ckpt``, which ``eval.py:23``, ``eval_uncondition.py:21``, ``train.py:49`` and
``train_uncondition.py:28`` all call -- rebuild the optimizer and ``load_state_dict`` the saved state (agent.py:193-196).  That state has
one entry per parameter TENSOR (264 for the 24-layer flow).  A flattened flow hands the optimizer ONE parameter, so without a bridge the
unedited ``load_ckpt`` raises on a checkpoint the authors published, and ``save_ckpt`` writes a state the reference cannot resume.

The bridge: the flat parameter carries its per-tensor layout (``tag``); every ``torch.optim.Optimizer`` constructed over a tagged parameter
gets two per-instance hooks (``Optimizer.register_state_dict_post_hook`` / ``register_load_state_dict_pre_hook``, torch's own extension
points):

  * ``state_dict()``       -> the per-tensor layout: per-element state (``exp_avg``, ``exp_avg_sq``, ``max_exp_avg_sq`` ...) sliced into the
                              tensors' shapes, per-parameter scalars (``step``) repeated, ``params`` renumbered, other parameters untouched;
  * ``load_state_dict(s)`` -> accepts BOTH layouts: a per-tensor state (the reference's, or one this bridge wrote) is concatenated back
                              into the one-entry form; a state already in the optimizer's own layout passes through.

Adam is elementwise with one ``step`` per group member, so stepping the flat tensor and stepping its 264 slices are the same arithmetic:
the round trip is exact.  torch has no construction-time hook for optimizers, so ``install()`` wraps ``torch.optim.Optimizer.__init__`` once
per process (the original runs first, unchanged; optimizers without a tagged parameter are not touched).  ``attach(optimizer)`` does the same
by hand for a parameter added later with ``add_param_group``.
"""
import functools
import threading

import torch

_TAG = "_rnf_flat_layout"
_ATTACHED = "_rnf_flat_bridge"
_lock = threading.Lock()
_installed = False


def tag(param, layout):
    """``layout``: [(state-dict key, offset, shape)] of the per-tensor slices of ``param`` in the reference's ``parameters()`` order."""
    setattr(param, _TAG, tuple((str(k), int(off), tuple(int(d) for d in shape)) for k, off, shape in layout))


def layout_of(param):
    return getattr(param, _TAG, None)


def _numel(shape):
    n = 1
    for d in shape:
        n *= d
    return n


def _is_per_element(v, total):
    return torch.is_tensor(v) and v.dim() >= 1 and v.numel() == total


def expand_state(param_groups, state):
    """Optimizer ``state_dict()`` in the optimizer's own layout -> the per-tensor layout.  ``param_groups``: the optimizer's live groups
    (their ``params`` are the Parameter objects, in the order ``state_dict()`` numbers them)."""
    if not any(layout_of(p) for g in param_groups for p in g["params"]):
        return state
    saved_groups = state["param_groups"]
    if len(saved_groups) != len(param_groups) or any(len(s["params"]) != len(g["params"]) for s, g in zip(saved_groups, param_groups)):
        return state                                                   # not this optimizer's own layout (already per tensor): leave it
    out_state, out_groups, nxt = {}, [], 0
    for sg, g in zip(saved_groups, param_groups):
        ids, names = [], []
        have_names = "param_names" in sg
        for j, (old, p) in enumerate(zip(sg["params"], g["params"])):
            lay = layout_of(p)
            st = state["state"].get(old)
            if lay is None:
                if st is not None:
                    out_state[nxt] = st
                ids.append(nxt)
                if have_names:
                    names.append(sg["param_names"][j])
                nxt += 1
                continue
            total = p.numel()
            prefix = sg["param_names"][j][:-len("_flat")] if have_names and sg["param_names"][j].endswith("_flat") else ""
            for key, off, shape in lay:
                n = _numel(shape)
                if st:
                    out_state[nxt] = {k: (v.reshape(-1)[off:off + n].reshape(shape).clone() if _is_per_element(v, total) else
                                          (v.clone() if torch.is_tensor(v) else v)) for k, v in st.items()}
                ids.append(nxt)
                if have_names:
                    names.append(prefix + key)
                nxt += 1
        ng = dict(sg, params=ids)
        if have_names:
            ng["param_names"] = names
        out_groups.append(ng)
    return {"state": out_state, "param_groups": out_groups}


def flatten_state(param_groups, state):
    """The inverse, for ``load_state_dict``: a per-tensor state -> the layout of an optimizer whose ``param_groups`` hold flat parameters.
    A state already in the optimizer's own layout is returned as it is; anything else is left for torch to refuse."""
    if not any(layout_of(p) for g in param_groups for p in g["params"]):
        return state
    saved_groups = state["param_groups"]
    if len(saved_groups) != len(param_groups):
        return state
    own = [len(g["params"]) for g in param_groups]
    per_tensor = [sum(len(layout_of(p) or (None,)) for p in g["params"]) for g in param_groups]
    saved = [len(s["params"]) for s in saved_groups]
    if saved == own or saved != per_tensor:
        return state
    out_state, out_groups, nxt = {}, [], 0
    for sg, g in zip(saved_groups, param_groups):
        ids, names, at = [], [], 0
        have_names = "param_names" in sg
        for p in g["params"]:
            lay = layout_of(p)
            if lay is None:
                old = sg["params"][at]
                if old in state["state"]:
                    out_state[nxt] = state["state"][old]
                if have_names:
                    names.append(sg["param_names"][at])
                at += 1
            else:
                olds = sg["params"][at:at + len(lay)]
                parts = [state["state"].get(o) for o in olds]
                if any(s is not None for s in parts):
                    if any(s is None for s in parts):
                        raise ValueError("optimizer state covers only some tensors of the flattened flow")
                    merged = {}
                    for k, v in parts[0].items():
                        if torch.is_tensor(v) and tuple(v.shape) == lay[0][2] and v.dim() >= 1:
                            for s, (key, _, shape) in zip(parts, lay):
                                if tuple(s[k].shape) != shape:
                                    raise ValueError(f"optimizer state '{k}' of {key} has shape {tuple(s[k].shape)}, the flow's tensor {shape}")
                            # (the blob is in the kernels' order, the entries in parameters() order: every slice goes to ITS offset)
                            blob = torch.empty(p.numel(), dtype=v.dtype, device=v.device)
                            for s, (_, off, shape) in zip(parts, lay):
                                blob[off:off + _numel(shape)] = s[k].reshape(-1)
                            merged[k] = blob
                        else:
                            if torch.is_tensor(v) and any(not torch.equal(s[k].cpu(), v.cpu()) for s in parts[1:]):
                                raise ValueError(f"optimizer state '{k}' differs between the tensors of one flow (per-tensor {k} cannot be "
                                                 "carried by a flattened parameter): load it with RNF_FLAT_PARAMS=0")
                            merged[k] = v.clone() if torch.is_tensor(v) else v
                    out_state[nxt] = merged
                if have_names:
                    first = sg["param_names"][at]
                    names.append(first[:len(first) - len(lay[0][0])] + "_flat" if first.endswith(lay[0][0]) else "_flat")
                at += len(lay)
            ids.append(nxt)
            nxt += 1
        ng = dict(sg, params=ids)
        if have_names:
            ng["param_names"] = names
        out_groups.append(ng)
    return {"state": out_state, "param_groups": out_groups}


def _post_state_dict(optimizer, state):
    return expand_state(optimizer.param_groups, state)


def _pre_load_state_dict(optimizer, state):
    return flatten_state(optimizer.param_groups, state)


def attach(optimizer) -> bool:
    """Register the two hooks on ``optimizer`` when one of its parameters is a tagged flat parameter (once per optimizer)."""
    if optimizer.__dict__.get(_ATTACHED):
        return True
    if not any(layout_of(p) for g in optimizer.param_groups for p in g["params"]):
        return False
    optimizer.register_state_dict_post_hook(_post_state_dict)
    optimizer.register_load_state_dict_pre_hook(_pre_load_state_dict)
    optimizer.__dict__[_ATTACHED] = True
    return True


def install():
    """Wrap ``torch.optim.Optimizer.__init__`` (once): after the original constructor, ``attach`` the optimizer.  Called by
    ``Flow.flatten_parameters``; a process that never flattens a flow never gets here."""
    global _installed
    with _lock:
        if _installed:
            return
        original = torch.optim.Optimizer.__init__

        @functools.wraps(original)
        def __init__(self, *args, **kwargs):
            original(self, *args, **kwargs)
            attach(self)
        __init__._rnf_original = original
        torch.optim.Optimizer.__init__ = __init__
        _installed = True


def uninstall():
    """Undo ``install`` (tests)."""
    global _installed
    with _lock:
        cur = torch.optim.Optimizer.__init__
        if _installed and hasattr(cur, "_rnf_original"):
            torch.optim.Optimizer.__init__ = cur._rnf_original
        _installed = False
