"""Differentiable ``Flow.forward``: what ``loss.backward()`` needs in the reference's training loop (agent.py:75-92).

The reference lets autograd trace every PyTorch op of every layer.  Here the forward is the fused HIP stack kernel (which
also saves the rotation entering each layer) and the backward is ONE launch of ``flow_train_backward_kernel``
(csrc/train_kernels.h) that recomputes each layer from its saved input and applies hand-derived reverse-mode formulas
(csrc/so3_grad.h).  This module is only the glue: it hands the layers' parameter tensors to ``torch.autograd.Function`` so
that optimizers, ``zero_grad`` and parameterisations built from tiny host-side tensor ops (LU, SVD) keep working unchanged.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib, runtime

ORTHOGONAL_FLAG = 1 << 8        # train_desc kind flag: the 4x4 matrix is orthogonal, its layer reports ldj = 0 (flow/rottrans.py:21)


class TrainPlan:
    """Layer table of the plain parameter blob for one flow (include/rnf_hip.h "training")."""

    def __init__(self, layers, perm_rows, segments, feat_dim):
        L = _lib.lib()
        self.tensor_counts = []
        desc = np.zeros((len(layers), 3), dtype=np.int32)
        off = 0
        for i, layer in enumerate(layers):
            kind = layer._rnf_kind
            desc[i] = (kind | (ORTHOGONAL_FLAG if getattr(layer, "_rnf_orthogonal", False) else 0), perm_rows[i], off)
            off += L.rnf_plain_layer_floats(kind, segments, feat_dim)
        self.desc = np.ascontiguousarray(desc)
        self.total = off
        self.segments = segments
        self.feat_dim = feat_dim


def train_tensors(layers):
    """Flat list of the tensors that make up the plain blob, in blob order, each differentiable w.r.t. the module parameters."""
    out = []
    for layer in layers:
        fn = getattr(layer, "_rnf_train_tensors", None)
        if fn is None:
            raise NotImplementedError(f"{type(layer).__name__} has no backward kernel yet (training path)")
        out.extend(fn())
    return out


def mlp_train_tensors(net):
    """ConditionalTransform parameters in plain-blob order (flow/condition.py:14-22)."""
    ts = [net.fc_first.weight, net.fc_first.bias]
    for j in (1, 3, 5):
        ts += [net.layers[j].weight, net.layers[j].bias]
    return ts + [net.fc_last.weight, net.fc_last.bias]


class _FlowForwardFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, packed, plan, rotation, feature, *tensors):
        rot, feat = runtime._check_inputs(rotation, feature, packed)
        n = rot.shape[0]
        dev = rot.device
        L = _lib.lib()
        out_rot = torch.empty_like(rot)
        out_ldj = torch.empty(n, dtype=torch.float32, device=dev)
        states = torch.empty((packed.n_layers, n, 9), dtype=torch.float32, device=dev)
        if n:
            ws = runtime.workspace(dev, L.rnf_workspace_bytes(n, packed.n_cond))
            with torch.cuda.device(dev):
                stream = torch.cuda.current_stream(dev).cuda_stream
                _lib.check(L.rnf_flow_forward_train(rot.data_ptr(), feat.data_ptr() if feat is not None else None, n,
                                                    packed.feat_padded, packed.blob.data_ptr(), packed.desc.ctypes.data,
                                                    packed.n_layers, packed.segments, out_rot.data_ptr(), out_ldj.data_ptr(),
                                                    states.data_ptr(), ws.data_ptr(), ws.numel(), stream))
        plain = torch.cat([t.detach().to(device=dev, dtype=torch.float32).reshape(-1) for t in tensors]) if tensors else \
            torch.zeros(0, device=dev)
        if plain.numel() != plan.total:
            raise RuntimeError(f"plain parameter blob has {plain.numel()} floats, layer table expects {plan.total}")
        feat_plain = None
        if packed.n_cond:
            feat_plain = feature.reshape(n, -1).to(device=dev, dtype=torch.float32).contiguous()
        ctx.packed, ctx.plan = packed, plan
        ctx.rot_shape = rotation.shape
        ctx.feat_shape = feature.shape if feature is not None else None
        ctx.shapes = [(t.shape, t.device, t.dtype) for t in tensors]
        ctx.save_for_backward(states, feat_plain, plain)
        return out_rot.reshape(rotation.shape), out_ldj

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_rot, g_ldj):
        states, feat_plain, plain = ctx.saved_tensors
        packed, plan = ctx.packed, ctx.plan
        n = states.shape[1]
        dev = states.device
        L = _lib.lib()
        grads = torch.zeros_like(plain)
        g_rot_in = torch.zeros((n, 9), dtype=torch.float32, device=dev)
        want_gfeat = feat_plain is not None and ctx.needs_input_grad[3]
        g_feat = torch.zeros_like(feat_plain) if want_gfeat else None
        scratch = torch.zeros(max(packed.n_layers, 1), dtype=torch.float32, device=dev)
        if n:
            g_rot_c = g_rot.reshape(n, 9).to(torch.float32).contiguous() if g_rot is not None else None
            g_ldj_c = (g_ldj.to(torch.float32).contiguous() if g_ldj is not None
                       else torch.zeros(n, dtype=torch.float32, device=dev))
            ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
            with torch.cuda.device(dev):
                stream = torch.cuda.current_stream(dev).cuda_stream
                _lib.check(L.rnf_flow_backward(states.data_ptr(), ptr(feat_plain), n, plan.feat_dim, plain.data_ptr(),
                                               plan.desc.ctypes.data, packed.n_layers, plan.segments, ptr(g_rot_c),
                                               g_ldj_c.data_ptr(), grads.data_ptr(), g_rot_in.data_ptr(), ptr(g_feat),
                                               scratch.data_ptr(), stream))
        outs = []
        off = 0
        for i, (shape, device, dtype) in enumerate(ctx.shapes):
            cnt = int(np.prod(shape)) if len(shape) else 1
            g = None
            if ctx.needs_input_grad[4 + i]:
                g = grads[off: off + cnt].reshape(shape).to(device=device, dtype=dtype)
            outs.append(g)
            off += cnt
        g_rotation = g_rot_in.reshape(ctx.rot_shape) if ctx.needs_input_grad[2] else None
        g_feature = g_feat.reshape(ctx.feat_shape) if want_gfeat else None
        return (None, None, g_rotation, g_feature, *outs)


def needs_grad(module, rotation, feature) -> bool:
    if not torch.is_grad_enabled():
        return False
    return rotation.requires_grad or (feature is not None and feature.requires_grad) or any(
        p.requires_grad for p in module.parameters())


def flow_forward(module, layers, perm_rows, packed, rotation, feature):
    """Differentiable (rotation', ldj) for a stack of layers; called by runtime.run_flow when a gradient is required."""
    if packed.segments > 64 and any(l._rnf_kind == runtime.KIND_MOBIUS for l in layers):
        raise NotImplementedError("training path: at most 64 segments")
    plan = TrainPlan(layers, perm_rows, packed.segments, packed.feat_dim)
    tensors = train_tensors(layers)
    return _FlowForwardFn.apply(packed, plan, rotation, feature, *tensors)
