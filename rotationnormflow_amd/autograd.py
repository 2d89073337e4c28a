"""Differentiable ``Flow.forward``: what ``loss.backward()`` needs in the reference's training loop (agent.py:75-92).

The reference lets autograd trace every PyTorch op of every layer.  Here the forward is one HIP launch that also saves the rotation
entering each layer -- for the reference's batch sizes the 16-rotation kernel of csrc/train_block16.h straight from the parameters (it
keeps the conditioner activations too), otherwise the packed stack kernel -- and the backward is ONE launch of the reverse sweep
(csrc/train_block16.h below 6144 rotations, csrc/train_kernels.h above) applying hand-derived reverse-mode formulas (csrc/so3_grad.h)
to the saved activations or to a recompute from the saved inputs.  This module is only the glue: it hands the layers' parameter tensors
to ``torch.autograd.Function`` so that optimizers, ``zero_grad`` and parameterisations built from tiny host-side tensor ops (LU, SVD) keep
working unchanged.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib, runtime

ORTHOGONAL_FLAG = 1 << 8        # layer-table kind flag: the 4x4 matrix is orthogonal, its layer reports ldj = 0 (flow/rottrans.py:21)


class TrainPlan:
    """Everything about one flow that depends only on its structure: the three layer tables (kernel blob, device packer,
    backward sweep) and the sizes of the two blobs (include/rnf_hip.h "training")."""

    def __init__(self, layers, perm_rows, device, precision):
        L = _lib.lib()
        n = len(layers)
        shapes = [layer._rnf_shape() for layer in layers]          # (kind, segments | 0, feature_dim | 0)
        ks = {k for kind, k, f in shapes if kind == runtime.KIND_MOBIUS}
        fs = {f for kind, k, f in shapes if f}
        if len(ks) > 1:
            raise ValueError("all Moebius layers of one flow must have the same number of segments")
        if len(fs) > 1:
            raise ValueError("all conditional layers of one flow must share feature_dim")
        self.segments = ks.pop() if ks else 8
        self.feat_dim = fs.pop() if fs else 0
        self.feat_padded = runtime.pad8(self.feat_dim)
        if self.segments > 512:
            raise NotImplementedError("training path: at most 512 segments (the backward kernel keeps the conditioner outputs of a "
                                      "16-rotation block in LDS)")
        self.precision = precision
        self.prec = runtime._PRECISIONS[precision]
        self.n_layers = n
        self.desc = np.zeros((n, runtime.DESC_STRIDE), dtype=np.int32)       # kernel blob table (runtime.pack_layers)
        self.pack_desc = np.zeros((n, 4), dtype=np.int32)                    # rnf_pack_flow_device
        self.train_desc = np.zeros((n, 3), dtype=np.int32)                   # rnf_flow_backward
        plain_off = rec_off = 0
        slot = 0
        side_slot = 0
        rec_sizes = []
        for i, (layer, (kind, k, f)) in enumerate(zip(layers, shapes)):
            flagged = kind | (ORTHOGONAL_FLAG if getattr(layer, "_rnf_orthogonal", False) else 0)
            self.desc[i] = (kind, perm_rows[i], rec_off, -1, -1, self.prec, -1, -1)     # no fp32 fallback images in training
            self.pack_desc[i] = (flagged, plain_off, rec_off, -1)
            self.train_desc[i] = (flagged, perm_rows[i], plain_off)
            if kind in runtime.SIDE_KINDS:                 # per-sample matrices from the caller: an empty record, no plain parameters;
                self.desc[i, 2] = side_slot                # the kernels find the layer's slot in the side buffer here
                self.train_desc[i, 0] = flagged | (side_slot << 16)
                side_slot += 1
                rec_sizes.append(4)
                rec_off += 4
                continue
            if kind == runtime.KIND_MOBIUS:
                size = L.rnf_mobius_packed_floats(self.segments)
            elif kind == runtime.KIND_COND16 or kind in runtime.COND9_KINDS:
                size = L.rnf_cond16_packed_floats()                # Condition9*: the same record with a 9-row fc_last
            elif kind == runtime.KIND_GS9:
                size = L.rnf_gs_packed_floats(3)
            elif kind == runtime.KIND_GS36:
                size = L.rnf_gs_packed_floats(6)
            elif kind == runtime.KIND_COND36:
                size = L.rnf_cond36_packed_floats()
            else:
                size = L.rnf_affine16_packed_floats()
            rec_sizes.append(size)
            rec_off += (size + 3) // 4 * 4
            plain_off += L.rnf_plain_layer_floats(kind, self.segments, self.feat_dim)
            if f:
                self.desc[i, 3] = slot
                slot += 1
        fsize = L.rnf_featproj_packed_floats(self.feat_padded) if self.feat_dim else 0
        for i, (kind, k, f) in enumerate(shapes):
            if f and kind not in runtime.SIDE_KINDS:
                self.desc[i, 4] = self.pack_desc[i, 3] = rec_off
                rec_off += (fsize + 3) // 4 * 4
        self.n_cond = slot
        self.n_side = side_slot
        self.n_mlp = sum(1 for kind, k, f in shapes if kind in (runtime.KIND_MOBIUS, runtime.KIND_COND16))
        self.plain_floats = plain_off
        self.blob_floats = max(rec_off, 4)
        self.desc = np.ascontiguousarray(self.desc)
        self.train_desc_inverse = np.ascontiguousarray(self.train_desc[::-1])    # rnf_flow_inverse_backward: iteration order of the inverse pass
        # packer status word: checked one call later through pinned memory, so that no step waits for the device
        self.flags = torch.zeros(1, dtype=torch.int32, device=device)
        self.flags_host = torch.zeros(1, dtype=torch.int32).pin_memory() if torch.cuda.is_available() else torch.zeros(1, dtype=torch.int32)
        self.flags_event = None
        self.feature_ms = None

    # batches below this many rotations take the 16-rotation training forward (measured crossover against the fused stack kernel:
    # profiles/README.md "Training"); RNF_TRAIN_FORWARD=stack|block16 forces one of them
    PLAIN_FORWARD_BELOW = 12288

    def plain_forward(self, n):
        """Whether a training forward of ``n`` rotations runs from the plain blob on 16-rotation workgroups (rnf_flow_forward_train_plain):
        flows of Moebius / constant 4x4 / Condition16Trans layers, at most 200 layers."""
        import os
        mode = os.environ.get("RNF_TRAIN_FORWARD", "")
        if mode == "stack" or self.n_side or self.n_layers > 200 or self.n_layers < 1:
            return False
        kinds = set(int(k) & 15 for k in self.train_desc[:, 0])
        if not kinds <= {runtime.KIND_MOBIUS, runtime.KIND_AFFINE16, runtime.KIND_COND16}:
            return False
        return mode == "block16" or n < self.PLAIN_FORWARD_BELOW

    # below this many rotations the plain forward also saves the conditioner activations (2 KB per rotation and layer at K = 64) and the
    # 16-rotation backward sweep reads them back instead of recomputing them (measured, profiles/README.md "Training": unconditional recipe
    # -5 % at 1024, even at 2048, +3 % at 4096; conditional F = 256, whose recompute includes the feature product: -18 % at 1024, -4 % at
    # 4096 -- up to where the 16-rotation sweep is used at all); RNF_TRAIN_ACTS=0|1 forces
    ACTS_BELOW = 2048
    ACTS_BELOW_CONDITIONAL = 6144

    def save_activations(self, n):
        import os
        mode = os.environ.get("RNF_TRAIN_ACTS", "")
        if mode == "0":
            return False
        return mode == "1" or n < (self.ACTS_BELOW_CONDITIONAL if self.n_cond else self.ACTS_BELOW)

    def calibrate(self, feature):
        """Mean square of the feature entries, measured ONCE per plan on the first batch it sees (a training run keeps its feature
        distribution): the data-dependent input of the device packer's equalisation (csrc/equalize.h)."""
        if self.feature_ms is None and feature is not None and not torch.cuda.is_current_stream_capturing():
            self.feature_ms = runtime.calibration_ms(feature)             # 1 unless runtime.set_feature_calibration("first-batch")

    def check_flags(self):
        if self.flags_event is not None and self.flags_event.query():
            bits = int(self.flags_host[0])
            self.flags_event = None
            if bits & 1:
                raise runtime.HalfRangeError("a weight left the fp16 range during training (the split-precision kernels produced inf/NaN "
                                             "for that step); continue with rotationnormflow_amd.set_precision('fp32')")
            if bits & 2:
                raise RuntimeError("a 4x4 affine matrix became singular during training")

    def pack(self, plain, stream, with_fallback=False):
        """plain blob -> fresh kernel blob, on the device (one launch).  ``with_fallback`` (inference from live parameters: training-mode
        modules under no_grad, nn.DataParallel replicas): a second launch appends the exact-fp32 images of the same layers, which the
        range guard of the split-precision kernels re-runs a call on (``desc_with_fallback``)."""
        L = _lib.lib()
        capturing = torch.cuda.is_current_stream_capturing()      # inside a HIP graph capture: no event queries, no pinned read-back
        if not capturing:
            self.check_flags()
        both = with_fallback and self.prec == _lib.PREC_F16X2
        blob = torch.empty(self.blob_floats * (2 if both else 1), dtype=torch.float32, device=plain.device)
        self.flags.zero_()
        if plain.numel() == 0:                             # a stack of side layers only: nothing to read, but the pointer must be valid
            plain = torch.zeros(4, dtype=torch.float32, device=blob.device)
        old_ms = L.rnf_set_feature_ms(self.feature_ms or 1.0)
        try:
            _lib.check(L.rnf_pack_flow_device(plain.data_ptr(), self.pack_desc.ctypes.data, self.n_layers, self.segments, self.feat_dim,
                                              self.prec, blob.data_ptr(), self.flags.data_ptr(), stream))
            if both:
                _lib.check(L.rnf_pack_flow_device(plain.data_ptr(), self.pack_desc.ctypes.data, self.n_layers, self.segments, self.feat_dim,
                                                  _lib.PREC_FP32, blob.data_ptr() + 4 * self.blob_floats, self.flags.data_ptr(), stream))
        finally:
            L.rnf_set_feature_ms(old_ms)
        if self.flags_event is None and not capturing:
            self.flags_host.copy_(self.flags, non_blocking=True)
            self.flags_event = torch.cuda.Event()
            self.flags_event.record()
        return blob


def desc_with_fallback(plan):
    """Layer table of a blob packed by ``plan.pack(..., with_fallback=True)``: columns 6 / 7 point at the exact-fp32 images behind the
    split-precision ones (include/rnf_hip.h); side layers keep -1."""
    desc = plan.desc.copy()
    if plan.prec != _lib.PREC_F16X2:
        return desc
    for i in range(plan.n_layers):
        if runtime.kind_has_mlp(int(desc[i, 0])):
            desc[i, 6] = plan.blob_floats + desc[i, 2]
            if desc[i, 4] >= 0:
                desc[i, 7] = plan.blob_floats + desc[i, 4]
    return np.ascontiguousarray(desc)


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
    """ConditionalTransform parameters in plain-blob order (flow/condition.py:14-22).  Read through the modules' own dictionaries on
    every call (a replaced Parameter or sub-module is seen; nn.Module.__getattr__ / Sequential.__getitem__ cost 0.4 ms per training step
    over the 24 conditioners of the reference's recipe, the dictionaries 0.1 ms)."""
    mods = net._modules
    seq = mods["layers"]._modules
    out = []
    for lin in (mods["fc_first"], seq["1"], seq["3"], seq["5"], mods["fc_last"]):
        prm = lin._parameters
        w, b = prm.get("weight"), prm.get("bias")
        if w is None or b is None:                         # a parametrised / pruned Linear computes the attribute
            w, b = lin.weight, lin.bias
        out.append(w)
        out.append(b)
    return out


class _FlowFn(torch.autograd.Function):
    """Differentiable ``Flow.forward`` (direction 0) and ``Flow.inverse`` (direction 1; flow/flow.py:53-92).  The stack kernel saves the
    rotation entering every layer (iteration position); the backward is one launch of the reverse sweep, which for the inverse pass walks
    the layers in the order that pass visited them, with MobiusFlow.inverse differentiated by BinFind.backward's implicit-function rule
    (flow/mobiusflow.py:247-273).  ``side``: [n_side, n, 16] per-sample matrices of the side layers (built differentiably by the caller
    from the HIP conditioner's outputs with the reference's own tensor ops) or None; its gradient comes back from the sweep."""

    @staticmethod
    def forward(ctx, plan, grad_sync, direction, rotation, feature, side, *tensors):
        rot, feat = runtime._check_inputs(rotation, feature, plan)
        n = rot.shape[0]
        dev = rot.device
        L = _lib.lib()
        out_rot = torch.empty_like(rot)
        out_ldj = torch.empty(n, dtype=torch.float32, device=dev)
        states = torch.empty((plan.n_layers, n, 9), dtype=torch.float32, device=dev)
        f32 = torch.float32                               # (grad mode is off inside Function.forward: no detach needed)
        if len(tensors) == 1 and tensors[0].is_cuda and tensors[0].dtype is f32 and tensors[0].is_contiguous():
            plain = tensors[0].reshape(-1)                # a flattened flow (Flow.flatten_parameters): the parameter storage, in place
        else:
            plain = torch.cat([t.reshape(-1) if (t.is_cuda and t.dtype is f32) else t.to(device=dev, dtype=f32).reshape(-1)
                               for t in tensors]) if tensors else torch.zeros(0, device=dev)
        if plain.numel() != plan.plain_floats:
            raise RuntimeError(f"plain parameter blob has {plain.numel()} floats, layer table expects {plan.plain_floats}")
        side_c = None
        if plan.n_side:
            if side is None or tuple(side.shape) != (plan.n_side, n, 16):
                raise RuntimeError("side-layer matrices are missing or mis-shaped")
            side_c = side.to(device=dev, dtype=f32).contiguous()
        feat_plain = None
        if plan.feat_dim and feature is not None:
            feat_plain = feature.reshape(n, plan.feat_dim).to(device=dev, dtype=torch.float32).contiguous()
        acts = None
        if n and direction == 0 and plan.plain_forward(n):
            # small batches: the forward on 16-rotation workgroups straight from the plain blob (csrc/train_block16.h), no packing launch;
            # below ACTS_BELOW rotations it also leaves the conditioner activations for the backward sweep (no recompute there)
            if plan.save_activations(n):
                acts = torch.empty(L.rnf_train_acts_floats(n, plan.n_mlp, plan.segments), dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                stream = torch.cuda.current_stream(dev).cuda_stream
                _lib.check(L.rnf_flow_forward_train_plain(rot.data_ptr(), feat_plain.data_ptr() if (feat_plain is not None and plan.n_cond) else None,
                                                          n, plan.feat_dim if plan.n_cond else 0, plain.data_ptr(), plan.train_desc.ctypes.data,
                                                          plan.n_layers, plan.segments, out_rot.data_ptr(), out_ldj.data_ptr(),
                                                          states.data_ptr(), acts.data_ptr() if acts is not None else None, stream))
        elif n:
            ws = runtime.workspace(dev, L.rnf_workspace_bytes_segments(n, plan.n_cond, plan.segments))     # (+ the K > 128 stash of an inverse pass)
            fptr = feat.data_ptr() if feat is not None else None
            with torch.cuda.device(dev):
                stream = torch.cuda.current_stream(dev).cuda_stream
                if plan.feat_dim:
                    plan.calibrate(feat)
                blob = plan.pack(plain, stream)
                if plan.n_side:
                    _lib.check(L.rnf_flow_train_side(direction, rot.data_ptr(), fptr, n, plan.feat_padded, side_c.data_ptr(), blob.data_ptr(),
                                                     plan.desc.ctypes.data, plan.n_layers, plan.segments, out_rot.data_ptr(),
                                                     out_ldj.data_ptr(), states.data_ptr(), ws.data_ptr(), ws.numel(), stream))
                else:
                    fn = L.rnf_flow_inverse_train if direction else L.rnf_flow_forward_train
                    _lib.check(fn(rot.data_ptr(), fptr, n, plan.feat_padded, blob.data_ptr(), plan.desc.ctypes.data, plan.n_layers,
                                  plan.segments, out_rot.data_ptr(), out_ldj.data_ptr(), states.data_ptr(), ws.data_ptr(), ws.numel(), stream))
        ctx.plan = plan
        ctx.grad_sync = grad_sync
        ctx.direction = direction
        ctx.rot_shape = rotation.shape
        ctx.feat_shape = feature.shape if feature is not None else None
        ctx.shapes = [(t.shape, t.device, t.dtype) for t in tensors]
        ctx.sizes = [t.numel() for t in tensors]
        ctx.save_for_backward(states, out_rot if direction else None, feat_plain, plain, side_c, acts)
        return out_rot.reshape(rotation.shape), out_ldj

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_rot, g_ldj):
        states, out_rot, feat_plain, plain, side_c, acts = ctx.saved_tensors
        plan = ctx.plan
        n = states.shape[1]
        dev = states.device
        L = _lib.lib()
        FIRST = 6                                         # index of the first parameter tensor among forward's arguments
        want_w = any(ctx.needs_input_grad[FIRST:])        # no parameter requires grad: input gradients only, weight products skipped
        grads = torch.zeros_like(plain) if want_w else None
        g_rot_in = torch.zeros((n, 9), dtype=torch.float32, device=dev)
        want_gfeat = feat_plain is not None and plan.n_cond > 0 and ctx.needs_input_grad[4]
        g_feat = torch.zeros_like(feat_plain) if want_gfeat else None
        g_side = torch.zeros_like(side_c) if side_c is not None else None
        scratch = torch.zeros(max(plan.n_layers, 1), dtype=torch.float32, device=dev)
        if n:
            g_rot_c = g_rot.reshape(n, 9).to(torch.float32).contiguous() if g_rot is not None else None
            g_ldj_c = (g_ldj.to(torch.float32).contiguous() if g_ldj is not None
                       else torch.zeros(n, dtype=torch.float32, device=dev))
            ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
            tdesc = plan.train_desc_inverse if ctx.direction else plan.train_desc
            with torch.cuda.device(dev):
                stream = torch.cuda.current_stream(dev).cuda_stream
                if plan.n_side:
                    _lib.check(L.rnf_flow_backward_side(ctx.direction, states.data_ptr(), ptr(out_rot), ptr(feat_plain) if plan.n_cond else None, n,
                                                        plan.feat_dim if plan.n_cond else 0, ptr(plain) if plain.numel() else None,
                                                        tdesc.ctypes.data, plan.n_layers, plan.segments, side_c.data_ptr(), g_side.data_ptr(),
                                                        ptr(g_rot_c), g_ldj_c.data_ptr(), ptr(grads), g_rot_in.data_ptr(), ptr(g_feat),
                                                        scratch.data_ptr(), stream))
                elif ctx.direction:
                    _lib.check(L.rnf_flow_inverse_backward(states.data_ptr(), out_rot.data_ptr(), ptr(feat_plain), n, plan.feat_dim,
                                                           plain.data_ptr(), tdesc.ctypes.data, plan.n_layers, plan.segments,
                                                           ptr(g_rot_c), g_ldj_c.data_ptr(), ptr(grads), g_rot_in.data_ptr(), ptr(g_feat),
                                                           scratch.data_ptr(), stream))
                elif acts is not None:
                    _lib.check(L.rnf_flow_backward_saved(states.data_ptr(), acts.data_ptr(), ptr(feat_plain) if plan.n_cond else None, n,
                                                         plan.feat_dim if plan.n_cond else 0, plain.data_ptr(), tdesc.ctypes.data,
                                                         plan.n_layers, plan.segments, ptr(g_rot_c), g_ldj_c.data_ptr(), ptr(grads),
                                                         g_rot_in.data_ptr(), ptr(g_feat), scratch.data_ptr(), stream))
                else:
                    _lib.check(L.rnf_flow_backward(states.data_ptr(), ptr(feat_plain), n, plan.feat_dim, plain.data_ptr(),
                                                   tdesc.ctypes.data, plan.n_layers, plan.segments, ptr(g_rot_c),
                                                   g_ldj_c.data_ptr(), ptr(grads), g_rot_in.data_ptr(), ptr(g_feat),
                                                   scratch.data_ptr(), stream))
        needs = ctx.needs_input_grad
        if want_w:
            runtime.note_training_step()                  # an optimizer step follows: host-packed blobs are stale from now on
            if ctx.grad_sync is not None:                 # data-parallel training: ONE collective for every parameter gradient
                ctx.grad_sync(grads)
        pieces = torch.split(grads, ctx.sizes) if (ctx.sizes and want_w) else ()
        outs = []
        for i, (shape, device, dtype) in enumerate(ctx.shapes):
            g = None
            if want_w and needs[FIRST + i]:
                g = pieces[i].view(shape)
                if device != dev or dtype is not torch.float32:
                    g = g.to(device=device, dtype=dtype)
            outs.append(g)
        g_rotation = g_rot_in.reshape(ctx.rot_shape) if needs[3] else None
        g_feature = g_feat.reshape(ctx.feat_shape) if want_gfeat else None
        return (None, None, None, g_rotation, g_feature, g_side if needs[5] else None, *outs)


# packer status words of _CondMLPFn calls: read back through pinned memory and checked at the next call (no step waits for the device)
_pending_mlp_flags = []


def _check_mlp_flags():
    keep = []
    for ev, host in _pending_mlp_flags:
        if not ev.query():
            keep.append((ev, host))
        elif int(host[0]) & 1:
            _pending_mlp_flags[:] = keep
            raise runtime.HalfRangeError("a weight of a side layer's conditioner left the fp16 range during training (that step produced "
                                         "inf/NaN); continue with rotationnormflow_amd.set_precision('fp32')")
    _pending_mlp_flags[:] = keep


class _CondMLPFn(torch.autograd.Function):
    """One ConditionalTransform(F -> n_out <= 16) on the GPU, differentiable: forward = device packer + rnf_cond_mlp_forward, backward =
    rnf_cond_mlp_backward (the training backward kernel with the layer math replaced by dL/d(outputs)).  The networks of the side layers
    (ConditionLU's three, ConditionRot's one) go through it when a gradient is required."""

    @staticmethod
    def forward(ctx, feature, n_out, *tensors):
        L = _lib.lib()
        dev = feature.device
        f32 = torch.float32
        feat = feature.to(f32).contiguous()
        n, F = feat.shape
        ts = [t.to(device=dev, dtype=f32) for t in tensors]
        plain = torch.cat([t.reshape(-1) for t in ts])
        wl, bl = ts[-2], ts[-1]                                # pad fc_last to the 16 rows of a Condition16Trans record
        plain16 = torch.cat([t.reshape(-1) for t in ts[:-2]] + [wl.reshape(-1), wl.new_zeros((16 - n_out) * 64), bl, bl.new_zeros(16 - n_out)])
        Fp = runtime.pad8(F)
        prec = runtime._PRECISIONS[runtime.device_precision()]
        rec = (L.rnf_cond16_packed_floats() + 3) // 4 * 4
        blob = torch.empty(rec + L.rnf_featproj_packed_floats(Fp), dtype=f32, device=dev)
        pack_desc = np.array([[runtime.KIND_COND16, 0, 0, rec]], dtype=np.int32)
        capturing = torch.cuda.is_current_stream_capturing()
        if not capturing:
            _check_mlp_flags()
        flags = torch.zeros(1, dtype=torch.int32, device=dev)
        out = torch.empty((n, 16), dtype=f32, device=dev)
        fpad = torch.nn.functional.pad(feat, (0, Fp - F)) if Fp != F else feat
        if n:
            ws = runtime.workspace(dev, L.rnf_workspace_bytes(n, 1))
            with torch.cuda.device(dev):
                stream = torch.cuda.current_stream(dev).cuda_stream
                _lib.check(L.rnf_pack_flow_device(plain16.data_ptr(), pack_desc.ctypes.data, 1, 8, F, prec, blob.data_ptr(), flags.data_ptr(), stream))
                _lib.check(L.rnf_cond_mlp_forward(fpad.data_ptr(), n, Fp, blob.data_ptr(), 0, rec, prec, out.data_ptr(), ws.data_ptr(), ws.numel(),
                                                  stream))
                if prec and not capturing and len(_pending_mlp_flags) < 64:
                    host = torch.zeros(1, dtype=torch.int32).pin_memory()
                    host.copy_(flags, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record()
                    _pending_mlp_flags.append((ev, host))
        ctx.n_out = n_out
        ctx.shapes = [(t.shape, t.device, t.dtype) for t in tensors]
        ctx.sizes = [t.numel() for t in tensors]
        ctx.feat_meta = (feature.shape, feature.dtype)
        ctx.save_for_backward(feat, plain)
        return out[:, :n_out].contiguous()

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g_out):
        feat, plain = ctx.saved_tensors
        L = _lib.lib()
        dev = feat.device
        n, F = feat.shape
        want_w = any(ctx.needs_input_grad[2:])
        grads = torch.zeros_like(plain) if want_w else None
        g_feat = torch.zeros_like(feat) if ctx.needs_input_grad[0] else None
        g = g_out.to(torch.float32).contiguous()
        scratch = torch.zeros(1, dtype=torch.float32, device=dev)
        if n:
            ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
            with torch.cuda.device(dev):
                _lib.check(L.rnf_cond_mlp_backward(feat.data_ptr(), n, F, plain.data_ptr(), ctx.n_out, g.data_ptr(), ptr(grads), ptr(g_feat),
                                                   scratch.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
        if want_w:
            runtime.note_training_step()
        pieces = torch.split(grads, ctx.sizes) if want_w else ()
        outs = []
        for i, (shape, device, dtype) in enumerate(ctx.shapes):
            gi = None
            if want_w and ctx.needs_input_grad[2 + i]:
                gi = pieces[i].view(shape)
                if device != dev or dtype is not torch.float32:
                    gi = gi.to(device=device, dtype=dtype)
            outs.append(gi)
        shape, dtype = ctx.feat_meta
        return (g_feat.reshape(shape).to(dtype) if g_feat is not None else None, None, *outs)


def cond_mlp(net, feature, n_out):
    """Differentiable ConditionalTransform(feature) -> [n, n_out] through the HIP conditioner."""
    return _CondMLPFn.apply(feature, n_out, *mlp_train_tensors(net))


def _side_tensor(plan, layers, rotation, feature, inverse):
    """[n_side, n, 16] matrices of the flow's side layers, built with autograd on (each layer's ``_rnf_side(feature, grad=True)``)."""
    if not plan.n_side:
        return None
    if feature is None:
        raise AssertionError("The input feature is needed in this module")
    n = rotation.reshape(-1, 9).shape[0]
    feat = feature.reshape(n, -1).to(device=rotation.device, dtype=torch.float32)
    mats = []
    for layer in layers:
        if layer._rnf_shape()[0] in runtime.SIDE_KINDS:
            m = layer._rnf_side(feat, grad=True)
            mats.append(torch.nn.functional.pad(m, (0, 16 - m.shape[1])) if m.shape[1] < 16 else m)
    return torch.stack(mats)


def _plan_for(module, layers, perm_rows, rotation):
    """(``rotation`` only supplies the device.)"""
    if not rotation.is_cuda:
        raise RuntimeError("rotationnormflow_amd runs on the GPU only (HIP kernels, no CPU fallback): got a CPU tensor")
    for layer in layers:
        if not hasattr(layer, "_rnf_train_tensors"):
            raise NotImplementedError(f"{type(layer).__name__} has no backward kernel yet (training path); evaluate it under torch.no_grad()")
    key = (str(rotation.device), runtime.device_precision(), tuple(perm_rows), tuple(l._rnf_shape() for l in layers))
    cached = getattr(module, "_rnf_train_plan", None)
    if cached is None or cached[0] != key:
        cached = (key, TrainPlan(layers, perm_rows, rotation.device, runtime.device_precision()))
        module._rnf_train_plan = cached
    fixed = getattr(module, "_feature_ms_fixed", None)          # Flow.set_feature_scale / dist.calibrate_feature_scale: one value for all ranks
    if fixed is not None:
        cached[1].feature_ms = fixed
    return cached[1]


def _parameter_inputs(module, layers):
    """What _FlowFn differentiates with respect to: the ONE flat parameter of a flattened flow (Flow.flatten_parameters), else every
    training tensor of every layer (264 for the reference's raw.yml recipe)."""
    flat = getattr(module, "_parameters", {}).get("_flat")
    return [flat] if flat is not None else train_tensors(layers)


def flow_inverse(module, layers, perm_rows, rotation, feature):
    """Differentiable ``Flow.inverse``.  ``layers`` / ``perm_rows`` in FLOW order; the inverse pass walks them back to front."""
    plan = _plan_for(module, layers, perm_rows, rotation)
    side = _side_tensor(plan, layers, rotation, feature, True)
    return _FlowFn.apply(plan, getattr(module, "_rnf_grad_sync", None), 1, rotation, feature, side, *_parameter_inputs(module, layers))


def flow_forward(module, layers, perm_rows, rotation, feature):
    """Differentiable (rotation', ldj) for a stack of layers; called by runtime.run_flow when a gradient is required."""
    plan = _plan_for(module, layers, perm_rows, rotation)
    side = _side_tensor(plan, layers, rotation, feature, False)
    return _FlowFn.apply(plan, getattr(module, "_rnf_grad_sync", None), 0, rotation, feature, side, *_parameter_inputs(module, layers))
