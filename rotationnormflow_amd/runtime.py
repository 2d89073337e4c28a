"""Host runtime: pack module parameters into the kernel blob, keep it fresh, launch through the C ABI.

PyTorch is used here only as plumbing: device memory (tensors), the current HIP stream, and parameter storage.
"""
from __future__ import annotations

import math
import os
import threading

import numpy as np
import torch

from . import _lib

KIND_MOBIUS, KIND_AFFINE16, KIND_COND16, KIND_GS9, KIND_GS36 = 1, 2, 3, 4, 5
KIND_COND9_GS, KIND_COND9_SMITH, KIND_COND9_POLAR_L, KIND_COND9_POLAR_R, KIND_COND36 = 6, 7, 8, 9, 10
COND9_KINDS = (KIND_COND9_GS, KIND_COND9_SMITH, KIND_COND9_POLAR_L, KIND_COND9_POLAR_R)
# layers whose per-sample matrix is built on the host side with the reference's own batched torch ops and handed to the kernels in a side
# buffer (include/rnf_hip.h RNF_LAYER_SIDE*): Condition16TransLU, ConditionRot, Condition9TransLU
KIND_SIDE16, KIND_SIDE16_ROT, KIND_SIDE9 = 11, 12, 13
SIDE_KINDS = (KIND_SIDE16, KIND_SIDE16_ROT, KIND_SIDE9)
DESC_STRIDE = 8            # include/rnf_hip.h RNF_DESC_STRIDE: kind, perm_row, param, cond_slot, feat, precision, fallback param, fallback feat

# Arithmetic of the conditioner GEMMs (include/rnf_hip.h RNF_PREC_*): "f16x2" = split-precision fp16 MFMA (22-bit
# operands, fp32 accumulate; default), "fp32" = exact fp32 MFMA, "bf16x3" (round 6) = every operand as three bf16 terms (24 bits, fp32's
# range: nothing to equalise, calibrate, audit or guard), six bf16 MFMAs per product-sum.  Environment override: RNF_PRECISION.
_PRECISIONS = {"fp32": _lib.PREC_FP32, "f16x2": _lib.PREC_F16X2, "bf16x3": _lib.PREC_BF16X3}
_precision = os.environ.get("RNF_PRECISION", "f16x2")


_fallback_precision = os.environ.get("RNF_FALLBACK", "bf16x3")       # arithmetic of the guard's re-run images of host-packed f16x2 flows


def device_precision() -> str:
    """The arithmetic of the paths whose kernel images are built ON THE DEVICE (training passes, nn.DataParallel replicas, the side layers'
    conditioners): "bf16x3" images exist for host-packed flows only, those paths then run the exact-fp32 kernels."""
    return "fp32" if _precision == "bf16x3" else _precision


def set_precision(name: str):
    """Select the arithmetic used for flows packed from now on ("f16x2", "bf16x3" or "fp32")."""
    global _precision
    if name not in _PRECISIONS:
        raise ValueError(f"precision must be one of {sorted(_PRECISIONS)}")
    _precision = name


def get_precision() -> str:
    return _precision

# flow/flow.py:13-15 -- the 6x3 permutation table; row m is the cyclic shift (m, m+1, m+2) mod 3
PERMUTE_ROWS = ((0, 1, 2), (1, 2, 0), (2, 0, 1), (0, 1, 2), (1, 2, 0), (2, 0, 1))


_prefetched = threading.local()      # id(parameter) -> host fp32 copy, filled by pack_layers for the duration of one packing


def _np32(t) -> np.ndarray:
    hit = getattr(_prefetched, "map", {}).get(id(t))
    if hit is not None:
        return hit
    return np.ascontiguousarray(t.detach().to("cpu", torch.float32).numpy())


def _prefetch_parameters(layers):
    """One device->host copy for all parameters of the stack instead of one per tensor (a training step repacks every iteration)."""
    # (a flattened flow -- Flow.flatten_parameters -- keeps its per-layer tensors as buffers that are views of one parameter)
    params = [p for layer in layers for p in list(layer.parameters()) + [b for b in layer.buffers() if b.is_floating_point()]]
    dev = [p for p in params if p.is_cuda]
    if len(dev) < 2:
        return {}
    flat = torch.cat([p.detach().reshape(-1).to(torch.float32) for p in dev]).cpu().numpy()
    out, off = {}, 0
    for p in dev:
        n = p.numel()
        out[id(p)] = flat[off: off + n].reshape(tuple(p.shape))
        off += n
    return out


def _pad_cols(w: np.ndarray, first: int, feat: int, feat_padded: int) -> np.ndarray:
    """[64, first+feat] -> [64, first+feat_padded] with zero columns appended (feature_dim not a multiple of 8)."""
    if feat == feat_padded:
        return w
    out = np.zeros((w.shape[0], first + feat_padded), dtype=np.float32)
    out[:, : first + feat] = w
    return out


def pad8(f: int) -> int:
    return (f + 7) // 8 * 8


class PackedFlow:
    """Device blob + host layer table for one (module list, device, parameter version)."""

    def __init__(self, blob, desc, n_cond, feat_dim, feat_padded, segments, precision="fp32"):
        self.precision = precision            # arithmetic the MLP images were packed for
        self.blob = blob                      # torch.float32 [P] on the device
        self.desc = desc                      # np.int32 [L, 6] (host, C-contiguous; passed by pointer)
        self.n_layers = desc.shape[0]
        self.n_cond = n_cond
        self.feat_dim = feat_dim
        self.feat_padded = feat_padded
        self.segments = segments
        self.side_layers = []                 # layer modules with _rnf_side(feature) -> [n, 16], in side-slot order
        self.feature_ms = 1.0                 # mean square of a feature entry the conditional layers were equalised for
        self.audit = 0.0                      # host packers: worst pack-time audit value of the split-precision images


class HalfRangeError(RuntimeError):
    """A weight does not fit the fp16 range: the flow must be packed for the exact fp32 kernels instead."""


def quantise_feature_ms(ms: float) -> float:
    """The calibration value the packers see: the measured mean square rounded to 1/16 of a binade.  The equalisation only takes
    power-of-two decisions from it (csrc/equalize.h), so nothing is lost, and two measurements of the same data that differ by summation
    order (one GPU over the whole batch / the all-reduced sums of N shards) land on the SAME value -- identical packed images on every rank."""
    if not (1e-20 < ms < 1e20):
        return 1.0
    return float(2.0 ** (round(math.log2(ms) * 16.0) / 16.0))


_SQ_CHUNK = 1 << 24                   # entries per reduction step of feature_square_sum (64 MB of fp32: its fp64 transient is 128 MB)


def feature_square_sum(feature) -> torch.Tensor:
    """float64 [2] = {sum of squares, count} of a feature batch on its device (no host sync): what ranks all-reduce to agree on one
    calibration (dist.calibrate_feature_scale)."""
    f = feature.detach().reshape(-1)
    # Accumulated in fp64 in bounded chunks.  (Round 5 called torch.linalg.vector_norm(f, 2, dtype=float64), which casts its WHOLE input to
    # fp64 first -- ADVICE r5: a 2 GB feature batch, C5's 2^20 x 512, allocated 4 GB of transient HBM at every pack / calibration.)
    ss = torch.zeros((), dtype=torch.float64, device=f.device)
    for chunk in f.split(_SQ_CHUNK):
        ss += chunk.to(torch.float64).square_().sum()
    return torch.stack((ss, torch.tensor(float(f.numel()), dtype=torch.float64, device=f.device)))


def feature_mean_square(feature) -> float:
    """Mean square of the entries of a feature batch: the one data-dependent input of the packers' equalisation (csrc/equalize.h).
    One device reduction + one scalar read-back; called when a parameter version is packed (which copies every parameter to the host
    anyway), never per evaluation."""
    if feature is None or feature.numel() == 0:
        return 1.0
    sq = feature_square_sum(feature)
    return quantise_feature_ms(float(sq[0] / sq[1]))


# ---- where the calibration input of the equalisation comes from ------------------------------------------------------------------------------
# "weights" (default since round 6): nothing is measured -- m_f = 1 (features of unit scale, the regime of normalised backbone features)
#     unless the caller fixed a value (Flow.set_feature_scale / Flow.calibrate_feature_scale / dist.calibrate_feature_scale) or the
#     checkpoint came with a sidecar (harness.build_flow_from_checkpoint reads ``<ckpt>.rnf.json``).  The packed images -- and with them every
#     rotation's result, bit for bit -- are a function of the weights and that one explicit number: two processes that see different first
#     batches agree.  Features far from the assumed scale are caught by the launch guard (exact-fp32 re-run: correct, ~3x slower) and a
#     RuntimeWarning names the call that fixes it.
# "first-batch" (rounds 3 - 5): m_f is measured on the first feature batch a parameter version is packed for, and measured again when the
#     guard keeps firing (GuardWatch).  Best accuracy without any call, but a process-local, data-dependent state.
_feature_calibration = os.environ.get("RNF_FEATURE_CALIBRATION", "weights")


def set_feature_calibration(mode: str) -> None:
    global _feature_calibration
    if mode not in ("weights", "first-batch"):
        raise ValueError(f"unknown feature calibration mode {mode!r} (weights, first-batch)")
    _feature_calibration = mode


def get_feature_calibration() -> str:
    return _feature_calibration


def calibration_ms(feature) -> float:
    """The m_f a pack without an explicit value uses: 1 ("weights" mode), or the batch's measured mean square ("first-batch")."""
    return feature_mean_square(feature) if _feature_calibration == "first-batch" else 1.0


def expand_shared_rows(feature, n_rot: int, feature_repeat: int):
    """``feature`` holds n_rot / feature_repeat rows, row r conditioning rotations [r Q, (r + 1) Q): materialise the reference's
    ``feature.repeat`` (agent.py:240-244) for the paths that take one feature row per rotation."""
    if n_rot % feature_repeat:
        raise ValueError(f"{n_rot} rotations are not a multiple of feature_repeat={feature_repeat}")
    return feature.reshape(n_rot // feature_repeat, -1).repeat_interleave(feature_repeat, dim=0)


def pack_layers(layers, perm_rows, device, precision=None, feature_ms=1.0) -> PackedFlow:
    """layers: product layer modules (each has ``_rnf_kind`` and ``_rnf_pack``); perm_rows: forward permutation row per layer.
    precision None = the module-wide setting, falling back to "fp32" when a weight is outside the fp16 range (or the pack-time audit
    refuses the split-precision image).  ``feature_ms``: mean square of a feature entry the equalisation of conditional layers assumes."""
    if precision is None:
        try:
            return pack_layers(layers, perm_rows, device, _precision, feature_ms)
        except HalfRangeError:
            return pack_layers(layers, perm_rows, device, "fp32", feature_ms)
    prec = _PRECISIONS[precision]
    L = _lib.lib()
    _prefetched.map = _prefetch_parameters(layers)
    old_ms = L.rnf_set_feature_ms(float(feature_ms))
    try:
        _audit.worst = 0.0
        blob, desc, slot, feat_dim, segments = _pack_layers(layers, perm_rows, prec, L)
        audit = _audit.worst
        if precision == "f16x2" and _guard_fallback:
            # the same layers once more as STRICT images behind the split-precision ones: the library re-runs a call on them, on the device,
            # when its guard fires (an fp16 operand overflowed, features far from the packed scale, ...; include/rnf_hip.h desc columns 5 - 7).
            # Round 6: bf16x3 images (RNF_FALLBACK=fp32 keeps the exact fp32-input MFMA images of rounds 2 - 5): a fired guard costs 1.8x the
            # guarded time instead of 3.3x.
            fbp = _lib.PREC_FP32 if _fallback_precision == "fp32" else _lib.PREC_BF16X3
            blob32, desc32, _, _, _ = _pack_layers(layers, perm_rows, fbp, L)
            base = (blob.size + 3) // 4 * 4
            mlp = np.array([kind_has_mlp(int(k)) for k in desc[:, 0]])
            desc[mlp, 6] = base + desc32[mlp, 2]
            has_feat = desc32[:, 4] >= 0
            desc[has_feat, 7] = base + desc32[has_feat, 4]
            desc[mlp, 5] |= fbp << 8                      # bits 8..15 of the precision column: arithmetic of the fallback records
            blob = np.concatenate([blob, np.zeros(base - blob.size, np.float32), blob32])
    finally:
        _prefetched.map = {}
        L.rnf_set_feature_ms(old_ms)
    packed = PackedFlow(torch.from_numpy(blob).to(device), np.ascontiguousarray(desc), slot, feat_dim, pad8(feat_dim), segments, precision)
    packed.feature_ms = float(feature_ms)
    packed.audit = audit                     # worst conditioner-output error of the packed split-precision images on the probe inputs (0 for fp32 images)
    for i, layer in enumerate(layers):
        if layer._rnf_kind in SIDE_KINDS:
            packed.desc[i, 2] = len(packed.side_layers)          # param offset column = slot in the side buffer
            packed.desc[i, 6] = -1
            packed.side_layers.append(layer)
            if not feat_dim:
                packed.feat_dim = layer.feature_dim              # a flow whose only conditional layers are side layers
                packed.feat_padded = pad8(layer.feature_dim)
    return packed


_guard_fallback = os.environ.get("RNF_GUARD", "1") != "0"


def stamp_rootfinder_order(packed, order) -> None:
    """Bits 16..17 of desc column 5 on the Moebius layers (include/rnf_hip.h): first-pass order of the inverse root finder -- None the
    library's default, 3 or 4.  The descriptor is host memory handed over by pointer on every call, so this is all it takes."""
    code = {None: 0, 3: 1, 4: 2}[order]
    if getattr(packed, "rf_code", 0) != code:
        m = packed.desc[:, 0] == KIND_MOBIUS
        packed.desc[m, 5] = (packed.desc[m, 5] & ~(3 << 16)) | (code << 16)
        packed.rf_code = code


def kind_has_mlp(kind: int) -> bool:
    return kind in (KIND_MOBIUS, KIND_COND16, KIND_COND9_GS, KIND_COND9_SMITH, KIND_COND9_POLAR_L, KIND_COND9_POLAR_R, KIND_COND36)


def fallback_fired(device) -> bool:
    """True when the last guarded split-precision call on this (device, current stream) was re-run on the exact-fp32 kernels.
    Synchronises (diagnostics / tests only)."""
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    with _ws_lock:
        ws = _workspaces.get(key)
    if ws is None:
        return False
    return bool(ws[4095 * 8 + 4: 4095 * 8 + 8].view(torch.int32).item())


class GuardWatch:
    """Notices, without ever waiting for the device, when the range guard keeps re-running a flow's launches on the exact-fp32 kernels
    (correct, but ~3x slower, and silent -- ADVICE r3): after every guarded call of a conditional flow the guard word of the workspace is
    copied to pinned memory behind an event; a later call reads the words that have landed.  ``REPEAT`` consecutive fired calls warn once
    and ask the caller to re-calibrate (the usual cause: features far larger than the batch the equalisation was calibrated on, so that the
    projected x0 trips kX0Guard; Flow._packed then measures the feature scale again on the batch at hand and re-packs)."""
    REPEAT = 3
    MAX_RECALIBRATIONS = 2             # per watch: a guard that still fires after that is not a calibration problem

    def __init__(self):
        self.pending = False           # a copy of the guard word is in flight (one pinned word and one event, reused for every call)
        self.host = None
        self.event = None
        self.streak = 0
        self.warned = False
        self.recalibrations = 0
        self.device = None             # the device the pinned word / event belong to

    def __deepcopy__(self, memo):
        return GuardWatch()

    def __reduce__(self):
        return (GuardWatch, ())

    def poll(self) -> bool:
        """-> True when a re-calibration is due."""
        if self.pending and self.event.query():
            fired = bool(int(self.host[0]))
            self.pending = False
            self.streak = self.streak + 1 if fired else 0
            if self.streak >= self.REPEAT:
                self.streak = 0
                if self.recalibrations >= self.MAX_RECALIBRATIONS:
                    return False       # re-packing again would produce the same images: stay on the (correct) fp32 re-runs, quietly
                auto = _feature_calibration == "first-batch"
                if auto:
                    self.recalibrations += 1
                if not self.warned:
                    import warnings
                    warnings.warn("rotationnormflow_amd: the range guard re-ran several consecutive launches of this flow on the exact-fp32 "
                                  "kernels (features far from the scale the packed images were calibrated for, or an activation beyond "
                                  "the fp16 range): results are correct but ~3x slower; "
                                  + ("re-calibrating on the current batch -- Flow.set_feature_scale() fixes a scale"
                                     if auto else "call Flow.calibrate_feature_scale(features) once (or Flow.set_feature_scale / a checkpoint "
                                     "sidecar, harness.write_feature_scale) to pack the images for this data")
                                  + ", set_precision('fp32') avoids the guarded path", RuntimeWarning)
                    self.warned = True
                return auto            # "weights" mode: nothing is re-packed behind the caller's back
            if not fired:              # a quiet launch: the cap counts CONSECUTIVE failed re-calibrations, not a lifetime total
                self.recalibrations = 0
        return False

    def watch(self, ws):
        if self.pending or torch.cuda.is_current_stream_capturing():     # (still watching at the re-calibration cap: a quiet launch lifts it again, poll())
            return
        if self.device is not None and self.device != ws.device:
            return                     # one watch, one device (nn.DataParallel replicas get their own: Flow._replicate_for_data_parallel)
        self.device = ws.device
        if self.host is None:
            self.host = torch.zeros(1, dtype=torch.int32).pin_memory()
            self.event = torch.cuda.Event()
        self.host.copy_(ws[4095 * 8 + 4: 4095 * 8 + 8].view(torch.int32), non_blocking=True)
        self.event.record()
        self.pending = True


def _pack_layers(layers, perm_rows, prec, L):
    records, feat_records = [], []
    desc = np.zeros((len(layers), DESC_STRIDE), dtype=np.int32)
    feat_dim = 0
    segments = 8
    slot = 0
    for i, layer in enumerate(layers):
        kind = layer._rnf_kind
        rec, feat_rec, fdim, segs = layer._rnf_pack(L, prec)
        if segs:
            if segments != 8 and segs != segments and any(l._rnf_kind == KIND_MOBIUS for l in layers[:i]):
                raise ValueError("all Moebius layers of one flow must have the same number of segments")
            segments = segs
        desc[i, 0] = kind
        desc[i, 1] = perm_rows[i]
        desc[i, 3] = -1
        desc[i, 4] = -1
        desc[i, 5] = prec
        desc[i, 6] = -1
        desc[i, 7] = -1
        if feat_rec is not None:
            if feat_dim and fdim != feat_dim:
                raise ValueError("all conditional layers of one flow must share feature_dim")
            feat_dim = fdim
            desc[i, 3] = slot
            slot += 1
        records.append(rec)
        feat_records.append(feat_rec)
    off = 0
    for i, rec in enumerate(records):
        desc[i, 2] = off
        off += (rec.size + 3) // 4 * 4
    for i, rec in enumerate(feat_records):
        if rec is not None:
            desc[i, 4] = off
            off += (rec.size + 3) // 4 * 4
    blob = np.zeros(max((off + 3) // 4 * 4, 4), dtype=np.float32)
    for i, rec in enumerate(records):
        blob[desc[i, 2]: desc[i, 2] + rec.size] = rec
    for i, rec in enumerate(feat_records):
        if rec is not None:
            blob[desc[i, 4]: desc[i, 4] + rec.size] = rec
    return blob, desc, slot, feat_dim, segments


_audit = threading.local()           # worst pack-time audit value (rnf_last_pack_audit) of the split-precision images of the flow being packed


def _check_pack(L, rc):
    if rc == 2:
        raise HalfRangeError(L.rnf_last_error().decode())
    _lib.check(rc)
    _audit.worst = max(getattr(_audit, "worst", 0.0), float(L.rnf_last_pack_audit()))


# ---- per-layer packers (called by the layer modules) -----------------------------------------------------------
def pack_mobius(L, cond, K, feature_dim, prec=_lib.PREC_FP32):
    """cond: ConditionalTransform(3+F, 4K).  -> (layer record, feature-projection record | None)"""
    F = feature_dim
    Fp = pad8(F)
    if K <= 0:
        raise ValueError(f"segments={K} must be positive")
    rec = np.empty(L.rnf_mobius_packed_floats_prec(K, prec), dtype=np.float32)
    frec = np.empty(L.rnf_featproj_packed_floats(Fp), dtype=np.float32) if F else None
    arrs = [_pad_cols(_np32(cond.fc_first.weight), 3, F, Fp), _np32(cond.fc_first.bias)]
    for j in (1, 3, 5):
        arrs += [_np32(cond.layers[j].weight), _np32(cond.layers[j].bias)]
    arrs += [_np32(cond.fc_last.weight), _np32(cond.fc_last.bias)]
    _check_pack(L, L.rnf_pack_mobius(*[a.ctypes.data for a in arrs], K, Fp, prec, rec.ctypes.data,
                                     frec.ctypes.data if frec is not None else None))
    return rec, frec


def pack_cond16(L, net, feature_dim, prec=_lib.PREC_FP32, n_out=16):
    F = feature_dim
    Fp = pad8(F)
    rec = np.empty(L.rnf_cond_packed_floats_prec(n_out, prec), dtype=np.float32)
    frec = np.empty(L.rnf_featproj_packed_floats(Fp), dtype=np.float32)
    arrs = [_pad_cols(_np32(net.fc_first.weight), 0, F, Fp), _np32(net.fc_first.bias)]
    for j in (1, 3, 5):
        arrs += [_np32(net.layers[j].weight), _np32(net.layers[j].bias)]
    arrs += [_np32(net.fc_last.weight), _np32(net.fc_last.bias)]
    fn = {16: L.rnf_pack_cond16, 9: L.rnf_pack_cond9, 36: L.rnf_pack_cond36}[n_out]
    _check_pack(L, fn(*[a.ctypes.data for a in arrs], Fp, prec, rec.ctypes.data, frec.ctypes.data))
    return rec, frec


def pack_affine16(L, mat):
    rec = np.empty(L.rnf_affine16_packed_floats(), dtype=np.float32)
    m = _np32(mat).reshape(16)
    _lib.check(L.rnf_pack_affine16(m.ctypes.data, rec.ctypes.data))
    return rec


def pack_gs(L, mat, n):
    """Uncondition9Trans (n=3) / Uncondition36Trans (n=6): [M | M^-1]."""
    rec = np.empty(L.rnf_gs_packed_floats(n), dtype=np.float32)
    m = _np32(mat).reshape(n * n)
    _lib.check(L.rnf_pack_gs(m.ctypes.data, n, rec.ctypes.data))
    return rec


def pack_rot16(L, rot_mat):
    rec = np.empty(L.rnf_affine16_packed_floats(), dtype=np.float32)
    m = _np32(rot_mat).reshape(16)
    _lib.check(L.rnf_pack_rot16(m.ctypes.data, rec.ctypes.data))
    return rec


# ---- per-sample matrix layers ("side" layers) ---------------------------------------------------------------------------------------
class SideNet:
    """Evaluates one ConditionalTransform(F, <= 16 outputs) on the GPU through rnf_cond_mlp_forward (record = rnf_pack_cond16 with the
    unused fc_last rows zero), with its own version-keyed packed copy."""

    def __init__(self, net, feature_dim, n_out):
        self.net, self.feature_dim, self.n_out = net, feature_dim, n_out
        self.cache = PackCache()

    def _pack(self, device, feature_ms=1.0):
        L = _lib.lib()
        prec_name = device_precision()
        old_ms = L.rnf_set_feature_ms(float(feature_ms))
        try:
            for name in ([prec_name, "fp32"] if prec_name != "fp32" else ["fp32"]):
                try:
                    rec, frec = pack_cond16(L, _Padded16(self.net, self.n_out), self.feature_dim, _PRECISIONS[name])
                    break
                except HalfRangeError:
                    continue
        finally:
            L.rnf_set_feature_ms(old_ms)
        blob = np.concatenate([rec, np.zeros((-rec.size) % 4, np.float32), frec])
        return (torch.from_numpy(blob).to(device), (rec.size + 3) // 4 * 4, _PRECISIONS[name])

    def __call__(self, feature):
        """feature [n, F] (cuda, fp32) -> [n, n_out].  With autograd on and a parameter (or the feature) requiring grad: the differentiable
        route (autograd.cond_mlp: device packer + rnf_cond_mlp_forward, backward rnf_cond_mlp_backward)."""
        if torch.is_grad_enabled() and (feature.requires_grad or any(p.requires_grad for p in self.net.parameters())):
            from . import autograd
            return autograd.cond_mlp(self.net, feature, self.n_out)
        dev = feature.device
        blob, feat_off, prec = self.cache.get(self.net, dev, lambda: self._pack(dev, calibration_ms(feature)))
        L = _lib.lib()
        n = feature.shape[0]
        Fp = pad8(self.feature_dim)
        f = feature.to(torch.float32)
        if Fp != self.feature_dim:
            f = torch.nn.functional.pad(f, (0, Fp - self.feature_dim))
        f = f.contiguous()
        out = torch.empty((n, 16), dtype=torch.float32, device=dev)
        if n:
            ws = workspace(dev, L.rnf_workspace_bytes(n, 1))
            with torch.cuda.device(dev):
                _lib.check(L.rnf_cond_mlp_forward(f.data_ptr(), n, Fp, blob.data_ptr(), 0, feat_off, prec, out.data_ptr(), ws.data_ptr(), ws.numel(),
                                                  torch.cuda.current_stream(dev).cuda_stream))
        return out[:, : self.n_out]


class _Padded16:
    """View of a ConditionalTransform whose fc_last has fewer than 16 rows as one with 16 (zero rows appended): what rnf_pack_cond16 packs."""

    def __init__(self, net, n_out):
        self.fc_first, self.layers = net.fc_first, net.layers
        w, b = net.fc_last.weight.detach(), net.fc_last.bias.detach()
        self.fc_last = type("L", (), {})()
        self.fc_last.weight = torch.cat([w, w.new_zeros(16 - n_out, w.shape[1])]) if n_out < 16 else w
        self.fc_last.bias = torch.cat([b, b.new_zeros(16 - n_out)]) if n_out < 16 else b


def build_side_buffer(packed, feature, n, device, inverse):
    """[n_side_layers, n, 16] per-sample matrices of the flow's side layers for THIS batch (the matrices of ConditionLU depend on the
    first rows of the batch: flow/squeezetrans.py:127)."""
    if feature is None:
        raise AssertionError("The input feature is needed in this module")
    feat = feature.reshape(n, -1).to(device=device, dtype=torch.float32)
    side = torch.zeros((len(packed.side_layers), n, 16), dtype=torch.float32, device=device)
    for slot, layer in enumerate(packed.side_layers):
        m = layer._rnf_side(feat, grad=False)                    # [n, 16] or [n, 9]
        side[slot, :, : m.shape[1]] = m
    return side


# ---- parameter-version keyed cache -------------------------------------------------------------------------------
# Bumped by every backward pass through the training path (autograd.py).  Part of the cache key because fused optimizers
# (torch._fused_adam_ and friends) update parameters in place WITHOUT bumping ``Tensor._version``: after a training step every
# host-packed blob is rebuilt on next use, whatever the optimizer did.
_train_epoch = 0


def note_training_step():
    global _train_epoch
    _train_epoch += 1


def params_key(module, device, params=None):
    """Changes whenever any parameter is modified in place through autograd-visible ops (optimizer step, load_state_dict) or replaced.
    NOT by writes through ``.data`` (they do not bump ``Tensor._version``): see ``PackCache`` / ``Flow.invalidate``."""
    if params is None:
        params = list(module.parameters())
    return (str(device), _precision, _train_epoch) + tuple((id(p), p._version, p.data_ptr()) for p in params)


class PackCache:
    """Packed blobs of one module, one entry per device, keyed on the parameter versions; thread safe (nn.DataParallel replicas share
    this object and call from worker threads).  The module tree is walked once and kept as a flat list of ``(submodule._parameters,
    name)`` slots; every call checks that each slot still holds the SAME Parameter object (a flat loop of identity tests, no tree walk)
    and re-walks when one was replaced (``module.weight = nn.Parameter(...)``, parametrize / prune, conversions that swap parameter
    objects): the versions and addresses of the OLD objects never change, so without this check the stale blob would be served (ADVICE
    r2).  Flows register no parameters after construction.  ``invalidate()`` drops everything: the hook for parameter edits the key cannot
    see (``p.data.copy_()``, EMA swaps, fused optimizers stepped outside this library's backward)."""

    def __init__(self):
        self._lock = threading.Lock()
        self._entries = {}            # str(device) -> (key, packed)
        self._params = None
        self._slots = None

    def invalidate(self):
        with self._lock:
            self._entries.clear()
            self._params = None
            self._slots = None

    # a copied / unpickled module starts with an empty cache of its own (the lock and the device blobs do not travel)
    def __deepcopy__(self, memo):
        return PackCache()

    def __reduce__(self):
        return (PackCache, ())

    def _walk(self, module):
        self._slots = [(m._parameters, name) for m in module.modules() for name, p in m._parameters.items() if p is not None]
        # floating-point buffers too: the per-layer tensors of a flattened flow (Flow.flatten_parameters) are buffers that view one
        # parameter and share its version counter; a layer called on its own has no parameter left to key on
        self._slots += [(m._buffers, name) for m in module.modules() for name, b in m._buffers.items() if b is not None and b.is_floating_point()]
        self._params = [d[name] for d, name in self._slots]

    def _live(self):
        for (d, name), p in zip(self._slots, self._params):
            if d.get(name) is not p:
                return False
        return True

    def peek(self, device):
        """The packed flow cached for ``device`` (whatever parameter version it was built for), or None."""
        with self._lock:
            hit = self._entries.get(str(device))
            return None if hit is None else hit[1]

    def get(self, module, device, builder):
        with self._lock:
            if self._params is None or not self._live():
                self._walk(module)
            key = params_key(module, device, self._params)
            hit = self._entries.get(str(device))
            if hit is None or hit[0] != key:
                self._walk(module)                                # a miss is rare (once per parameter version): re-walk, then pack
                key = params_key(module, device, self._params)
                hit = (key, builder())
                self._entries[str(device)] = hit
            return hit[1]


# ---- workspace ------------------------------------------------------------------------------------------------------
_ws_lock = threading.Lock()
_workspaces: dict = {}


def workspace(device, nbytes: int) -> torch.Tensor:
    """Grow-only scratch per (device, stream) for the feature projection and the NLL partials."""
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    with _ws_lock:
        ws = _workspaces.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = torch.empty(max(nbytes, 1 << 16), dtype=torch.uint8, device=device)
            _workspaces[key] = ws
        return ws


# ---- launch ---------------------------------------------------------------------------------------------------------
def _check_inputs(rotation, feature, packed: PackedFlow, feature_repeat=None):
    if not rotation.is_cuda:
        raise RuntimeError("rotationnormflow_amd runs on the GPU only (HIP kernels, no CPU fallback): got a CPU tensor")
    if rotation.dim() < 2 or rotation.shape[-1] != 3 or rotation.shape[-2] != 3:
        raise ValueError(f"rotation must be [N,3,3], got {tuple(rotation.shape)}")
    rot = rotation.reshape(-1, 3, 3)
    if rot.dtype != torch.float32:
        rot = rot.float()                                  # reference silently produces fp32 (mobiusflow.py:80)
    rot = rot.contiguous()
    feat = None
    if packed.n_cond:
        assert feature is not None, "The input feature is needed in this module"          # mobiusflow.py:48-49
        if feature_repeat:
            if rot.shape[0] % feature_repeat:
                raise ValueError(f"{rot.shape[0]} rotations are not a multiple of feature_repeat={feature_repeat}")
            feat = feature.reshape(rot.shape[0] // feature_repeat, packed.feat_dim)
        else:
            feat = feature.reshape(rot.shape[0], -1) if rot.shape[0] else feature.reshape(0, packed.feat_dim)   # (-1 is ambiguous for 0 rows)
        if feat.shape[1] != packed.feat_dim:
            raise ValueError(f"feature has {feat.shape[1]} columns, flow expects {packed.feat_dim}")
        feat = feat.to(device=rot.device, dtype=torch.float32)
        if packed.feat_padded != packed.feat_dim:
            feat = torch.nn.functional.pad(feat, (0, packed.feat_padded - packed.feat_dim))
        feat = feat.contiguous()
    return rot, feat


def _needs_grad(rotation, feature, module) -> bool:
    if not torch.is_grad_enabled():
        return False
    return rotation.requires_grad or (feature is not None and feature.requires_grad) or any(
        p.requires_grad for p in module.parameters())


def _refuse_autograd(rotation, feature, module, what):
    if _needs_grad(rotation, feature, module):
        raise NotImplementedError(
            f"rotationnormflow_amd: {what} has no backward kernel (Flow.forward and Flow.inverse are differentiable); call it under "
            "torch.no_grad() -- there is deliberately no PyTorch fallback path")


def _watch_guard(module, packed, ws):
    """Conditional flows packed for the guarded split-precision kernels: keep an eye on the guard word (GuardWatch)."""
    if packed.n_cond and packed.precision == "f16x2" and _guard_fallback:
        watch = getattr(module, "_rnf_guard_watch", None)
        if watch is None:
            watch = module.__dict__.setdefault("_rnf_guard_watch", GuardWatch())
        watch.watch(ws)


def run_flow(module, packed, rotation, feature, inverse=False, train_layers=None, train_rows=None, feature_repeat=None):
    """-> (rotation' [N,3,3], ldj [N]) through rnf_flow_forward / rnf_flow_inverse.
    When a gradient is required (training, agent.py:75-92) the forward direction goes through autograd.flow_forward, which
    needs the layer modules and their permutation rows (``train_layers``, ``train_rows``) and packs on the device itself.
    ``packed`` may be a callable that builds the host-packed flow on demand.
    ``feature_repeat`` = Q: ``feature`` has N / Q rows and row r conditions rotations [r Q, (r + 1) Q) (pose estimation, agent.py:238-263;
    evaluation only) -- the feature projection then runs once per row instead of once per rotation."""
    if _needs_grad(rotation, feature, module):
        if feature_repeat and feature is not None:
            # eval.py:464-480 (nll_grad) differentiates log p w.r.t. the QUERY rotations with every image feature repeated number_queries
            # times (agent.py:240-244 materialises feature.repeat): the differentiable kernels take one feature row per rotation, so the
            # rows are expanded here -- autograd's repeat_interleave sums the row gradients back onto the shared rows
            feature = expand_shared_rows(feature, rotation.reshape(-1, 3, 3).shape[0], feature_repeat)
        from . import autograd
        if inverse:                                        # BinFind.backward (flow/mobiusflow.py:247-273) and friends
            return autograd.flow_inverse(module, train_layers, train_rows, rotation, feature)
        return autograd.flow_forward(module, train_layers, train_rows, rotation, feature)
    if callable(packed):
        packed = packed()
    shared = bool(feature_repeat) and packed.n_cond > 0
    if packed.side_layers and feature_repeat and feature is not None:
        # ConditionRot / ConditionLU build their per-sample matrices from the feature rows with batched torch ops: expand the shared rows
        # (the reference's feature.repeat, agent.py:240-244) and take the one-row-per-rotation path
        feature = expand_shared_rows(feature, rotation.reshape(-1, 3, 3).shape[0], feature_repeat)
        feature_repeat, shared = None, False
    rot, feat = _check_inputs(rotation, feature, packed, feature_repeat if shared else None)
    n = rot.shape[0]
    L = _lib.lib()
    out_rot = torch.empty_like(rot)
    out_ldj = torch.empty(n, dtype=torch.float32, device=rot.device)
    if n == 0:                                             # empty batch: nothing to launch (data_ptr() would be null)
        return out_rot.reshape(rotation.shape), out_ldj
    with torch.cuda.device(rot.device):
        stream = torch.cuda.current_stream(rot.device).cuda_stream
        if shared:
            ws = workspace(rot.device, L.rnf_workspace_bytes_shared(n, packed.n_cond, feature_repeat))
            fn = L.rnf_flow_inverse_shared if inverse else L.rnf_flow_forward_shared
            _lib.check(fn(rot.data_ptr(), feat.data_ptr(), n, packed.feat_padded, feature_repeat, packed.blob.data_ptr(),
                          packed.desc.ctypes.data, packed.n_layers, packed.segments, out_rot.data_ptr(), out_ldj.data_ptr(),
                          ws.data_ptr(), ws.numel(), stream))
        elif packed.side_layers:
            side = build_side_buffer(packed, feature, n, rot.device, inverse)
            ws = workspace(rot.device, L.rnf_workspace_bytes_segments(n, max(packed.n_cond, 1), packed.segments if inverse else 0))
            fn = L.rnf_flow_inverse_side if inverse else L.rnf_flow_forward_side
            _lib.check(fn(rot.data_ptr(), feat.data_ptr() if feat is not None else None, n, packed.feat_padded, side.data_ptr(),
                          packed.blob.data_ptr(), packed.desc.ctypes.data, packed.n_layers, packed.segments,
                          out_rot.data_ptr(), out_ldj.data_ptr(), ws.data_ptr(), ws.numel(), stream))
        else:
            ws = workspace(rot.device, L.rnf_workspace_bytes_segments(n, packed.n_cond, packed.segments) if inverse else L.rnf_workspace_bytes(n, packed.n_cond))
            fn = L.rnf_flow_inverse if inverse else L.rnf_flow_forward
            _lib.check(fn(rot.data_ptr(), feat.data_ptr() if feat is not None else None, n, packed.feat_padded,
                          packed.blob.data_ptr(), packed.desc.ctypes.data, packed.n_layers, packed.segments,
                          out_rot.data_ptr(), out_ldj.data_ptr(), ws.data_ptr(), ws.numel(), stream))
            _watch_guard(module, packed, ws)
    return out_rot.reshape(rotation.shape), out_ldj


def run_log_prob(module, packed: PackedFlow, rotation, feature, fisher_A=None, fisher_c=None,
                 want_rotation=False, want_ldj=False, want_logp=True, feature_repeat=None):
    """Fused Flow.forward + base log-density + NLL sum.  -> dict(logp, sum [2] float64 device tensor, rotation, ldj)"""
    _refuse_autograd(rotation, feature, module, "the fused log_prob evaluation")
    if packed.side_layers and feature_repeat and feature is not None:      # see run_flow: side layers take one feature row per rotation
        feature = expand_shared_rows(feature, rotation.reshape(-1, 3, 3).shape[0], feature_repeat)
        feature_repeat = None
    shared = bool(feature_repeat) and packed.n_cond > 0
    rot, feat = _check_inputs(rotation, feature, packed, feature_repeat if shared else None)
    n = rot.shape[0]
    L = _lib.lib()
    dev = rot.device
    out_rot = torch.empty_like(rot) if want_rotation else None
    out_ldj = torch.empty(n, dtype=torch.float32, device=dev) if want_ldj else None
    out_lp = torch.empty(n, dtype=torch.float32, device=dev) if want_logp else None
    out_sum = torch.zeros(2, dtype=torch.float64, device=dev)
    if n == 0:
        return dict(logp=out_lp, sum=out_sum, rotation=out_rot, ldj=out_ldj)
    B = 0
    if fisher_A is not None:
        fisher_A = fisher_A.reshape(-1, 3, 3).to(device=dev, dtype=torch.float32).contiguous()
        fisher_c = fisher_c.reshape(-1).to(device=dev, dtype=torch.float32).contiguous()
        B = fisher_A.shape[0]
    ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        if shared:
            ws = workspace(dev, L.rnf_workspace_bytes_shared(n, packed.n_cond, feature_repeat))
            _lib.check(L.rnf_flow_log_prob_shared(rot.data_ptr(), ptr(feat), n, packed.feat_padded, feature_repeat, packed.blob.data_ptr(),
                                                  packed.desc.ctypes.data, packed.n_layers, packed.segments,
                                                  ptr(fisher_A), ptr(fisher_c), B, ptr(out_rot), ptr(out_ldj), ptr(out_lp),
                                                  out_sum.data_ptr(), ws.data_ptr(), ws.numel(), stream))
        elif packed.side_layers:
            side = build_side_buffer(packed, feature, n, dev, False)
            ws = workspace(dev, L.rnf_workspace_bytes(n, max(packed.n_cond, 1)))
            _lib.check(L.rnf_flow_log_prob_side(rot.data_ptr(), ptr(feat), n, packed.feat_padded, side.data_ptr(), packed.blob.data_ptr(),
                                                packed.desc.ctypes.data, packed.n_layers, packed.segments,
                                                ptr(fisher_A), ptr(fisher_c), B, ptr(out_rot), ptr(out_ldj), ptr(out_lp),
                                                out_sum.data_ptr(), ws.data_ptr(), ws.numel(), stream))
        else:
            ws = workspace(dev, L.rnf_workspace_bytes(n, packed.n_cond))
            _lib.check(L.rnf_flow_log_prob(rot.data_ptr(), ptr(feat), n, packed.feat_padded, packed.blob.data_ptr(),
                                           packed.desc.ctypes.data, packed.n_layers, packed.segments,
                                           ptr(fisher_A), ptr(fisher_c), B, ptr(out_rot), ptr(out_ldj), ptr(out_lp),
                                           out_sum.data_ptr(), ws.data_ptr(), ws.numel(), stream))
            _watch_guard(module, packed, ws)
    return dict(logp=out_lp, sum=out_sum, rotation=out_rot, ldj=out_ldj)
