"""Mirror of the reference's flow/squeezetrans.py: the 4x4 quaternion-affine family (constant, LU-parameterised, feature-conditioned)
and the 3x3 / 6x6 Gram-Schmidt ablation layers, unconditional and conditional -- every class of the registry is built (DESIGN.md
section 3.7; the conditional LU layers build their batch-coupled per-sample matrices with a HIP kernel into the stack kernel's side buffer)."""
import numpy as np
import torch
import torch.nn as nn

from .. import runtime
from .condition import ConditionalTransform
from .mobiusflow import _SingleLayer


class Uncondition16Trans(nn.Module, _SingleLayer):
    """q' = M q, R' = R(q'/|q'|), ldj = log|det M| - 4 log|q'| with one learned 4x4 M (flow/squeezetrans.py:161-174)."""

    _rnf_kind = runtime.KIND_AFFINE16

    def __init__(self):
        super().__init__()
        self.mat = nn.Parameter(torch.eye(4).unsqueeze(0) + torch.randn(1, 4, 4) * 1e-3)
        self._cache = runtime.PackCache()

    def _rnf_pack(self, L, prec=0):
        return runtime.pack_affine16(L, self.mat), None, 0, 0

    def _rnf_shape(self):
        return (self._rnf_kind, 0, 0)

    def _rnf_train_tensors(self):
        return [self.mat]

    def forward(self, rotation, permute=None, feature=None):
        return self._single(rotation, permute, None, inverse=False)

    def inverse(self, rotation, permute=None, feature=None):
        return self._single(rotation, permute, None, inverse=True)


class Condition16Trans(nn.Module, _SingleLayer):
    """M = I + reshape(MLP(feature), 4, 4) per sample (flow/squeezetrans.py:41-55)."""

    _rnf_kind = runtime.KIND_COND16

    def __init__(self, feature_dim):
        super().__init__()
        self.feature_dim = feature_dim
        self.net = ConditionalTransform(feature_dim, 16)
        self._cache = runtime.PackCache()

    def _rnf_pack(self, L, prec=0):
        rec, frec = runtime.pack_cond16(L, self.net, self.feature_dim, prec)
        return rec, frec, self.feature_dim, 0

    def _rnf_shape(self):
        return (self._rnf_kind, 0, self.feature_dim)

    def _rnf_train_tensors(self):
        from ..autograd import mlp_train_tensors
        return mlp_train_tensors(self.net)

    def forward(self, rotation, permute=None, feature=None):
        return self._single(rotation, permute, feature, inverse=False)

    def inverse(self, rotation, permute=None, feature=None):
        return self._single(rotation, permute, feature, inverse=True)


class UnconditionLU(nn.Module):
    """Glow-style LU parameterisation of an invertible matrix (flow/squeezetrans.py:58-95): P (L*mask + I) (U*mask + diag(sign e^s)).
    Same buffers (w_p, u_mask, l_mask, s_sign, l_eye) and parameters (w_l, w_s, w_u) as the reference => same state-dict keys.
    ``forward()`` assembles the [1,n,n] matrix with a handful of tiny host-side tensor ops (parameter preprocessing; the
    per-sample arithmetic runs in the HIP kernels)."""

    def __init__(self, in_channel):
        super().__init__()
        import numpy as np
        from scipy import linalg as la
        weight = 1e-3 * np.random.randn(in_channel, in_channel) + np.eye(in_channel)
        q, _ = la.qr(weight)
        w_p, w_l, w_u = la.lu(q.astype(np.float32))
        w_s = np.diag(w_u).copy()
        w_u = np.triu(w_u, 1)
        u_mask = np.triu(np.ones_like(w_u), 1)
        self.register_buffer("w_p", torch.from_numpy(w_p))
        self.register_buffer("u_mask", torch.from_numpy(u_mask))
        self.register_buffer("l_mask", torch.from_numpy(u_mask.T.copy()))
        self.register_buffer("s_sign", torch.sign(torch.from_numpy(w_s)))
        self.register_buffer("l_eye", torch.eye(in_channel))
        self.w_l = nn.Parameter(torch.from_numpy(w_l))
        self.w_s = nn.Parameter(torch.from_numpy(w_s).abs().log())
        self.w_u = nn.Parameter(torch.from_numpy(w_u))

    def forward(self):
        weight = (self.w_p @ (self.w_l * self.l_mask + self.l_eye)
                  @ ((self.w_u * self.u_mask) + torch.diag(self.s_sign * torch.exp(self.w_s))))
        return weight.unsqueeze(0)


class Uncondition16TransLU(nn.Module, _SingleLayer):
    """calculate_16 with the LU-parameterised 4x4 matrix (flow/squeezetrans.py:146-158): the constant-matrix affine kernel."""

    _rnf_kind = runtime.KIND_AFFINE16

    def __init__(self):
        super().__init__()
        self.mat = UnconditionLU(4)
        self._cache = runtime.PackCache()

    def _rnf_pack(self, L, prec=0):
        with torch.no_grad():
            return runtime.pack_affine16(L, self.mat()), None, 0, 0

    def _rnf_shape(self):
        return (self._rnf_kind, 0, 0)

    def _rnf_train_tensors(self):
        return [self.mat()]            # autograd chains dL/dM through the LU assembly (a few tiny tensor ops)

    def forward(self, rotation, permute=None, feature=None):
        return self._single(rotation, permute, None, inverse=False)

    def inverse(self, rotation, permute=None, feature=None):
        return self._single(rotation, permute, None, inverse=True)


class _GramSchmidtLayer(nn.Module, _SingleLayer):
    """Shared body of the 3x3 / 6x6 ablation layers: R' = Gram-Schmidt of the transformed first two columns, log-det from three
    tangent directions (calculate_9 / calculate_36, flow/squeezetrans.py:199-231, 293-331); the inverse pass applies M^-1."""

    _rnf_n = 3

    def _matrix(self):
        raise NotImplementedError

    def _rnf_pack(self, L, prec=0):
        with torch.no_grad():
            return runtime.pack_gs(L, self._matrix(), self._rnf_n), None, 0, 0

    def _rnf_shape(self):
        return (self._rnf_kind, 0, 0)

    def _train_tensors(self):
        """The plain blob holds M: [9] + 3 floats of padding, or [36].  Both log-dets have closed forms (csrc/so3_math.h gs9_apply,
        csrc/so3_grad.h gs36_closed_form) whose reverse modes are gs9_backward / gs36_backward."""
        m = self._matrix()
        return [m.reshape(9), m.new_zeros(3)] if self._rnf_n == 3 else [m.reshape(36)]

    def forward(self, rotation, permute=None, feature=None):
        return self._single(rotation, permute, None, inverse=False)

    def inverse(self, rotation, permute=None, feature=None):
        return self._single(rotation, permute, None, inverse=True)


class Uncondition9Trans(_GramSchmidtLayer):
    """calculate_9 with one learned 3x3 matrix (flow/squeezetrans.py:250-261)."""

    _rnf_kind = runtime.KIND_GS9

    def __init__(self):
        super().__init__()
        self.mat = nn.Parameter(torch.eye(3) + torch.randn(3, 3) * 1e-3)
        self._cache = runtime.PackCache()

    def _matrix(self):
        return self.mat

    def _rnf_train_tensors(self):
        return self._train_tensors()


class Uncondition9TransLU(_GramSchmidtLayer):
    """calculate_9 with the LU-parameterised 3x3 matrix (flow/squeezetrans.py:264-275)."""

    _rnf_kind = runtime.KIND_GS9

    def __init__(self):
        super().__init__()
        self.mat = UnconditionLU(3)
        self._cache = runtime.PackCache()

    def _matrix(self):
        return self.mat()[0]

    def _rnf_train_tensors(self):
        return self._train_tensors()            # autograd chains dL/dM through the LU assembly


class Uncondition36Trans(_GramSchmidtLayer):
    """calculate_36 with one learned 6x6 matrix acting on the first two columns of R (flow/squeezetrans.py:350-361)."""

    _rnf_kind = runtime.KIND_GS36
    _rnf_n = 6

    def __init__(self):
        super().__init__()
        self.mat = nn.Parameter(torch.eye(6) + torch.randn(6, 6) * 1e-3)
        self._cache = runtime.PackCache()

    def _matrix(self):
        return self.mat

    def _rnf_train_tensors(self):
        return self._train_tensors()


class _Conditional9(nn.Module, _SingleLayer):
    """Shared body of the conditional 3x3 ablation layers: M = I + reshape(MLP(feature), 3, 3) per sample; the per-sample map (Gram-Schmidt
    with tangent log-det, Smith rotation, polar rotation) runs in the extended instantiation of the fused stack kernel."""

    _rnf_outputs = 9

    def __init__(self, feature_dim):
        super().__init__()
        self.feature_dim = feature_dim
        self.net = ConditionalTransform(feature_dim, self._rnf_outputs)
        self._cache = runtime.PackCache()

    def _rnf_pack(self, L, prec=0):
        rec, frec = runtime.pack_cond16(L, self.net, self.feature_dim, prec, n_out=self._rnf_outputs)
        return rec, frec, self.feature_dim, 0

    def _rnf_shape(self):
        return (self._rnf_kind, 0, self.feature_dim)

    def _rnf_train_tensors(self):
        from ..autograd import mlp_train_tensors
        return mlp_train_tensors(self.net)

    def forward(self, rotation, permute=None, feature=None):
        return self._single(rotation, permute, feature, inverse=False)

    def inverse(self, rotation, permute=None, feature=None):
        return self._single(rotation, permute, feature, inverse=True)


class Condition36Trans(_Conditional9):
    """calculate_36 with a per-sample 6x6 matrix I + reshape(MLP(feature), 6, 6); the inverse pass inverts it per sample in registers
    (flow/squeezetrans.py:334-347)."""
    _rnf_kind = runtime.KIND_COND36
    _rnf_outputs = 36


class Condition9Trans(_Conditional9):
    """calculate_9 with a per-sample matrix; the inverse pass inverts it per sample (flow/squeezetrans.py:234-247)."""
    _rnf_kind = runtime.KIND_COND9_GS


class _SideLayer(nn.Module, _SingleLayer):
    """Layers whose per-sample matrix is formed with the reference's own batched torch ops on outputs of the HIP conditioner
    (runtime.SideNet) and handed to the stack kernel in a side buffer (include/rnf_hip.h RNF_LAYER_SIDE*)."""

    _rnf_side_layer = True
    _rnf_no_graph = True           # per-batch torch ops (and, for ConditionRot, a host SVD) sit between the conditioner and the stack kernel

    def _rnf_pack(self, L, prec=0):
        return np.zeros(4, dtype=np.float32), None, 0, 0

    def _rnf_shape(self):
        return (self._rnf_kind, 0, self.feature_dim)

    def _rnf_train_tensors(self):
        return []              # no parameters in the flow's plain blob: the gradient reaches the networks through the side matrices

    def forward(self, rotation, permute=None, feature=None):
        return self._single(rotation, permute, feature, inverse=False)

    def inverse(self, rotation, permute=None, feature=None):
        return self._single(rotation, permute, feature, inverse=True)


class ConditionLU(nn.Module):
    """flow/squeezetrans.py:94-131: P (L * l_mask + I) (U * u_mask + diag) with L, U and the diagonal predicted from the feature by three
    ConditionalTransforms.  Buffer / parameter names are the checkpoint contract.  The reference's expression is reproduced as defined,
    including ``torch.diag`` of the 2-D [N, C] tensor: that call takes the diagonal ACROSS THE BATCH (entry i of sample i, i < C) and the
    C-vector is then broadcast onto every row of every sample's upper factor -- the weight of one sample depends on the first C samples
    of the batch it travels in (and a batch of fewer than C rows fails to broadcast, as in the reference).  The three MLPs run through the
    HIP conditioner (runtime.SideNet), the assembly and its backward through ``rnf_condlu_matrices`` / ``rnf_condlu_backward`` (round 6)."""

    _rnf_batch_coupled = True

    def __init__(self, in_channel, feature_dim):
        super().__init__()
        import scipy.linalg as la
        self.in_channel = in_channel
        weight = np.random.randn(in_channel, in_channel)
        q, _ = la.qr(weight)
        w_p, w_l, w_u = la.lu(q.astype(np.float32))
        w_s = np.diag(w_u)
        u_mask = np.triu(np.ones_like(w_u), 1)
        self.register_buffer("w_p", torch.from_numpy(w_p))
        self.register_buffer("u_mask", torch.from_numpy(u_mask))
        self.register_buffer("l_mask", torch.from_numpy(u_mask.T.copy()))
        self.register_buffer("s_sign", torch.sign(torch.from_numpy(np.copy(w_s))))
        self.register_buffer("l_eye", torch.eye(in_channel))
        self.w_l_net = ConditionalTransform(feature_dim, in_channel * in_channel)
        self.w_u_net = ConditionalTransform(feature_dim, in_channel * in_channel)
        self.w_s_net = ConditionalTransform(feature_dim, in_channel)
        self._nets = None
        self._consts_cache = None
        self._feature_dim = feature_dim

    def _consts(self, device):
        """w_p | l_mask | u_mask | l_eye | s_sign as one device vector (the layout rnf_condlu_matrices reads), rebuilt when a buffer changes."""
        bufs = (self.w_p, self.l_mask, self.u_mask, self.l_eye, self.s_sign)
        key = (str(device),) + tuple((id(b), b._version, b.data_ptr()) for b in bufs)
        if self._consts_cache is None or self._consts_cache[0] != key:
            self._consts_cache = (key, torch.cat([b.detach().reshape(-1).to(device=device, dtype=torch.float32) for b in bufs]).contiguous())
        return self._consts_cache[1]

    def side(self, feature, add_identity=False):
        """-> [n, 16]: the C x C matrix of every sample row-major in the leading C*C floats (one slot of the stack kernel's side buffer).
        The three conditioners and the assembly (csrc/rnf_api.hip condlu_assemble_kernel, with its backward) are HIP kernels; no torch
        compute op runs between the call and the launches (round 6; until round 5 the assembly was the reference's einsum / torch.diag)."""
        if not feature.is_cuda:
            raise RuntimeError("rotationnormflow_amd runs on the GPU only (HIP kernels, no CPU fallback): got a CPU tensor")
        C = self.in_channel
        if self._nets is None:
            self._nets = (runtime.SideNet(self.w_l_net, self._feature_dim, C * C), runtime.SideNet(self.w_u_net, self._feature_dim, C * C),
                          runtime.SideNet(self.w_s_net, self._feature_dim, C))
        wl, wu, ws = (net(feature) for net in self._nets)
        return _CondLUFn.apply(wl, wu, ws, self._consts(feature.device), C, bool(add_identity))

    def forward(self, feature):
        """The reference's call (flow/squeezetrans.py:120-131): weight [n, C, C]."""
        C = self.in_channel
        return self.side(feature)[:, : C * C].reshape(-1, C, C)


def _rows(t):
    """(tensor, row stride) of a 2-D float32 cuda tensor whose rows are dense: the conditioner hands out views of [n, 16] buffers."""
    if t.dtype is not torch.float32 or t.stride(1) != 1:
        t = t.to(torch.float32).contiguous()
    return t, t.stride(0)


class _CondLUFn(torch.autograd.Function):
    """ConditionLU's matrices from the three conditioners' outputs (rnf_condlu_matrices / rnf_condlu_backward): P (L * l_mask + I)
    (U * u_mask + dvec) with the reference's batch-coupled dvec = torch.diag(s_sign * exp(ws)) (flow/squeezetrans.py:120-131)."""

    @staticmethod
    def forward(ctx, wl, wu, ws, consts, C, add_identity):
        from .. import _lib
        (wl, sl), (wu, su), (ws, ss) = _rows(wl.detach()), _rows(wu.detach()), _rows(ws.detach())
        n, dev = wl.shape[0], wl.device
        if 0 < n < C:          # the reference's broadcast of the C-vector torch.diag(...) against [n, C, C] fails for fewer than C rows
            raise RuntimeError(f"The size of tensor a ({n}) must match the size of tensor b ({C}) at non-singleton dimension 2")
        out = torch.empty((n, 16), dtype=torch.float32, device=dev)
        if n:
            with torch.cuda.device(dev):
                _lib.check(_lib.lib().rnf_condlu_matrices(wl.data_ptr(), wu.data_ptr(), ws.data_ptr(), sl, su, ss, n, C, consts.data_ptr(),
                                                          int(add_identity), out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
        ctx.save_for_backward(wl, wu, ws, consts)
        ctx.meta = (C, sl, su, ss)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        from .. import _lib
        wl, wu, ws, consts = ctx.saved_tensors
        C, sl, su, ss = ctx.meta
        n, dev = wl.shape[0], wl.device
        g = g.to(torch.float32).contiguous()
        g_wl = torch.empty((n, C * C), dtype=torch.float32, device=dev)
        g_wu = torch.empty((n, C * C), dtype=torch.float32, device=dev)
        g_ws = torch.empty((n, C), dtype=torch.float32, device=dev)
        scratch = torch.empty(4, dtype=torch.float32, device=dev)
        if n:
            with torch.cuda.device(dev):
                _lib.check(_lib.lib().rnf_condlu_backward(wl.data_ptr(), wu.data_ptr(), ws.data_ptr(), sl, su, ss, n, C, consts.data_ptr(),
                                                          g.data_ptr(), g_wl.data_ptr(), g_wu.data_ptr(), g_ws.data_ptr(), scratch.data_ptr(),
                                                          torch.cuda.current_stream(dev).cuda_stream))
        return g_wl, g_wu, g_ws, None, None, None


class Condition16TransLU(_SideLayer):
    """calculate_16 with the per-sample ConditionLU(4) matrix; the inverse pass inverts it per sample (flow/squeezetrans.py:134-144)."""
    _rnf_kind = runtime.KIND_SIDE16
    _rnf_batch_coupled = True

    def __init__(self, feature_dim):
        super().__init__()
        self.feature_dim = feature_dim
        self.net = ConditionLU(4, feature_dim)
        self._cache = runtime.PackCache()

    _rnf_no_graph = False          # round 6: conditioners, assembly and their backward are HIP kernels -- capturable into a HIP graph

    def _rnf_side(self, feature, grad=False):
        with torch.set_grad_enabled(grad):
            return self.net.side(feature)


class Condition9TransLU(_SideLayer):
    """calculate_9 with I + the per-sample ConditionLU(3) matrix (flow/squeezetrans.py:264-277; note the + I, which the unconditional
    Uncondition9TransLU does not have)."""
    _rnf_kind = runtime.KIND_SIDE9
    _rnf_batch_coupled = True

    def __init__(self, feature_dim):
        super().__init__()
        self.feature_dim = feature_dim
        self.net = ConditionLU(3, feature_dim)
        self._cache = runtime.PackCache()

    _rnf_no_graph = False

    def _rnf_side(self, feature, grad=False):
        with torch.set_grad_enabled(grad):
            return self.net.side(feature, add_identity=True)          # + I (flow/squeezetrans.py:269-271) inside the assembly kernel


# ---- the reference's module-level functions under their own names (flow/squeezetrans.py:10-38,200-231) ---------------------------------
# The layer classes above never call these (their maps run inside the fused kernels); they exist for code that calls the functions
# directly.  calculate_16 / calculate_9 take a matrix per rotation -- or one matrix for all -- and run on the stack kernel's side-matrix
# path (the one the conditional LU layers use), differentiable w.r.t. the rotations and the matrices like every other layer.

class _ExplicitMatrix(_SideLayer):
    """A side layer whose per-rotation matrices are handed in by the caller (as the `feature` argument of the single-layer launch)."""
    _rnf_no_graph = False

    def __init__(self, kind, size):
        super().__init__()
        self._rnf_kind = kind
        self.feature_dim = size
        self._cache = runtime.PackCache()

    def _rnf_side(self, feature, grad=False):
        return feature


_explicit_layers = {}


def _calculate(kind, size, mat, rotation):
    if not rotation.is_cuda:
        raise RuntimeError("rotationnormflow_amd runs on the GPU only (HIP kernels, no CPU fallback): got a CPU tensor")
    rotation = rotation.reshape(-1, 3, 3)
    mats = mat.reshape(-1, size).to(device=rotation.device, dtype=torch.float32)
    if mats.shape[0] == 1:
        mats = mats.expand(rotation.shape[0], size)
    if mats.shape[0] != rotation.shape[0]:
        raise ValueError(f"{mats.shape[0]} matrices for {rotation.shape[0]} rotations")
    layer = _explicit_layers.get(kind)
    if layer is None:
        layer = _explicit_layers[kind] = _ExplicitMatrix(kind, size)
    return layer._single(rotation, None, mats.contiguous(), inverse=False)


def calculate_16(mat, rotation):
    """flow/squeezetrans.py:33-38: q' = M q / |M q| on the rotation's quaternion, log-det = log|det M| - 4 log|M q|.  mat [N,4,4] (or one
    [4,4] / [1,4,4] for all) -> (rotation' [N,3,3], ldj [N])."""
    return _calculate(runtime.KIND_SIDE16, 16, mat, rotation)


def calculate_9(mat, rotation):
    """flow/squeezetrans.py:200-231: Gram-Schmidt of the first two columns of M R with the closed-form tangent log-det.  mat [N,9] /
    [N,3,3] (or one for all) -> (rotation', ldj)."""
    return _calculate(runtime.KIND_SIDE9, 9, mat, rotation)


def my_det_3_3(A):
    """flow/squeezetrans.py:10-14 (cofactor expansion; any leading dimensions)."""
    return (A[..., 0, 0] * (A[..., 1, 1] * A[..., 2, 2] - A[..., 1, 2] * A[..., 2, 1])
            + A[..., 0, 1] * (A[..., 1, 2] * A[..., 2, 0] - A[..., 1, 0] * A[..., 2, 2])
            + A[..., 0, 2] * (A[..., 1, 0] * A[..., 2, 1] - A[..., 1, 1] * A[..., 2, 0]))


def my_det_4_4(A):
    """flow/squeezetrans.py:17-22: expansion along the first row."""
    rows = A[..., 1:, :]
    minor = lambda cols: my_det_3_3(rows[..., cols])
    return (A[..., 0, 0] * minor([1, 2, 3]) - A[..., 0, 1] * minor([0, 2, 3])
            + A[..., 0, 2] * minor([0, 1, 3]) - A[..., 0, 3] * minor([0, 1, 2]))
