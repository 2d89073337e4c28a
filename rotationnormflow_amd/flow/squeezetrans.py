"""Mirror of the reference's flow/squeezetrans.py: the 4x4 quaternion-affine family (constant, LU-parameterised, feature-conditioned)
and the unconditional 3x3 / 6x6 Gram-Schmidt ablation layers.  The remaining conditional variants are declared for the registry and
fail loudly at construction."""
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
        """3x3 only: the plain blob holds M [9] + 3 floats of padding (the log-det has a closed form, csrc/so3_math.h gs9_apply, whose
        reverse mode is csrc/so3_grad.h gs9_backward); the 6x6 layer has no backward kernel."""
        m = self._matrix()
        return [m.reshape(9), m.new_zeros(3)]

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


def _not_built(name, where):
    class _Unbuilt(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()
            raise NotImplementedError(
                f"{name} ({where}) has no HIP kernel yet and rotationnormflow_amd has no PyTorch fallback; "
                "built affine layers: the Uncondition* family and Condition16Trans")
    _Unbuilt.__name__ = _Unbuilt.__qualname__ = name
    return _Unbuilt


# declared so that the registry (flow/affineflow.py:5-73) resolves every name; constructing them fails loudly
Condition16TransLU = _not_built("Condition16TransLU", "flow/squeezetrans.py:130-143")
Condition9TransLU = _not_built("Condition9TransLU", "flow/squeezetrans.py:278-291")
