"""Mirror of the reference's flow/rottrans.py (SVD / Smith rotation layers, ldj = 0).  Every class of the registry is built (DESIGN.md
section 3.7).  Unconditional layers: a constant orthogonal 4x4 on the quaternion, prepared on the host from the 3x3 / 4x4 parameter and
run by the quaternion kernel.  Conditional 3x3 layers: per-sample polar / Smith rotation inside the fused stack kernel.  ConditionRot:
the per-sample matrices U^T V are built by the reference's own ``torch.svd`` call (its result depends on the SVD routine's sign
conventions) and handed to the kernel as a side buffer."""
import os

import torch
import torch.nn as nn

from .. import runtime
from .mobiusflow import _SingleLayer

_CONDROT_SVD_ON_DEVICE = os.environ.get("RNF_CONDROT_SVD", "host") == "device"


class UnconditionRot(nn.Module, _SingleLayer):
    """Rotate the quaternion by the orthogonal 4x4 matrix U^T V of the SVD of a learned 4x4 parameter; log-det 0
    (flow/rottrans.py:8-35).  The SVD (a 4x4, once per parameter version) is parameter preprocessing on the host, done with the
    same ``torch.svd`` call as the reference because U^T V depends on the SVD's arbitrary column signs; the per-sample arithmetic
    is the constant-matrix quaternion kernel."""

    _rnf_kind = runtime.KIND_AFFINE16
    _rnf_orthogonal = True

    def __init__(self):
        super().__init__()
        self.rot = nn.Parameter(torch.randn((1, 4, 4)) * 1e-3 + torch.eye(4).unsqueeze(0))
        self._cache = runtime.PackCache()

    def _rnf_pack(self, L, prec=0):
        with torch.no_grad():
            U, S, V = torch.svd(self.rot.detach().cpu().float())
            rot_mat = U.transpose(-1, -2) @ V
        return runtime.pack_rot16(L, rot_mat), None, 0, 0

    def _rnf_shape(self):
        return (self._rnf_kind, 0, 0)

    _rnf_host_preprocess = True       # the training tensors go through a host SVD: a device->host copy per iteration, no stream capture

    def _rnf_train_tensors(self):
        U, S, V = torch.svd(self.rot.cpu().float())      # differentiable; the 4x4 SVD stays on the host (see _rnf_pack)
        return [U.transpose(-1, -2) @ V]

    def forward(self, rotation, permute=None, feature=None):
        return self._single(rotation, permute, None, inverse=False)

    def inverse(self, rotation, permute=None, feature=None):
        return self._single(rotation, permute, None, inverse=True)

def _quaternion_of(P):
    """Unit quaternion (w, x, y, z) of a proper 3x3 rotation, differentiable torch ops on the host (largest-component branch)."""
    m = P
    tr = m[0, 0] + m[1, 1] + m[2, 2]
    cands = torch.stack([1 + tr, 1 + m[0, 0] - m[1, 1] - m[2, 2], 1 - m[0, 0] + m[1, 1] - m[2, 2], 1 - m[0, 0] - m[1, 1] + m[2, 2]])
    b = int(torch.argmax(cands))
    if b == 0:
        q = torch.stack([cands[0], m[2, 1] - m[1, 2], m[0, 2] - m[2, 0], m[1, 0] - m[0, 1]])
    elif b == 1:
        q = torch.stack([m[2, 1] - m[1, 2], cands[1], m[0, 1] + m[1, 0], m[0, 2] + m[2, 0]])
    elif b == 2:
        q = torch.stack([m[0, 2] - m[2, 0], m[0, 1] + m[1, 0], cands[2], m[1, 2] + m[2, 1]])
    else:
        q = torch.stack([m[1, 0] - m[0, 1], m[0, 2] + m[2, 0], m[1, 2] + m[2, 1], cands[3]])
    return q / q.norm()


def _left_matrix(p):
    """4x4 matrix of q -> p (x) q: the rotation R(p) applied on the LEFT of R(q)."""
    w, x, y, z = p
    return torch.stack([torch.stack([w, -x, -y, -z]), torch.stack([x, w, -z, y]), torch.stack([y, z, w, -x]), torch.stack([z, -y, x, w])])


def _right_matrix(p):
    """4x4 matrix of q -> q (x) p: the rotation R(p) applied on the RIGHT of R(q)."""
    w, x, y, z = p
    return torch.stack([torch.stack([w, -x, -y, -z]), torch.stack([x, w, z, -y]), torch.stack([y, -z, w, x]), torch.stack([z, y, -x, w])])


class _ConstantRotationLayer(nn.Module, _SingleLayer):
    """The unconditional 3x3 "rotation" ablations (flow/rottrans.py:72-165) multiply R by ONE constant rotation Q(mat) on the left or
    on the right and report log-det 0:

      * Uncondition9RotL: U V^T of svd(mat R) = polar(mat) R            (R is orthogonal, so the polar factor factorises)
      * Uncondition9RotR: U V^T of svd(R mat) = R polar(mat)
      * Uncondition9RotRSmith: R GramSchmidt(mat)

    A constant left / right rotation is a constant orthogonal 4x4 matrix on the quaternion, i.e. exactly the UnconditionRot kernel.
    The reference runs a batched 3x3 SVD per sample for the first two; here the 3x3 polar factor is parameter preprocessing on the
    host (one tiny SVD per parameter version, differentiable for training)."""

    _rnf_kind = runtime.KIND_AFFINE16
    _rnf_orthogonal = True
    _left = False

    def __init__(self):
        super().__init__()
        self.mat = nn.Parameter(torch.eye(3) + torch.randn(3, 3) * 1e-3)
        self._cache = runtime.PackCache()

    def _rotation(self, m):
        raise NotImplementedError

    def _quat_matrix(self):
        Q = self._rotation(self.mat.cpu().float())
        if float(torch.det(Q.detach())) < 0:
            raise ValueError(f"{type(self).__name__}: det(mat) < 0, the layer would output reflections")
        p = _quaternion_of(Q)
        return (_left_matrix(p) if self._left else _right_matrix(p)).unsqueeze(0)

    def _rnf_pack(self, L, prec=0):
        with torch.no_grad():
            return runtime.pack_rot16(L, self._quat_matrix()), None, 0, 0

    def _rnf_shape(self):
        return (self._rnf_kind, 0, 0)

    _rnf_host_preprocess = True       # polar factor / Gram-Schmidt + quaternion on the host: a device->host copy per training iteration

    def _rnf_train_tensors(self):
        return [self._quat_matrix()]

    def forward(self, rotation, permute=None, feature=None):
        return self._single(rotation, permute, None, inverse=False)

    def inverse(self, rotation, permute=None, feature=None):
        return self._single(rotation, permute, None, inverse=True)


def _polar(m):
    U, S, V = torch.svd(m)
    return U @ V.transpose(-1, -2)


class Uncondition9RotL(_ConstantRotationLayer):
    """flow/rottrans.py:94-105."""
    _left = True

    def _rotation(self, m):
        return _polar(m)


class Uncondition9RotR(_ConstantRotationLayer):
    """flow/rottrans.py:124-135."""

    def _rotation(self, m):
        return _polar(m)


class Uncondition9RotRSmith(_ConstantRotationLayer):
    """flow/rottrans.py:154-165 (calculate_9_r_smith, :85-96)."""

    def _rotation(self, m):
        m0 = m[:, 0] / m[:, 0].norm()
        m1 = m[:, 1] - (m0 * m[:, 1]).sum() * m0
        m1 = m1 / m1.norm()
        return torch.stack([m0, m1, torch.linalg.cross(m0, m1)], dim=-1)


from .squeezetrans import _Conditional9, _SideLayer  # noqa: E402
from .condition import ConditionalTransform  # noqa: E402


class Condition9RotL(_Conditional9):
    """Polar rotation of the per-sample matrix applied on the left (flow/rottrans.py:108-121); log-det 0."""
    _rnf_kind = runtime.KIND_COND9_POLAR_L


class Condition9RotR(_Conditional9):
    """Polar rotation of the per-sample matrix applied on the right (flow/rottrans.py:138-151); log-det 0."""
    _rnf_kind = runtime.KIND_COND9_POLAR_R


class Condition9RotRSmith(_Conditional9):
    """R times the Gram-Schmidt rotation of the per-sample matrix, its transpose for the inverse (flow/rottrans.py:168-181); log-det 0."""
    _rnf_kind = runtime.KIND_COND9_SMITH


class ConditionRot(_SideLayer):
    """flow/rottrans.py:37-66: rot = U^T V of the batched SVD of I + reshape(net(feature), 4, 4), applied to the quaternion; log-det 0;
    the inverse pass applies its transpose.  U^T V (not the polar factor U V^T) depends on the sign conventions of the SVD routine, so
    the layer is defined by the routine.  Evaluation: net(feature) and the per-sample U^T V both run on the GPU, the latter through
    csrc/svd4_lapack.h -- LAPACK's dense-SVD path (sgebd2, sorgbr, sbdsqr) restated for 4x4 so that its sign conventions are those of the
    reference's ``torch.svd`` (tests/test_svd4.py).  Training: torch differentiates its own ``torch.svd`` on the host, as the reference does
    (RNF_CONDROT_SVD=device keeps that call on the GPU, with hipSOLVER's conventions)."""
    _rnf_kind = runtime.KIND_SIDE16_ROT
    _rnf_host_preprocess = True

    def __init__(self, feature_dim):
        super().__init__()
        self.feature_dim = feature_dim
        self.net = ConditionalTransform(feature_dim, 16)
        self._cache = runtime.PackCache()
        self._net = None

    def _rnf_side(self, feature, grad=False):
        if self._net is None:
            self._net = runtime.SideNet(self.net, self.feature_dim, 16)
        if not grad:
            # inference: the conditioner on the GPU, then U^T V per sample on the GPU with LAPACK's sign conventions (csrc/svd4_lapack.h):
            # no device -> host copy, no host LAPACK, stream-ordered (round 2 ran torch.svd on the host here)
            from .. import _lib
            with torch.no_grad():
                out = self._net(feature).to(torch.float32).contiguous()          # [n, 16]
                rot = torch.empty_like(out)
                if out.shape[0]:
                    with torch.cuda.device(out.device):
                        _lib.check(_lib.lib().rnf_condrot_matrices(out.data_ptr(), out.shape[0], rot.data_ptr(),
                                                                   torch.cuda.current_stream(out.device).cuda_stream))
            return rot
        with torch.set_grad_enabled(True):                # training: torch differentiates its own SVD, as the reference does (host LAPACK)
            mat = self._net(feature).reshape(-1, 4, 4) + torch.eye(4, device=feature.device)
            if _CONDROT_SVD_ON_DEVICE:
                U, S, V = torch.svd(mat)
                return (U.transpose(-1, -2) @ V).reshape(-1, 16)
            U, S, V = torch.svd(mat.cpu())
            return (U.transpose(-1, -2) @ V).reshape(-1, 16).to(feature.device)
