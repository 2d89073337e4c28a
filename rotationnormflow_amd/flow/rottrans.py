"""Mirror of the reference's flow/rottrans.py (SVD / Smith rotation layers, ldj = 0).  Every class of the registry is built (DESIGN.md
section 3.7).  Unconditional layers: a constant orthogonal 4x4 on the quaternion, prepared on the host from the 3x3 / 4x4 parameter and
run by the quaternion kernel.  Conditional 3x3 layers: per-sample polar / Smith rotation inside the fused stack kernel.  ConditionRot:
the per-sample matrices U^T V come from the device restatement of LAPACK's 4x4 SVD path (the result depends on the SVD routine's sign
conventions) and are handed to the kernel as a side buffer; the same routine's factors give the backward."""
import torch
import torch.nn as nn

from .. import runtime
from .mobiusflow import _SingleLayer

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


_pending_svd_flags = []          # (event, pinned host int32, device flag): convergence flags of earlier ConditionRot calls


def _check_svd_flags(wait: bool = False):
    """Raise if the 4x4 QR iteration of an EARLIER ConditionRot call did not converge (csrc/svd4_lapack.h returns false after LAPACK's
    sweep limit; the slot then holds the last sweep's factors).  Like the sampler's flag (utils/fisher.py) the failure surfaces one call
    later, so that no call waits for the device."""
    if torch.cuda.is_current_stream_capturing():            # inside a HIP graph capture: no event queries
        return
    keep, bad = [], False
    for ev, host, flag in _pending_svd_flags:
        if wait:
            ev.synchronize()
        if ev.query():
            bad = bad or bool(int(host[0]))
        else:
            keep.append((ev, host, flag))
    _pending_svd_flags[:] = keep
    if bad:
        raise RuntimeError("ConditionRot: the SVD of a per-sample 4x4 matrix did not converge in an earlier call (LAPACK's sbdsqr sweep limit, "
                           "csrc/svd4_lapack.h): the conditioner's output is not a usable matrix (NaN / inf entries?)")


def condrot_failures(wait: bool = True):
    """Raise if any earlier ConditionRot evaluation reported a non-converged SVD (waits for outstanding calls by default)."""
    _check_svd_flags(wait)


def _watch_flag(flag):
    if torch.cuda.is_current_stream_capturing():
        return
    host = torch.zeros(1, dtype=torch.int32).pin_memory()
    host.copy_(flag, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    _pending_svd_flags.append((ev, host, flag))


class _CondRotFn(torch.autograd.Function):
    """rot = U^T V of svd(I + reshape(out, 4, 4)) per sample on the device (csrc/svd4_lapack.h) with its analytic backward, so that training
    and evaluation see ONE routine's sign choices (round 3 trained through the host's torch.svd and evaluated through the device
    routine, which agree on 99.8 % of the matrices only) and a flow with this layer needs no device -> host copy per iteration.

    With M = U S V^T, P = U^T dM V, wU = U^T dU and wV = V^T dV (antisymmetric): P_ij = wU_ij s_j - s_i wV_ij off the diagonal, hence
        wU_ij = (s_j P_ij + s_i P_ji) / (s_j^2 - s_i^2),   wV_ij = (s_i P_ij + s_j P_ji) / (s_j^2 - s_i^2),   d rot = -wU rot + rot wV.
    Backward of <G, d rot>: a = A - A^T with A = -G rot^T, b = B - B^T with B = rot^T G; X_ij = (a_ij s_j + b_ij s_i) / (s_j^2 - s_i^2) for
    i != j (0 on the diagonal) and dL/dM = U X V^T.  Like torch.svd's own backward this is singular where two singular values coincide (U^T V
    itself is discontinuous there); the denominators are floored at 1e-20."""

    @staticmethod
    def forward(ctx, out16):
        from .. import _lib
        _check_svd_flags()
        out = out16.detach().to(torch.float32).contiguous()
        n = out.shape[0]
        dev = out.device
        rot = torch.empty_like(out)
        U, S, VT = torch.empty_like(out), torch.empty((n, 4), dtype=torch.float32, device=dev), torch.empty_like(out)
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        if n:
            with torch.cuda.device(dev):
                _lib.check(_lib.lib().rnf_condrot_svd(out.data_ptr(), n, rot.data_ptr(), U.data_ptr(), S.data_ptr(), VT.data_ptr(), flag.data_ptr(),
                                                      torch.cuda.current_stream(dev).cuda_stream))
                _watch_flag(flag)
        ctx.save_for_backward(rot, U, S, VT)
        return rot

    @staticmethod
    def backward(ctx, g):
        return condrot_grad(*ctx.saved_tensors, g)


def condrot_grad(rot, U, S, VT, g):
    """dL/d(out16) of rot = U^T V given dL/d(rot) = g (see _CondRotFn): batched 4x4 torch ops on whatever device the factors live on
    (rot [n,16], U [n,16] row-major, S [n,4], VT [n,16] = V^T row-major)."""
    n = rot.shape[0]
    R = rot.reshape(n, 4, 4)
    G = g.reshape(n, 4, 4).to(R.dtype)
    A = -G @ R.transpose(-1, -2)
    B = R.transpose(-1, -2) @ G
    a = A - A.transpose(-1, -2)
    b = B - B.transpose(-1, -2)
    si, sj = S[:, :, None], S[:, None, :]
    den = sj * sj - si * si
    den = torch.where(den.abs() < 1e-20, torch.full_like(den, 1e-20), den)
    X = (a * sj + b * si) / den
    X = X * (1.0 - torch.eye(4, device=X.device, dtype=X.dtype))
    dM = U.reshape(n, 4, 4) @ X @ VT.reshape(n, 4, 4)
    return dM.reshape(n, 16)


class ConditionRot(_SideLayer):
    """flow/rottrans.py:37-66: rot = U^T V of the batched SVD of I + reshape(net(feature), 4, 4), applied to the quaternion; log-det 0;
    the inverse pass applies its transpose.  U^T V (not the polar factor U V^T) depends on the sign conventions of the SVD routine, so
    the layer is defined by the routine: net(feature) and the per-sample U^T V both run on the GPU, the latter through
    csrc/svd4_lapack.h -- LAPACK's dense-SVD path (sgebd2, sorgbr, sbdsqr) restated for 4x4 so that its sign conventions are those of the
    reference's ``torch.svd`` (tests/test_svd4.py) -- in evaluation AND in training (round 4: ``_CondRotFn`` differentiates U^T V from the
    device routine's own factors; round 3 trained through the host's torch.svd).  No device -> host copy in either mode; a non-converged QR
    iteration raises one call later (``condrot_failures``)."""
    _rnf_kind = runtime.KIND_SIDE16_ROT
    _rnf_no_graph = False          # nothing between the conditioner and the stack kernel leaves the device: capturable into a HIP graph

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
            from .. import _lib
            _check_svd_flags()
            with torch.no_grad():
                out = self._net(feature).to(torch.float32).contiguous()          # [n, 16]
                rot = torch.empty_like(out)
                if out.shape[0]:
                    flag = torch.zeros(1, dtype=torch.int32, device=out.device)
                    with torch.cuda.device(out.device):
                        _lib.check(_lib.lib().rnf_condrot_matrices(out.data_ptr(), out.shape[0], rot.data_ptr(), flag.data_ptr(),
                                                                   torch.cuda.current_stream(out.device).cuda_stream))
                        _watch_flag(flag)
            return rot
        with torch.set_grad_enabled(True):                # training: the same device routine, differentiated analytically (_CondRotFn)
            return _CondRotFn.apply(self._net(feature).to(torch.float32))
