"""Mirror of the reference's utils/fisher.py: ``MatrixFisherN`` (log-density on the GPU through rnf_fisher_log_prob).

log p(R) = tr(A^T R) - (s0+s1+s2) - log norm, norm = 1/sqrt(8 pi (s0+s1)(s1+s2)(s0+s2))   (utils/fisher.py:93-97,217-232; norm_type 1)
with s the *proper* singular values of A (last one sign-flipped by det(U) det(V), utils/fisher.py:67-76); norm_type 0 is the
small-s approximation of utils/fisher.py:88-91.  Differentiable w.r.t. the rotations and w.r.t. A.
"""
import math

import torch

from .. import _lib


def device_proper_svd(A):
    """A [B,3,3] on the GPU -> (U, V, s, lam) fp32 device tensors through rnf_fisher_proper_svd (fp64 Jacobi, one thread per matrix):
    stream-ordered, no host synchronisation (utils/fisher.py:53-76,151-158)."""
    A32 = A.detach().reshape(-1, 3, 3).to(torch.float32).contiguous()
    B, dev = A32.shape[0], A32.device
    U = torch.empty(B, 3, 3, dtype=torch.float32, device=dev)
    V = torch.empty(B, 3, 3, dtype=torch.float32, device=dev)
    s = torch.empty(B, 3, dtype=torch.float32, device=dev)
    lam = torch.empty(B, 4, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().rnf_fisher_proper_svd(A32.data_ptr(), B, U.data_ptr(), V.data_ptr(), s.data_ptr(), lam.data_ptr(),
                                                    torch.cuda.current_stream(dev).cuda_stream))
    return U, V, s, lam


def proper_singular_values(A):
    """[B,3,3] -> [B,3] in fp64 (utils/fisher.py:67-76), by host LAPACK on an fp64 copy of A -- also for a GPU-resident A (one tiny
    device -> host copy).  This serves the lazily computed ``MatrixFisherN.norm`` attribute only (a diagnostic: the density path uses the
    device kernels and never calls it), where round 3's fp32 device values cost ``sum(S) * 6e-8`` of relative accuracy and dropped the
    precision of a float64 A."""
    A64 = A.detach().to("cpu", torch.float64)
    U, S, Vh = torch.linalg.svd(A64)
    S = S.clone()
    S[:, 2] = S[:, 2] * torch.det(U) * torch.det(Vh)
    return S


def quaternion_to_matrix(q):
    """Real-part-first quaternions -> rotation matrices (semantics of pytorch3d.transforms.quaternion_to_matrix)."""
    w, x, y, z = q.unbind(-1)
    s2 = 2.0 / (q * q).sum(-1)
    m = torch.stack([1 - s2 * (y * y + z * z), s2 * (x * y - z * w), s2 * (x * z + y * w),
                     s2 * (x * y + z * w), 1 - s2 * (x * x + z * z), s2 * (y * z - x * w),
                     s2 * (x * z - y * w), s2 * (y * z + x * w), 1 - s2 * (x * x + y * y)], -1)
    return m.reshape(q.shape[:-1] + (3, 3))


def _norm_from_singular_values(S, norm_type):
    """matrix_fisher_norm_N (utils/fisher.py:79-97) for the two closed-form approximations; type 0 keeps the reference's batch-global
    ``(S**2).sum()``."""
    if norm_type == 0:
        return (1.0 + (S ** 2).sum() / 6.0 + S[:, 0] * S[:, 1] * S[:, 2] / 6.0) / S.sum(-1).exp()
    return 1.0 / torch.sqrt(8 * math.pi * (S[:, 0] + S[:, 1]) * (S[:, 2] + S[:, 1]) * (S[:, 0] + S[:, 2]))


class _FisherLogProb(torch.autograd.Function):
    """log p(R) = tr(A^T R) - c through rnf_fisher_log_prob; d/dR = g A through rnf_fisher_log_prob_backward (training with a
    matrix-Fisher base differentiates it w.r.t. the flow's output rotation, agent.py:58-64); d/dA -- needed when A is predicted by a
    network and kept in the graph, agent.py:57-65 -- through rnf_fisher_log_prob_backward_param (derivative of the normaliser included)."""

    @staticmethod
    def forward(ctx, inputs, A_param, A, c, norm_type):
        R = inputs.reshape(-1, 3, 3).to(torch.float32).contiguous()
        n, B = R.shape[0], A.shape[0]
        out = torch.empty(n, dtype=torch.float32, device=R.device)
        with torch.cuda.device(R.device):
            _lib.check(_lib.lib().rnf_fisher_log_prob(R.data_ptr(), n, A.data_ptr(), c.data_ptr(), B, out.data_ptr(),
                                                      torch.cuda.current_stream(R.device).cuda_stream))
        ctx.save_for_backward(A, R)
        ctx.in_shape, ctx.in_dtype, ctx.norm_type = inputs.shape, inputs.dtype, norm_type
        ctx.A_shape, ctx.A_dtype, ctx.A_device = A_param.shape, A_param.dtype, A_param.device
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        A, R = ctx.saved_tensors
        g = g.to(torch.float32).contiguous()
        n, B = g.shape[0], A.shape[0]
        L = _lib.lib()
        g_rot = g_A = None
        with torch.cuda.device(g.device):
            stream = torch.cuda.current_stream(g.device).cuda_stream
            if ctx.needs_input_grad[0]:
                g_rot = torch.empty((n, 3, 3), dtype=torch.float32, device=g.device)
                _lib.check(L.rnf_fisher_log_prob_backward(g.data_ptr(), n, A.data_ptr(), B, g_rot.data_ptr(), stream))
                g_rot = g_rot.reshape(ctx.in_shape).to(ctx.in_dtype)
            if ctx.needs_input_grad[1]:
                g_A = torch.empty((B, 3, 3), dtype=torch.float32, device=g.device)
                scratch = torch.empty(int(L.rnf_fisher_scratch_bytes(B)) // 8, dtype=torch.float64, device=g.device)
                _lib.check(L.rnf_fisher_log_prob_backward_param(g.data_ptr(), R.data_ptr(), n, A.data_ptr(), B, ctx.norm_type, scratch.data_ptr(),
                                                            scratch.numel() * 8, g_A.data_ptr(), stream))
                g_A = g_A.reshape(ctx.A_shape).to(device=ctx.A_device, dtype=ctx.A_dtype)
        return g_rot, g_A, None, None, None


# fail flags of earlier _sample calls (a proposal loop that exhausted its 4096 attempts leaves the identity rotation in that slot): read
# back through pinned memory and checked when their copy has landed -- at the next _sample call or by sampler_failures() -- so that
# sampling never waits for the device
_pending_flags = []


def _check_sampler_flag(wait: bool = False):
    keep = []
    bad = False
    for ev, host, flag in _pending_flags:
        if wait:
            ev.synchronize()
        if ev.query():
            bad = bad or bool(int(host[0]))
        else:
            keep.append((ev, host, flag))
    _pending_flags[:] = keep
    if bad:
        raise RuntimeError("MatrixFisherN._sample: a rejection loop exhausted its 4096 proposals in an earlier call (utils/fisher.py:117-207 "
                           "oversamples 8x and retries; here the slot was left at the identity rotation) -- the parameter matrix is far outside "
                           "the range the ACG envelope covers")


def sampler_failures(wait: bool = True):
    """Raise if any earlier ``MatrixFisherN._sample`` call reported an exhausted rejection loop (waits for outstanding calls by default)."""
    _check_sampler_flag(wait)


class MatrixFisherN(torch.nn.Module):
    """MatrixFisherN(A [B,3,3], norm_type=1).  ``_log_prob(R [N,3,3])`` broadcasts row b over N/B consecutive samples."""

    def __init__(self, A, norm_type=1, approx_num=None):
        super().__init__()
        if norm_type not in (0, 1, 2):
            raise NotImplementedError("normaliser approximations 0, 1 (closed forms) and 2 (Monte-Carlo) are built; type 3 indexes ROWS of the "
                                      "[N,3] singular values and cannot serve MatrixFisherN's batched A (utils/fisher.py:102-113)")
        self.norm_type = int(norm_type)
        self.A = A.reshape(-1, 3, 3)
        if self.norm_type == 2:
            # utils/fisher.py:98-101: mean over approx_num uniform rotations, ONE matrix (the reference's broadcast); on the device with
            # a Philox stream keyed from torch's generator (statistical parity with pytorch3d.random_rotations)
            if approx_num is None or int(approx_num) <= 0:
                raise TypeError("norm_type=2 needs approx_num (the reference passes it to random_rotations, utils/fisher.py:99)")
            if self.A.shape[0] != 1:
                raise RuntimeError("norm_type=2 serves one matrix: the reference broadcasts [approx_num,3,3] against [N,3,3] (utils/fisher.py:100)")
            if not self.A.is_cuda:
                raise RuntimeError("rotationnormflow_amd runs on the GPU only (no CPU fallback): construct MatrixFisherN with A on the GPU")
            A32 = self.A.detach().to(torch.float32).contiguous()
            self._c = torch.empty(1, dtype=torch.float32, device=A32.device)
            scratch = torch.empty(1, dtype=torch.float64, device=A32.device)
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            with torch.cuda.device(A32.device):
                _lib.check(_lib.lib().rnf_fisher_log_const_mc(A32.data_ptr(), 1, int(approx_num), seed, scratch.data_ptr(), 8, self._c.data_ptr(),
                                                              torch.cuda.current_stream(A32.device).cuda_stream))
            self._norm = None
            return
        if self.A.is_cuda:
            # per-sample A from a network (agent.py:57-60): constants on the device, no host SVD and no device->host sync
            A32 = self.A.detach().to(torch.float32).contiguous()
            L = _lib.lib()
            self._c = torch.empty(A32.shape[0], dtype=torch.float32, device=A32.device)
            scratch = torch.empty(2, dtype=torch.float64, device=A32.device)
            with torch.cuda.device(A32.device):
                _lib.check(L.rnf_fisher_log_const_nt(A32.data_ptr(), A32.shape[0], self.norm_type, scratch.data_ptr(), 16, self._c.data_ptr(),
                                                     torch.cuda.current_stream(A32.device).cuda_stream))
            self._norm = None
        else:
            S = proper_singular_values(self.A)
            norm = _norm_from_singular_values(S, self.norm_type)
            self._norm = norm.to(dtype=self.A.dtype)
            self._c = (S.sum(-1) + norm.log()).to(torch.float32)      # log p = tr(A^T R) - c

    @property
    def norm(self):
        """The reference's ``self.norm`` (utils/fisher.py:215); computed lazily when A lives on the GPU (needs the singular values)."""
        if self._norm is None:
            S = proper_singular_values(self.A)
            if self.norm_type == 2:                          # c = sum S + log norm
                self._norm = (self._c.detach().to(S.device, torch.float64) - S.sum(-1)).exp().to(device=self.A.device, dtype=self.A.dtype)
            else:
                self._norm = _norm_from_singular_values(S, self.norm_type).to(device=self.A.device, dtype=self.A.dtype)
        return self._norm

    def log_const(self):
        return self._c

    def _log_prob(self, inputs, context=9):
        if not inputs.is_cuda:
            raise RuntimeError("rotationnormflow_amd runs on the GPU only (no CPU fallback)")
        if self.norm_type == 2 and torch.is_grad_enabled() and self.A.requires_grad:
            raise NotImplementedError("rotationnormflow_amd: no gradient w.r.t. A through the Monte-Carlo normaliser (norm_type=2); use 0 or 1")
        if inputs.shape[-1] == 4:
            if torch.is_grad_enabled() and inputs.requires_grad:
                raise NotImplementedError("rotationnormflow_amd: quaternion inputs are not differentiable here; pass rotation matrices")
            inputs = quaternion_to_matrix(inputs)
        dev = inputs.device
        A = self.A.detach().to(device=dev, dtype=torch.float32).contiguous()
        c = self._c.to(dev).contiguous()
        return _FisherLogProb.apply(inputs, self.A, A, c, self.norm_type)

    def log_prob(self, inputs, context=None):
        return self._log_prob(inputs)

    def _sample(self, num_samples, context=9):
        """[B, num_samples, 3, 3] rotations (context=9) or quaternions (context=4) ~ MF(A) on A's device.  Exact rejection
        sampler (Bingham through an ACG envelope, utils/fisher.py:117-207) on the GPU with a counter-based Philox stream whose
        key is drawn from torch's default generator, so ``torch.manual_seed`` makes it reproducible."""
        A = self.A
        if not A.is_cuda:
            raise RuntimeError("rotationnormflow_amd runs on the GPU only (no CPU fallback): construct MatrixFisherN with A on the GPU")
        dev = A.device
        U32, V32, _, lam32 = device_proper_svd(A)            # proper SVD + Bingham parameters on the device: no host LAPACK, no sync
        B = A.shape[0]
        out = torch.empty(B, num_samples, 3, 3, dtype=torch.float32, device=dev)
        _check_sampler_flag()                                # a failure of an EARLIER call surfaces here, without a device wait now
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().rnf_fisher_sample(U32.data_ptr(), V32.data_ptr(), lam32.data_ptr(), B, num_samples, seed,
                                                    out.data_ptr(), flag.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
            if not torch.cuda.is_current_stream_capturing():
                host = torch.zeros(1, dtype=torch.int32).pin_memory()
                host.copy_(flag, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                _pending_flags.append((ev, host, flag))
        if context == 9:
            return out
        if context == 4:                                     # utils/fisher.py:242-243: matrix_to_quaternion(result)
            quat = torch.empty(B, num_samples, 4, dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                _lib.check(_lib.lib().rnf_matrix_to_quaternion(out.data_ptr(), B * num_samples, quat.data_ptr(),
                                                               torch.cuda.current_stream(dev).cuda_stream))
            return quat
        return None

    def sample(self, num_samples, context=None):
        return self._sample(num_samples)


# ---- the reference's module-level helpers under their own names (utils/fisher.py:14-207) -----------------------------------------------
# MatrixFisherN above never calls these (it goes to the kernels directly); they exist so that code written against the reference's module
# finds the same names with the same argument meaning.  GPU tensors go through the device kernels; the SVD helpers also accept host tensors
# (torch's LAPACK, as the reference itself does) because they are plain linear algebra, not part of the density path.

def quat_to_rotmat(quat):
    """[B,4] (w, x, y, z), normalised first -> [B,3,3] (utils/fisher.py:14-46)."""
    return quaternion_to_matrix(quat)


def proper_svd_N(A):
    """[N,3,3] -> (U, S, V) with det(U) = det(V) = +1 and the last singular value carrying the sign (utils/fisher.py:67-76)."""
    A = A.reshape(-1, 3, 3)
    if A.is_cuda:
        U, V, s, _ = device_proper_svd(A)
        return U.to(A.dtype), s.to(A.dtype), V.to(A.dtype)
    U, S, Vh = torch.linalg.svd(A.detach())
    V = Vh.transpose(-1, -2).clone()
    U, S = U.clone(), S.clone()
    du, dv = torch.det(U), torch.det(V)
    U[:, :, 2] *= du[:, None]
    V[:, :, 2] *= dv[:, None]
    S[:, 2] *= du * dv
    return U, S, V


def proper_svd(A, clone=False):
    """One [3,3] matrix -> (U [3,3], S [3], V [3,3]) (utils/fisher.py:48-64; `clone` is accepted and has no effect: nothing aliases A here)."""
    U, S, V = proper_svd_N(A.reshape(1, 3, 3))
    return U[0], S[0], V[0]


def matrix_fisher_norm_N(A, type_approx=0, approx_num=17890714):
    """The approximated normalising constant of MF(A), A [N,3,3] -> [N] (utils/fisher.py:79-115): type 0 / 1 closed forms (type 0 keeps
    the reference's batch-global ``(S**2).sum()``), type 2 Monte-Carlo over `approx_num` uniform rotations (ONE matrix, on the GPU).
    Type 3 is refused, see MatrixFisherN."""
    A = A.reshape(-1, 3, 3)
    if type_approx in (0, 1):
        S = proper_svd_N(A)[1] if A.is_cuda else proper_singular_values(A).to(A.dtype)
        return _norm_from_singular_values(S, type_approx)
    if type_approx == 2:
        return MatrixFisherN(A, 2, approx_num).norm
    raise NotImplementedError("matrix_fisher_norm_N: type_approx 0, 1 and 2 are built (type 3: see MatrixFisherN)")


def sample_matrix_fisher(A, num_samples, b=1.5, oversampling_ratio=8):
    """[num_samples,3,3] rotations ~ MF(A) for ONE [3,3] matrix on the GPU (utils/fisher.py:175-207).  The device sampler is an exact
    per-sample rejection loop with the reference's envelope (b = 1.5): `oversampling_ratio` has nothing to size here, other `b` are refused."""
    if float(b) != 1.5:
        raise NotImplementedError("sample_matrix_fisher: the device sampler is built for the reference's envelope parameter b = 1.5")
    return MatrixFisherN(A.reshape(1, 3, 3))._sample(num_samples)[0]
