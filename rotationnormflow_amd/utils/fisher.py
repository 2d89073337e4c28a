"""Mirror of the reference's utils/fisher.py: ``MatrixFisherN`` (log-density on the GPU through rnf_fisher_log_prob).

log p(R) = tr(A^T R) - (s0+s1+s2) - log norm, norm = 1/sqrt(8 pi (s0+s1)(s1+s2)(s0+s2))   (utils/fisher.py:93-97,217-232)
with s the *proper* singular values of A (last one sign-flipped by det(U) det(V), utils/fisher.py:67-76).
"""
import math

import torch

from .. import _lib


def proper_singular_values(A):
    """[B,3,3] -> [B,3] (utils/fisher.py:67-76).  O(B) host-side parameter preprocessing in torch, fp64 internally."""
    A64 = A.detach().to("cpu", torch.float64)               # tiny [B,3,3]: LAPACK on the host (no rocSOLVER start-up cost)
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


class _FisherLogProb(torch.autograd.Function):
    """log p(R) = tr(A^T R) - c through rnf_fisher_log_prob; d/dR = A through rnf_fisher_log_prob_backward (training with a
    matrix-Fisher base differentiates it w.r.t. the flow's output rotation, agent.py:58-64)."""

    @staticmethod
    def forward(ctx, inputs, A, c):
        R = inputs.reshape(-1, 3, 3).to(torch.float32).contiguous()
        n, B = R.shape[0], A.shape[0]
        out = torch.empty(n, dtype=torch.float32, device=R.device)
        with torch.cuda.device(R.device):
            _lib.check(_lib.lib().rnf_fisher_log_prob(R.data_ptr(), n, A.data_ptr(), c.data_ptr(), B, out.data_ptr(),
                                                      torch.cuda.current_stream(R.device).cuda_stream))
        ctx.save_for_backward(A)
        ctx.in_shape, ctx.in_dtype = inputs.shape, inputs.dtype
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        (A,) = ctx.saved_tensors
        g = g.to(torch.float32).contiguous()
        n = g.shape[0]
        g_rot = torch.empty((n, 3, 3), dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            _lib.check(_lib.lib().rnf_fisher_log_prob_backward(g.data_ptr(), n, A.data_ptr(), A.shape[0], g_rot.data_ptr(),
                                                               torch.cuda.current_stream(g.device).cuda_stream))
        return g_rot.reshape(ctx.in_shape).to(ctx.in_dtype), None, None


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
        if norm_type != 1:
            raise NotImplementedError("only the default normaliser approximation norm_type=1 is built (utils/fisher.py:93-97)")
        self.A = A.reshape(-1, 3, 3)
        if self.A.is_cuda:
            # per-sample A from a network (agent.py:57-60): constants on the device, no host SVD and no device->host sync
            A32 = self.A.detach().to(torch.float32).contiguous()
            self._c = torch.empty(A32.shape[0], dtype=torch.float32, device=A32.device)
            with torch.cuda.device(A32.device):
                _lib.check(_lib.lib().rnf_fisher_log_const(A32.data_ptr(), A32.shape[0], self._c.data_ptr(),
                                                           torch.cuda.current_stream(A32.device).cuda_stream))
            self._norm = None
        else:
            S = proper_singular_values(self.A)
            norm = 1.0 / torch.sqrt(8 * math.pi * (S[:, 0] + S[:, 1]) * (S[:, 2] + S[:, 1]) * (S[:, 0] + S[:, 2]))
            self._norm = norm.to(dtype=self.A.dtype)
            self._c = (S.sum(-1) + norm.log()).to(torch.float32)      # log p = tr(A^T R) - c

    @property
    def norm(self):
        """The reference's ``self.norm`` (utils/fisher.py:215); computed lazily when A lives on the GPU (needs the singular values)."""
        if self._norm is None:
            S = proper_singular_values(self.A)
            self._norm = (1.0 / torch.sqrt(8 * math.pi * (S[:, 0] + S[:, 1]) * (S[:, 2] + S[:, 1]) * (S[:, 0] + S[:, 2]))).to(
                device=self.A.device, dtype=self.A.dtype)
        return self._norm

    def log_const(self):
        return self._c

    def _log_prob(self, inputs, context=9):
        if not inputs.is_cuda:
            raise RuntimeError("rotationnormflow_amd runs on the GPU only (no CPU fallback)")
        if torch.is_grad_enabled() and self.A.requires_grad:
            raise NotImplementedError("rotationnormflow_amd: the gradient of the matrix-Fisher density w.r.t. A (it needs the derivative of "
                                      "the normaliser) is not built; detach A (the reference trains the flow on a frozen A, agent.py:58-60)")
        if inputs.shape[-1] == 4:
            if torch.is_grad_enabled() and inputs.requires_grad:
                raise NotImplementedError("rotationnormflow_amd: quaternion inputs are not differentiable here; pass rotation matrices")
            inputs = quaternion_to_matrix(inputs)
        dev = inputs.device
        A = self.A.detach().to(device=dev, dtype=torch.float32).contiguous()
        c = self._c.to(dev).contiguous()
        return _FisherLogProb.apply(inputs, A, c)

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
        A64 = A.detach().to("cpu", torch.float64)           # host LAPACK for the tiny SVDs, as in proper_singular_values
        U, S, Vh = torch.linalg.svd(A64)
        V = Vh.transpose(-1, -2)
        dU, dV = torch.det(U), torch.det(V)
        U = U.clone(); V = V.clone(); S = S.clone()
        U[:, :, 2] *= dU[:, None]                                   # proper SVD, utils/fisher.py:53-64
        V[:, :, 2] *= dV[:, None]
        S[:, 2] *= dU * dV
        lam = torch.stack([torch.zeros_like(S[:, 0]), 2 * (S[:, 1] + S[:, 2]), 2 * (S[:, 0] + S[:, 2]), 2 * (S[:, 0] + S[:, 1])], -1)
        U32, V32, lam32 = (t.to(dev, torch.float32).contiguous() for t in (U, V, lam))
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
        if context == 4:
            raise NotImplementedError("context=4 (quaternion output) is not built; convert with your own matrix_to_quaternion")
        return None

    def sample(self, num_samples, context=None):
        return self._sample(num_samples)
