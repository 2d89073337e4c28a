"""CPU: csrc/so3_math.h compiled for the host (tests/csrc/host_math.cpp) against float64 numpy / the oracle.
Checks the algebra the kernels use per sample: branch-free softplus, the [0,2pi) arctangent, small-range sincos, the
cofactor 4x4 inverse, calculate_16, and the 2-D in-plane Moebius layer against the oracle's 3-D formulation."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import flow_oracle as orc
from rotationnormflow_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "host_math.cpp")
OUT = os.path.join(HERE, "csrc", "_host_math.so")
HDR = os.path.join(os.path.dirname(HERE), "rotationnormflow_amd", "csrc", "so3_math.h")


@pytest.fixture(scope="module")
def hm():
    if not os.path.exists(OUT) or os.path.getmtime(OUT) < max(os.path.getmtime(SRC), os.path.getmtime(HDR)):
        subprocess.run(["/opt/rocm/bin/hipcc", "-x", "hip", "--cuda-host-only", "-O2", "-ffp-contract=off", "-shared", "-fPIC",
                        "-o", OUT, SRC], check=True)
    return C.CDLL(OUT)


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def test_softplus(hm):
    x = f32(np.concatenate([np.linspace(-30, 30, 20001), [-100, -87.5, 19.999, 20.0, 20.001, 88.0, 0.0, -0.0]]))
    y = np.empty_like(x)
    hm.hm_softplus(ptr(x), ptr(y), x.size)
    want = np.logaddexp(0.0, x.astype(np.float64))
    rel = np.abs(y - want) / np.maximum(want, 1e-300)
    assert rel[x > -80].max() < 3e-7
    assert np.abs(y - want)[x <= -80].max() < 1e-37                     # below the fp32 normal range: 0 vs denormal
    assert np.array_equal(y[x > 20.0], x[x > 20.0])                    # reference's threshold behaviour


def test_angle_0_2pi(hm):
    rng = np.random.default_rng(0)
    t = rng.uniform(0, 2 * np.pi, 200000)
    rad = np.exp(rng.uniform(-3, 3, t.size))
    y, x = f32(rad * np.sin(t)), f32(rad * np.cos(t))
    special = np.array([[0, 1], [1, 0], [0, -1], [-1, 0], [1, 1], [-1, 1], [-1, -1], [1, -1], [1e-20, -1], [-1e-20, -1]], np.float32)
    y, x = np.concatenate([y, special[:, 0]]), np.concatenate([x, special[:, 1]])
    o = np.empty_like(x)
    hm.hm_angle(ptr(y), ptr(x), ptr(o), x.size)
    want = np.arctan2(y.astype(np.float64), x.astype(np.float64))
    want = np.where(want < 0, want + 2 * np.pi, want)
    err = np.abs(o - want)
    err = np.minimum(err, 2 * np.pi - err)
    assert err.max() < 6e-7                                             # ~1 ulp at 2*pi
    assert (o >= 0).all() and (o <= np.float32(2 * np.pi)).all()


def test_sincos_small(hm):
    x = f32(np.linspace(-1.0, 8.0, 100001))
    s, c = np.empty_like(x), np.empty_like(x)
    hm.hm_sincos(ptr(x), ptr(s), ptr(c), x.size)
    assert np.abs(s - np.sin(x.astype(np.float64))).max() < 2e-7
    assert np.abs(c - np.cos(x.astype(np.float64))).max() < 2e-7


def test_inv4(hm):
    rng = np.random.default_rng(1)
    m = f32(np.eye(4)[None] + 0.4 * rng.standard_normal((500, 4, 4)))
    o, det = np.empty_like(m), np.empty(500, np.float32)
    hm.hm_inv4(ptr(m), ptr(o), ptr(det), 500)
    m64 = m.astype(np.float64)
    want_det = np.linalg.det(m64)
    ok = np.abs(want_det) > 0.05
    assert np.abs(det - want_det)[ok].max() < 1e-5
    assert np.abs(o - np.linalg.inv(m64))[ok].max() < 2e-4
    assert np.abs(np.einsum("nij,njk->nik", o.astype(np.float64), m64) - np.eye(4))[ok].max() < 5e-5


@pytest.mark.parametrize("entry", ["hm_affine16", "hm_affine16_table"])
def test_affine16_matches_oracle(hm, entry):
    """calculate_16 with a constant M: the quaternion form (conditional layers) and the packer's 10x10 table form (unconditional layers)."""
    rng = np.random.default_rng(2)
    M = f32(np.eye(4) + 0.25 * rng.standard_normal((4, 4)))
    R = synth.uniform_rotations(4096, seed=8)
    Ro, ldj = np.empty_like(R), np.empty(4096, np.float32)
    lad = float(np.log(abs(np.linalg.det(M.astype(np.float64)))))
    getattr(hm, entry)(ptr(M), C.c_float(lad), ptr(R), ptr(Ro), ptr(ldj), 4096)
    wR, wl = orc.affine16(torch.from_numpy(M).double()[None], torch.from_numpy(R).double())
    assert np.abs(Ro - wR.numpy()).max() < 2e-6
    assert np.abs(ldj - wl.numpy()).max() < 2e-6


@pytest.mark.parametrize("nm", [3, 6])
def test_gram_schmidt_layers_match_oracle(hm, nm):
    """calculate_9 / calculate_36 (3x3 and 6x6 ablation layers): kernel math (host build) against the oracle restatement."""
    rng = np.random.default_rng(nm)
    M = f32(np.eye(nm) + 0.3 * rng.standard_normal((nm, nm)))
    R = synth.uniform_rotations(4096, seed=9 + nm)
    Ro, ldj = np.empty_like(R), np.empty(4096, np.float32)
    hm.hm_gs(ptr(M), nm, ptr(R), ptr(Ro), ptr(ldj), 4096)
    fn = orc.gs9 if nm == 3 else orc.gs36
    wR, wl = fn(torch.from_numpy(M).double(), torch.from_numpy(R).double())
    assert np.abs(Ro - wR.numpy()).max() < 3e-6
    err = np.abs(ldj - wl.numpy())
    # the log amplifies fp32 rounding where the tangent volume collapses (|det| -> 0; ldj down to -10 with this harsh M): the
    # yardstick is the oracle's own fp32 evaluation against fp64 (p99 1.4e-5, max 2e-3 on the 6x6 case)
    ref32 = np.abs(fn(torch.from_numpy(M), torch.from_numpy(R))[1].numpy() - wl.numpy())
    assert np.quantile(err, 0.99) < max(2.0 * np.quantile(ref32, 0.99), 1e-5)
    assert err.max() < max(2.0 * ref32.max(), 1e-5)
    assert np.abs(np.exp(ldj.astype(np.float64)) - np.exp(wl.numpy())).max() < 1e-4
    assert np.abs(np.einsum("nji,njk->nik", Ro.astype(np.float64), Ro.astype(np.float64)) - np.eye(3)).max() < 2e-6


@pytest.mark.parametrize("perm_row", [0, 1, 2, 4])
def test_inplane_mobius_layer_matches_oracle_3d_formulation(hm, perm_row):
    n, K = 2048, 64
    cfg = orc.make_config(layers=1, rot="None", first_affine=0)
    w = synth.fill_state_dict(orc.state_shapes(cfg), seed=31 + perm_row, regime="trained")
    R = synth.uniform_rotations(n, seed=12)
    perm = orc.PERMUTE_ROWS[perm_row]
    p = {k: torch.from_numpy(v).double() for k, v in w.items()}
    R64 = torch.from_numpy(R).double()
    cond = orc.conditioner(R64[..., perm[1]], p, "layers.0.conditioner")
    wantR, wantl = orc.mobius_forward(R64, perm, None, p, "layers.0.conditioner", K)
    c32 = f32(cond.numpy())
    Ro, ldj = np.empty_like(R), np.empty(n, np.float32)
    hm.hm_mobius_forward(ptr(R), ptr(c32), K, perm_row, ptr(Ro), ptr(ldj), n)
    assert np.abs(ldj - wantl.numpy()).max() < 3e-6
    assert np.abs(Ro - wantR.numpy()).max() < 3e-6


def test_condrot_backward_formula_matches_autograd_of_the_svd():
    """flow/rottrans.py condrot_grad (the analytic backward of rot = U^T V, flow/rottrans.py:42-53 of the reference differentiates
    torch.svd): against fp64 autograd of torch.svd on the SAME factors, for random matrices near the identity (the layer's regime)."""
    import torch
    from rotationnormflow_amd.flow.rottrans import condrot_grad
    torch.manual_seed(3)
    n = 200
    M = (torch.eye(4)[None] + 0.3 * torch.randn(n, 4, 4)).double().requires_grad_(True)
    U, S, V = torch.svd(M)
    rot = U.transpose(-1, -2) @ V
    G = torch.randn(n, 4, 4).double()
    (rot * G).sum().backward()
    got = condrot_grad(rot.detach().reshape(n, 16), U.detach().reshape(n, 16), S.detach(), V.detach().transpose(-1, -2).reshape(n, 16), G.reshape(n, 16))
    want = M.grad.reshape(n, 16)
    gap = (S.detach()[:, :-1] - S.detach()[:, 1:]).min(1).values                  # the derivative is singular where two singular values meet
    ok = gap > 1e-3
    assert ok.sum() > 150
    err = (got - want).abs().amax(1) / want.abs().amax(1).clamp_min(1e-6)
    assert err[ok].max() < 1e-8


def test_reference_named_fisher_helpers_on_host_tensors():
    """utils/fisher.py's module-level helpers under their own names (host tensors: torch's LAPACK, as the reference does) against the
    oracle's restatements (oracle/flow_oracle.py, pinned to the reference)."""
    from rotationnormflow_amd.utils import fisher as F
    torch.manual_seed(3)
    A = torch.randn(5, 3, 3, dtype=torch.float64) * 3
    A[1] = -A[1] @ A[1].T                                  # a negative determinant: the last proper singular value is negative
    U, S, V = F.proper_svd_N(A)
    assert (U @ torch.diag_embed(S) @ V.transpose(-1, -2) - A).abs().max().item() < 1e-12
    assert (torch.det(U) - 1).abs().max().item() < 1e-12 and (torch.det(V) - 1).abs().max().item() < 1e-12
    assert torch.allclose(S, orc.proper_singular_values(A), atol=1e-12) and S[1, 2].item() < 0
    u, s, v = F.proper_svd(A[2])
    uo, so, vo = orc.proper_svd(A[2])
    assert torch.allclose(s, so, atol=1e-12) and torch.allclose(u @ torch.diag(s) @ v.T, A[2], atol=1e-12)
    for t in (0, 1):
        assert torch.allclose(F.matrix_fisher_norm_N(A.abs(), t), orc.fisher_norm(orc.proper_singular_values(A.abs()), t), rtol=1e-12)
    with pytest.raises(NotImplementedError):
        F.matrix_fisher_norm_N(A, 3)
    q = torch.randn(7, 4, dtype=torch.float64)
    assert torch.allclose(F.quat_to_rotmat(q), orc.quaternion_to_matrix(q / q.norm(dim=1, keepdim=True)), atol=1e-12)
