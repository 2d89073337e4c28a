"""CPU: csrc/svd4_lapack.h (the dense-SVD path of LAPACK restated for 4x4, run per sample on the device by ConditionRot) reproduces
``torch.svd``'s SIGN conventions: U^T V -- which is not a function of the matrix alone (flow/rottrans.py:37-66) -- equals the reference's."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "host_svd4.cpp")
OUT = os.path.join(HERE, "csrc", "_host_svd4.so")
HDR = os.path.join(os.path.dirname(HERE), "rotationnormflow_amd", "csrc", "svd4_lapack.h")


@pytest.fixture(scope="module")
def hs():
    if not os.path.exists(OUT) or os.path.getmtime(OUT) < max(os.path.getmtime(SRC), os.path.getmtime(HDR)):
        subprocess.run(["/opt/rocm/bin/hipcc", "-x", "hip", "--cuda-host-only", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", OUT, SRC], check=True)
    return C.CDLL(OUT)


def _utv(hs, M):
    A = np.ascontiguousarray(M.reshape(-1, 16), dtype=np.float32)
    rot = np.empty_like(A)
    sv = np.empty((A.shape[0], 4), np.float32)
    bad = hs.hs_utv(A.ctypes.data_as(C.c_void_p), rot.ctypes.data_as(C.c_void_p), sv.ctypes.data_as(C.c_void_p), A.shape[0])
    return rot.reshape(-1, 4, 4), sv, bad


@pytest.mark.parametrize("spread", [0.05, 0.5, 3.0])
def test_utv_has_lapacks_signs(hs, spread):
    """I + spread * N(0,1): near-identity (the layer at initialisation), the trained-like regime of the fixtures, and far from identity."""
    torch.manual_seed(int(spread * 100))
    M = (torch.eye(4) + spread * torch.randn(20000, 4, 4)).float()
    U, S, V = torch.svd(M)
    want = (U.transpose(1, 2) @ V).numpy()
    got, sv, bad = _utv(hs, M.numpy())
    assert bad == 0
    assert np.abs(sv - S.numpy()).max() < 2e-5 * max(1.0, float(S.max()))
    err = np.abs(got - want).reshape(len(got), -1).max(1)
    # Identical rotation for >= 99.7 % of the matrices (measured 99.8 - 99.9 %).  The rest are razor-edge decisions of the QR iteration
    # (sweep direction, deflation and shift tests compare quantities that differ by rounding between two implementations of the same
    # algorithm -- MKL against netlib against this header, each pair ~0.1 % apart): another, equally valid sign pattern, as between the
    # reference's own fp32 and fp64 runs.
    assert np.mean(err < 1e-4) > 0.997, np.mean(err < 1e-4)
    # orthogonality of the result
    assert np.abs(np.einsum("nij,nkj->nik", got, got) - np.eye(4)).max() < 1e-5


def test_utv_on_degenerate_inputs(hs):
    """Identity, a diagonal matrix with a negative entry, a rank-deficient matrix: finite, orthogonal, LAPACK's answer where it is unique."""
    M = np.stack([np.eye(4), np.diag([2.0, -1.0, 0.5, 3.0]), np.diag([1.0, 1.0, 0.0, 2.0]) + 0 * np.eye(4)]).astype(np.float32)
    got, sv, bad = _utv(hs, M)
    assert bad == 0 and np.isfinite(got).all()
    U, S, V = torch.svd(torch.from_numpy(M))
    want = (U.transpose(1, 2) @ V).numpy()
    assert np.abs(got[:2] - want[:2]).max() < 1e-6
    assert np.abs(np.einsum("nij,nkj->nik", got, got) - np.eye(4)).max() < 1e-5
