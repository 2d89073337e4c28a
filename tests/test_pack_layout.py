"""CPU: the host packers + the kernel's fragment dataflow (numpy MFMA emulator) reproduce the oracle's conditioner."""
import numpy as np
import pytest
import torch

from oracle import flow_oracle as orc
from rotationnormflow_amd import _lib, runtime, synth
from rotationnormflow_amd.flow.condition import ConditionalTransform
from tests import mfma_emulator as emu


def _filled_mlp(ni, no, seed, gain=3.0):
    torch.manual_seed(seed)
    m = ConditionalTransform(ni, no)
    with torch.no_grad():
        for p in m.parameters():
            p.mul_(gain if p.dim() == 2 else 1.0).add_(0.05 * torch.randn_like(p))
    return m


def _oracle_mlp(m, x):
    p = {"c." + k: v.detach().double() for k, v in m.state_dict().items()}
    return orc.conditioner(torch.as_tensor(x, dtype=torch.float64), p, "c").numpy()


@pytest.mark.parametrize("K", [8, 64, 128, 10, 20])
def test_unconditional_mobius_record_matches_oracle(K):
    m = _filled_mlp(3, 4 * K, seed=K)
    rec, frec = runtime.pack_mobius(_lib.lib(), m, K, 0)
    assert frec is None and rec.size == 12736 + ((K + 7) // 8) * 2080          # K % 8 != 0: last tile zero padded
    y = synth.uniform_rotations(32, seed=3)[:, :, 1]
    got = emu.conditioner_from_record(rec, y, K)
    want = _oracle_mlp(m, y)
    assert np.abs(got - want).max() < 1e-5 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("F", [8, 40, 256])
def test_conditional_mobius_record_and_feature_projection(F):
    K = 16
    m = _filled_mlp(3 + F, 4 * K, seed=100 + F)
    rec, frec = runtime.pack_mobius(_lib.lib(), m, K, F)
    Fp = runtime.pad8(F)
    assert frec.size == _lib.lib().rnf_featproj_packed_floats(Fp) >= 2 * (Fp // 8) * 256 + 64
    y = synth.uniform_rotations(32, seed=4)[:, :, 2]
    feat = synth.features(32, F, seed=9)
    featp = np.zeros((32, Fp), np.float32)
    featp[:, :F] = feat
    g = emu.featproj_from_record(frec, featp.astype(np.float64), Fp)
    tt = emu.mlp_head(np.asarray(rec, np.float64), y, cinit=g)
    out = np.zeros((32, 4 * K))
    for tau in range(K // 8):
        o = emu.last_tile(np.asarray(rec, np.float64), tau, tt)
        for gi in range(4):
            for c in range(4):
                k = 8 * tau + 2 * gi + emu.H
                out[emu.J, np.where(c == 0, k, K + 3 * k + (c - 1))] = o[4 * gi + c] * (emu.S_UNSCALE if c == 0 else 1.0)
    want = _oracle_mlp(m, np.concatenate([y, feat], axis=1))
    assert np.abs(out - want).max() < 1e-5 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("F", [8, 40, 256])
def test_split_precision_feature_projection(F):
    """f16x2 feature-projection image (k-steps of 16, zero padded) reproduces W0[:, 3:] f + b0."""
    K = 8
    m = _filled_mlp(3 + F, 4 * K, seed=200 + F)
    rec, frec = runtime.pack_mobius(_lib.lib(), m, K, F, _lib.PREC_F16X2)
    feat = synth.features(32, F, seed=9)
    g = emu.featproj_from_record_h(frec, feat.astype(np.float64), F)
    W0 = m.fc_first.weight.detach().double().numpy()
    want = feat.astype(np.float64) @ W0[:, 3:].T + m.fc_first.bias.detach().double().numpy()      # [32, 64]
    for ot in range(2):
        for r in range(16):
            rows = 32 * ot + emu.rho(r, emu.H)
            assert np.abs(g[ot][r] - want[emu.J, rows]).max() < 3e-6 * max(1.0, np.abs(want).max())


def test_cond16_record_yields_matrix_rows_on_lane_halves():
    F = 24
    m = _filled_mlp(F, 16, seed=7, gain=1.0)
    rec, frec = runtime.pack_cond16(_lib.lib(), m, F)
    feat = synth.features(32, F, seed=5).astype(np.float64)
    g = emu.featproj_from_record(frec, feat, F)
    tt = emu.mlp_head(np.asarray(rec, np.float64), np.zeros((32, 3)), cinit=g)
    o = emu.last_tile(np.asarray(rec, np.float64), 0, tt)
    want = _oracle_mlp(m, feat).reshape(32, 4, 4)
    for gi in range(2):
        for c in range(4):
            for h in range(2):
                lanes = emu.LANES[emu.H == h]
                assert np.abs(o[4 * gi + c][lanes] - want[emu.J[lanes], 2 * gi + h, c]).max() < 1e-6
    assert np.abs(o[8:]).max() == 0.0          # rows 16..31 are zero padding


def test_affine16_record():
    rng = np.random.default_rng(0)
    M = (np.eye(4) + 0.3 * rng.standard_normal((4, 4))).astype(np.float32)
    rec = runtime.pack_affine16(_lib.lib(), torch.from_numpy(M)[None])
    assert rec.size == 244
    assert np.array_equal(rec[:16], M.ravel())
    assert abs(rec[16] - np.log(abs(np.linalg.det(M.astype(np.float64))))) < 1e-6
    assert np.abs(rec[17:33].reshape(4, 4) - np.linalg.inv(M.astype(np.float64))).max() < 1e-6
    assert abs(rec[33] + rec[16]) < 1e-6
    # the 10x10 tables: [|q'|^2 ; |q'|^2 R'] = T [1 ; R] for q' = M q(R), checked against the quaternion formulas in fp64
    from oracle import flow_oracle as orc
    R = synth.uniform_rotations(64, seed=4).astype(np.float64)
    for off, mat in ((36, M.astype(np.float64)), (140, np.linalg.inv(M.astype(np.float64)))):
        # two halves of 52 floats: five rows of the table, log|det|, the orthogonal flag (so3_math.h affine16_table)
        halves = rec[off:off + 104].reshape(2, 52)
        T = halves[:, :50].reshape(10, 10).astype(np.float64)
        out = np.concatenate([np.ones((64, 1)), R.reshape(64, 9)], axis=1) @ T.T
        q = orc.matrix_to_quaternion(torch.from_numpy(R)).numpy() @ mat.T
        want = orc.quaternion_to_matrix(torch.from_numpy(q)).numpy()
        assert np.abs(out[:, 0] - (q * q).sum(1)).max() < 1e-6
        assert np.abs(out[:, 1:] / out[:, :1] - want.reshape(64, 9)).max() < 1e-6
        assert (halves[:, 50] == rec[16 if off == 36 else 33]).all() and (halves[:, 51] == 0.0).all()


@pytest.mark.parametrize("K", [8, 64, 20])
def test_split_precision_record_matches_oracle(K):
    """f16x2 image + the fp16 MFMA lane maps + hi/lo operand split reproduce the fp64 conditioner to ~2^-22."""
    m = _filled_mlp(3, 4 * K, seed=40 + K)
    rec, _ = runtime.pack_mobius(_lib.lib(), m, K, 0, _lib.PREC_F16X2)
    assert rec.size == 12736 + ((K + 7) // 8) * 2080
    y = synth.uniform_rotations(32, seed=3)[:, :, 1]
    got = emu.conditioner_from_record_h(rec, y, K)
    want = _oracle_mlp(m, y)
    scale = max(1.0, np.abs(want).max())
    assert np.abs(got - want).max() < 2e-6 * scale
    # and it is NOT just fp16: a single-term fp16 image would be ~1e-3 off
    rec32, _ = runtime.pack_mobius(_lib.lib(), m, K, 0, _lib.PREC_FP32)
    assert np.abs(emu.conditioner_from_record(rec32, y, K) - want).max() < 1e-5 * scale


def test_split_precision_refuses_weights_outside_fp16_range():
    m = _filled_mlp(3, 32, seed=2)
    with torch.no_grad():
        m.fc_last.weight[3, 5] = 7.0e4
    with pytest.raises(runtime.HalfRangeError):
        runtime.pack_mobius(_lib.lib(), m, 8, 0, _lib.PREC_F16X2)
    runtime.pack_mobius(_lib.lib(), m, 8, 0, _lib.PREC_FP32)             # exact path still packs


def test_pack_rejects_bad_sizes():
    m = _filled_mlp(3, 40, seed=1)
    with pytest.raises(ValueError):
        runtime.pack_mobius(_lib.lib(), m, 0, 0)
    L = _lib.lib()
    assert L.rnf_mobius_packed_floats(0) == -1 and L.rnf_featproj_packed_floats(12) == -1
    assert L.rnf_mobius_packed_floats(10) == 12736 + 2 * 2080
