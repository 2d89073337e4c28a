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
    """f16x2 feature-projection image (k-steps of 16, zero padded) reproduces W0[:, 3:] f; the bias b0 sits -- with the same per-row power of
    two -- in the bias slot of the fc_first image of the layer record (round 5: the projection kernels start their accumulators at zero)."""
    K = 8
    m = _filled_mlp(3 + F, 4 * K, seed=200 + F)
    rec, frec = runtime.pack_mobius(_lib.lib(), m, K, F, _lib.PREC_F16X2)
    feat = synth.features(32, F, seed=9)
    g = emu.featproj_from_record_h(frec, feat.astype(np.float64), F)
    W0 = m.fc_first.weight.detach().double().numpy()
    want = feat.astype(np.float64) @ W0[:, 3:].T                                                  # [32, 64]
    b0 = m.fc_first.bias.detach().double().numpy()
    # the packer scales row i of the projection (with x0_i) by a power of two (csrc/equalize.h): recover it per row, then compare
    got = np.zeros_like(want)
    for ot in range(2):
        for r in range(16):
            got[emu.J, 32 * ot + emu.rho(r, emu.H)] = g[ot][r]
    for i in range(64):
        k = np.argmax(np.abs(want[:, i]))
        e = np.log2(got[k, i] / want[k, i])
        assert abs(e - round(e)) < 1e-4, (i, e)
        assert np.abs(got[:, i] * 2.0 ** -round(e) - want[:, i]).max() < 3e-6 * max(1.0, np.abs(want).max())
        ot, lane = i // 32, 32 + i % 32                           # fc_first image: float2 per lane, element 1 of lane-half 1 = the bias slot
        assert abs(float(rec[(ot * 64 + lane) * 2 + 1]) * 2.0 ** -round(e) - b0[i]) < 1e-6 * max(1.0, abs(b0[i])), i


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


# ---- pack-time equalisation of the split-precision images (csrc/equalize.h) --------------------------------------------
def _relu_rescale(m, c0, c1, c2):
    """Move a ConditionalTransform along its ReLU-rescaling orbit (same function, flow/condition.py:24-30): x0 -> c0 x0, x1 -> c1 x1,
    x2 -> c2 x2, x3 -> c0 x3 (the residual ties x3 to x0), fc_last undoes c0.  c* are scalars or per-unit vectors [64]."""
    c0, c1, c2 = (torch.as_tensor(c, dtype=torch.float32) * torch.ones(64) for c in (c0, c1, c2))
    with torch.no_grad():
        m.fc_first.weight.mul_(c0[:, None]); m.fc_first.bias.mul_(c0)
        m.layers[1].weight.mul_(c1[:, None] / c0[None, :]); m.layers[1].bias.mul_(c1)
        m.layers[3].weight.mul_(c2[:, None] / c1[None, :]); m.layers[3].bias.mul_(c2)
        m.layers[5].weight.mul_(c0[:, None] / c2[None, :]); m.layers[5].bias.mul_(c0)
        m.fc_last.weight.div_(c0[None, :])
    return m


@pytest.mark.parametrize("s", [2.0 ** -8, 2.0 ** -4, 2.0 ** 6])
def test_split_precision_record_is_invariant_under_power_of_two_relu_rescaling(s):
    """The verdict's counter-example: hidden layer 1 x s, layer 2 x s, layer 3 x s^-2 (identical function).  Without the equalisation
    the unscaled fp16 lo terms lose 8 bits at s = 2^-8 (conditioner error 9e-3); with it the packer lands on the SAME canonical record."""
    K = 64
    base = _filled_mlp(3, 4 * K, seed=77)
    rec1, _ = runtime.pack_mobius(_lib.lib(), base, K, 0, _lib.PREC_F16X2)
    m = _relu_rescale(_filled_mlp(3, 4 * K, seed=77), 1.0, s, s * s)
    rec, _ = runtime.pack_mobius(_lib.lib(), m, K, 0, _lib.PREC_F16X2)
    assert np.array_equal(rec.view(np.uint32), rec1.view(np.uint32))
    y = synth.uniform_rotations(32, seed=3)[:, :, 1]
    want = _oracle_mlp(base, y)
    assert np.abs(emu.conditioner_from_record_h(rec, y, K) - want).max() < 2e-6 * max(1.0, np.abs(want).max())
    assert _lib.lib().rnf_last_pack_audit() < 1e-6


def test_split_precision_survives_per_unit_imbalance_with_features():
    """Per-unit, non-power-of-two orbit elements over 12 binades, conditional layer (feature projection rows scale with x0)."""
    K, F = 16, 40
    rng = np.random.default_rng(5)
    base = _filled_mlp(3 + F, 4 * K, seed=81)
    m = _relu_rescale(_filled_mlp(3 + F, 4 * K, seed=81), *(2.0 ** rng.uniform(-8, 4, 64) for _ in range(3)))
    rec, frec = runtime.pack_mobius(_lib.lib(), m, K, F, _lib.PREC_F16X2)
    assert _lib.lib().rnf_last_pack_audit() < 1.5e-6
    y = synth.uniform_rotations(32, seed=4)[:, :, 2]
    feat = synth.features(32, F, seed=9)
    g = emu.featproj_from_record_h(frec, feat.astype(np.float64), F)
    got = emu.conditioner_from_record_h(rec, y, K, cinit=g)
    want = _oracle_mlp(base, np.concatenate([y, feat], axis=1))
    want_m = _oracle_mlp(m, np.concatenate([y, feat], axis=1))
    assert np.abs(want_m - want).max() < 1e-4 * np.abs(want).max()             # same function (fp32 rounding of the rescaled weights)
    assert np.abs(got - want_m).max() < 3e-6 * max(1.0, np.abs(want_m).max())


def test_audit_refuses_an_unequalised_imbalanced_layer():
    """rnf_set_equalize(0) splits the weights as given: the pack-time audit must then refuse the s = 2^-8 layer (return code 2 ->
    HalfRangeError -> the flow is packed for the exact-fp32 kernels), and accept the balanced one."""
    L = _lib.lib()
    old = L.rnf_set_equalize(0)
    try:
        runtime.pack_mobius(L, _filled_mlp(3, 256, seed=77), 64, 0, _lib.PREC_F16X2)
        assert L.rnf_last_pack_audit() < 1e-6
        with pytest.raises(runtime.HalfRangeError):
            runtime.pack_mobius(L, _relu_rescale(_filled_mlp(3, 256, seed=77), 1.0, 2.0 ** -8, 2.0 ** -16), 64, 0, _lib.PREC_F16X2)
        assert L.rnf_last_pack_audit() > 1e-4
    finally:
        L.rnf_set_equalize(old)


def test_feature_mean_square_is_an_input_of_the_equalisation():
    """rnf_set_feature_ms: a layer whose features are 40x larger than unit scale (and whose feature weights are 40x smaller).  Told the
    features' mean square, the packer lands on (almost) the record of the unit-scale twin -- the same canonical point of the orbit; left at
    the default it scales x0 up by 2^5 too much and pushes fc_last's columns 2^5 towards the fp16 floor."""
    K, F = 16, 40
    L = _lib.lib()
    base = _filled_mlp(3 + F, 4 * K, seed=91)
    twin = _filled_mlp(3 + F, 4 * K, seed=91)
    with torch.no_grad():
        twin.fc_first.weight[:, 3:] /= 40.0
    rec0, frec0 = runtime.pack_mobius(L, base, K, F, _lib.PREC_F16X2)
    old = L.rnf_set_feature_ms(1600.0)
    try:
        rec1, frec1 = runtime.pack_mobius(L, twin, K, F, _lib.PREC_F16X2)
    finally:
        L.rnf_set_feature_ms(old)
    rec2, _ = runtime.pack_mobius(L, twin, K, F, _lib.PREC_F16X2)                       # default: mean square 1
    tile0 = slice(emu.MOB_HEAD, emu.MOB_HEAD + emu.TILE_BIAS)                            # first fc_last tile: fp16 hi / lo images
    last0 = rec0[tile0].view(np.float16).astype(np.float64)
    last2 = rec2[tile0].view(np.float16).astype(np.float64)
    assert np.array_equal(rec0[emu.MOB_HID:], rec1[emu.MOB_HID:])                        # hidden layers and fc_last: the same canonical images
    assert np.abs(last2).max() <= np.abs(last0).max() / 2                               # uncalibrated: x0 scaled up, fc_last's columns scaled down
    assert L.rnf_set_feature_ms(1.0) == 1.0                                             # restored; out-of-range values fall back to 1
    L.rnf_set_feature_ms(float("nan"))
    assert L.rnf_set_feature_ms(1.0) == 1.0
    # and the calibrated record reproduces the oracle on 40x features
    y = synth.uniform_rotations(32, seed=4)[:, :, 2]
    feat = synth.features(32, F, seed=9) * np.float32(40.0)
    g = emu.featproj_from_record_h(frec1, feat.astype(np.float64), F)
    got = emu.conditioner_from_record_h(rec1, y, K, cinit=g)
    want = _oracle_mlp(twin, np.concatenate([y, feat], axis=1))
    assert np.abs(got - want).max() < 3e-6 * max(1.0, np.abs(want).max())
