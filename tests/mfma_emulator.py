"""numpy emulation of the kernel's MFMA dataflow (test infrastructure).

Emulates v_mfma_f32_32x32x2_f32 with the lane maps of cdna_hip_programming.md section 3 and replays, from a PACKED layer record,
exactly the loads / operand choices that csrc/flow_kernels.h makes (mlp_head, last_tile).  Used on CPU to validate the
host packers and the fragment layout without a GPU.
"""
import numpy as np

LANES = np.arange(64)
J = LANES & 31
H = LANES >> 5

MOB_FIRST, MOB_HID, MOB_HB, MOB_HEAD, TILE_FLOATS, TILE_BIAS = 0, 256, 256 + 12288, 12736, 2080, 2048


# the fc_last rows of the segment weights' pre-activations are packed times log2(e) (csrc/layout.h S_PRESCALE); consumers multiply by ln 2
S_UNSCALE = 0.693147180559945309


def rho(r, h):
    return (r & 3) + 8 * (r >> 2) + 4 * h


def mfma(a, b, acc):
    """a, b: [64] per-lane operands; acc: [16, 64] per-lane accumulator registers.  D = A.B + C."""
    A = np.zeros((32, 2), np.float64)
    B = np.zeros((2, 32), np.float64)
    A[J, H] = a
    B[H, J] = b
    D = A @ B                                   # [32, 32]
    out = acc.copy()
    for r in range(16):
        out[r] += D[rho(r, H), J]
    return out


def bias16(rec, off):
    """load_bias16: 16 floats per lane-half starting at off + h*16."""
    acc = np.zeros((16, 64))
    for r in range(16):
        acc[r] = rec[off + H * 16 + r]
    return acc


def gemm_tile64(rec, off, tin, acc, relu):
    """gemm_tile64: rec[off + (tg*64 + lane)*4 + c], B operand = register (tg&3)*4+c of tile tg>>2."""
    for tg in range(8):
        a4 = rec[off + (tg * 64 + LANES)[:, None] * 4 + np.arange(4)[None, :]]      # [64, 4]
        t, r0 = tg >> 2, (tg & 3) * 4
        for c in range(4):
            b = tin[t][r0 + c]
            if relu:
                b = np.maximum(b, 0)
            acc = mfma(a4[:, c], b, acc)
    return acc


def mlp_head(rec, y, cinit=None):
    """y: [32,3] one wave's samples.  Returns tt (2 tiles of [16,64])."""
    y = np.asarray(y, np.float64)
    bA = np.where(H == 1, y[J, 1], y[J, 0])
    bB = np.where(H == 1, 1.0, y[J, 2])
    x0 = []
    for ot in range(2):
        a2 = rec[MOB_FIRST + ((ot * 64 + LANES)[:, None] * 2 + np.arange(2)[None, :])]
        c = np.zeros((16, 64)) if cinit is None else cinit[ot].copy()
        c = mfma(a2[:, 0], bA, c)
        c = mfma(a2[:, 1], bB, c)
        x0.append(c)
    hin = x0
    for L in range(3):
        hout = []
        for ot in range(2):
            c = bias16(rec, MOB_HB + (L * 2 + ot) * 32)
            hout.append(gemm_tile64(rec, MOB_HID + (L * 2 + ot) * 2048, hin, c, relu=True))
        hin = hout
    return [np.maximum(x0[t] + hin[t], 0) for t in range(2)]


def last_tile(rec, tau, tt):
    off = MOB_HEAD + tau * TILE_FLOATS
    return gemm_tile64(rec, off, tt, bias16(rec, off + TILE_BIAS), relu=False)


def conditioner_from_record(rec, y, K):
    """Full ConditionalTransform output [32, 4K] in the reference's row order, via the kernel's dataflow."""
    rec = np.asarray(rec, np.float64)
    tt = mlp_head(rec, y)
    out = np.zeros((32, 4 * K))
    for tau in range((K + 7) // 8):                  # K % 8 != 0: the last tile is zero padded, its pad rows must come out 0
        o = last_tile(rec, tau, tt)
        for g in range(4):
            for c in range(4):
                k = 8 * tau + 2 * g + H
                real = k < K
                assert np.abs(o[4 * g + c][~real]).max(initial=0.0) == 0.0
                row = np.where(c == 0, k, K + 3 * k + (c - 1))
                out[J[real], row[real]] = o[4 * g + c][real] * (S_UNSCALE if c == 0 else 1.0)      # each (sample, row): exactly one lane
    return out


def featproj_from_record(frec, feat, F):
    """featproj_kernel for one wave: feat [32,F] -> G fragments (2 tiles of [16,64])."""
    frec = np.asarray(frec, np.float64)
    ng = F // 8
    tiles = []
    for ot in range(2):
        acc = bias16(frec, 2 * ng * 256 + ot * 32)
        for u in range(ng):
            a4 = frec[((ot * ng + u) * 64 + LANES)[:, None] * 4 + np.arange(4)[None, :]]
            for c in range(4):
                b = feat[J, 8 * u + 4 * H + c]
                acc = mfma(a4[:, c], b, acc)
        tiles.append(acc)
    return tiles


# ---- split-precision (f16x2) dataflow: v_mfma_f32_32x32x16_f16 lane maps + the hi/lo operand split --------------------
def mfma_h(a8, b8, acc):
    """a8, b8: [64, 8] per-lane operands (values already fp16-representable); A[i][k=8h+j], B[k=8h+j][col]."""
    A = np.zeros((32, 16), np.float64)
    B = np.zeros((16, 32), np.float64)
    for j in range(8):
        A[J, 8 * H + j] = a8[:, j]
        B[8 * H + j, J] = b8[:, j]
    D = A @ B
    out = acc.copy()
    for r in range(16):
        out[r] += D[rho(r, H), J]
    return out


def split_act(x2, relu=True):
    """x2: two [16,64] register tiles -> (hi[4][64,8], lo[4][64,8]) as in flow_kernels.h split_act."""
    hi, lo = [], []
    for s in range(4):
        vals = np.stack([x2[s >> 1][8 * (s & 1) + j] for j in range(8)], axis=1).astype(np.float32)     # [64, 8]
        if relu:
            vals = np.maximum(vals, 0)
        h = vals.astype(np.float16)
        l = (vals - h.astype(np.float32)).astype(np.float16)       # unscaled: lands in the fp16 subnormal range, floor 2^-24
        hi.append(h.astype(np.float64))
        lo.append(l.astype(np.float64))
    return hi, lo


def gemm_tile64_h(rec32, off_floats, act, acc):
    """rec32: the packed record as float32 array; weights at float offset `off_floats`: [4 s][hi,lo][64 lanes] 8 x fp16.
    One accumulator takes all three products (hi.hi, hi.lo, lo.hi) as in flow_kernels.h gemm_tile64_h."""
    halves = rec32[off_floats: off_floats + 2048].view(np.float16).astype(np.float64).reshape(4, 2, 64, 8)
    hi, lo = act
    for s in range(4):
        acc = mfma_h(halves[s, 0], hi[s], acc)
        acc = mfma_h(halves[s, 0], lo[s], acc)
        acc = mfma_h(halves[s, 1], hi[s], acc)
    return acc


def conditioner_from_record_h(rec32, y, K, cinit=None):
    """ConditionalTransform output [32, 4K] through the split-precision dataflow of Mlp<1> (fc_first stays fp32).
    cinit: feature-projection fragments (featproj_from_record_h) that initialise the fc_first accumulator of a conditional layer."""
    rec32 = np.ascontiguousarray(rec32, dtype=np.float32)
    rec = rec32.astype(np.float64)
    y = np.asarray(y, np.float64)
    bA = np.where(H == 1, y[J, 1], y[J, 0])
    bB = np.where(H == 1, 1.0, y[J, 2])
    x0 = []
    for ot in range(2):
        a2 = rec[MOB_FIRST + ((ot * 64 + LANES)[:, None] * 2 + np.arange(2)[None, :])]
        c = mfma(a2[:, 0], bA, np.zeros((16, 64)) if cinit is None else cinit[ot].copy())
        x0.append(mfma(a2[:, 1], bB, c))
    act = split_act(x0)
    hcur = x0
    for L in range(3):
        hcur = [gemm_tile64_h(rec32, MOB_HID + (L * 2 + ot) * 2048, act, bias16(rec, MOB_HB + (L * 2 + ot) * 32)) for ot in range(2)]
        if L < 2:
            act = split_act(hcur)
    act = split_act([x0[t] + hcur[t] for t in range(2)])
    out = np.zeros((32, 4 * K))
    for tau in range((K + 7) // 8):
        off = MOB_HEAD + tau * TILE_FLOATS
        o = gemm_tile64_h(rec32, off, act, bias16(rec, off + TILE_BIAS))
        for g in range(4):
            for c in range(4):
                k = 8 * tau + 2 * g + H
                real = k < K
                row = np.where(c == 0, k, K + 3 * k + (c - 1))
                out[J[real], row[real]] = o[4 * g + c][real] * (S_UNSCALE if c == 0 else 1.0)
    return out


def featproj_from_record_h(frec32, feat, F):
    """featproj_kernel<PREC=1> for one wave: feat [32, F] (F % 8 == 0) -> G fragments (2 tiles of [16,64])."""
    frec32 = np.ascontiguousarray(frec32, dtype=np.float32)
    ns = (F + 15) // 16
    tiles = []
    for ot in range(2):
        acc1 = bias16(frec32.astype(np.float64), 2 * ns * 512 + ot * 32)
        acc2 = np.zeros((16, 64))
        halves = frec32[ot * ns * 512:(ot + 1) * ns * 512].view(np.float16).astype(np.float64).reshape(ns, 2, 64, 8)
        for s in range(ns):
            k = 16 * s + 8 * H[:, None] + np.arange(8)[None, :]                      # [64, 8] feature index per lane/elem
            vals = np.where(k < F, feat[J[:, None], np.minimum(k, F - 1)], 0.0).astype(np.float32)
            bh = vals.astype(np.float16)
            bl = ((vals - bh.astype(np.float32)) * np.float32(4096.0)).astype(np.float16)
            acc1 = mfma_h(halves[s, 0], bh.astype(np.float64), acc1)
            acc2 = mfma_h(halves[s, 0], bl.astype(np.float64), acc2)
            acc2 = mfma_h(halves[s, 1], bh.astype(np.float64), acc2)
        tiles.append(acc1 + acc2 / 4096.0)
    return tiles
