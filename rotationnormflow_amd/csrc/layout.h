// layout.h -- packed-parameter layout shared by the host packers (api) and the gfx950 kernels.
//
// All GEMMs of the conditioner MLP (flow/condition.py:24-30) run on v_mfma_f32_32x32x2_f32 in the orientation
//     D[out feature i][sample j] += A[i][k] * B[k][j]        (A = weights, B = activations)
// Lane l = (j = l & 31, h = l >> 5).  Hardware maps (cdna_hip_programming.md section 3):
//     A operand: lane (i,h) supplies A[i][k = h]      B operand: lane (j,h) supplies B[k = h][j]
//     D/C      : lane (j,h), register r (0..15) holds D[row = rho(r,h)][col = j],  rho(r,h) = (r&3) + 8*(r>>2) + 4*h
// so a 64-feature activation is two f32x16 tiles per lane and register r of tile t holds feature 32t + rho(r,h).
// Feeding that register as the B operand of k-step (t,r) of the next layer therefore needs NO lane movement as
// long as the A operand of that step is W[out][32t + rho(r,h)].  Four consecutive steps r = 4g..4g+3 read
// W[out][32t + 8g + 4h + 0..3]: one contiguous float4 of the row-major weight row, so the LDS image of a
// [OUT][64] weight matrix is, per 32-row out tile `ot` and per k-group tg = 4t+g (0..7):
//     float4 image[ot][tg][lane] = W[32*ot + (lane&31)][8*tg + 4*(lane>>5) + 0..3]
// read with one conflict-free ds_read_b128 per 4 MFMAs.
//
// fc_first (3 inputs + bias = K of 4 = two k-steps): float2 per lane, image[ot][lane] = (W0[o][h], h ? b0[o] : W0[o][2])
// with the B operand (h ? y1 : y0) then (h ? 1 : y2).  For a conditional layer the bias slot holds 0 and the bias is
// folded into the feature projection G = W0[:,3:] f + b0 that initialises the accumulator.
//
// Bias of a later layer initialises the accumulator: bias image[ot][h][r] = b[32*ot + rho(r,h)] (16 floats per half).
//
// fc_last rows are permuted so that every lane ends up with whole segments: packed row P = 32*tau + 8g + 4h + c
// (tile tau, register 4g+c of lane-half h) is segment k = 8*tau + 2g + h, component c (0: raw weight, 1..3: w_k[c-1]),
// i.e. reference row (c == 0 ? k : K + 3k + c - 1)   (flow/mobiusflow.py:58-61).
#pragma once
#include <stdint.h>
#if defined(__HIPCC__)
#define RNF_LAYOUT_INLINE __host__ __device__ inline constexpr
#else
#define RNF_LAYOUT_INLINE inline constexpr
#endif

namespace rnf {

// split-precision operand pairs x = hi + lo / SCALE (flow_kernels.h): the 64-wide conditioner GEMMs keep lo UNSCALED (it lives in
// the fp16 subnormal range, absolute resolution 2^-24) so that all three products accumulate into ONE MFMA accumulator; the
// feature projection scales lo by 2^12 and combines two accumulators.
// The fc_last rows (and biases) of a Moebius layer that produce the segment weights' pre-activations s_k are PACKED pre-multiplied by
// log2(e): the forward split-precision segment evaluates softplus(s) / ln 2 = log2(1 + 2^(s log2 e)) and saves the multiply (one VALU
// instruction per segment on the VALU-issue-bound kernel); every other consumer multiplies the matrix output by ln 2 first.
constexpr float S_PRESCALE = 1.44269504088896341f, S_UNSCALE = 0.693147180559945309f;
constexpr float W_LO_SCALE = 1.0f;
constexpr float FEAT_LO_SCALE = 4096.0f;


constexpr int HID = 64;                   // flow/condition.py:9
constexpr int WAVE = 64;
constexpr int TILE_SAMPLES = 32;          // samples per wave (one MFMA column tile)

// ---- Moebius layer image (floats) ----
constexpr int MOB_FIRST = 0;                              // [2][64] float2                      = 256
constexpr int MOB_FIRST_FLOATS = 2 * 64 * 2;
constexpr int MOB_HID = MOB_FIRST + MOB_FIRST_FLOATS;     // [3][2][8][64] float4                = 12288
constexpr int MOB_HID_FLOATS = 3 * 2 * 8 * 64 * 4;
constexpr int MOB_HB = MOB_HID + MOB_HID_FLOATS;          // [3][2][2][16]                       = 192
constexpr int MOB_HB_FLOATS = 3 * 2 * 2 * 16;
constexpr int MOB_HEAD_FLOATS = MOB_HB + MOB_HB_FLOATS;   // 12736: everything before fc_last
constexpr int MOB_LAST = MOB_HEAD_FLOATS;                 // per tile tau: [8][64] float4 (2048) + bias [2][16] (32)
constexpr int MOB_LAST_TILE_FLOATS = 8 * 64 * 4 + 32;     // 2080
constexpr int MOB_LAST_TILE_BIAS = 8 * 64 * 4;            // bias offset inside a tile record
constexpr int MOB_MAX_TILES_IN_LDS = 8;                   // fc_last tiles resident at once (K = 64 -> all of them)

inline constexpr int64_t mobius_packed_floats(int K) { return MOB_HEAD_FLOATS + (int64_t)((K + 7) / 8) * MOB_LAST_TILE_FLOATS; }   // last tile zero padded

// ---- the same record per arithmetic (round 6) ----
// PREC 0 (exact fp32) and 1 (fp16 hi + lo pairs) store a [32 x 64] weight tile in 2048 floats (the constants above).  PREC 2 ("bf16x3"):
// every weight as THREE bf16 terms hi + mid + lo (24 significant bits, fp32's exponent range: no equalisation, no audit, no calibration, no
// range guard), tile image [k-step s (4)][hi, mid, lo][lane (64)] 8 x bf16 = 3072 floats, element j of lane (i, h) of k-step s = the weight
// the fp16 image holds there.  A K = 64 layer is then 43,712 floats = 171 KiB -- more than the 160 KiB of LDS -- so these kernels stage
// synchronously and keep four fc_last tiles resident at a time (flow_kernels.h).
template <int PREC>
struct Lay {
    static constexpr int W_TILE = PREC == 2 ? 4 * 3 * 64 * 4 : 8 * 64 * 4;
    static constexpr int FIRST = MOB_FIRST;
    static constexpr int HID = MOB_FIRST + MOB_FIRST_FLOATS;
    static constexpr int HB = HID + 3 * 2 * W_TILE;
    static constexpr int HEAD_FLOATS = HB + MOB_HB_FLOATS;
    static constexpr int LAST = HEAD_FLOATS;
    static constexpr int LAST_TILE_BIAS = W_TILE;
    static constexpr int LAST_TILE_FLOATS = W_TILE + 32;
    static constexpr int MAX_TILES_IN_LDS = PREC == 2 ? 4 : MOB_MAX_TILES_IN_LDS;
};
static_assert(Lay<0>::HEAD_FLOATS == MOB_HEAD_FLOATS && Lay<1>::LAST_TILE_FLOATS == MOB_LAST_TILE_FLOATS && Lay<1>::HB == MOB_HB, "Lay<0/1> are the MOB_* constants");
inline constexpr int64_t mobius_packed_floats_p(int K, int prec) {
    return prec == 2 ? Lay<2>::HEAD_FLOATS + (int64_t)((K + 7) / 8) * Lay<2>::LAST_TILE_FLOATS : mobius_packed_floats(K);
}
inline constexpr int64_t cond_packed_floats_p(int tiles, int prec) {      // Condition16Trans / Condition9* (1 fc_last tile), Condition36Trans (2)
    return prec == 2 ? Lay<2>::HEAD_FLOATS + (int64_t)tiles * Lay<2>::LAST_TILE_FLOATS : MOB_HEAD_FLOATS + (int64_t)tiles * MOB_LAST_TILE_FLOATS;
}

// ---- unconditional 4x4 affine record ----
// [0..15] M row-major, [16] log|det M|, [17..32] M^-1, [33] log|det M^-1|, [34] 1.0 if M is orthogonal (log-det exactly 0), [35] 0,
// [36..139] forward block: the 10x10 table of M in two halves of 52 floats (so3_math.h affine16_table: five rows, log|det M|, the orthogonal flag);
// [140..243] the same block for M^-1 (the inverse pass).  The forward kernel stages one block in LDS next to the layer images.
constexpr int AFF_TABLE_FWD = 36, AFF_TABLE_INV = 140, AFF_TABLE_FLOATS = 104;
constexpr int AFF_FLOATS = 244;
constexpr int AFF_TABLE_LDS_STRIDE = 112;                 // floats between the two blocks staged in LDS (AFF_TABLE_FLOATS rounded up)

// ---- 3x3 / 6x6 Gram-Schmidt layers (Uncondition9Trans, Uncondition36Trans): [M row-major | M^-1 row-major] ----
constexpr int GS9_FLOATS = 20;                            // 9 + 9, padded to a multiple of 4
constexpr int GS36_FLOATS = 72;                           // 36 + 36

// ---- Condition16Trans record: same head as a Moebius layer with an all-zero fc_first image (its whole first layer
// is the feature projection), then ONE fc_last tile whose rows are M entries: packed row 8g + 4h + c (g = 0,1) is
// M[2g + h][c]; rows 16..31 are zero padding.
constexpr int64_t COND16_FLOATS = MOB_HEAD_FLOATS + MOB_LAST_TILE_FLOATS;
constexpr int64_t COND36_FLOATS = MOB_HEAD_FLOATS + 2 * MOB_LAST_TILE_FLOATS;

// ---- feature projection record (per layer that consumes the feature vector), F padded to a multiple of 8 ----
// [2][F/8][64] float4 weight image of W0[:, 3:] (or W0 for Condition16Trans), then bias image [2][2][16]
// split precision: [2][ceil(F/16)][hi, lo][64] 8 x fp16 (512 floats per k-step of 16), then the same bias image.
// Both layouts fit in the size returned here (equal when F % 16 == 0).
inline constexpr int64_t featproj_packed_floats(int F) { return F <= 0 ? 0 : (int64_t)2 * ((F + 15) / 16) * 512 + 64; }

// ---- feature-projection scratch G (device workspace) ----
// G[cond_slot][sample_group of 32][ot (2)][q (4)][lane (64)] float4 : register 4q..4q+3 of tile ot of lane
constexpr int G_FLOATS_PER_GROUP = 2 * 4 * 64 * 4;        // 2048 floats = 64 features x 32 samples

// layer kinds (== RNF_LAYER_* of include/rnf_hip.h)
constexpr int RNF_KIND_MOBIUS = 1, RNF_KIND_AFFINE16 = 2, RNF_KIND_COND16 = 3, RNF_KIND_GS9 = 4, RNF_KIND_GS36 = 5;
// conditional 3x3 layers: M = I + reshape(MLP(feature), 3, 3) per sample (Condition9Trans, Condition9RotRSmith, Condition9RotL, Condition9RotR)
constexpr int RNF_KIND_COND9_GS = 6, RNF_KIND_COND9_SMITH = 7, RNF_KIND_COND9_POLAR_L = 8, RNF_KIND_COND9_POLAR_R = 9, RNF_KIND_COND9_LAST = 9;
// Condition36Trans: M = I + reshape(MLP(feature), 6, 6) per sample; TWO fc_last tiles (packed row P of tile 0 is output P, rows 0..3 of
// tile 1 are outputs 32..35)
constexpr int RNF_KIND_COND36 = 10;
// per-sample matrices handed in by the caller (side buffer [slot][n][16], slot = desc cond_slot... see include/rnf_hip.h): the layers whose
// matrix the reference builds with batched torch ops the kernels do not restate (ConditionRot: SVD; ConditionLU: its batch-coupled
// torch.diag) -- 4x4 on the quaternion with log-det, 4x4 orthogonal (log-det 0), 3x3 Gram-Schmidt with the tangent log-det
constexpr int RNF_KIND_SIDE16 = 11, RNF_KIND_SIDE16_ROT = 12, RNF_KIND_SIDE9 = 13, RNF_KIND_LAST = 13;
RNF_LAYOUT_INLINE bool kind_is_side(int kind) { return kind >= RNF_KIND_SIDE16 && kind <= RNF_KIND_SIDE9; }
// training path only (train_kernels.h): a conditioner MLP on its own -- dL/d(outputs) comes from the caller, the kernel back-propagates it
// to the MLP's parameters and to the feature (rnf_cond_mlp_backward; the networks of the side layers)
constexpr int RNF_KIND_MLP_ONLY = 14;
RNF_LAYOUT_INLINE bool kind_is_cond9(int kind) { return kind >= RNF_KIND_COND9_GS && kind <= RNF_KIND_COND9_LAST; }
RNF_LAYOUT_INLINE bool kind_has_mlp(int kind) { return kind == RNF_KIND_MOBIUS || kind == RNF_KIND_COND16 || kind_is_cond9(kind) || kind == RNF_KIND_COND36; }
// fc_last tiles of one layer record (KT = segments / 8 for a Moebius layer)
RNF_LAYOUT_INLINE int kind_last_tiles(int kind, int KT) { return kind == RNF_KIND_MOBIUS ? KT : (kind == RNF_KIND_COND36 ? 2 : 1); }

// layer descriptor columns (include/rnf_hip.h)
constexpr int D_KIND = 0, D_PERM = 1, D_PARAM = 2, D_SLOT = 3, D_FEAT = 4, D_PREC = 5;   // D_PREC: RNF_PREC_* of include/rnf_hip.h
// fallback records (exact fp32 images of the same layers) for flows packed in split precision: offsets in the same blob, or -1
constexpr int D_PARAM_FB = 6, D_FEAT_FB = 7, D_STRIDE = 8;

inline constexpr int rho(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

}  // namespace rnf
