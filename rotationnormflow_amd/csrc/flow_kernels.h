// flow_kernels.h -- the fused layer-stack kernels (gfx950 / CDNA4).
//
// One workgroup = NW waves; every wave owns 32 rotations (one MFMA column tile).  Lane l = (j = l&31, h = l>>5):
// both lanes of a pair (j, j+32) hold the full per-sample state (3x3 rotation + running log-det, registers, kept
// across the whole layer stack); the conditioner MLP runs on v_mfma_f32_32x32x2_f32 with the weights of the
// current layer staged in LDS (layout.h), its 4K outputs are produced 32 rows at a time and consumed immediately
// by the segment math, the two lanes of a pair each taking half of the segments (4 per fc_last tile).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "layout.h"
#include "so3_math.h"

namespace rnf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define RNF_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

constexpr int MAX_LAYERS = 400;

// kernel arguments (passed by value; the layer table sits in the kernarg segment -> scalar loads)
struct FlowArgs {
    const float *rot_in;      // [n,9]
    const float *blob;        // packed parameters
    const float *G;           // feature-projection scratch for this chunk (or nullptr)
    float *rot_out;           // [n,9] or nullptr
    float *ldj_out;           // [n] or nullptr
    float *logp_out;          // [n] or nullptr
    double *partials;         // [gridDim.x] or nullptr
    const float *fisher_A;    // [B,9] or nullptr
    const float *fisher_c;    // [B]
    long long n;              // samples in this launch
    long long sample_base;    // global index of sample 0 (for the Fisher row lookup)
    long long fisher_div;     // samples per Fisher row (n_total / B)
    long long g_groups;       // 32-sample groups per cond slot in G
    long long g_div;          // > 0: feature rows are SHARED, row = (sample_base + sample) / g_div, and G holds one 64-float record per
    long long g_rows;         //      (slot, feature row) instead of one fragment tile per (slot, 32-sample group)
    int fair_off;                // >= 0: float offset in LDS of the per-wave progress words (SIMD fairness governor on)
    unsigned long long *stamps;  // diagnostic builds only (RNF_STAMPS): per-phase cycle sums, else nullptr
    float *states;            // training forward: [n_layers][states_n][9] rotation at the input of every layer, else nullptr
    long long states_n;       // rotations in the whole call (chunks write at sample_base)
    int n_layers;
    int KT;                   // fc_last tiles = ceil(segments / 8)
    int K;                    // segments of the Moebius layers (pad segments of the last tile are masked when K % 8 != 0)
    int rf_first4;            // inverse pass: the root finder's FIRST pass takes the fourth-order step (mobius_inv_finish; set per FLOW by the launcher)
    float min_wsum;           // kMinWeightSum * K, set by the launcher: a kernel argument stays in a scalar register (computed in the kernel it was a
                              // loop-invariant VECTOR register that the 128-register instantiations spilled and reloaded -- behind an s_waitcnt vmcnt(0)
                              // that also waited for the LDS-DMA just issued -- in every layer finish)
    int tab_off;              // >= 0 (DMA staging): float offset in LDS of two AFF_TABLE_LDS_STRIDE-float buffers for constant-affine blocks
    // Range guard of the split-precision kernels (an fp16 operand beyond 65504 turns into inf and the sample's log-det into NaN):
    //   guard_mode 1: set guard[0] when a sample ends with a non-finite log-det or rotation;
    //   guard_mode 2 (the exact-fp32 instantiation launched right behind): run only if guard[0] != 0, and note the re-run in guard[1].
    int *guard;
    int guard_mode;
    const float *side;        // per-sample matrices of the RNF_KIND_SIDE* layers: [side slot][side_n][16] floats, or nullptr
    long long side_n;         // rotations in the whole call (chunks index at sample_base)
    // FUSED instantiation (conditional flows, feature_dim <= FUSED_MAX_F): the feature projection runs INSIDE the stack kernel
    // inverse with more than 128 segments (KT > 16, the 4-wave instantiation): the parameters of the segments beyond the 64 per lane that fit in
    // registers live in this stash, [gridDim.x * NW][4 (KT - 16)][64 lanes] float4 (sp, ur, uv, q), rewritten every layer (L2-resident)
    float4 *inv_stash;
    const float *feat;        // [n, feat_F] features of this launch
    float *stash;             // [gridDim.x * NW][G_FLOATS_PER_GROUP]: each wave's projected features of the NEXT layer (stays in L2)
    int feat_F;               // feature columns (multiple of 8)
    int feat_base, feat_stride;   // blob offset (floats) of cond slot 0's projection record, and the distance between consecutive slots
    int pa_off;               // float offset in LDS of the projection-weight buffer (FUSED_PA_FLOATS)
    // per layer: x = kind | perm_row << 4 | (cond_slot + 1) << 8 | (position of the next MLP layer + 1) << 16 ; y = param offset (floats)
    int2 layers[MAX_LAYERS];
};

__device__ __forceinline__ float4 lds_f4(const float *base, int idx4) {
    return reinterpret_cast<const float4 *>(base)[idx4];
}

// cooperative global -> LDS copy of `nfloats` (multiple of 4) floats (synchronous staging: the callers put a barrier on both sides).
// Round 6: LDS-DMA pieces (global_load_lds_dwordx4, 1 KiB per wave and instruction, no VGPR round trip), ALL requested before the one wait --
// the load -> ds_write loop this replaces was one exposed L2 round trip per 16 bytes and thread (a bf16x3 layer image: 15 of them).
#ifndef RNF_STAGE_DMA
#define RNF_STAGE_DMA 1
#endif
__device__ __forceinline__ void stage_floats(float *dst, const float *src, int nfloats, int tid, int nthreads) {
#if RNF_STAGE_DMA
    const int n4 = nfloats >> 2, lane = tid & 63, nw64 = nthreads & ~63;
    for (int base = __builtin_amdgcn_readfirstlane(tid & ~63); base < n4; base += nw64) {       // `base` is wave uniform
        const int idx = base + lane;
        if (idx < n4)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + 4 * (size_t)idx),
                                             (__attribute__((address_space(3))) void *)(dst + 4 * base), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    const float4 *s = reinterpret_cast<const float4 *>(src);
    float4 *d = reinterpret_cast<float4 *>(dst);
    for (int i = tid; i < (nfloats >> 2); i += nthreads) d[i] = s[i];
#endif
}

__device__ __forceinline__ f32x16 load_bias16(const float *bias_half /* 16 floats of this lane-half */) {
    f32x16 c;
    float4 b0 = lds_f4(bias_half, 0), b1 = lds_f4(bias_half, 1), b2 = lds_f4(bias_half, 2), b3 = lds_f4(bias_half, 3);
    c[0] = b0.x; c[1] = b0.y; c[2] = b0.z; c[3] = b0.w;
    c[4] = b1.x; c[5] = b1.y; c[6] = b1.z; c[7] = b1.w;
    c[8] = b2.x; c[9] = b2.y; c[10] = b2.z; c[11] = b2.w;
    c[12] = b3.x; c[13] = b3.y; c[14] = b3.z; c[15] = b3.w;
    return c;
}

// accumulator init of fc_first from the feature projection scratch (global memory, fragment order).
// NT: non-temporal loads -- for the instantiations that read a layer's scratch tile exactly ONCE (KEEPX0): the 1.7 - 2.8 GB stream per
// launch then stops evicting the layer images from L2 (C5: -2.7 %); the 16-wave instantiations read every tile a second time for the
// residual and want it cached (nt there: +1.3 % on C4).
template <bool NT = false>
__device__ __forceinline__ f32x16 load_g16(const float *g_tile /* [4][64] float4 of this (group, ot) */, int lane) {
    typedef float f4v __attribute__((ext_vector_type(4)));
    const f4v *gv = reinterpret_cast<const f4v *>(g_tile);
    f4v b0, b1, b2, b3;
    if constexpr (NT) {
        b0 = __builtin_nontemporal_load(gv + lane); b1 = __builtin_nontemporal_load(gv + 64 + lane);
        b2 = __builtin_nontemporal_load(gv + 128 + lane); b3 = __builtin_nontemporal_load(gv + 192 + lane);
    } else {
        b0 = gv[lane]; b1 = gv[64 + lane]; b2 = gv[128 + lane]; b3 = gv[192 + lane];
    }
    f32x16 c;
    c[0] = b0.x; c[1] = b0.y; c[2] = b0.z; c[3] = b0.w;
    c[4] = b1.x; c[5] = b1.y; c[6] = b1.z; c[7] = b1.w;
    c[8] = b2.x; c[9] = b2.y; c[10] = b2.z; c[11] = b2.w;
    c[12] = b3.x; c[13] = b3.y; c[14] = b3.z; c[15] = b3.w;
    return c;
}

// Where a wave finds its feature-projection values.  ROWS = false (default kernels): fragment tiles of its 32-sample group (p = tile of
// out tile 0).  ROWS = true (extended instantiation): feature rows may be shared by runs of consecutive samples (pose estimation: one
// image feature, many query rotations) and `rows` selects the 64-float record of each lane's own feature row,
// [out tile][lane half][16 accumulator registers].
template <bool ROWS>
struct GFrag {
    static constexpr bool AS_BIAS = false;
    const float *p;
    bool rows;
    __device__ __forceinline__ explicit operator bool() const { return p != nullptr; }
    template <bool NT = false>
    __device__ __forceinline__ f32x16 load(int ot, int lane, int h) const {
        if (!ROWS || !rows) return load_g16<NT>(p + ot * (4 * 64 * 4), lane);
        const float4 *q = reinterpret_cast<const float4 *>(p + (ot * 2 + h) * 16);
        const float4 b0 = q[0], b1 = q[1], b2 = q[2], b3 = q[3];
        f32x16 c;
        c[0] = b0.x; c[1] = b0.y; c[2] = b0.z; c[3] = b0.w;
        c[4] = b1.x; c[5] = b1.y; c[6] = b1.z; c[7] = b1.w;
        c[8] = b2.x; c[9] = b2.y; c[10] = b2.z; c[11] = b2.w;
        c[12] = b3.x; c[13] = b3.y; c[14] = b3.z; c[15] = b3.w;
        return c;
    }
};

// ROWS instantiations (round 5): SHARED feature rows on the lean / inverse kernels.  Pose estimation evaluates one image feature against
// Q query rotations (agent.py:238-263, eval.py:322-347: `feature.repeat` over number_queries), so row r of the projection scratch (a 64-float
// record per (layer, row), [out tile][lane half][16 accumulator registers]) serves rotations [r Q, (r + 1) Q).  With Q >= 32 the 32
// rotations of a wave sit in at most two rows, and then G is not a per-sample tile at all but a BIAS VECTOR per row: it enters x0 as one more
// exact-fp32 matrix step of fc_first,
//     x0[o][j] += G[row0][o] * [row(j) == row0] + G[row0 + 1][o] * [row(j) == row0 + 1]            (K = 2: lane-half h carries row0 + h)
// A operand: lane (i, h) holds G[row0 + h][32 ot + i] -- ONE float per lane and out tile, loaded once per layer; B operand: the 0 / 1
// indicator of the lane's own sample.  No 16-register accumulator tile is loaded, kept (KEEPX0) or re-read for the residual: the residual
// repeats the same matrix step.  Everything else about the rows is wave uniform (scalar registers).
struct GFragRows {
    static constexpr bool AS_BIAS = true;
    const float *p;           // G + slot * g_rows * 64 (wave uniform), or nullptr for an unconditional layer
    int row0, rem0, gdiv, last;   // first rotation of the wave: its row and position inside the row; rotations per row; last valid row
    __device__ __forceinline__ explicit operator bool() const { return p != nullptr; }
    __device__ __forceinline__ float aop(int ot, int lane, int h) const {
        const int i = lane & 31;
        const int row = min(row0 + h, last);
        // position of output row 32 ot + i inside the record: lane half (i >> 2) & 1, accumulator register (i & 3) + 4 (i >> 3)   (layout.h rho)
        const unsigned idx = (unsigned)row * 64u + (unsigned)(ot * 32 + ((i >> 2) & 1) * 16 + (i & 3) + 4 * (i >> 3));
        return p[idx];
    }
    __device__ __forceinline__ float bop(int lane, int h) const {
        const int over = (rem0 + (lane & 31)) >= gdiv ? 1 : 0;       // this lane's sample sits in row0 + over
        return over == h ? 1.0f : 0.0f;
    }
    template <bool NT = false>
    __device__ __forceinline__ f32x16 load(int, int, int) const {       // (interface of the tile-shaped sources; never called for AS_BIAS)
        return f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    }
};

// LEAN instantiations (unconditional Moebius / constant-affine stacks): there is no feature projection at all
struct NoG {
    static constexpr bool AS_BIAS = false;
    __device__ __forceinline__ constexpr explicit operator bool() const { return false; }
    template <bool NT = false>
    __device__ __forceinline__ f32x16 load(int, int, int) const {
        return f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    }
};

// one 64 -> 32 output tile: acc += W_tile[32 x 64] . relu?(in)   (in = two f32x16 register tiles)
template <bool RELU>
__device__ __forceinline__ f32x16 gemm_tile64(const float *w_tile /* [8][64] float4 */, int lane, const f32x16 (&in)[2],
                                              f32x16 acc) {
#pragma unroll
    for (int tg = 0; tg < 8; ++tg) {
        float4 a = lds_f4(w_tile, tg * 64 + lane);
        const int t = tg >> 2, r0 = (tg & 3) * 4;
        float b0 = in[t][r0], b1 = in[t][r0 + 1], b2 = in[t][r0 + 2], b3 = in[t][r0 + 3];
        if (RELU) { b0 = fmaxf(b0, 0.f); b1 = fmaxf(b1, 0.f); b2 = fmaxf(b2, 0.f); b3 = fmaxf(b3, 0.f); }
        acc = RNF_MFMA(a.x, b0, acc);
        acc = RNF_MFMA(a.y, b1, acc);
        acc = RNF_MFMA(a.z, b2, acc);
        acc = RNF_MFMA(a.w, b3, acc);
    }
    return acc;
}

// ConditionalTransform up to the input of fc_last (flow/condition.py:24-29): tt = relu(x0 + L5(relu(L3(relu(L1(relu(x0)))))))
// x0 = fc_first(y (+) feature): the y part and (unconditional) bias run as two K=2 MFMA steps, the feature part and
// (conditional) bias arrive pre-multiplied in `cinit`.
// nan_in: set when x0 comes out NaN (a NaN input: every output row mixes every input).  fmaxf launders a NaN into 0, so without the flag a
// NaN feature row or rotation would yield finite garbage where the reference's torch.relu propagates the NaN (SURVEY 8(b) "errors").
__device__ __forceinline__ void mlp_head(const float *lds, int lane, int h, float y0, float y1, float y2,
                                         const f32x16 (&cinit)[2], f32x16 (&tt)[2], bool &nan_in) {
    const float bA = h ? y1 : y0;
    const float bB = h ? 1.0f : y2;
    f32x16 x0[2];
#pragma unroll
    for (int ot = 0; ot < 2; ++ot) {
        float2 a = reinterpret_cast<const float2 *>(lds + MOB_FIRST)[ot * 64 + lane];
        f32x16 c = RNF_MFMA(a.x, bA, cinit[ot]);
        x0[ot] = RNF_MFMA(a.y, bB, c);
    }
    nan_in |= x0[0][0] != x0[0][0];
    f32x16 hin[2] = {x0[0], x0[1]};
#pragma unroll
    for (int L = 0; L < 3; ++L) {
        f32x16 hout[2];
#pragma unroll
        for (int ot = 0; ot < 2; ++ot) {
            f32x16 c = load_bias16(lds + MOB_HB + ((L * 2 + ot) * 2 + h) * 16);
            hout[ot] = gemm_tile64<true>(lds + MOB_HID + (L * 2 + ot) * (8 * 64 * 4), lane, hin, c);
        }
        hin[0] = hout[0];
        hin[1] = hout[1];
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) tt[t][r] = fmaxf(x0[t][r] + hin[t][r], 0.0f);
}

__device__ __forceinline__ f32x16 last_tile(const float *tile_rec, int lane, int h, const f32x16 (&tt)[2]) {
    f32x16 c = load_bias16(tile_rec + MOB_LAST_TILE_BIAS + h * 16);
    return gemm_tile64<false>(tile_rec, lane, tt, c);
}

// ------------------------------------------------------------------------------------------------------------
// Split-precision path (PREC = 1, "f16x2"): every fp32 operand x of the 64-wide GEMMs is carried as two fp16 terms
//     x = hi + lo,   hi = fp16(x) (RN),  lo = fp16(x - hi)   (unscaled: lo lives in the fp16 subnormal range, whose spacing 2^-24 is
//                                                              the absolute floor of the pair; 22 significant bits above 2^-2)
// and each fp32 product-sum as THREE v_mfma_f32_32x32x16_f16 into ONE fp32 accumulator:
//     acc += Ahi.Bhi;  acc += Ahi.Blo;  acc += Alo.Bhi                                      (Alo.Blo ~ 2^-24 dropped)
// (gfx950's fp16 MFMA keeps subnormal inputs; with lo pre-scaled by 2^12 instead, a second accumulator and 16 combining FMAs per
// tile were needed: +9 % run time and +29 registers.)
// Why: on gfx950 the f32-input MFMA runs on the same FMA datapath as the VALU (measured: VALU issue stalls for the
// ~48-64 cycles an f32 MFMA occupies, profiles/r1/mfma_valu_coissue_microbench.txt), so the exact-fp32 kernel can never
// overlap the segment math with the conditioner GEMMs; the f16 MFMA runs on the real matrix cores at 16x the rate.
// Measured effect on parity: none beyond fp32 rounding noise.
// Range: an activation or weight >= 65520 in magnitude overflows fp16 and the result turns into inf/NaN (loudly); the
// host packer refuses weights outside the fp16 range and falls back to the exact PREC = 0 kernels.
//
// Operand maps of v_mfma_f32_32x32x16_f16: lane (r = l&31, h = l>>5) element j (0..7) is A[row r][k = 8h + j] and
// B[k = 8h + j][col r].  K-step s = 2t + s' of a 64-feature activation consumes accumulator registers 8s'..8s'+7 of
// tile t of the producing layer; register 8s'+j of lane-half h is feature 32t + 16s' + 8(j>>2) + 4h + (j&3), so the
// weight image stores for lane (i, h):  elem j = W[32*ot + i][32t + 16s' + 8(j>>2) + 4h + (j&3)]   (hi image, lo image).
// ------------------------------------------------------------------------------------------------------------
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#define RNF_MFMA_H(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

struct ActFrag {
    h8 hi[4], lo[4];          // B fragments of one 64-feature activation, k-step s = 2*tile + half
};

typedef float f2 __attribute__((ext_vector_type(2)));

template <bool RELU>
__device__ __forceinline__ void split_act(const f32x16 (&x)[2], ActFrag &f) {
    // -1.0 the optimiser cannot see through: fma(m1, hi, v) then stays one v_fma_mixlo/mixhi_f16 that takes the fp16 `hi` directly
    // (a literal -1 is folded into v - float(hi): v_cvt_f32_f16 + v_sub_f32 + a second v_cvt_pk_f16_f32)
    float m1 = -1.0f;
    asm("" : "+s"(m1));
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            f2 v = {x[s >> 1][8 * (s & 1) + j], x[s >> 1][8 * (s & 1) + j + 1]};
            if (RELU) {       // integer max(bits, 0) IS relu on the bit pattern (negative floats are negative integers): one instruction,
                              // where fmaxf / fmed3 cost two (they first quiet a possible sNaN)
                v.x = __int_as_float(max(__float_as_int(v.x), 0));
                v.y = __int_as_float(max(__float_as_int(v.y), 0));
            }
            const h2 ph = __builtin_convertvector(v, h2);                 // v_cvt_pk_f16_f32 (RN)
            const h2 pl = {(_Float16)__builtin_fmaf(m1, (float)ph[0], v.x), (_Float16)__builtin_fmaf(m1, (float)ph[1], v.y)};
            f.hi[s][j] = ph[0]; f.hi[s][j + 1] = ph[1];
            f.lo[s][j] = pl[0]; f.lo[s][j + 1] = pl[1];
        }
    }
}

// one value pair of an accumulator tile -> element pair `e` (0..3) of one k-step fragment (the 5 instructions of split_act)
__device__ __forceinline__ void split_pair(float a, float b, float m1, h8 &hi, h8 &lo, int e) {
    a = __int_as_float(max(__float_as_int(a), 0));
    b = __int_as_float(max(__float_as_int(b), 0));
    const f2 v = {a, b};
    const h2 ph = __builtin_convertvector(v, h2);
    const h2 pl = {(_Float16)__builtin_fmaf(m1, (float)ph[0], a), (_Float16)__builtin_fmaf(m1, (float)ph[1], b)};
    hi[2 * e] = ph[0]; hi[2 * e + 1] = ph[1];
    lo[2 * e] = pl[0]; lo[2 * e + 1] = pl[1];
}

__device__ __forceinline__ h8 lds_h8(const float *base, int idx16) {
    return reinterpret_cast<const h8 *>(base)[idx16];
}

// one 64 -> 32 output tile in split precision; w_tile = [4 k-steps][hi, lo][64 lanes] h8  (2048 floats, as in fp32)
__device__ __forceinline__ f32x16 gemm_tile64_h(const float *w_tile, int lane, const ActFrag &in, f32x16 acc) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const h8 ah = lds_h8(w_tile, (s * 2 + 0) * 64 + lane);
        const h8 al = lds_h8(w_tile, (s * 2 + 1) * 64 + lane);
        acc = RNF_MFMA_H(ah, in.hi[s], acc);
        acc = RNF_MFMA_H(ah, in.lo[s], acc);
        acc = RNF_MFMA_H(al, in.hi[s], acc);
    }
    return acc;
}

// Hidden layers in slot order (split-precision path).  The operand split of an output tile (8 value pairs x 5 VALU) is written behind
// the matrix instructions of the NEXT tile instead of between two matrix bursts (where the compiler actually places the FILL 2 work:
// see the note at RNF_PIN_H):
//   FILL 1 (tile 0 of a layer): the previous layer's tile 1 (`src`) becomes fragments f[2], f[3] during k-steps 0 and 1 -- which only
//           need f[0], f[1]; k-steps 2, 3 then consume the fresh fragments;
//   FILL 2 (tile 1 of a layer): this layer's tile 0 (`src`) becomes f[0] during k-step 1 and f[1] during k-step 2 -- each fragment is
//           overwritten right after the last matrix instruction that reads its previous content has been issued, so the fragments
//           need no second set of registers.
// Matrix instruction m = 3 * k-step + term; one sched_barrier per slot keeps the order (the compiler would cluster the 12 MFMAs).
template <int M, int FILL>
__device__ __forceinline__ void hidden_slot(const float *w_tile, int lane, ActFrag &f, f32x16 &acc, const f32x16 &src, float m1, h8 &ah,
                                            h8 &al) {
    constexpr int ks = M / 3, term = M % 3;
    if constexpr (term == 0) acc = RNF_MFMA_H(ah, f.hi[ks], acc);
    else if constexpr (term == 1) acc = RNF_MFMA_H(ah, f.lo[ks], acc);
    else acc = RNF_MFMA_H(al, f.hi[ks], acc);
#ifdef RNF_EARLY_AH        // each operand is re-fetched right behind ITS last reader: ah one matrix instruction earlier than al
    if constexpr (term == 1 && ks + 1 < 4) ah = lds_h8(w_tile, ((ks + 1) * 2 + 0) * 64 + lane);
    if constexpr (term == 2 && ks + 1 < 4) al = lds_h8(w_tile, ((ks + 1) * 2 + 1) * 64 + lane);
#else
    if constexpr (term == 2 && ks + 1 < 4) {
        ah = lds_h8(w_tile, ((ks + 1) * 2 + 0) * 64 + lane);
        al = lds_h8(w_tile, ((ks + 1) * 2 + 1) * 64 + lane);
    }
#endif
    // which fragment this k-step group fills (-1: none) and from which half of `src`
    constexpr int dst = FILL == 1 ? (ks == 0 ? 2 : (ks == 1 ? 3 : -1)) : (FILL == 2 ? (ks == 1 ? 0 : (ks == 2 ? 1 : -1)) : -1);
    if constexpr (dst >= 0) {
        constexpr int half = dst & 1;                              // fragment 2t + half <- registers 8 * half .. 8 * half + 7 of tile t
        if constexpr (term == 0) {
            split_pair(src[8 * half + 0], src[8 * half + 1], m1, f.hi[dst], f.lo[dst], 0);
        } else if constexpr (term == 1) {
            split_pair(src[8 * half + 2], src[8 * half + 3], m1, f.hi[dst], f.lo[dst], 1);
        } else {
            split_pair(src[8 * half + 4], src[8 * half + 5], m1, f.hi[dst], f.lo[dst], 2);
            split_pair(src[8 * half + 6], src[8 * half + 7], m1, f.hi[dst], f.lo[dst], 3);
        }
#ifdef RNF_PIN_H                                                    // FILL 2 results are consumed by the next layer only: unpinned, the optimiser
        asm volatile("" : "+v"(f.hi[dst]), "+v"(f.lo[dst]));       // moves that split behind the tile (see tile_step_h); pinned: 0.8 % slower
#endif
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (M + 1 < 12) hidden_slot<M + 1, FILL>(w_tile, lane, f, acc, src, m1, ah, al);
}

template <int FILL>
__device__ __forceinline__ void hidden_tile(const float *w_tile, int lane, ActFrag &f, f32x16 &acc, const f32x16 &src, float m1) {
    h8 ah = lds_h8(w_tile, 0 * 64 + lane);
    h8 al = lds_h8(w_tile, 1 * 64 + lane);
    __builtin_amdgcn_sched_barrier(0);
    hidden_slot<0, FILL>(w_tile, lane, f, acc, src, m1, ah, al);
}

// DEEP (round 6; workgroups of at most 8 waves, i.e. at most two waves per SIMD): the operand pairs of TWO k-steps are in flight.  hidden_slot
// requests the operands of k-step s + 1 behind the LAST matrix instruction of k-step s and the very next instruction needs them: with four
// waves per SIMD the partners cover that LDS round trip, with two (every inverse kernel) each k-step of the dependent chain waited for it
// (ds_read, s_waitcnt lgkmcnt, three matrix instructions: ~220 cycles per k-step where the matrix work is 96).  Here the pair of k-step
// s + 2 -- of the NEXT tile for s = 2, 3: the six hidden tiles of a layer are contiguous in LDS -- is requested instead, one whole k-step
// ahead; 8 more registers, which these instantiations have in the hidden phase (no segment state is live there).  Same matrix
// instructions in the same order: bit-identical results.
template <int M, int FILL, bool HAS_NEXT>
__device__ __forceinline__ void hidden_slot2(const float *w_tile, int lane, ActFrag &f, f32x16 &acc, const f32x16 &src, float m1, h8 (&ah)[2],
                                             h8 (&al)[2]) {
    constexpr int ks = M / 3, term = M % 3, p = ks & 1;
    if constexpr (term == 0) acc = RNF_MFMA_H(ah[p], f.hi[ks], acc);
    else if constexpr (term == 1) acc = RNF_MFMA_H(ah[p], f.lo[ks], acc);
    else acc = RNF_MFMA_H(al[p], f.hi[ks], acc);
    if constexpr (term == 2) {
        if constexpr (ks < 2) {
            ah[p] = lds_h8(w_tile, ((ks + 2) * 2 + 0) * 64 + lane);
            al[p] = lds_h8(w_tile, ((ks + 2) * 2 + 1) * 64 + lane);
        } else if constexpr (HAS_NEXT) {
            ah[p] = lds_h8(w_tile + 8 * 64 * 4, ((ks - 2) * 2 + 0) * 64 + lane);
            al[p] = lds_h8(w_tile + 8 * 64 * 4, ((ks - 2) * 2 + 1) * 64 + lane);
        }
    }
    constexpr int dst = FILL == 1 ? (ks == 0 ? 2 : (ks == 1 ? 3 : -1)) : (FILL == 2 ? (ks == 1 ? 0 : (ks == 2 ? 1 : -1)) : -1);
    if constexpr (dst >= 0) {
        constexpr int half = dst & 1;
        if constexpr (term == 0) {
            split_pair(src[8 * half + 0], src[8 * half + 1], m1, f.hi[dst], f.lo[dst], 0);
        } else if constexpr (term == 1) {
            split_pair(src[8 * half + 2], src[8 * half + 3], m1, f.hi[dst], f.lo[dst], 1);
        } else {
            split_pair(src[8 * half + 4], src[8 * half + 5], m1, f.hi[dst], f.lo[dst], 2);
            split_pair(src[8 * half + 6], src[8 * half + 7], m1, f.hi[dst], f.lo[dst], 3);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (M + 1 < 12) hidden_slot2<M + 1, FILL, HAS_NEXT>(w_tile, lane, f, acc, src, m1, ah, al);
}
template <int FILL, bool HAS_NEXT>
__device__ __forceinline__ void hidden_tile2(const float *w_tile, int lane, ActFrag &f, f32x16 &acc, const f32x16 &src, float m1, h8 (&ah)[2], h8 (&al)[2]) {
    __builtin_amdgcn_sched_barrier(0);
    hidden_slot2<0, FILL, HAS_NEXT>(w_tile, lane, f, acc, src, m1, ah, al);
}

// SIMD fairness governor.  The two waves a workgroup places on one SIMD (w and w^4) run the same instruction stream; the
// hardware arbitrates by age, so the older one runs ahead, reaches the layer barrier early and leaves its partner to finish
// alone (a single wave cannot fill the VALU: every instruction waits for the previous one of its dependent chain).  Each wave
// publishes a progress counter in LDS at a few points per layer and takes the lower issue priority while it is ahead of its
// partner, so the pair advances together and the barrier wait shrinks.  Only changes timing, never results.  (8-wave workgroups
// only: with 16 waves, four per SIMD, the hardware arbitration alone measured 2 % faster than any governor.)
struct Fair {
    float *lds;
    int wave, off, prog;
    __device__ __forceinline__ void tick() {
        if (off < 0) return;
        ++prog;
        volatile int *pw = reinterpret_cast<volatile int *>(lds + off);
        pw[wave] = prog;
        const int other = __builtin_amdgcn_readfirstlane(pw[wave ^ 4]);
        if (prog > other) __builtin_amdgcn_s_setprio(0);
        else __builtin_amdgcn_s_setprio(2);
    }
};

// -DRNF_V_NOFAIR only (measured and NOT the default): the instantiations that never run the governor compiled without its
// code.  The default build gives every instantiation `Fair` with a run-time `off < 0` test (the launcher enables it for the general 8-wave
// forward split-precision kernel only).
struct NoFair {
    // (the tick points stay scheduling fences: without them the scheduler interleaves the hidden layers more freely and the 128-register
    // instantiations spill three times as much)
    __device__ __forceinline__ void tick() { __builtin_amdgcn_sched_barrier(0); }
};

// The two precisions behind one interface: Act = what the hidden stack hands to fc_last.
template <int PREC>
struct Mlp;

template <>
struct Mlp<0> {
    struct Act { f32x16 t[2]; };
    // g: this wave's feature-projection fragments for the layer (global memory), or nullptr for an unconditional layer
    template <class GF, bool KEEPX0 = false, class FairT = Fair, bool DEEP = false>
    static __device__ __forceinline__ void head(const float *lds, int lane, int h, float y0, float y1, float y2,
                                                const GF &g, Act &out, FairT &, bool &bad, const f32x16 * = nullptr) {
        f32x16 cinit[2];
        if (g) {
            cinit[0] = g.load(0, lane, h);
            cinit[1] = g.load(1, lane, h);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) { cinit[0][r] = 0.f; cinit[1][r] = 0.f; }
        }
        mlp_head(lds, lane, h, y0, y1, y2, cinit, out.t, bad);
    }
    static __device__ __forceinline__ f32x16 last(const float *tile_rec, int lane, int h, const Act &a) {
        return last_tile(tile_rec, lane, h, a.t);
    }
};

constexpr float kX0Guard = 64.0f;

template <>
struct Mlp<1> {
    typedef ActFrag Act;
    // fc_first (K = 3 + bias) stays on the two exact fp32 MFMA steps (4 of the layer's 176 matrix instructions).  x0 is
    // NOT kept in registers for the residual: it is recomputed at the end with the last hidden output as the accumulator
    // input (x0 + h3 in one fma chain), which frees 32 VGPRs through the hidden layers.
    static __device__ __forceinline__ f32x16 first_tile(const float *lds, int ot, int lane, float bA, float bB, f32x16 c) {
        const float2 a = reinterpret_cast<const float2 *>(lds + MOB_FIRST)[ot * 64 + lane];
        c = RNF_MFMA(a.x, bA, c);
        return RNF_MFMA(a.y, bB, c);
    }
    // `bad`: set when a hidden layer comes out NaN.  An operand beyond the fp16 range splits into (inf, -inf) and turns EVERY output of
    // the next layer into NaN (each output row meets each input); the integer ReLU of split_pair would then launder a NaN with the sign
    // bit set into 0, so one accumulator register per hidden layer is tested before it is split (1 VALU each).  An overflow of the last
    // activation needs no test: the fc_last outputs, the segment sums and finally the log-det are NaN (flow_stack_kernel's guard).
    // KEEPX0 (instantiations with 256 registers per lane: 8-wave workgroups): x0 stays in registers for the residual, so a conditional
    // layer reads its projected features from the scratch ONCE (round 2: twice -- C5's inverse pass moved 2 x 42 x 256 bytes per rotation).
    // pre != nullptr: the projected features of this layer were already fetched into registers (flow_stack_kernel, LEAN = 2: issued behind
    // the previous layer's barrier B2, so that the HBM latency of the scratch read -- the first instruction of the hidden phase otherwise,
    // +11 k cycles per layer in the round-2 stamps -- is spent under the layer finish and the affine layer)
    // ONE definition of the residual x0 + x3 (flow/condition.py:29) for every instantiation, so that a rotation's result does not depend on
    // the workgroup width its launch picked: fc_first(y) is accumulated onto x3 in the matrix unit, then the layer's projected features G
    // are added -- from the registers that kept them (KEEPX0: one read of the scratch) or from a second read of the scratch.
    template <bool KEEPX0, class GF>
    static __device__ __forceinline__ void residual(const float *lds, int ot, int lane, int h, float bA, float bB, const GF &g,
                                                    const f32x16 (&gk)[2], f32x16 &a, const float (&ga)[2], float gb) {
        a = first_tile(lds, ot, lane, bA, bB, a);
        if constexpr (GF::AS_BIAS) {                              // shared rows: G as one more fc_first step (see GFragRows)
            if (g) a = RNF_MFMA(ga[ot], gb, a);
        } else if (g) {
            if constexpr (KEEPX0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) a[r] += gk[ot][r];
            } else {
                const f32x16 gg = g.load(ot, lane, h);
#pragma unroll
                for (int r = 0; r < 16; ++r) a[r] += gg[r];
            }
        }
    }
    template <class GF, bool KEEPX0 = false, class FairT = Fair, bool DEEP = false>
    static __device__ __forceinline__ void head(const float *lds, int lane, int h, float y0, float y1, float y2,
                                                const GF &g, Act &out, FairT &fair, bool &bad, const f32x16 *pre = nullptr) {
        const float bA = h ? y1 : y0;
        const float bB = h ? 1.0f : y2;
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float m1 = -1.0f;
        asm("" : "+s"(m1));                                     // see split_act
        ActFrag &f = out;                                       // the fragments are rewritten in place, layer after layer
        f32x16 x0k[2];                                          // KEEPX0 only (dead otherwise): the projected features G of this layer
        float ga[2] = {0.f, 0.f}, gb = 0.f;                     // AS_BIAS only: the row records' A operands and the sample's row indicator
        if constexpr (GF::AS_BIAS) {                            // (for the tile-shaped sources the same re-derivation changes nothing: C4 7.87 / 7.93 ms)
            // the lane-derived offsets of this layer's reads are re-derived here: hoisted out of the layer loop they are registers the
            // 128-register instantiations spill and reload behind an s_waitcnt vmcnt(0)
            asm volatile("" : "+v"(lane), "+v"(h));
        }
        {
            f32x16 x0[2];
            if constexpr (GF::AS_BIAS) {
                if (g) { ga[0] = g.aop(0, lane, h); ga[1] = g.aop(1, lane, h); gb = g.bop(lane, h); }
#pragma unroll
                for (int ot = 0; ot < 2; ++ot) {
                    x0[ot] = first_tile(lds, ot, lane, bA, bB, zero);
                    if (g) x0[ot] = RNF_MFMA(ga[ot], gb, x0[ot]);
                }
            } else {
#pragma unroll
                for (int ot = 0; ot < 2; ++ot) {
                    const f32x16 gin = pre ? pre[ot] : (g ? g.template load<KEEPX0>(ot, lane, h) : zero);      // KEEPX0: the only read of this tile
                    if constexpr (KEEPX0) x0k[ot] = gin;
                    x0[ot] = first_tile(lds, ot, lane, bA, bB, gin);
                }
            }
            // a feature beyond the fp16 range (the whole projection row is NaN), or features so much larger than the packer's equalisation
            // assumed (x0 is normalised to an rms of 1/4 .. 1/2; kX0Guard = 64) that fc_last's down-scaled columns would lose bits
            if (g) bad |= !(fmaxf(fmaxf(fabsf(x0[0][0]), fabsf(x0[0][9])), fmaxf(fabsf(x0[1][3]), fabsf(x0[1][14]))) < kX0Guard);
            split_act<true>(x0, f);
        }
        auto w = [&](int L, int ot) { return lds + MOB_HID + (L * 2 + ot) * (8 * 64 * 4); };
        auto bias = [&](int L, int ot) { return load_bias16(lds + MOB_HB + ((L * 2 + ot) * 2 + h) * 16); };
        f32x16 a0, a1, b0, b1;
        if constexpr (DEEP) {                                   // see hidden_slot2: the same tiles with two operand pairs in flight
            h8 ah[2], al[2];
            ah[0] = lds_h8(w(0, 0), 0 * 64 + lane); al[0] = lds_h8(w(0, 0), 1 * 64 + lane);
            ah[1] = lds_h8(w(0, 0), 2 * 64 + lane); al[1] = lds_h8(w(0, 0), 3 * 64 + lane);
            a0 = bias(0, 0); a1 = bias(0, 1);
            hidden_tile2<0, true>(w(0, 0), lane, f, a0, a0, m1, ah, al);
            bad |= a0[0] != a0[0];
            b0 = bias(1, 0);
            hidden_tile2<2, true>(w(0, 1), lane, f, a1, a0, m1, ah, al);
            b1 = bias(1, 1);
            hidden_tile2<1, true>(w(1, 0), lane, f, b0, a1, m1, ah, al);
            bad |= b0[0] != b0[0];
            a0 = bias(2, 0);
            hidden_tile2<2, true>(w(1, 1), lane, f, b1, b0, m1, ah, al);
            a1 = bias(2, 1);
            hidden_tile2<1, true>(w(2, 0), lane, f, a0, b1, m1, ah, al);
            bad |= a0[0] != a0[0];
            residual<KEEPX0>(lds, 0, lane, h, bA, bB, g, x0k, a0, ga, gb);
            hidden_tile2<2, false>(w(2, 1), lane, f, a1, a0, m1, ah, al);
        } else {
        // layer 0: tile 0 bare (its input is complete), tile 1 carries the split of tile 0
        a0 = bias(0, 0); a1 = bias(0, 1);
        hidden_tile<0>(w(0, 0), lane, f, a0, a0, m1);
        bad |= a0[0] != a0[0];
        hidden_tile<2>(w(0, 1), lane, f, a1, a0, m1);
        fair.tick();
        // layer 1: tile 0 carries the split of layer 0's tile 1, tile 1 the split of its own tile 0
        b0 = bias(1, 0); b1 = bias(1, 1);
        hidden_tile<1>(w(1, 0), lane, f, b0, a1, m1);
        bad |= b0[0] != b0[0];
        hidden_tile<2>(w(1, 1), lane, f, b1, b0, m1);
        fair.tick();
        // layer 2, then the residual x0 + h3 (flow/condition.py:29) tile by tile: tile 0's residual + split ride behind tile 1's MFMAs
        a0 = bias(2, 0);
        a1 = bias(2, 1);
        hidden_tile<1>(w(2, 0), lane, f, a0, b1, m1);
        bad |= a0[0] != a0[0];
        residual<KEEPX0>(lds, 0, lane, h, bA, bB, g, x0k, a0, ga, gb);
        hidden_tile<2>(w(2, 1), lane, f, a1, a0, m1);
        fair.tick();
        }
        residual<KEEPX0>(lds, 1, lane, h, bA, bB, g, x0k, a1, ga, gb);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            split_pair(a1[2 * e], a1[2 * e + 1], m1, f.hi[2], f.lo[2], e);
            split_pair(a1[8 + 2 * e], a1[8 + 2 * e + 1], m1, f.hi[3], f.lo[3], e);
        }
    }
    static __device__ __forceinline__ f32x16 last(const float *tile_rec, int lane, int h, const Act &a) {
        f32x16 c = load_bias16(tile_rec + MOB_LAST_TILE_BIAS + h * 16);
        return gemm_tile64_h(tile_rec, lane, a, c);
    }
};

// ------------------------------------------------------------------------------------------------------------
// PREC = 2, "bf16x3" (round 6): the STRICT arithmetic.  Every fp32 operand of the 64-wide GEMMs -- weight and activation -- is carried as
// three bf16 terms x = hi + mid + lo (8 + 8 + 8 significant bits: all 24 of an fp32, with fp32's exponent range), and every fp32
// product-sum as SIX v_mfma_f32_32x32x16_bf16 into one fp32 accumulator, all terms down to 2^-16 of the leading one:
//     acc += Ah.Bh + Ah.Bm + Am.Bh + Ah.Bl + Al.Bh + Am.Bm                         (Am.Bl, Al.Bm ~ 2^-24, Al.Bl ~ 2^-32 dropped)
// Nothing about it depends on the data or on where the weights sit on their ReLU-rescaling orbit: no equalisation, no audit, no feature
// calibration, no range guard (an activation overflows where fp32 itself does).  Twice the matrix instructions of the fp16 pairs and a
// third more operand bytes (layout.h Lay<2>: 171 KiB per K = 64 layer, hence synchronous staging and four fc_last tiles at a time), but
// on the real matrix cores: unlike the fp32-input MFMA (PREC 0), which shares the VALU's FMA datapath, it co-issues with the segment math.
// The operand split truncates (hi = top 16 bits of x, mid = top 16 bits of x - hi, lo = top 16 bits of the rest): three same-signed
// terms, residual < 2^-24 |x|; 11 VALU per value pair.
// ------------------------------------------------------------------------------------------------------------
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
#define RNF_MFMA_B(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
typedef unsigned int u4v __attribute__((ext_vector_type(4)));

struct ActFrag3 {
    u4v hi[4], mid[4], lo[4];     // B fragments of one 64-feature activation (8 x bf16 per fragment), k-step s = 2 * tile + half (48 registers)
};
__device__ __forceinline__ b8 as_b8(const u4v &v) { return __builtin_bit_cast(b8, v); }

// one value pair -> element pair `e` (0..3) of the three fragments of one k-step (13 VALU with the ReLU)
template <bool RELU>
__device__ __forceinline__ void split3_pair(float a, float b, u4v &hi, u4v &mid, u4v &lo, int e) {
    if (RELU) { a = relu_bits(a); b = relu_bits(b); }
    const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
    const float ra = a - __uint_as_float(ua & 0xffff0000u), rb = b - __uint_as_float(ub & 0xffff0000u);        // exact
    const unsigned va = __float_as_uint(ra), vb = __float_as_uint(rb);
    const float sa = ra - __uint_as_float(va & 0xffff0000u), sb = rb - __uint_as_float(vb & 0xffff0000u);      // exact
    hi[e] = __builtin_amdgcn_perm(ub, ua, 0x07060302u);                   // {a[31:16], b[31:16]}: element 2e = a, 2e + 1 = b
    mid[e] = __builtin_amdgcn_perm(vb, va, 0x07060302u);
    lo[e] = __builtin_amdgcn_perm(__float_as_uint(sb), __float_as_uint(sa), 0x07060302u);
}
// the same pair in two slices (7 + 6 VALU), one behind each of two consecutive matrix instructions (hidden_slot_b3)
struct Split3State { float ra, rb; };
__device__ __forceinline__ void split3_stage_a(float a, float b, Split3State &st, u4v &hi, int e) {
    a = relu_bits(a); b = relu_bits(b);
    const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
    st.ra = a - __uint_as_float(ua & 0xffff0000u);
    st.rb = b - __uint_as_float(ub & 0xffff0000u);
    hi[e] = __builtin_amdgcn_perm(ub, ua, 0x07060302u);
}
__device__ __forceinline__ void split3_stage_b(const Split3State &st, u4v &mid, u4v &lo, int e) {
    const unsigned va = __float_as_uint(st.ra), vb = __float_as_uint(st.rb);
    const float sa = st.ra - __uint_as_float(va & 0xffff0000u), sb = st.rb - __uint_as_float(vb & 0xffff0000u);
    mid[e] = __builtin_amdgcn_perm(vb, va, 0x07060302u);
    lo[e] = __builtin_amdgcn_perm(__float_as_uint(sb), __float_as_uint(sa), 0x07060302u);
}
template <bool RELU>
__device__ __forceinline__ void split3_act(const f32x16 (&x)[2], ActFrag3 &f) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e) split3_pair<RELU>(x[s >> 1][8 * (s & 1) + 2 * e], x[s >> 1][8 * (s & 1) + 2 * e + 1], f.hi[s], f.mid[s], f.lo[s], e);
}
__device__ __forceinline__ b8 lds_b8(const float *base, int idx16) { return reinterpret_cast<const b8 *>(base)[idx16]; }

// The A operands (hi, mid, lo images of one k-step of a weight tile) of TWO k-steps in flight: set s & 1 serves k-step s.
struct Ops3 {
    b8 a[2][3];
    __device__ __forceinline__ void load(int set, const float *w_tile, int ks, int lane) {
#pragma unroll
        for (int t = 0; t < 3; ++t) a[set][t] = lds_b8(w_tile, (ks * 3 + t) * 64 + lane);
    }
};
// matrix instruction M (0..23) of a tile: k-step M / 6, the six products smallest first: Am.Bm, Al.Bh, Ah.Bl, Am.Bh, Ah.Bm, Ah.Bh
template <int M>
__device__ __forceinline__ f32x16 b3_mfma(const Ops3 &o, const ActFrag3 &in, f32x16 acc) {
    constexpr int ks = M / 6, t = M % 6, p = ks & 1;
    if constexpr (t == 0) return RNF_MFMA_B(o.a[p][1], as_b8(in.mid[ks]), acc);
    else if constexpr (t == 1) return RNF_MFMA_B(o.a[p][2], as_b8(in.hi[ks]), acc);
    else if constexpr (t == 2) return RNF_MFMA_B(o.a[p][0], as_b8(in.lo[ks]), acc);
    else if constexpr (t == 3) return RNF_MFMA_B(o.a[p][1], as_b8(in.hi[ks]), acc);
    else if constexpr (t == 4) return RNF_MFMA_B(o.a[p][0], as_b8(in.mid[ks]), acc);
    else return RNF_MFMA_B(o.a[p][0], as_b8(in.hi[ks]), acc);
}
// behind the last instruction of k-step ks: its operand set is free -- request k-step ks + 2 of this tile, or k-step ks - 2 of the next one
template <int M>
__device__ __forceinline__ void b3_prefetch(Ops3 &o, const float *w_tile, const float *w_next, int lane) {
    constexpr int ks = M / 6, t = M % 6, p = ks & 1;
    if constexpr (t == 5) {
        if constexpr (ks < 2) o.load(p, w_tile, ks + 2, lane);
        else if (w_next) o.load(p, w_next, ks - 2, lane);
    }
}

// Hidden tiles, the operand split of the previous tile written behind the matrix instructions (the scheme of hidden_slot; 24 instructions
// per tile here and 13 VALU per value pair, cut in two slices of 7 and 6: one slice behind each of 16 consecutive matrix instructions --
// 8 + 26 issue cycles against the 32 of the instruction; with a whole pair behind every second one the stream measured VALU + matrix time):
//   FILL 1 (tile 0 of a layer): `src` (the previous layer's tile 1) -> fragments 2 (slots 0..7) and 3 (slots 8..15): k-steps 0, 1 read 0 and 1 only;
//   FILL 2 (tile 1 of a layer): `src` (this layer's tile 0) -> fragment 0 (slots 6..13, behind k-step 0, its last reader) and 1 (slots 14..21).
template <int M, int FILL>
__device__ __forceinline__ void hidden_slot_b3(const float *w_tile, const float *w_next, int lane, ActFrag3 &f, f32x16 &acc, const f32x16 &src, Ops3 &o,
                                               Split3State &st) {
    acc = b3_mfma<M>(o, f, acc);
    b3_prefetch<M>(o, w_tile, w_next, lane);
    constexpr int first = FILL == 1 ? 0 : (FILL == 2 ? 6 : -1);
    if constexpr (FILL != 0 && M >= first && M < first + 16) {
        constexpr int q = (M - first) / 2;                        // 0..7: value pair q of src
        constexpr int dst = (FILL == 1 ? 2 : 0) + q / 4, e = q % 4;
        if constexpr (((M - first) % 2) == 0) split3_stage_a(src[2 * q], src[2 * q + 1], st, f.hi[dst], e);
        else split3_stage_b(st, f.mid[dst], f.lo[dst], e);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (M + 1 < 24) hidden_slot_b3<M + 1, FILL>(w_tile, w_next, lane, f, acc, src, o, st);
}
template <int FILL>
__device__ __forceinline__ void hidden_tile_b3(const float *w_tile, const float *w_next, int lane, ActFrag3 &f, f32x16 &acc, const f32x16 &src, Ops3 &o) {
    Split3State st;
    __builtin_amdgcn_sched_barrier(0);
    hidden_slot_b3<0, FILL>(w_tile, w_next, lane, f, acc, src, o, st);
}

// one 64 -> 32 output tile without fillers (fc_last of the inverse pass and of the non-Moebius conditioners)
template <int M = 0>
__device__ __forceinline__ void plain_slot_b3(const float *w_tile, int lane, const ActFrag3 &in, f32x16 &acc, Ops3 &o) {
    acc = b3_mfma<M>(o, in, acc);
    b3_prefetch<M>(o, w_tile, nullptr, lane);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (M + 1 < 24) plain_slot_b3<M + 1>(w_tile, lane, in, acc, o);
}
__device__ __forceinline__ f32x16 gemm_tile64_b3(const float *w_tile, int lane, const ActFrag3 &in, f32x16 acc) {
    Ops3 o;
    o.load(0, w_tile, 0, lane);
    o.load(1, w_tile, 1, lane);
    __builtin_amdgcn_sched_barrier(0);
    plain_slot_b3<0>(w_tile, lane, in, acc, o);
    return acc;
}

template <>
struct Mlp<2> {
    typedef ActFrag3 Act;
    // RING staging (flow_stack_kernel): a layer image is consumed in UNITS that stream through three LDS regions --
    //     Ha = [fc_first | hidden tiles 0..2 | hidden biases]    Hb = [hidden tiles 3..5 | hidden biases]    L_g = fc_last tiles 4g .. 4g + 3
    // (float offsets inside a region)
    static constexpr int HA_TILES = Lay<2>::FIRST + MOB_FIRST_FLOATS, HA_BIAS = HA_TILES + 3 * Lay<2>::W_TILE, HA_FLOATS = HA_BIAS + MOB_HB_FLOATS;
    static constexpr int HB_BIAS = 3 * Lay<2>::W_TILE, HB_FLOATS = HB_BIAS + MOB_HB_FLOATS;
    static constexpr int REGION_FLOATS = 4 * Lay<2>::LAST_TILE_FLOATS;
    static_assert(HA_FLOATS <= REGION_FLOATS && HB_FLOATS <= REGION_FLOATS, "every unit fits a region");
    // Same structure as Mlp<1>::head: fc_first on the two exact fp32 MFMA steps, x0 recomputed for the residual (Mlp<1>::residual: ONE
    // definition of x0 + x3 for every arithmetic), the operand splits behind the matrix instructions of the next tile (hidden_slot_b3).
    // ctl.begin_unit() -> base of the next unit (waits for it, frees the previous one, requests the one after next).
    // `bad`: a NaN x0 (NaN feature row / rotation): the integer ReLU would launder it (the reference propagates NaN).
    template <class GF, bool KEEPX0, class Ctl>
    static __device__ __forceinline__ void head_ring(Ctl &ctl, int lane, int h, float y0, float y1, float y2, const GF &g, Act &out, bool &bad) {
        typedef Lay<2> L2;
        const float bA = h ? y1 : y0;
        const float bB = h ? 1.0f : y2;
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        f32x16 x0k[2];                                          // KEEPX0 only
        float ga[2] = {0.f, 0.f}, gb = 0.f;                     // AS_BIAS only
        const float *ua = ctl.begin_unit();                     // Ha
        auto wa = [&](int i) { return ua + HA_TILES + i * L2::W_TILE; };                       // hidden tiles 0..2 = (L0,0) (L0,1) (L1,0)
        auto bias_a = [&](int i) { return load_bias16(ua + HA_BIAS + (i * 2 + h) * 16); };
        Ops3 o;
        o.load(0, wa(0), 0, lane);                              // (their LDS latency sits under fc_first and the split of x0)
        o.load(1, wa(0), 1, lane);
        ActFrag3 &f = out;
        // fc_first (float2 per lane at the head of the unit), twice: x0 here, the residual at the end -- its image is copied out of the unit
        // (2 registers per lane and out tile) because unit Ha's region is gone by then
        float2 w0[2];
#pragma unroll
        for (int ot = 0; ot < 2; ++ot) w0[ot] = reinterpret_cast<const float2 *>(ua + L2::FIRST)[ot * 64 + lane];
        auto first_tile = [&](int ot, f32x16 c) { c = RNF_MFMA(w0[ot].x, bA, c); return RNF_MFMA(w0[ot].y, bB, c); };
        {
            f32x16 x0[2];
            if constexpr (GF::AS_BIAS) {
                if (g) { ga[0] = g.aop(0, lane, h); ga[1] = g.aop(1, lane, h); gb = g.bop(lane, h); }
#pragma unroll
                for (int ot = 0; ot < 2; ++ot) {
                    x0[ot] = first_tile(ot, zero);
                    if (g) x0[ot] = RNF_MFMA(ga[ot], gb, x0[ot]);
                }
            } else {
#pragma unroll
                for (int ot = 0; ot < 2; ++ot) {
                    const f32x16 gin = g ? g.template load<KEEPX0>(ot, lane, h) : zero;
                    if constexpr (KEEPX0) x0k[ot] = gin;
                    x0[ot] = first_tile(ot, gin);
                }
            }
            bad |= x0[0][0] != x0[0][0];
            split3_act<true>(x0, f);
        }
        // layer 0: tile 0 bare (its input is complete), tile 1 carries the split of tile 0
        f32x16 a0 = bias_a(0), a1 = bias_a(1);
        hidden_tile_b3<0>(wa(0), wa(1), lane, f, a0, a0, o);
        f32x16 b0 = bias_a(2);
        hidden_tile_b3<2>(wa(1), wa(2), lane, f, a1, a0, o);
        // layer 1: tile 0 carries the split of layer 0's tile 1 ...
        hidden_tile_b3<1>(wa(2), nullptr, lane, f, b0, a1, o);
        const float *ub = ctl.begin_unit();                     // Hb
        auto wb = [&](int i) { return ub + i * L2::W_TILE; };                                   // hidden tiles 3..5 = (L1,1) (L2,0) (L2,1)
        auto bias_b = [&](int i) { return load_bias16(ub + HB_BIAS + (i * 2 + h) * 16); };
        o.load(0, wb(0), 0, lane);
        o.load(1, wb(0), 1, lane);
        f32x16 b1 = bias_b(3);
        a0 = bias_b(4);
        // ... tile 1 the split of its own tile 0
        hidden_tile_b3<2>(wb(0), wb(1), lane, f, b1, b0, o);
        // layer 2, then the residual x0 + x3 (flow/condition.py:29) tile by tile: tile 0's residual + split ride behind tile 1's instructions
        a1 = bias_b(5);
        hidden_tile_b3<1>(wb(1), wb(2), lane, f, a0, b1, o);
        auto residual = [&](int ot, f32x16 &acc) {              // Mlp<1>::residual with the copied fc_first image
            acc = first_tile(ot, acc);
            if constexpr (GF::AS_BIAS) {
                if (g) acc = RNF_MFMA(ga[ot], gb, acc);
            } else if (g) {
                if constexpr (KEEPX0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] += x0k[ot][r];
                } else {
                    const f32x16 gg = g.load(ot, lane, h);
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] += gg[r];
                }
            }
        };
        residual(0, a0);
        hidden_tile_b3<2>(wb(2), nullptr, lane, f, a1, a0, o);
        residual(1, a1);
#pragma unroll
        for (int q = 0; q < 8; ++q) split3_pair<true>(a1[2 * q], a1[2 * q + 1], f.hi[2 + q / 4], f.mid[2 + q / 4], f.lo[2 + q / 4], q % 4);
    }
    static __device__ __forceinline__ f32x16 last(const float *tile_rec, int lane, int h, const Act &a) {
        return gemm_tile64_b3(tile_rec, lane, a, load_bias16(tile_rec + Lay<2>::LAST_TILE_BIAS + h * 16));
    }
};

// sum over the two lanes (j, j + 32) of a rotation.  Round 6: v_permlane32_swap (one VALU instruction: every lane gets both halves' values)
// instead of ds_bpermute_b32 (__shfl_xor: an LDS round trip on the lgkm counter, ~100 cycles in front of every layer finish and of every
// root-finder step).  half 0's value + half 1's value in both lanes: the same bits as own + partner's.
#ifndef RNF_PAIRSUM_SWAP
#define RNF_PAIRSUM_SWAP 1
#endif
__device__ __forceinline__ float pair_sum(float v) {
#if RNF_PAIRSUM_SWAP
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
#else
    return v + __shfl_xor(v, 32, 64);
#endif
}

// ------------------------------------------------------------------------------------------------------------
// Moebius layer (flow/mobiusflow.py:46-224), split into the three pieces the staging
// pipeline needs: head (reads the H part of the LDS image), tiles (reads the L part), finish (no LDS).
// ------------------------------------------------------------------------------------------------------------
struct MobiusCtx {
    Frame f;
    v3f y;            // the conditioning column (unchanged by the layer)
    float zc, zs, zth; // forward: the input column x as a unit point (cos, sin) of the frame and its angle (== pi)
    float target;     // inverse: angle of the given column
    int p0, p2;
    bool cyc;         // branch of mobiusflow.py:75 / :172
};

// S7: the context carries the frame PRE-SCALED by 0.7 (so3_math.h seg_s7_stage); mobius_fwd_finish<true> undoes the scale
template <int DIR, bool S7 = false>
__device__ __forceinline__ void mobius_begin(const Rot &R, int perm_row, MobiusCtx &c) {
    const int p0 = perm_row % 3, p1 = (perm_row + 1) % 3;                        // flow/flow.py:13-15
    c.p0 = p0;
    c.p2 = (perm_row + 2) % 3;
    c.cyc = (p1 - p0 == 1) || (p1 - p0 == -2);
    const v3f x = get_col(R, p0);
    c.y = get_col(R, p1);
    if constexpr (S7 && DIR == 0) {                     // forward split-precision path: only the 0.7-scaled frame is used (the input point
        c.f = make_frame_scaled(x, c.y, kSquash);       // is (-1, 0) in its own frame by construction)
        c.zc = -1.f; c.zs = 0.f; c.zth = kPi; c.target = 0.f;
        return;
    }
    c.f = make_frame(x, c.y);
#ifndef RNF_INV_PI
#define RNF_INV_PI 1
#endif
    if constexpr (RNF_INV_PI && DIR == 1) {
        // Inverse pass (round 6): the given column expressed in its own frame IS (-|x|, 0) -- r = -x/|x| and v = (y x r)/|y x r| is orthogonal
        // to r whatever y is -- so the reference's target angle atan2(tx.v, tx.r) wrapped to [0, 2 pi) (mobiusflow.py:157-167) is pi: exactly
        // in its fp64 run, pi or a neighbouring fp32 number in its fp32 run (tx.v is rounding noise ~ 1e-8 beside tx.r = -1); the snap to 0
        // near 2 pi never applies.  The forward split-precision path has used the same fact since round 2 (S7 branch above).  ~45 VALU
        // instructions per layer (two dot products, a reciprocal square root, an octant arctangent) for a constant.
        c.zc = -1.f; c.zs = 0.f; c.zth = kPi; c.target = kPi;
        if (S7) c.f = scale_frame(c.f, kSquash);
        return;
    }
    const float xr = dot3(x, c.f.r), xv = dot3(x, c.f.v);
    const float inv = hw_rsq(fmaf(xv, xv, xr * xr));
    c.zc = xr * inv;
    c.zs = xv * inv;
    c.zth = angle_0_2pi(xv, xr);
    if (DIR) {   // target angle of the given column (== pi by construction), wrapped and snapped (mobiusflow.py:157-167)
        c.target = fabsf(c.zth - kTwoPi) < 1e-4f ? 0.f : c.zth;
    }
    if (S7) c.f = scale_frame(c.f, kSquash);
}

// HALF (split-precision kernels): A accumulates sp * atan(t), mobius_fwd_finish<true> adds the constant part (so3_math.h)
template <bool HALF, bool SAFE = true>
__device__ __forceinline__ void segments4(const f32x16 &o, const MobiusCtx &c, float &S, float &A, float &J) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if constexpr (HALF) segment_fwd_s7<SAFE>(o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3], c.f, S, A, J);      // c.f is the 0.7-scaled frame
        else segment_fwd_pi<false>(S_UNSCALE * o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3], c.f, S, A, J);
    }
}

// The last fc_last tile of a layer whose segment count K is not a multiple of 8: the packer pads the tile with zero rows, and the pad
// segments (k = 8 tau + 2 g + h >= K) must carry weight 0 -- softplus(0) = ln 2 is not 0 -- so their contribution is masked out AFTER the
// activation.  nv = number of real segments among this lane's four (g < nv).
template <bool HALF, bool SAFE = true>
__device__ __forceinline__ void segments4_tail(const f32x16 &o, const MobiusCtx &c, float &S, float &A, float &J, int nv) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        float s1 = 0.f, a1 = 0.f, j1 = 0.f;
        if constexpr (HALF) segment_fwd_s7<SAFE>(o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3], c.f, s1, a1, j1);
        else segment_fwd_pi<false>(S_UNSCALE * o[4 * g], o[4 * g + 1], o[4 * g + 2], o[4 * g + 3], c.f, s1, a1, j1);
        const bool real = g < nv;
        S += real ? s1 : 0.f;
        A += real ? a1 : 0.f;
        J += real ? j1 : 0.f;
    }
}
// real segments among the four of lane-half h in the LAST tile of a K-segment layer (4 when K % 8 == 0)
__device__ __forceinline__ int tail_segments(int K, int h) {
    const int rem = K - 8 * ((K + 7) / 8 - 1) - h;          // segments k = base + 2 g + h, g = 0..3, real while 2 g + h < K - base
    return min(4, max(0, (rem + 1) / 2));
}
template <bool HALF, bool SAFE = true>
__device__ __forceinline__ void segments4_last(const f32x16 &o, const MobiusCtx &c, float &S, float &A, float &J, int K, int h) {
    if (K & 7) segments4_tail<HALF, SAFE>(o, c, S, A, J, tail_segments(K, h));     // wave-uniform branch
    else segments4<HALF, SAFE>(o, c, S, A, J);
}

// forward, all fc_last tiles resident (K <= 64): software pipelined BY HAND.  Tile tau+1's 32 dependent MFMAs (64
// cycles each on the matrix pipe) are interleaved one-for-one with 32 slices (4 segments x 8 stages, ~10 VALU issue
// slots each, so3_math.h seg_stage) of tile tau's segment math, every MFMA + slice pair fenced with sched_barrier(0).  Left to
// itself hipcc emits the 32 MFMAs back to back followed by ~300 VALU instructions (and ignores a
// sched_group_barrier pipeline for this block), so each wave alternates between matrix-only and VALU-only stretches
// and the two waves of a SIMD, released together by the layer barriers, leave the matrix pipe idle in lockstep.
template <int K>
__device__ __forceinline__ void tile_step(f32x16 &nxt, const float4 (&a)[8], const f32x16 (&tt)[2], const f32x16 &cur,
                                          SegState (&seg)[4], const MobiusCtx &c, float &S, float &A, float &J) {
    constexpr int tg = K >> 2, q = K & 3;
    const float av = q == 0 ? a[tg].x : (q == 1 ? a[tg].y : (q == 2 ? a[tg].z : a[tg].w));
    nxt = RNF_MFMA(av, tt[K >> 4][K & 15], nxt);
    constexpr int g = K >> 3, st = K & 7;             // slice `st` of segment `g` rides behind MFMA number K
    seg_stage<st>(seg[g], S_UNSCALE * cur[4 * g], cur[4 * g + 1], cur[4 * g + 2], cur[4 * g + 3], c.f, c.zc, c.zs, c.zth, S, A, J);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (K + 1 < 32) tile_step<K + 1>(nxt, a, tt, cur, seg, c, S, A, J);
}

// one pipelined tile: nxt <- fc_last tile `rec` (32 MFMAs), while the segment math of the finished tile `cur` runs
// (exact fp32 path; the split-precision path needs no hand interleave: its MFMAs run on the matrix cores)
__device__ __forceinline__ void tile_pipe(const float *rec, int lane, int h, const f32x16 (&tt)[2], f32x16 &nxt,
                                          const f32x16 &cur, const MobiusCtx &c, float &S, float &A, float &J) {
    nxt = load_bias16(rec + MOB_LAST_TILE_BIAS + h * 16);
    float4 a[8];
#pragma unroll
    for (int tg = 0; tg < 8; ++tg) a[tg] = lds_f4(rec, tg * 64 + lane);
    SegState seg[4];
    __builtin_amdgcn_sched_barrier(0);
    tile_step<0>(nxt, a, tt, cur, seg, c, S, A, J);
}

// split-precision counterpart of tile_step: slot m = 3 * (segment g) + (slice st) carries matrix instruction m of the next tile
// (k-step m / 3; term hi.hi, hi.lo, lo.hi) and slice st of segment g of the finished tile; the operands of k-step s + 1 are fetched
// behind the last matrix instruction of k-step s into the same registers (one look-ahead, 8 registers instead of 32).
template <int M, bool SAFE>
__device__ __forceinline__ void tile_step_h(const float *rec, int lane, f32x16 &nxt, h8 &ah, h8 &al, const ActFrag &in,
                                            const f32x16 &cur, SegS7 (&seg)[4], const MobiusCtx &c, float &S, float &A, float &J) {
    constexpr int ks = M / 3, term = M % 3;
    if constexpr (term == 0) nxt = RNF_MFMA_H(ah, in.hi[ks], nxt);
    else if constexpr (term == 1) nxt = RNF_MFMA_H(ah, in.lo[ks], nxt);
    else nxt = RNF_MFMA_H(al, in.hi[ks], nxt);
#ifdef RNF_EARLY_AH
    if constexpr (term == 1 && ks + 1 < 4) ah = lds_h8(rec, ((ks + 1) * 2 + 0) * 64 + lane);
    if constexpr (term == 2 && ks + 1 < 4) al = lds_h8(rec, ((ks + 1) * 2 + 1) * 64 + lane);
#else
    if constexpr (term == 2 && ks + 1 < 4) {          // operands of the next k-step: the matrix instruction has read its registers at
        ah = lds_h8(rec, ((ks + 1) * 2 + 0) * 64 + lane);     // issue, the slice below and the other waves cover the LDS latency
        al = lds_h8(rec, ((ks + 1) * 2 + 1) * 64 + lane);
    }
#endif
    constexpr int g = M / 3, st = M % 3;
    seg_s7_stage<st, SAFE>(seg[g], cur[4 * g], cur[4 * g + 1], cur[4 * g + 2], cur[4 * g + 3], c.f, S, A, J);
    // NOTE: the slice's results are only consumed slots later, and sched_barrier orders the machine scheduler, not the IR passes: the
    // optimiser sinks this arithmetic past the fences into one VALU block behind the 12 matrix instructions.  Pinning each slice in
    // its slot (-DRNF_PIN_L) gives the interleaved stream the source suggests and measured 0.8 % SLOWER, so the shipped build does not.
#ifdef RNF_PIN_L
    if constexpr (st == 0) asm volatile("" : "+v"(seg[g].a), "+v"(seg[g].b), "+v"(seg[g].n2), "+v"(seg[g].D));
    else if constexpr (st == 1) asm volatile("" : "+v"(seg[g].t), "+v"(seg[g].c), "+v"(seg[g].p), "+v"(seg[g].z));
    else asm volatile("" : "+v"(S), "+v"(A), "+v"(J));
#endif
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (M + 1 < 12) tile_step_h<M + 1, SAFE>(rec, lane, nxt, ah, al, in, cur, seg, c, S, A, J);   // (rec already holds the lane offset when lane == 0 is passed)
}

template <bool SAFE>
__device__ __forceinline__ void tile_pipe_h(const float *rec, int lane, int h, const ActFrag &tt, f32x16 &nxt, const f32x16 &cur,
                                            const MobiusCtx &c, float &S, float &A, float &J) {
    nxt = load_bias16(rec + MOB_LAST_TILE_BIAS + h * 16);
    h8 ah = lds_h8(rec, 0 * 64 + lane);
    h8 al = lds_h8(rec, 1 * 64 + lane);
    SegS7 seg[4];
    __builtin_amdgcn_sched_barrier(0);
    tile_step_h<0, SAFE>(rec, lane, nxt, ah, al, tt, cur, seg, c, S, A, J);
}
// The same tile addressed through two per-lane offsets the optimiser cannot fold (woff: floats from `base` to this lane's operand slot of
// the tile, boff: to its bias half): every LDS read of the tile then carries its position as an immediate offset.  With the tile position
// folded into constants instead, tiles beyond 64 KiB (the fc_last part starts at 50 KiB) need a v_add per read group: the DS offset field
// is 16 bits.
template <bool SAFE>
__device__ __forceinline__ void tile_pipe_h_off(const float *base, int woff, int boff, const ActFrag &tt, f32x16 &nxt, const f32x16 &cur,
                                                const MobiusCtx &c, float &S, float &A, float &J) {
    nxt = load_bias16(base + boff);
    const float *rec = base + woff;
    h8 ah = lds_h8(rec, 0 * 64);
    h8 al = lds_h8(rec, 1 * 64);
    SegS7 seg[4];
    __builtin_amdgcn_sched_barrier(0);
    tile_step_h<0, SAFE>(rec, 0, nxt, ah, al, tt, cur, seg, c, S, A, J);
}

template <int PREC, bool PINGPONG = false, bool FASTSP = PINGPONG, class FairT = Fair>
__device__ __forceinline__ void mobius_fwd_tiles(const float *lds, int KT, int K, int lane, int h, const typename Mlp<PREC>::Act &tt,
                                                 const MobiusCtx &c, float &S, float &A, float &J, FairT &fair) {
    const float *rec = lds + MOB_LAST;
    if constexpr (PREC == 0) {
        f32x16 bufA = last_tile(rec, lane, h, tt.t), bufB;
        int tau = 1;
        for (; tau + 1 < KT; tau += 2) {       // two tiles per trip: the accumulators ping-pong, no register copies
            tile_pipe(rec + tau * MOB_LAST_TILE_FLOATS, lane, h, tt.t, bufB, bufA, c, S, A, J);
            tile_pipe(rec + (tau + 1) * MOB_LAST_TILE_FLOATS, lane, h, tt.t, bufA, bufB, c, S, A, J);
        }
        if (tau < KT) {
            tile_pipe(rec + tau * MOB_LAST_TILE_FLOATS, lane, h, tt.t, bufB, bufA, c, S, A, J);
            segments4_last<false>(bufB, c, S, A, J, K, h);
        } else {
            segments4_last<false>(bufA, c, S, A, J, K, h);
        }
    } else {
        // tile tau+1's 12 matrix instructions and tile tau's segment math in one loop body (tile_pipe_h).  (`cur = nxt` costs 8
        // v_mov_b64 per tile; the two-tiles-per-trip ping-pong of the fp32 path avoids them but spills 10 registers under the 128
        // budget of the 16-wave instantiation and measured 1 % slower.)
        if constexpr (PINGPONG) {          // two tiles per trip, the accumulators change roles: no register copies.  LEAN kernels only, which the
                                           // launcher runs guarded: the one-piece softplus (so3_math.h seg_s7_stage SAFE = false)
            f32x16 bufA = Mlp<1>::last(rec, lane, h, tt), bufB;
            int tau = 1;
            int woff = MOB_LAST_TILE_FLOATS + 4 * lane, boff = MOB_LAST_TILE_FLOATS + MOB_LAST_TILE_BIAS + 16 * h;
            for (; tau + 1 < KT; tau += 2) {
                asm volatile("" : "+v"(woff), "+v"(boff));
                tile_pipe_h_off<false>(rec, woff, boff, tt, bufB, bufA, c, S, A, J);
                tile_pipe_h_off<false>(rec, woff + MOB_LAST_TILE_FLOATS, boff + MOB_LAST_TILE_FLOATS, tt, bufA, bufB, c, S, A, J);
                woff += 2 * MOB_LAST_TILE_FLOATS;
                boff += 2 * MOB_LAST_TILE_FLOATS;
            }
            if (tau < KT) {
                asm volatile("" : "+v"(woff), "+v"(boff));
                tile_pipe_h_off<false>(rec, woff, boff, tt, bufB, bufA, c, S, A, J);
                segments4_last<true, false>(bufB, c, S, A, J, K, h);
            } else {
                segments4_last<true, false>(bufA, c, S, A, J, K, h);
            }
            return;
        }
        f32x16 cur = Mlp<1>::last(rec, lane, h, tt);
        for (int tau = 1; tau < KT; ++tau) {
            f32x16 nxt;
            tile_pipe_h<!FASTSP>(rec + tau * MOB_LAST_TILE_FLOATS, lane, h, tt, nxt, cur, c, S, A, J);
            cur = nxt;
            fair.tick();
        }
        segments4_last<true, !FASTSP>(cur, c, S, A, J, K, h);
    }
}

// forward fc_last tile tau + 1 (24 matrix instructions) with the segment math of the finished tile tau behind them: slice st of segment g
// behind instruction 6 g + st (so3_math.h seg_s7_stage6: 6 - 10 VALU per slice)
template <int M>
__device__ __forceinline__ void last_slot_b3(const float *w_tile, const float *w_next, int lane, const ActFrag3 &in, f32x16 &nxt, Ops3 &o,
                                             const f32x16 &cur, SegS6 (&seg)[4], const MobiusCtx &c, float &S, float &A, float &J) {
    nxt = b3_mfma<M>(o, in, nxt);
    b3_prefetch<M>(o, w_tile, w_next, lane);
    constexpr int g = M / 6, st = M % 6;
    seg_s7_stage6<st>(seg[g], cur[4 * g], cur[4 * g + 1], cur[4 * g + 2], cur[4 * g + 3], c.f, S, A, J);
    // pinned in its slot: sched_barrier orders the machine scheduler only, the IR optimiser would sink the slices (their results are consumed
    // slots later) into one block behind the 24 matrix instructions -- where this kernel measured VALU time + matrix time
    if constexpr (st == 0) asm volatile("" : "+v"(seg[g].a), "+v"(seg[g].b));
    else if constexpr (st == 1) asm volatile("" : "+v"(seg[g].bb), "+v"(seg[g].e), "+v"(seg[g].num));
    else if constexpr (st == 2) asm volatile("" : "+v"(seg[g].t), "+v"(seg[g].c), "+v"(seg[g].z), "+v"(seg[g].ex));
    else if constexpr (st == 3) asm volatile("" : "+v"(seg[g].p), "+v"(seg[g].lg));
    else if constexpr (st == 4) asm volatile("" : "+v"(seg[g].p));
    else asm volatile("" : "+v"(S), "+v"(A), "+v"(J));
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (M + 1 < 24) last_slot_b3<M + 1>(w_tile, w_next, lane, in, nxt, o, cur, seg, c, S, A, J);
}

// forward tile phase of the bf16x3 kernels: one ring unit (four fc_last tiles) at a time, each pipelined -- tile tau + 1's matrix
// instructions with tile tau's segment math behind them (last_slot_b3)
template <class Ctl>
__device__ __forceinline__ void mobius_fwd_tiles_b3(Ctl &ctl, int KT, int K, int lane, int h, const ActFrag3 &tt,
                                                    const MobiusCtx &c, float &S, float &A, float &J) {
    typedef Lay<2> L2;
    for (int t0 = 0; t0 < KT; t0 += L2::MAX_TILES_IN_LDS) {
        const int nt = min(L2::MAX_TILES_IN_LDS, KT - t0);
        const float *rec = ctl.begin_unit();
        f32x16 cur = Mlp<2>::last(rec, lane, h, tt);
        for (int i = 1; i < nt; ++i) {
            const float *t = rec + i * L2::LAST_TILE_FLOATS;
            f32x16 nxt = load_bias16(t + L2::LAST_TILE_BIAS + h * 16);
            Ops3 o;
            o.load(0, t, 0, lane);
            o.load(1, t, 1, lane);
            SegS6 seg[4];
            __builtin_amdgcn_sched_barrier(0);
            last_slot_b3<0>(t, nullptr, lane, tt, nxt, o, cur, seg, c, S, A, J);
            cur = nxt;
        }
        if (t0 + nt == KT) segments4_last<true, true>(cur, c, S, A, J, K, h);
        else segments4<true, true>(cur, c, S, A, J);
    }
}

// forward, K > 64: fc_last tiles restaged synchronously 8 at a time (staging mode SYNC only)
template <int PREC>
__device__ __forceinline__ void mobius_fwd_tiles_restage(float *lds, const float *layer_params, int KT, int K, int lane, int h,
                                                         const typename Mlp<PREC>::Act &tt, const MobiusCtx &c, float &S,
                                                         float &A, float &J, int tid, int nthreads) {
    for (int tau = 0; tau < KT; ++tau) {
        if (tau > 0 && (tau % Lay<PREC>::MAX_TILES_IN_LDS) == 0) {
            __syncthreads();
            int nt = min(Lay<PREC>::MAX_TILES_IN_LDS, KT - tau);
            stage_floats(lds + Lay<PREC>::LAST, layer_params + Lay<PREC>::LAST + (size_t)tau * Lay<PREC>::LAST_TILE_FLOATS,
                         nt * Lay<PREC>::LAST_TILE_FLOATS, tid, nthreads);
            __syncthreads();
        }
        f32x16 o = Mlp<PREC>::last(lds + Lay<PREC>::LAST + (tau % Lay<PREC>::MAX_TILES_IN_LDS) * Lay<PREC>::LAST_TILE_FLOATS, lane, h, tt);
        if (tau + 1 == KT) segments4_last<PREC != 0>(o, c, S, A, J, K, h);
        else segments4<PREC != 0>(o, c, S, A, J);
    }
}

// FASTSP (kernels with the one-piece softplus, launched guarded): a weight sum below kMinWeightSum * K means every segment weight is so
// small that fl(1 + 2^s) has rounded the weights themselves (relative error 2^-24 / weight): flagged like an overflow, the exact-fp32
// re-run evaluates the full softplus.  (!(S >= x) also catches a NaN sum.)
constexpr float kMinWeightSum = 1.0f / 128.0f;
template <bool HALF, bool FASTSP = false>
__device__ __forceinline__ void mobius_fwd_finish(const MobiusCtx &c, float S, float A, float J, Rot &R, float &ldj, bool &bad, float min_s) {
    S = pair_sum(S);
    A = pair_sum(A);
    J = pair_sum(J);
    if constexpr (FASTSP) bad |= !(S >= min_s);
    const float invS = hw_rcp(S);
    float sn, cs;
    // HALF: theta' = pi + 2 A / S, and sin / cos of pi + d are -sin d, -cos d
    if (HALF) sincos_twice(A * invS, sn, cs);                  // |A / S| <= atan(0.98) < pi/4: no quadrant reduction (so3_math.h)
    else sincos_small(A * invS, sn, cs);
    if (HALF) { sn *= -kInvSquash; cs *= -kInvSquash; }                                   // the frame of the split-precision path is 0.7-scaled
    const v3f tx = c.f.v * sn + c.f.r * cs;
    const v3f tz = normalize3(c.cyc ? cross3(tx, c.y) : cross3(c.y, tx));               // mobiusflow.py:75-79
    set_col(R, c.p0, tx);
    set_col(R, c.p2, tz);
    ldj = fmaf(0.693147180559945309f, hw_log2(J * invS), ldj);        // J / S in [0.18, 5.7]: the hardware log2 (1 ulp) needs none of logf's range handling
}

// inverse: the 4*KT segment parameters of this lane stay in registers for the 15 bisection steps
template <int KT>
struct InvSegs {
    float sp[4 * KT], ur[4 * KT], uv[4 * KT];
    float q[4 * KT];          // sp * (1 - |u|^2): the theta-independent numerator of the segment's derivative term
};

// kt: tiles this layer really has (<= KT, the instantiation's capacity), K: its real segment count; slots beyond them get weight 0.
// KT > Lay<PREC>::MAX_TILES_IN_LDS (K > 64, synchronous staging only): the second half of the fc_last image is staged in the middle.
struct NoRing {};                       // (the staging of the PREC 0 / 1 kernels needs no unit controller)
template <int KT, int PREC, bool FASTSP = false, class Ctl = NoRing>
__device__ __forceinline__ void mobius_inv_tiles(float *lds, const float *layer_params, int kt, int K, int lane, int h, const typename Mlp<PREC>::Act &tt,
                                                 const MobiusCtx &c, InvSegs<KT> &sg, float &S, int tid, int nthreads, float4 *stash = nullptr, Ctl *ctl = nullptr) {
    constexpr bool RING = !std::is_same<Ctl, NoRing>::value;
    const float *tiles = lds + Lay<PREC>::LAST;                 // RING: the current unit (four tiles)
#pragma unroll
    for (int tau = 0; tau < KT; ++tau) {
        if (tau < kt) {                                         // wave uniform
            if constexpr (RING) {
                if ((tau % Lay<PREC>::MAX_TILES_IN_LDS) == 0) tiles = ctl->begin_unit();
            } else if constexpr (KT > Lay<PREC>::MAX_TILES_IN_LDS) {
                if (tau > 0 && (tau % Lay<PREC>::MAX_TILES_IN_LDS) == 0) {
                    __syncthreads();
                    stage_floats(lds + Lay<PREC>::LAST, layer_params + Lay<PREC>::LAST + (size_t)tau * Lay<PREC>::LAST_TILE_FLOATS,
                                 min(Lay<PREC>::MAX_TILES_IN_LDS, kt - tau) * Lay<PREC>::LAST_TILE_FLOATS, tid, nthreads);
                    __syncthreads();
                }
            }
            f32x16 o = Mlp<PREC>::last(tiles + (tau % Lay<PREC>::MAX_TILES_IN_LDS) * Lay<PREC>::LAST_TILE_FLOATS, lane, h, tt);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                squash_center(o[4 * g + 1], o[4 * g + 2], o[4 * g + 3], c.f, sg.ur[4 * tau + g], sg.uv[4 * tau + g]);
                // the matrix output is s log2 e (layout.h S_PRESCALE).  Split precision: softplus / ln 2 in one piece, as the forward segment
                // does -- the root finder and the log-det only use ratios of the weights -- in its overflow-safe form (once per segment and layer:
                // free next to the root finder)
                // FASTSP (round 4; guarded calls only): the lean one-piece form, 4 instructions instead of 11 -- the all-weights-tiny corner it
                // cannot resolve is flagged by mobius_inv_finish's weight-sum test and re-run on the exact-fp32 kernels, as in the forward pass
                float sp = PREC != 0 ? (FASTSP ? softplus2_lean(o[4 * g]) : softplus2_safe(o[4 * g])) : softplus(S_UNSCALE * o[4 * g]);
                if (8 * tau + 2 * g + h >= K) sp = 0.f;          // pad segment of a K % 8 != 0 layer: weight 0 AFTER the activation
                sg.sp[4 * tau + g] = sp;
                sg.q[4 * tau + g] = sp * (1.0f - fmaf(sg.uv[4 * tau + g], sg.uv[4 * tau + g], sg.ur[4 * tau + g] * sg.ur[4 * tau + g]));
                S += sp;
            }
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) sg.sp[4 * tau + g] = sg.ur[4 * tau + g] = sg.uv[4 * tau + g] = sg.q[4 * tau + g] = 0.f;
        }
        // keep the tiles in order: letting the scheduler hoist all 8 tiles' MFMAs (8 x 16 accumulators) on top of the
        // 96 live segment registers spills; the inverse is VALU-bound in the bisection anyway
        __builtin_amdgcn_sched_barrier(0);
    }
    // K > 8 KT segments (the largest instantiation only, KT = 16: K > 128): the remaining tiles' segment parameters go to the wave's stash
    if constexpr (KT == 16) {
        for (int tau = KT; tau < kt; ++tau) {
            if constexpr (RING) {
                if ((tau % Lay<PREC>::MAX_TILES_IN_LDS) == 0) tiles = ctl->begin_unit();
            } else if ((tau % Lay<PREC>::MAX_TILES_IN_LDS) == 0) {
                __syncthreads();
                stage_floats(lds + Lay<PREC>::LAST, layer_params + Lay<PREC>::LAST + (size_t)tau * Lay<PREC>::LAST_TILE_FLOATS,
                             min(Lay<PREC>::MAX_TILES_IN_LDS, kt - tau) * Lay<PREC>::LAST_TILE_FLOATS, tid, nthreads);
                __syncthreads();
            }
            const f32x16 o = Mlp<PREC>::last(tiles + (tau % Lay<PREC>::MAX_TILES_IN_LDS) * Lay<PREC>::LAST_TILE_FLOATS, lane, h, tt);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float ur, uv;
                squash_center(o[4 * g + 1], o[4 * g + 2], o[4 * g + 3], c.f, ur, uv);
                float sp = PREC != 0 ? (FASTSP ? softplus2_lean(o[4 * g]) : softplus2_safe(o[4 * g])) : softplus(S_UNSCALE * o[4 * g]);
                if (8 * tau + 2 * g + h >= K) sp = 0.f;
                stash[(size_t)(4 * (tau - KT) + g) * 64 + lane] = make_float4(sp, ur, uv, sp * (1.0f - fmaf(uv, uv, ur * ur)));
                S += sp;
            }
        }
        if (kt > KT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the root finder reads the stash back (same lane, same addresses)
    }
}

// Round 6: the same tile loop for the split-precision kernels with every fc_last tile resident (K <= 64), software pipelined BY HAND.
// Left to the compiler (mobius_inv_tiles at its 230-register budget) every k-step was  ds_read_b128 -> s_waitcnt lgkmcnt(0) -> MFMA  with
// ONE operand register quad reused for all eight reads of a tile: eight fully exposed LDS latencies inside a dependent chain of twelve
// matrix instructions, ~1000 of the ~1850 cycles a wave spent per tile (r5 stamps: the tile phase was 27 % of the kernel).  Here the
// operands of k-step s + 2 are requested behind the last matrix instruction of k-step s (two operand pairs in flight), the NEXT tile's
// first two k-steps and its bias (the other accumulator: the two change roles, no copies) are requested in front of the segment math of
// the finished tile, and the pad mask is a compare against a scalar (it was 32 hoisted lane masks in spilled SGPR pairs: two
// v_readlane per segment).  Same arithmetic per rotation as mobius_inv_tiles (the squash reciprocal folds the 0.7: one rounding moves).
template <int KT, bool FASTSP>
__device__ __forceinline__ void mobius_inv_tiles_pipe(const float *lds, int K, int lane, int h, const ActFrag &tt, const MobiusCtx &c,
                                                      InvSegs<KT> &sg, float &S) {
    static_assert(KT <= MOB_MAX_TILES_IN_LDS, "every fc_last tile resident");
    // Two per-lane offsets the optimiser cannot fold (see tile_pipe_h_off): every LDS read then carries its tile position as an immediate.
    // Folded into constants, the positions beyond 64 KiB (the DS offset field is 16 bits) become one address register per read group,
    // hoisted out of the layer loop and spilled (the first build of this function: 56 spilled registers, all of them LDS addresses).
    int woff = MOB_LAST + 4 * lane, boff = MOB_LAST + MOB_LAST_TILE_BIAS + 16 * h;
    asm volatile("" : "+v"(woff), "+v"(boff));
    constexpr int HALF = (KT + 1) / 2;                            // tiles from HALF on are addressed from a second pair of bases
    int woff2 = woff + HALF * MOB_LAST_TILE_FLOATS, boff2 = boff + HALF * MOB_LAST_TILE_FLOATS;
    asm volatile("" : "+v"(woff2), "+v"(boff2));
    auto wt = [&](int tau) { return tau < HALF ? lds + woff + tau * MOB_LAST_TILE_FLOATS : lds + woff2 + (tau - HALF) * MOB_LAST_TILE_FLOATS; };
    auto bt = [&](int tau) { return tau < HALF ? lds + boff + tau * MOB_LAST_TILE_FLOATS : lds + boff2 + (tau - HALF) * MOB_LAST_TILE_FLOATS; };
    f32x16 acc = load_bias16(bt(0));
    h8 ah[2], al[2];
    ah[0] = lds_h8(wt(0), 0 * 64); al[0] = lds_h8(wt(0), 1 * 64);
    ah[1] = lds_h8(wt(0), 2 * 64); al[1] = lds_h8(wt(0), 3 * 64);
#pragma unroll
    for (int tau = 0; tau < KT; ++tau) {                          // the caller guarantees that the layer has exactly KT tiles
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            __builtin_amdgcn_sched_barrier(0);
            acc = RNF_MFMA_H(ah[ks & 1], tt.hi[ks], acc);
            acc = RNF_MFMA_H(ah[ks & 1], tt.lo[ks], acc);
            acc = RNF_MFMA_H(al[ks & 1], tt.hi[ks], acc);
            if (ks < 2) {                                         // k-step ks + 2 of this tile
                ah[ks & 1] = lds_h8(wt(tau), ((ks + 2) * 2 + 0) * 64);
                al[ks & 1] = lds_h8(wt(tau), ((ks + 2) * 2 + 1) * 64);
            } else if (tau + 1 < KT) {                            // k-step ks - 2 of the next tile
                ah[ks & 1] = lds_h8(wt(tau + 1), ((ks - 2) * 2 + 0) * 64);
                al[ks & 1] = lds_h8(wt(tau + 1), ((ks - 2) * 2 + 1) * 64);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // stage A: everything that reads the accumulator (7 instructions per segment); behind it the accumulator's registers are free for
        // the next tile's bias, whose LDS latency then sits under stage B instead of in front of the next matrix chain
        float wr[4], wv[4], sx[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float w0 = acc[4 * g + 1], w1 = acc[4 * g + 2], w2 = acc[4 * g + 3];
            wr[g] = fmaf(w2, c.f.r.z, fmaf(w1, c.f.r.y, w0 * c.f.r.x));
            wv[g] = fmaf(w2, c.f.v.z, fmaf(w1, c.f.v.y, w0 * c.f.v.x));
            sx[g] = acc[4 * g];
        }
        asm volatile("" : "+v"(wr[0]), "+v"(wr[1]), "+v"(wr[2]), "+v"(wr[3]), "+v"(wv[0]), "+v"(wv[1]), "+v"(wv[2]), "+v"(wv[3]),
                          "+v"(sx[0]), "+v"(sx[1]), "+v"(sx[2]), "+v"(sx[3]));
        if (tau + 1 < KT) acc = load_bias16(bt(tau + 1));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            // 0.7 / (1 + |w|) = 1 / (1/0.7 + |w| / 0.7)   (flow/mobiusflow.py:72)
            const float sc = hw_rcp(fmaf(hw_sqrt(fmaf(wv[g], wv[g], wr[g] * wr[g])), kInvSquash, kInvSquash));
            const float ur = wr[g] * sc, uv = wv[g] * sc;
            float sp = FASTSP ? softplus2_lean(sx[g]) : softplus2_safe(sx[g]);
            if (tau == KT - 1) sp = (h < K - 8 * tau - 2 * g) ? sp : 0.f;      // pad segment of a K % 8 != 0 layer (last tile only): weight 0 AFTER the activation
            sg.ur[4 * tau + g] = ur;
            sg.uv[4 * tau + g] = uv;
            sg.sp[4 * tau + g] = sp;
            sg.q[4 * tau + g] = fmaf(-sp, fmaf(uv, uv, ur * ur), sp);
            S += sp;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

// Order of the inverse pass's root-finder iteration.  3 (shipped): Halley.  4 (-DRNF_RF_ORDER=4, measured and NOT shipped): Householder's
// method with the third derivative, four more packed instructions per segment pair and pass (+13 % per pass: 1.53 against 1.35 ms per pass
// over C5u's 42 layers x 2^20 rotations).  On sharply peaked weights (softmax of 6 x N(0,1) logits, centres 6 x N(0,1): the start is off by
// > 0.17 for a tenth of the rotations) it brings every wave from 2.85 passes to 2.0 with the same cell agreement (CPU emulation, 16384
// samples); on BASELINE's C5 / C5u weights the third-order iteration already needs 2.0 passes per wave (fixed-2-pass build 11.62 ms,
// shipped 11.64), so the fourth order only adds its per-pass cost: C5u 11.79 -> 12.26 ms, C5 22.59 -> 23.10 (profiles/README.md).
#ifndef RNF_RF_ORDER
#define RNF_RF_ORDER 3
#endif
// Root of BinFind (flow/mobiusflow.py:189-224).  The reference bisects f(theta) = sum_k wt_k phi_k(theta) - target on
// [pi/2, 3pi/2] exactly 15 times (its batch-global stop test max(b - a) < 1e-4 is data independent) and returns the LAST
// midpoint, i.e. the centre of the cell of the grid  pi/2 + n * pi/2^14  that contains the root:
//     theta_ref = pi/2 + (n + 1/2) * pi/2^14,   n = floor((theta* - pi/2) * 2^14 / pi).
// f is strictly increasing (f' = sum_k wt_k c_k >= (1-0.7)/(1+0.7) > 0) and the root lies strictly inside the bracket (a
// Moebius map with |w| < 0.7 moves a point by less than 2 asin 0.7 = 88.9 degrees), so instead of 15 passes over the
// segments the kernel finds theta* with a bracket-safeguarded Newton iteration (phi_k and its derivative c_k come out of the
// same evaluation) and then snaps it to that grid: the same returned iterate as the reference's bisection, except when
// theta* lies within rounding error of a cell boundary -- where the reference's own fp32 and fp64 runs disagree too.
// n_over > 0 (KT = 16 only): `stash` holds the parameters of this lane's segments beyond the 4 KT in registers (mobius_inv_tiles); every
// sum over the segments continues over them (one float4 load per segment and pass: a K > 128 inverse is rare, L2 absorbs it)
template <int KT>
__device__ __forceinline__ void mobius_inv_finish(const MobiusCtx &c, const InvSegs<KT> &sg, float S, Rot &R, float &ldj, const float4 *stash,
                                                  int n_over, int lane, float min_s, bool check_s, bool &bad, bool first4 = true) {
    S = pair_sum(S);
    // lean softplus: every weight tiny (or a NaN sum) -> exact-fp32 re-run.  (`bad` by reference and the test switched by a value: a
    // `bool *` that is either &bad or nullptr kept the flag in scratch memory, and every update of it was a load + store on the vector
    // memory counter in front of the scratch-tile loads of the hidden phase)
    bad |= check_s && !(S >= min_s);
    const float invS = hw_rcp(S);
    float lo = 0.5f * kPi, hi = 1.5f * kPi, th;
    {   // Starting point: the exact inverse of ONE Moebius map with the weighted mean centre m = sum wt_k u_k (the map with centre -m
        // applied to the target point).  To first order in the centres the mixture IS that map, and the Newton iteration then needs
        // 3.8 - 4.0 passes per wave on trained-like weights instead of 5.4 - 6.7 from theta = pi (tests/test_inverse_rootfinder.py).
        f2 mr2 = {0.f, 0.f}, mv2 = {0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4 * KT; s += 2) {
            const f2 sp = {sg.sp[s], sg.sp[s + 1]};
            mr2 = __builtin_elementwise_fma(sp, f2{sg.ur[s], sg.ur[s + 1]}, mr2);
            mv2 = __builtin_elementwise_fma(sp, f2{sg.uv[s], sg.uv[s + 1]}, mv2);
        }
        if constexpr (KT == 16)
            for (int s = 0; s < n_over; ++s) {
                const float4 p = stash[(size_t)s * 64 + lane];
                mr2.x = fmaf(p.x, p.y, mr2.x);
                mv2.x = fmaf(p.x, p.z, mv2.x);
            }
        const float mr = pair_sum(mr2.x + mr2.y) * invS;
        const float mv = pair_sum(mv2.x + mv2.y) * invS;
        if constexpr (RNF_INV_PI) {
            // target = pi (mobius_begin): z_target = (-1, 0), so (a, b) = (-m) conj(z_target) = (mr, mv), and |b / (1 - a)| <= tan(asin 0.7) < 1
            // (|m| < 0.7): the segments' own single-quadrant arctangent, no octant logic, no wrap
            th = fminf(fmaxf(fmaf(2.0f, atan_unit(-mv * hw_rcp(1.0f - mr)), kPi), lo + 1.0e-3f), hi - 1.0e-3f);
        } else {
            float st, ct;
            sincos_small(c.target, st, ct);
            const float a = -fmaf(mv, st, mr * ct);                     // (a, b) = (-m) conj(z_target)
            const float b = fmaf(mr, st, -mv * ct);
            float d = angle_0_2pi(-b, 1.0f - a);                        // 1 - a >= 0.3: the angle is in (-pi/2, pi/2), returned mod 2 pi
            d = d > kPi ? d - kTwoPi : d;
            th = fminf(fmaxf(fmaf(2.0f, d, c.target), lo + 1.0e-3f), hi - 1.0e-3f);
        }
    }
    bool done = false;
    float prev = 1.0f;                                                     // size of the previous step (1: none yet)
    // Round 6 (RNF_RF_CENTRE): every pass after the first is evaluated AT THE CENTRE of the grid cell its predecessor's step landed in.  A step
    // from a centre that stays inside the centre's own cell CONFIRMS the cell -- the root is there to ~ C (cell/2)^3 -- and the pass has just
    // summed f' at the very point the log-determinant needs it: such a lane is finished (`conf`, Jc) and the closing evaluation of f' at the
    // centre (8 packed + 2 v_rcp per segment pair, 0.38 of a pass) is run only by waves that hold a lane which converged WITHOUT confirming
    // (a root within the first step's error of a cell boundary, a far start on peaked weights).  CPU emulation, fp32 (32-rotation waves,
    // softmax-of-g-N(0,1) weights): waves that still run the closing evaluation g = 0.3: 0 %, g = 1: 8 %, g = 3 and 6: 95 - 100 % (as before);
    // passes per wave and cell agreement with the reference's bisection unchanged.  Which of the two sums serves a lane follows from the lane's own
    // iterates alone, so a rotation's log-determinant still does not depend on the wave it travels in.
#ifndef RNF_RF_CENTRE
#define RNF_RF_CENTRE 1
#endif
#ifndef RNF_RF_E1FOLD_CLOSE
#define RNF_RF_E1FOLD_CLOSE 1
#endif
    bool conf = false;
    float Jc = 0.f;
    const float cell = kPi * (1.0f / 16384.0f);
    auto cell_of = [&](float t) { return fminf(fmaxf(floorf((t - 0.5f * kPi) * (16384.0f / kPi)), 0.f), 16383.f); };
    // Round 6 (first4, a property of the FLOW: rnf_api.hip, include/rnf_hip.h desc column 5 bits 16..17; default off): the FIRST pass -- the one
    // that starts up to 0.2 - 0.5 rad from the root on sharply peaked weights -- takes a FOURTH-order step
    // (Householder's method with f3: four more packed instructions per segment pair, in this pass only) and measures the third-order
    // iteration's asymptotic error constant C = |3 f2^2 - 2 f1 f3| / (12 f1^2) on the way; every later pass is the Halley step of round 4 and
    // a lane stops when 4 C step^3 is below the fp32 spacing of theta (the constant from the derivatives, not from the ratio of two steps --
    // which a first step of another order would falsify).  CPU emulation in fp32 (tests/test_inverse_rootfinder.py; softmax-of-g-N(0,1)
    // weights, centres g N(0,1)): passes a wave of 32 rotations needs, g = 6: 2.85 -> 2.09 (91 % of the waves in two passes, 15 % before),
    // g = 12: 2.89 -> 2.26, g = 3: 2.15 -> 2.00, g = 1 (and BASELINE's synthetic weights): 2.00 -> 2.00; cell agreement with the reference's
    // bisection 99.95 -> 99.98 %.  Costs mild weights its extra sums (BASELINE's synthetic C5u +2 %, C5q +4 %), hence opt-in per flow.
#ifndef RNF_RF_FIRST4
#define RNF_RF_FIRST4 1
#endif
    float cerr = 80.0f;                                                    // 4 C (until the first pass has measured it: the old floor)
    auto pass = [&](auto o4_tag, int it) -> bool {
        constexpr bool O4 = (decltype(o4_tag)::value && RNF_RF_FIRST4) || RNF_RF_ORDER >= 4;
        float sn, cs;
        sincos_pi_band(th, sn, cs);                                   // th stays inside [pi/2, 3 pi/2] (the sign bracket)
        // per segment (so3_math.h mobius_angle): phi = th + 2 atan(-b / (1 - a)), c = (1 - |u|^2) / (b^2 + (1 - a)^2), (a, b) = u conj(z);
        // the constant parts are hoisted: sum sp phi = th S + 2 sum sp atan(.), sum sp c = sum q / (b^2 + (1 - a)^2)   (21 + 2 instead of 27 + 2).
        // Round 4: the SECOND derivative rides along -- d/dtheta of q / den is 2 q b / den^2 (da/dtheta = b, db/dtheta = -a) -- three more
        // packed instructions per segment pair, for a third-order (Halley) step: on trained-like weights a wave needs 2.9 passes instead of
        // 4.0 (tests/test_inverse_rootfinder.py), the pass costs 12 % more.
        float acc = 0.f, der = 0.f, dd = 0.f, d3 = 0.f;
        // two segments per instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32): no matrix instruction runs during the root finder,
        // so the packed fp32 VALU forms pay here (-6.5 % on the whole inverse pass) -- beside MFMAs they are an anti-lever
        {
            f2 acc2 = {0.f, 0.f}, der2 = {0.f, 0.f}, dd2 = {0.f, 0.f}, e3a = {0.f, 0.f}, e3b = {0.f, 0.f};
            const f2 sn2 = {sn, sn}, cs2 = {cs, cs};
#pragma unroll
            for (int s = 0; s < 4 * KT; s += 2) {
                const f2 ur = {sg.ur[s], sg.ur[s + 1]}, uv = {sg.uv[s], sg.uv[s + 1]};
                const f2 sp = {sg.sp[s], sg.sp[s + 1]}, q = {sg.q[s], sg.q[s + 1]};
                const f2 a = __builtin_elementwise_fma(uv, sn2, ur * cs2);
                const f2 b = __builtin_elementwise_fma(uv, cs2, -(ur * sn2));
                // (1 - a folded into two instructions, a only where the fourth-order sums need it: one instruction fewer on paper, 41 spilled
                // registers in this kernel -- the packed form wants its constant in a register pair)
                const f2 e1 = 1.0f - a;
                const f2 r1 = {hw_rcp(e1.x), hw_rcp(e1.y)};
                const f2 t = -b * r1;
                const f2 z = t * t;
                f2 p = __builtin_elementwise_fma(f2{2.456724578e-03f, 2.456724578e-03f}, z, f2{-1.440135792e-02f, -1.440135792e-02f});
                p = __builtin_elementwise_fma(p, z, f2{3.978122362e-02f, 3.978122362e-02f});
                p = __builtin_elementwise_fma(p, z, f2{-7.234857378e-02f, -7.234857378e-02f});
                p = __builtin_elementwise_fma(p, z, f2{1.049894609e-01f, 1.049894609e-01f});
                p = __builtin_elementwise_fma(p, z, f2{-1.416122920e-01f, -1.416122920e-01f});
                p = __builtin_elementwise_fma(p, z, f2{1.998590677e-01f, 1.998590677e-01f});
                p = __builtin_elementwise_fma(p, z, f2{-3.333259703e-01f, -3.333259703e-01f});
                p = __builtin_elementwise_fma(p, z, f2{9.999998864e-01f, 9.999998864e-01f});
                acc2 = __builtin_elementwise_fma(sp, p * t, acc2);
                const f2 den = __builtin_elementwise_fma(b, b, e1 * e1);
                const f2 r2 = {hw_rcp(den.x), hw_rcp(den.y)};
                const f2 cq = q * r2;                               // c = q / den, the segment's term of f'
                der2 = RNF_RF_CENTRE ? __builtin_elementwise_fma(q, r2, der2) : der2 + cq;     // (the closing evaluation's own instruction)
                if constexpr (O4) {                  // dc/dtheta = 2 c b / den,  d2c/dtheta2 = 2 c (4 b^2 / den - a) / den   (d den/dtheta = -2 b)
                    const f2 cr = cq * r2;
                    const f2 cb = cr * b;                           // c b / den
                    dd2 = dd2 + cb;
                    e3a = __builtin_elementwise_fma(cb, b * r2, e3a);
                    e3b = __builtin_elementwise_fma(cr, a, e3b);
                } else {
                    dd2 = __builtin_elementwise_fma(cq, b * r2, dd2);
                }
            }
            acc = acc2.x + acc2.y;
            der = der2.x + der2.y;
            dd = dd2.x + dd2.y;
            if constexpr (O4) d3 = fmaf(4.0f, e3a.x + e3a.y, -(e3b.x + e3b.y));
        }
        if constexpr (KT == 16)
            for (int s = 0; s < n_over; ++s) {                         // the same per-segment evaluation on the stashed parameters
                const float4 p = stash[(size_t)s * 64 + lane];
                const float a = fmaf(p.z, sn, p.y * cs), b = fmaf(p.z, cs, -(p.y * sn));
                const float e1 = 1.0f - a;
                const float t = -b * hw_rcp(e1);
                acc = fmaf(p.x, atan_unit(t), acc);
                const float r2 = hw_rcp(fmaf(b, b, e1 * e1));
                const float cq = p.w * r2;
                der = RNF_RF_CENTRE ? fmaf(p.w, r2, der) : der + cq;
                if constexpr (O4) {
                    const float cr = cq * r2, cb = cr * b;
                    dd += cb;
                    d3 += fmaf(4.0f * cb, b * r2, -(cr * a));
                } else {
                    dd = fmaf(cq, b * r2, dd);
                }
            }

        const float fx = fmaf(2.0f * pair_sum(acc), invS, th) - c.target;
        const float dfx = pair_sum(der) * invS;                            // f1 = df/dtheta  > 0
        const float ddfx = 2.0f * pair_sum(dd) * invS;                     // f2, the second derivative
        if (RNF_RF_CENTRE) { if (fx < 0.f) lo = fmaxf(lo, th); else hi = fminf(hi, th); }   // (a centre may lie just outside the bracket)
        else { if (fx < 0.f) lo = th; else hi = th; }
        // Halley: theta - f / (f1 - f f2 / (2 f1)); a non-positive denominator (far from the root) falls back to the Newton step
        const float hden = fmaf(-0.5f * fx * ddfx, hw_rcp(dfx), dfx);
        float nt = th - fx * hw_rcp(hden > 0.25f * dfx ? hden : dfx);
        if constexpr (O4) {
            // Fourth order (Householder's method with the third derivative f3):  theta - 3 f (2 f1^2 - f f2) / (6 f1 (f1^2 - f f2) + f^2 f3).
            // The starting point is off by > 0.17 for a tenth of the rotations (0.47 at worst); a third-order step brings those to ~ 1e-2,
            // one pass short of the fp32 spacing, so that 85 % of the waves ran a third pass for 6 % of their lanes.  The fourth-order step
            // gets every lane there in two (2.0 - 2.1 passes per wave instead of 2.85 in the CPU emulation) for four more packed instructions
            // per segment pair and pass.
            const float d3fx = 2.0f * pair_sum(d3) * invS;
            const float df2 = dfx * dfx, ffd = fx * ddfx;
            const float den4 = fmaf(6.0f * dfx, df2 - ffd, fx * fx * d3fx);
            const float num4 = 3.0f * fx * fmaf(2.0f, df2, -ffd);
            if (den4 > 1.5f * df2 * dfx) nt = th - num4 * hw_rcp(den4);    // a denominator below 1/4 of its value at the root: the Halley / Newton step
            // the Halley iteration's error constant here (e_next ~ C e^3), floored at 1, times a margin of 4
            cerr = 4.0f * fmaxf(fabsf(fmaf(3.0f * ddfx, ddfx, -2.0f * dfx * d3fx)) * hw_rcp(12.0f * df2), 1.0f);
        }
        if (!(nt >= lo && nt <= hi)) nt = 0.5f * (lo + hi);               // keep the iterate inside the sign bracket
        if (done) nt = th;                                                // a converged lane stays put
        // Convergence of order p (4; 3 for the Halley build): the error left behind a step of size d is ~ C d^p with C estimated from this step
        // and the previous one (d / d_prev^p, floored).  A lane stops when that prediction is below the fp32 spacing of theta (2.4e-7) -- the
        // pass that would only confirm it is not run -- or, as before, after a step of <= 1e-4.
        const float step = fabsf(nt - th);
        bool conv_now = false;
        if constexpr (RNF_RF_CENTRE && RNF_RF_FIRST4 && RNF_RF_ORDER < 4) {
            const bool was = done;
            if (it > 0) {
                const float c3 = first4 ? cerr : fmaxf(step * hw_rcp(prev * prev * prev), 20.0f);
                conv_now = !was && (step <= 1.0e-4f || (step <= 5.0e-3f && c3 * step * step * step <= 2.4e-7f));
                const bool same = !was && cell_of(nt) == cell_of(th);     // th is a centre here: the step stayed inside its cell
                if (same) { conf = true; Jc = der; nt = th; }             // (der: this lane's half of f' S at th; pair-summed at the end)
                done = was || conv_now || same;
            }
            // (a first pass ends no lane: one that started on the root confirms its cell in the second, which its wave runs anyway)
            prev = done ? prev : step;
            // the next evaluation point: the centre of the cell the step landed in; a lane that converged without confirming keeps its iterate
            th = was ? th : (conv_now && !conf) ? nt : conf ? th : fmaf(cell_of(nt) + 0.5f, cell, 0.5f * kPi);
#ifdef RNF_KO_FIXED_PASSES
            return it + 1 == RNF_KO_FIXED_PASSES;
#else
            return __all(done);
#endif
        } else
        if constexpr (RNF_RF_FIRST4 && RNF_RF_ORDER < 4) {
            if constexpr (O4) done = done || step <= 1.0e-4f;             // (fourth-order first pass: a lane that started on the root)
            else if (it == 0) {                                           // third-order first pass: nothing to extrapolate from yet
                done = done || step <= 1.0e-4f;
            } else {
                // C: measured by a fourth-order first pass; behind a third-order one, from the last two steps (d / d_prev^3, floored at 20)
                const float c3 = first4 ? cerr : fmaxf(step * hw_rcp(prev * prev * prev), 20.0f);
                done = done || step <= 1.0e-4f || (step <= 5.0e-3f && c3 * step * step * step <= 2.4e-7f);
            }
        } else if constexpr (O4) {
            const float p2 = prev * prev, s2 = step * step;
            const float c4 = fmaxf(step * hw_rcp(p2 * p2), 100.0f);
            done = done || step <= 1.0e-4f || (step <= 1.0e-2f && c4 * s2 * s2 <= 2.4e-7f);
        } else {
            const float c3 = fmaxf(step * hw_rcp(prev * prev * prev), 20.0f);
            done = done || step <= 1.0e-4f || (step <= 5.0e-3f && c3 * step * step * step <= 2.4e-7f);
        }
        prev = done ? prev : step;
        th = nt;
#ifdef RNF_KO_FIXED_PASSES          // timing-only diagnostic: every wave runs exactly this many passes
        return it + 1 == RNF_KO_FIXED_PASSES;
#else
        return __all(done);                                               // wave-uniform exit: typically 2 passes
#endif
    };
    // (first4: wave uniform, one value per flow -- a rotation's result never depends on the launch it travels in)
    if (!(first4 ? pass(std::true_type{}, 0) : pass(std::false_type{}, 0))) {
#pragma unroll 1
        for (int it = 1; it < 16; ++it)
            if (pass(std::false_type{}, it)) break;
    }
    const float mid = fmaf(cell_of(th) + 0.5f, cell, 0.5f * kPi);      // (a confirmed lane's th is this centre already, bit for bit)
    float sn, cs;
    sincos_pi_band(mid, sn, cs);
    float J = Jc;
    if (!RNF_RF_CENTRE || __any(!conf)) {
        f2 J2 = {0.f, 0.f};
        const f2 sn2 = {sn, sn}, cs2 = {cs, cs};
#pragma unroll
        for (int s = 0; s < 4 * KT; s += 2) {
            const f2 ur = {sg.ur[s], sg.ur[s + 1]}, uv = {sg.uv[s], sg.uv[s + 1]}, q = {sg.q[s], sg.q[s + 1]};
            f2 den;
            if constexpr (RNF_RF_E1FOLD_CLOSE) {
                // |1 - u conj(z)|^2 = b^2 + (1 - a)^2 = 2 (1 - a) + |u|^2 - 1: b is not needed here (6 instead of 8 instructions)
                const f2 e1 = __builtin_elementwise_fma(-uv, sn2, __builtin_elementwise_fma(-ur, cs2, f2{1.0f, 1.0f}));
                const f2 um = __builtin_elementwise_fma(ur, ur, __builtin_elementwise_fma(uv, uv, f2{-1.0f, -1.0f}));
                den = __builtin_elementwise_fma(f2{2.0f, 2.0f}, e1, um);
            } else {
                const f2 a = __builtin_elementwise_fma(uv, sn2, ur * cs2);
                const f2 b = __builtin_elementwise_fma(uv, cs2, -(ur * sn2));
                const f2 e1 = 1.0f - a;
                den = __builtin_elementwise_fma(b, b, e1 * e1);
            }
            J2 = __builtin_elementwise_fma(q, f2{hw_rcp(den.x), hw_rcp(den.y)}, J2);
        }
        float Jl = J2.x + J2.y;
        if constexpr (KT == 16)
            for (int s = 0; s < n_over; ++s) {
                const float4 p = stash[(size_t)s * 64 + lane];
                const float a = fmaf(p.z, sn, p.y * cs), b = fmaf(p.z, cs, -(p.y * sn));
                const float e1 = 1.0f - a;
                Jl = fmaf(p.w, hw_rcp(fmaf(b, b, e1 * e1)), Jl);
            }
        J = conf ? Jc : Jl;
    }
    J = pair_sum(J);
    const v3f xx = c.f.v * sn + c.f.r * cs;
    const v3f zz = normalize3(c.cyc ? cross3(xx, c.y) : cross3(c.y, xx));               // mobiusflow.py:172-176
    set_col(R, c.p0, xx);
    set_col(R, c.p2, zz);
    ldj = fmaf(-0.693147180559945309f, hw_log2(J * invS), ldj);                          // mobiusflow.py:183
}

// Condition16Trans (flow/squeezetrans.py:41-55): M = I + reshape(MLP(feature), 4, 4); the one fc_last tile leaves
// rows {h, 2+h} of M on lane-half h; the partner's two rows come over with 8 cross-lane moves.
template <bool INVERSE>
__device__ __forceinline__ void cond16_finish(const f32x16 &o, int h, Rot &R, float &ldj) {
    float M[16];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float mine = o[4 * g + c];
            float other = __shfl_xor(mine, 32, 64);
            M[4 * (2 * g) + c] = h ? other : mine;    // rows h and 2+h are this lane's
            M[4 * (2 * g + 1) + c] = h ? mine : other;
        }
    M[0] += 1.f; M[5] += 1.f; M[10] += 1.f; M[15] += 1.f;
    float Mi[16];
    float det = inv4(M, Mi);
    if (INVERSE) affine16_apply(Mi, -logf(fabsf(det)), R, ldj);
    else affine16_apply(M, logf(fabsf(det)), R, ldj);
}

// Conditional 3x3 layers (extended instantiation only): M = I + reshape(outputs 0..8, 3, 3); output i sits where output i of
// Condition16Trans sits (rows 0..3: lane-half 0 registers 0..3, rows 4..7: lane-half 1 registers 0..3, row 8: lane-half 0 register 4).
template <bool INVERSE>
__device__ __forceinline__ void cond9_finish(int kind, const f32x16 &o, int h, Rot &R, float &ldj) {
    float m[9];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float mine = o[c], other = __shfl_xor(mine, 32, 64);
        m[c] = h ? other : mine;
        m[4 + c] = h ? mine : other;
    }
    {
        const float mine = o[4], other = __shfl_xor(mine, 32, 64);
        m[8] = h ? other : mine;
    }
    m[0] += 1.f; m[4] += 1.f; m[8] += 1.f;
    if (kind == RNF_KIND_COND9_GS) {                       // Condition9Trans (squeezetrans.py:239-247)
        if (INVERSE) {
            float mi[9];
            inv3(m, mi);
            gs9_apply(mi, R, ldj);
        } else {
            gs9_apply(m, R, ldj);
        }
    } else if (kind == RNF_KIND_COND9_SMITH) {             // Condition9RotRSmith (rottrans.py:173-181): R GS(M), inverse R GS(M)^T
        v3f q0, q1, q2;
        smith3(m, q0, q1, q2);
        if (INVERSE) right_mul_cols(v3f{q0.x, q1.x, q2.x}, v3f{q0.y, q1.y, q2.y}, v3f{q0.z, q1.z, q2.z}, R);
        else right_mul_cols(q0, q1, q2, R);
    } else {                                               // Condition9RotL / 9RotR (rottrans.py:113-121, 143-151): polar(M) R, R polar(M)
        v3f p0, p1, p2;                                    // rows of P = polar(M); the inverse uses M^T, whose polar factor is P^T
        polar3(m, p0, p1, p2);
        const v3f t0 = v3f{p0.x, p1.x, p2.x}, t1 = v3f{p0.y, p1.y, p2.y}, t2 = v3f{p0.z, p1.z, p2.z};     // rows of P^T = columns of P
        if (kind == RNF_KIND_COND9_POLAR_L) {
            if (INVERSE) left_mul_rows(t0, t1, t2, R); else left_mul_rows(p0, p1, p2, R);
        } else {
            if (INVERSE) right_mul_cols(p0, p1, p2, R); else right_mul_cols(t0, t1, t2, R);   // columns of P^T are the rows of P
        }
    }
}

// Condition36Trans (extended instantiation only): M = I + reshape(outputs 0..35, 6, 6) from two fc_last tiles.  Packed row P of tile 0 is
// output P (register 4g + c of lane-half hh holds packed row 8g + 4hh + c), rows 0..3 of tile 1 are outputs 32..35 (lane-half 0).
template <bool INVERSE>
__device__ __forceinline__ void cond36_finish(const f32x16 &o0, const f32x16 &o1, int h, Rot &R, float &ldj) {
    float m[36];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float mine = o0[4 * g + c], other = __shfl_xor(mine, 32, 64);
            m[8 * g + c] = h ? other : mine;              // packed rows 8g + c belong to lane-half 0
            m[8 * g + 4 + c] = h ? mine : other;          // packed rows 8g + 4 + c to lane-half 1
        }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float mine = o1[c], other = __shfl_xor(mine, 32, 64);
        m[32 + c] = h ? other : mine;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) m[7 * i] += 1.0f;
    if (INVERSE) {
        float mi[36];
        inv6(m, mi);
        gs36_apply(mi, R, ldj);
    } else {
        gs36_apply(m, R, ldj);
    }
}

// ------------------------------------------------------------------------------------------------------------
// weight staging
//   SYNC: before each MLP layer all waves copy the layer's record global(L2) -> LDS between two barriers (any K).
//   DMA : K <= 64.  LDS-DMA (global_load_lds_dwordx4, no VGPR round trip) prefetch in two halves that ping-pong with
//         the two compute phases of a layer: while fc_last + segment math of layer l read the L part, the H part of
//         the next MLP layer streams in; while the hidden layers of layer l+1 read H, its L part streams in.
// ------------------------------------------------------------------------------------------------------------
// One 1 KiB LDS-DMA piece (16 bytes per lane).  hipcc counts a piece issued through the builtin as a pending store to LDS and puts
// `s_waitcnt vmcnt(0)` in front of the first ds_read behind it (visible in the .s in front of the affine layer's table read and of the first
// fc_last operand read).  Round 4 measured the same pieces issued from inline assembly, hidden from that bookkeeping (the protocol orders the
// data itself: dma_wait_all() + the layer barrier): no difference on any config (C2 4.248 / 4.243 ms, profiles/README.md) -- the images have
// landed by then -- so the compiler-tracked form stays.
__device__ __forceinline__ void dma_piece(const float *g_lane /* this lane's 16 source bytes */, float *lds_piece /* wave uniform */) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g_lane, (__attribute__((address_space(3))) void *)lds_piece, 16, 0, 0);
}
__device__ __forceinline__ void dma_floats(float *lds_dst, const float *g_src, int nfloats, int wave, int lane, int nwaves) {
    const int n4 = nfloats >> 2;
    for (int base = wave * 64; base < n4; base += nwaves * 64) {          // `base` is wave uniform: 1 KiB per instruction
        const int idx = base + lane;
        if (idx < n4) dma_piece(g_src + 4 * (size_t)idx, lds_dst + 4 * base);
    }
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ------------------------------------------------------------------------------------------------------------
// the fused stack kernel.  DIR 0 = Flow.forward (flow/flow.py:53-72), 1 = Flow.inverse (flow/flow.py:74-92).
// KT_INV: compile-time tile count for the inverse (0 for forward instantiations).  PIPE: staging mode DMA.
// ------------------------------------------------------------------------------------------------------------
// In-kernel phase stamps (diagnostic build only, -DRNF_STAMPS; cdna_hip_programming.md section 7 "In-kernel stamps").
// The shipped library is built without them: no stamp executes in the product kernel.
// timing-only diagnostic (-DRNF_KO_BARRIER): the two layer barriers of the DMA pipeline removed -- results are garbage, the guard is muted
#ifdef RNF_KO_BARRIER
#define RNF_LAYER_BARRIER() asm volatile("" ::: "memory")
#else
#define RNF_LAYER_BARRIER() __syncthreads()
#endif
#if defined(RNF_STAMPS) && defined(RNF_STAMPS_FINE)
#define RNF_STAMP_FINE(i) RNF_STAMP(i)
#else
#define RNF_STAMP_FINE(i)
#endif
#ifdef RNF_STAMPS
#define RNF_STAMP_DECL unsigned long long st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long st_t = clock64();
#define RNF_STAMP(i) { __builtin_amdgcn_sched_barrier(0); unsigned long long now_ = clock64(); st_acc[i] += now_ - st_t; st_t = now_; __builtin_amdgcn_sched_barrier(0); }
#define RNF_STAMP_FLUSH if (args.stamps && lane == 0) { for (int i_ = 0; i_ < 12; ++i_) atomicAdd(args.stamps + i_, st_acc[i_]); }
#else
#define RNF_STAMP_DECL
#define RNF_STAMP(i)
#define RNF_STAMP_FLUSH
#endif

// ------------------------------------------------------------------------------------------------------------
// FUSED instantiation: the feature projection of a conditional flow INSIDE the stack kernel (round 3; BASELINE configs[3]).
// The pre-pass of featproj_kernel.h writes G_l = W_l[:, 3:] f + b_l for every conditional layer to HBM and the stack kernel reads it back
// (12.8 KB per rotation for C4, 18x the algorithmic traffic of the whole evaluation).  Here every wave keeps the features of its 32 rotations
// in registers for the whole layer stack, as the fp16 hi / (2^12-scaled) lo B fragments of the projection (F <= 256: 128 VGPRs of the 256 an
// 8-wave workgroup has per lane -- MFMA B operands only, so they may sit in AGPRs), and computes G of the NEXT MLP layer while the current
// one runs: 2 x 16 k-steps x 3 matrix instructions per layer, weights streamed by LDS-DMA through ONE 32 KB buffer in two halves
// (out tile 0 behind barrier B1, out tile 1 behind B2; each half has a whole compute phase to land).  The 64 x 32 result goes to a
// per-wave 8 KB stash that is rewritten every layer and therefore lives in L2; the head of the next layer loads it exactly like the
// pre-pass scratch (GFrag).  HBM traffic: the features once (4 F bytes per rotation) instead of 2 x 256 bytes per rotation and layer.
// ------------------------------------------------------------------------------------------------------------
constexpr int FUSED_MAX_F = 256;
constexpr int FUSED_KSTEPS = FUSED_MAX_F / 16;
constexpr int FUSED_PA_W = FUSED_KSTEPS * 512;          // floats: one out tile of projection weights, [k-step][hi, lo][lane] 8 x fp16
constexpr int FUSED_PA_FLOATS = FUSED_PA_W + 32;        // + that out tile's bias image [2][16]

struct FeatFrag {
    h8 hi[FUSED_KSTEPS], lo[FUSED_KSTEPS];              // B fragments: k-step s covers features 16 s .. 16 s + 15, lane-half h supplies 16 s + 8 h + 0..7
};

__device__ __forceinline__ void fused_load_features(const float *feat, int F, long long sample, bool valid, int h, FeatFrag &f) {
    // four k-steps (8 x 16 bytes per lane) in flight at a time, fenced: unfenced, the scheduler issues all 32 loads first and the raw fp32
    // values (128 registers) sit on top of the 128 fragment registers they are converted into
    const float *row = feat + (valid ? sample : 0) * F + 8 * h;
#pragma unroll
    for (int s0 = 0; s0 < FUSED_KSTEPS; s0 += 4) {
        float4 p[8];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k0 = 16 * (s0 + u) + 8 * h;
            p[2 * u] = p[2 * u + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (valid && k0 < F) {                        // F % 8 == 0: an 8-group is entirely inside or outside
                p[2 * u] = *reinterpret_cast<const float4 *>(row + 16 * (s0 + u));
                p[2 * u + 1] = *reinterpret_cast<const float4 *>(row + 16 * (s0 + u) + 4);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const f2 v[4] = {f2{p[2 * u].x, p[2 * u].y}, f2{p[2 * u].z, p[2 * u].w}, f2{p[2 * u + 1].x, p[2 * u + 1].y}, f2{p[2 * u + 1].z, p[2 * u + 1].w}};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const h2 ph = __builtin_convertvector(v[q], h2);
                const h2 pl = __builtin_convertvector((v[q] - __builtin_convertvector(ph, f2)) * FEAT_LO_SCALE, h2);
                f.hi[s0 + u][2 * q] = ph[0]; f.hi[s0 + u][2 * q + 1] = ph[1];
                f.lo[s0 + u][2 * q] = pl[0]; f.lo[s0 + u][2 * q + 1] = pl[1];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// one out tile (32 of the 64 projected features) of one layer for this wave's 32 rotations: pa = the staged weights + bias of that tile
__device__ __forceinline__ void fused_project_half(const float *pa, int ns, int lane, int h, const FeatFrag &f, float *stash_tile, bool &bad) {
    f32x16 acc1 = load_bias16(pa + FUSED_PA_W + h * 16);
    f32x16 acc2 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // one k-step of operand look-ahead, fenced: left alone the scheduler hoists all 32 operand reads (128 registers) to the top of the
    // unrolled loop, on top of the 128 feature registers
    h8 ah = lds_h8(pa, 0 * 64 + lane), al = lds_h8(pa, 1 * 64 + lane);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < FUSED_KSTEPS; ++s) {
        if (s < ns) {                                     // wave uniform
            acc1 = RNF_MFMA_H(ah, f.hi[s], acc1);
            acc2 = RNF_MFMA_H(ah, f.lo[s], acc2);
            acc2 = RNF_MFMA_H(al, f.hi[s], acc2);
            if (s + 1 < ns) {
                ah = lds_h8(pa, ((s + 1) * 2 + 0) * 64 + lane);
                al = lds_h8(pa, ((s + 1) * 2 + 1) * 64 + lane);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    float4 *dst = reinterpret_cast<float4 *>(stash_tile) + lane;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        dst[q * 64] = make_float4(fmaf(acc2[4 * q], 1.0f / FEAT_LO_SCALE, acc1[4 * q]), fmaf(acc2[4 * q + 1], 1.0f / FEAT_LO_SCALE, acc1[4 * q + 1]),
                                  fmaf(acc2[4 * q + 2], 1.0f / FEAT_LO_SCALE, acc1[4 * q + 2]), fmaf(acc2[4 * q + 3], 1.0f / FEAT_LO_SCALE, acc1[4 * q + 3]));
    bad |= acc1[0] != acc1[0];                            // a feature beyond the fp16 range: (inf, -inf) pair, every product NaN
}

// RING staging of the bf16x3 kernels (PREC = 2): see flow_stack_kernel.  begin_unit() -> LDS base of the next unit of the layer image.
template <class F>
struct RingCtl {
    F &f;
    __device__ __forceinline__ const float *begin_unit() { return f(); }
};

// LEAN = 1: the stack holds Moebius and constant 4x4 affine layers only, nothing conditional, no saved states (BASELINE configs C1 / C2 / C3):
// every other layer kind, the feature-projection reads and the kind dispatch are compiled out.
// LEAN = 2 (round 3): the CONDITIONAL counterpart (BASELINE configs[3]): Moebius, constant 4x4 affine and Condition16Trans layers, EVERY
// MLP layer conditional (its projected features come from the scratch -- or, FUSED, from the wave's stash), no saved states, no governor.
// Both run guarded only (one-piece softplus).
// ROWS (round 5): shared feature rows (GFragRows) on a non-extended instantiation; the host launches it only with g_div >= 32.
template <int DIR, int KT_INV, int NW, bool PIPE, int PREC, bool EXT = false, int LEAN = 0, bool FUSED = false, bool ROWS = false>
__global__ __launch_bounds__(NW * 64) void flow_stack_kernel(const FlowArgs args) {
    static_assert(!FUSED || (DIR == 0 && PIPE && PREC == 1 && !EXT && LEAN == 2), "FUSED: forward, DMA staging, split precision, conditional lean stack");
    static_assert(!ROWS || (!EXT && !FUSED && LEAN != 1), "ROWS: the extended instantiation reads shared rows itself; lean-1 stacks have no features");
    static_assert(PREC != 2 || (!PIPE && LEAN == 0 && !FUSED && !ROWS), "bf16x3: synchronous staging (its layer image exceeds the LDS), general family");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (args.guard_mode == 2) {                                      // fp32 re-run of a split-precision call: only when its guard fired
        if (__hip_atomic_load(args.guard, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
        if (blockIdx.x == 0 && threadIdx.x == 0) args.guard[1] = 1;
    }
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    constexpr int NT = NW * 64;
    constexpr int TILE = NW * TILE_SAMPLES;
    constexpr int KTI = KT_INV > 0 ? KT_INV : 1;
#ifdef RNF_NO_KEEPX0
    constexpr bool KEEP_X0 = false;
#else
    constexpr bool KEEP_X0 = PREC != 0 && NW <= 8 && LEAN != 1;
#endif         // room for x0 beside the hidden layers (Mlp<1>::head), and something to re-read
#ifndef RNF_DEEP_H
#define RNF_DEEP_H 1
#endif
#ifndef RNF_DEEP_FWD
#define RNF_DEEP_FWD 0
#endif
    constexpr bool DEEP_H = RNF_DEEP_H && PREC == 1 && ((DIR == 1 && NW <= 8) || (RNF_DEEP_FWD && DIR == 0 && (RNF_DEEP_FWD >= 2 || NW <= 8)));      // Mlp<1>::head: two operand pairs in flight (hidden_slot2)
    // RING (PREC = 2, round 6): a bf16x3 layer image (171 KiB at K = 64) does not fit the LDS beside its successor, so it streams through THREE
    // regions of 48.5 KiB in units -- Ha (fc_first + hidden tiles 0..2), Hb (hidden tiles 3..5), then the fc_last tiles four at a time
    // (Mlp<2>): while unit u is read, unit u + 1 has landed or is landing and unit u + 2 is requested (LDS-DMA) into the region unit u - 1
    // just left; one workgroup barrier per unit (begin_unit: wait for the own pieces, barrier, request).  No layer phase waits for a copy.
    constexpr bool RING = PREC == 2;
    const long long ntiles = (args.n + TILE - 1) / TILE;
    const int KT = args.KT;                                          // DIR = 1: <= KT_INV, the capacity of this instantiation
    const int n_layers = args.n_layers;
    double dsum = 0.0;                                               // wave-uniform running sum of log p (kept in scalar registers)
    #ifdef RNF_V_NOFAIR
    constexpr bool FAIR_ON = NW == 8 && DIR == 0 && PREC == 1 && (LEAN == 0 || FUSED);
#else
    constexpr bool FAIR_ON = true;
#endif      // the launcher sets fair_off >= 0 for these only
    typedef typename std::conditional<FAIR_ON, Fair, NoFair>::type FairT;
    FairT fair;
    if constexpr (FAIR_ON) fair = Fair{lds, wave, args.fair_off, 0};
    RNF_STAMP_DECL
    // (Round 6, measured and NOT kept -- profiles/r6/ab_lag_*.jsonl: the workgroup's upper four waves run ONE PHASE BEHIND the lower four --
    // three barriers per layer, the partners of a SIMD always in different phases (hidden / tiles / root finder), the matrix-phase wave at
    // the higher issue priority; results bit-identical; C5u 11.67 -> 11.89 ms without, 11.72 ms with the priorities, C2's inverse 6.89 ->
    // 6.96: the lockstep kernel already overlaps a wave's hidden phase with its partner's root finder)
    // iteration position -> layer index, and the next position (> pos) whose layer owns an MLP image, or -1
    auto layer_at = [&](int pos) { return DIR ? (n_layers - 1 - pos) : pos; };
    auto next_mlp = [&](int pos) {
        for (int q = pos + 1; q < n_layers; ++q)
            if (kind_has_mlp(args.layers[layer_at(q)].x & 15)) return q;
        return -1;
    };
    auto l_floats = [&](int kind) { return kind_last_tiles(kind, KT) * Lay<PREC>::LAST_TILE_FLOATS; };
    const int first_mlp = next_mlp(-1);
    // DMA staging: the block of the constant-affine layer right behind the MLP layer at position q rides with that layer's fc_last
    // image into LDS buffer `parity` (two buffers: a slow wave may still read the previous block while the next one lands)
    auto stage_table = [&](int q, int parity) {
        if (args.tab_off < 0 || q < 0 || q + 1 >= n_layers) return;
        const int2 da = args.layers[layer_at(q + 1)];
        if ((da.x & 15) != RNF_KIND_AFFINE16 || wave != 0 || lane >= AFF_TABLE_FLOATS / 4) return;
        int ln = lane;
        asm volatile("" : "+v"(ln));                                  // keeps the per-lane address out of the long-lived registers
        const float *t = args.blob + da.y + (DIR ? AFF_TABLE_INV : AFF_TABLE_FWD) + 4 * ln;
        dma_piece(t, lds + args.tab_off + AFF_TABLE_LDS_STRIDE * parity);
    };
    auto has_table = [&](int q) { return args.tab_off >= 0 && q + 1 < n_layers && (args.layers[layer_at(q + 1)].x & 15) == RNF_KIND_AFFINE16; };
    // FUSED: cond slot -> its projection record; DMA of one out tile (weights + bias image) into the LDS buffer; the slot of a position
    const int fused_ns = FUSED ? (args.feat_F + 15) / 16 : 0;         // k-steps of 16 features
    auto fused_dma = [&](int slot, int half) {
        const float *rec = args.blob + args.feat_base + (size_t)slot * args.feat_stride;
        dma_floats(lds + args.pa_off, rec + (size_t)half * fused_ns * 512, fused_ns * 512, wave, lane, NW);
        if (wave == NW - 1 && lane < 8)
            dma_piece(rec + (size_t)2 * fused_ns * 512 + half * 32 + 4 * lane, lds + args.pa_off + FUSED_PA_W);
    };
    auto slot_at = [&](int pos) { return ((args.layers[layer_at(pos)].x >> 8) & 255) - 1; };
    auto next_in_tile = [&](int pos) { return ((args.layers[layer_at(pos)].x >> 16) & 1023) - 1; };   // next MLP position, -1 at the end
    float *const stash = FUSED ? args.stash + ((size_t)blockIdx.x * NW + wave) * G_FLOATS_PER_GROUP : nullptr;
    int seq = 0;                                                      // MLP layers this wave has been through (across tiles)
    int tab_parity = -1;                                              // >= 0: the next affine layer's block sits in LDS buffer tab_parity

    if (FAIR_ON && args.fair_off >= 0 && tid < NW) reinterpret_cast<int *>(lds + args.fair_off)[tid] = 0;
    if (PIPE && first_mlp >= 0) {                                  // prologue: image of the first MLP layer
        const int2 d = args.layers[layer_at(first_mlp)];
        dma_floats(lds, args.blob + d.y, Lay<PREC>::HEAD_FLOATS, wave, lane, NW);
        dma_floats(lds + Lay<PREC>::LAST, args.blob + d.y + Lay<PREC>::LAST, l_floats(d.x & 15), wave, lane, NW);
        stage_table(first_mlp, 0);
        if constexpr (FUSED) fused_dma(((d.x >> 8) & 255) - 1, 0);      // out tile 0 of the first layer's projection weights
        dma_wait_all();
        __syncthreads();
    }

#if defined(RNF_KO_BARRIER) && defined(RNF_KO_STAGGER)
    // timing-only diagnostic: the upper half of the workgroup's waves starts RNF_KO_STAGGER x 4096 cycles late (with the barriers knocked
    // out the offset persists): what would two wave groups half a layer apart -- one in its matrix-bound hidden phase while the other
    // is in its VALU-bound segment phase -- be worth?
    if (wave >= NW / 2) for (int i_ = 0; i_ < RNF_KO_STAGGER; ++i_) __builtin_amdgcn_s_sleep(64);
#endif
    // ---- RING state (dead code for PREC != 2) ----
    int ring_reg = 0;                                                 // region of the next unit to be read
    int d_pos = -1, d_part = 0, d_seq = 0;                            // DMA cursor: MLP layer position, unit of that layer, MLP layers passed
    long long d_tile = 0;
    // Tile order.  Default: workgroup b takes tiles b, b + grid, ...  ROWS (shared feature rows): XCD-AWARE -- workgroups are dealt to the 8
    // XCDs round-robin (`blockIdx.x % 8` labels the workgroups that share an XCD and its L2, MI355X_MICROARCH.md "Workgroup dispatch"), and
    // the per-(layer, row) records of consecutive rotations are the same few cache lines, so XCD-mate x takes the x-th CONTIGUOUS eighth of
    // the tiles: each L2 then holds one eighth of the row records (C4q: 1.6 MB of 13 MB) instead of all of them, every XCD for itself.
    // A pure placement choice: a rotation's result does not depend on the tile order (grids that are no multiple of 8 keep the default).
    // Balanced ranges: XCD-mate x walks tiles [x ntiles / 8, (x + 1) ntiles / 8) -- with ceil(ntiles / 8)-sized eighths the last XCD
    // got the remainder only (ntiles = 257 on a grid of 256: 26 tiles for its 32 workgroups while the others walked 33).
    const bool xcd_order = ROWS && (gridDim.x & 7) == 0;
    const long long tile_begin = xcd_order ? (ntiles * (long long)(blockIdx.x & 7)) >> 3 : 0;
    const long long tile_end = xcd_order ? (ntiles * (long long)((blockIdx.x & 7) + 1)) >> 3 : ntiles;
    const long long tile_step = xcd_order ? (gridDim.x >> 3) : gridDim.x;
    auto ring_issue = [&](int region) {                               // request the unit under the DMA cursor into `region`, advance the cursor
        if constexpr (RING) {
            if (d_pos < 0) return;
            typedef Mlp<2> M2;
            const int2 dl = args.layers[layer_at(d_pos)];
            const float *src = args.blob + dl.y;
            float *dst = lds + region * M2::REGION_FLOATS;
            const int tiles = kind_last_tiles(dl.x & 15, KT);
            const int nparts = 2 + (tiles + 3) / 4;
            if (d_part == 0) {
                dma_floats(dst, src, M2::HA_BIAS, wave, lane, NW);
                dma_floats(dst + M2::HA_BIAS, src + Lay<2>::HB, MOB_HB_FLOATS, wave, lane, NW);
            } else if (d_part == 1) {
                dma_floats(dst, src + M2::HA_BIAS, 3 * Lay<2>::W_TILE, wave, lane, NW);
                dma_floats(dst + M2::HB_BIAS, src + Lay<2>::HB, MOB_HB_FLOATS, wave, lane, NW);
            } else {
                const int t0 = 4 * (d_part - 2);
                dma_floats(dst, src + Lay<2>::LAST + t0 * Lay<2>::LAST_TILE_FLOATS, min(4, tiles - t0) * Lay<2>::LAST_TILE_FLOATS, wave, lane, NW);
                if (d_part == nparts - 1) stage_table(d_pos, d_seq & 1);      // the block of the constant-affine layer behind this one
            }
            if (++d_part == nparts) {
                d_part = 0;
                ++d_seq;
                int q = ((dl.x >> 16) & 1023) - 1;                    // next MLP layer of this tile
                if (q < 0) {
                    d_tile += tile_step;
                    q = d_tile < tile_end ? first_mlp : -1;
                }
                d_pos = q;
            }
        }
    };
    auto ring_begin = [&]() -> const float * {
        dma_wait_all();
        RNF_LAYER_BARRIER();                                          // every wave is past unit u - 1 and its pieces of unit u have landed
        ring_issue((ring_reg + 2) % 3);
        const float *base = lds + ring_reg * Mlp<2>::REGION_FLOATS;
        ring_reg = (ring_reg + 1) % 3;
        return base;
    };
    RingCtl<decltype(ring_begin)> ring{ring_begin};
    if constexpr (RING) {
        d_tile = xcd_order ? tile_begin + (blockIdx.x >> 3) : blockIdx.x;
        d_pos = d_tile < tile_end ? first_mlp : -1;
        ring_issue(0);
        ring_issue(1);
    }
    for (long long tile = xcd_order ? tile_begin + (blockIdx.x >> 3) : blockIdx.x; tile < tile_end; tile += tile_step) {
        const long long group = tile * NW + wave;                 // 32-sample group index inside this launch
        const long long sample0 = group * TILE_SAMPLES;           // wave uniform
        const bool valid = sample0 + j < args.n;
        // this lane's sample index, re-derived where it is used: a 64-bit index (and the addresses the compiler pre-computes from it)
        // kept live across the layer stack costs register pairs that end up in scratch in the 128-register instantiation
        auto sample_now = [&]() {
            int jj = j;
            asm volatile("" : "+v"(jj));
            return sample0 + jj;
        };
        const bool more_tiles = tile + tile_step < tile_end;
        int g_row0 = 0, g_rem0 = 0;                               // ROWS: feature row of the wave's first rotation, its position inside the row
        if constexpr (ROWS) {                                     // (wave uniform; the host keeps sample_base + n below 2^31 for these launches)
            const unsigned s0 = (unsigned)(args.sample_base + sample0), gd = (unsigned)args.g_div;
            g_row0 = __builtin_amdgcn_readfirstlane((int)(s0 / gd));
            g_rem0 = __builtin_amdgcn_readfirstlane((int)(s0 - (unsigned)g_row0 * gd));
        }

        Rot R;
        R.c0 = v3f{1.f, 0.f, 0.f}; R.c1 = v3f{0.f, 1.f, 0.f}; R.c2 = v3f{0.f, 0.f, 1.f};
        if (valid) {
            const float *src = args.rot_in + (sample0 + j) * 9;      // row-major [3][3]
            R.c0 = v3f{src[0], src[3], src[6]};
            R.c1 = v3f{src[1], src[4], src[7]};
            R.c2 = v3f{src[2], src[5], src[8]};
        }
        float ldj = 0.f;
        bool bad = false;                                         // split-precision kernels: a hidden layer came out NaN (Mlp<1>::head)
        f32x16 gpre[2];                                           // LEAN = 2: the next MLP layer's projected features, fetched at the top of the affine layer in front of it
        bool have_gpre = false;
        FeatFrag ff;                                              // FUSED only (dead otherwise)
        if constexpr (FUSED) {
            // this tile's features, and the projection of the FIRST MLP layer (the layers behind it are projected one layer ahead, below):
            // out tile 0 is in the LDS buffer (prologue / last layer of the previous tile), out tile 1 is fetched here -- the one DMA
            // latency per tile that nothing hides
            fused_load_features(args.feat, args.feat_F, sample0 + j, valid, h, ff);
            const int s0 = slot_at(first_mlp);
            dma_wait_all();
            __syncthreads();
            fused_project_half(lds + args.pa_off, fused_ns, lane, h, ff, stash, bad);
            __syncthreads();
            fused_dma(s0, 1);
            dma_wait_all();
            __syncthreads();
            fused_project_half(lds + args.pa_off, fused_ns, lane, h, ff, stash + G_FLOATS_PER_GROUP / 2, bad);
            __syncthreads();
            int q2 = next_in_tile(first_mlp);
            if (q2 < 0 && more_tiles) q2 = first_mlp;
            if (q2 >= 0) fused_dma(slot_at(q2), 0);
            RNF_STAMP(10)                                         // 10: FUSED tile start (features + first layer's projection)
        }
        RNF_STAMP(7)                                              // 7: tile prologue / epilogue

        for (int pos = 0; pos < n_layers; ++pos) {
            const int2 d = args.layers[layer_at(pos)];
            const int kind = d.x & 15, perm_row = (d.x >> 4) & 15, slot = ((d.x >> 8) & 255) - 1;
            const float *params = args.blob + d.y;
            if (!LEAN && args.states && valid && h == 0) {       // saved for train_kernels.h (the backward sweep recomputes from here), both directions
                float *dst = args.states + ((size_t)pos * args.states_n + args.sample_base + sample_now()) * 9;
                dst[0] = R.c0.x; dst[1] = R.c1.x; dst[2] = R.c2.x;
                dst[3] = R.c0.y; dst[4] = R.c1.y; dst[5] = R.c2.y;
                dst[6] = R.c0.z; dst[7] = R.c1.z; dst[8] = R.c2.z;
            }

            if (kind == RNF_KIND_AFFINE16) {
#ifdef RNF_GPRE      // measured and NOT shipped (profiles/r3/README.md): 32 (or 16) prefetch registers do not fit beside the table math / layer finish of the
                     // 128-register instantiation (51 / 31 spills): 9.10 ms against 7.99 ms on C4 with the 16-register variant issued behind B2
                if constexpr (LEAN == 2 && !FUSED && DIR == 0) {      // the scratch read of the NEXT MLP layer flies under this layer's table math
                    const int q1 = ((d.x >> 16) & 1023) - 1;
                    if (q1 >= 0) {
                        const float *gp = args.G + ((size_t)slot_at(q1) * args.g_groups + group) * G_FLOATS_PER_GROUP;
                        gpre[0] = load_g16(gp, lane);
                        gpre[1] = load_g16(gp + 4 * 64 * 4, lane);
                        have_gpre = true;
                    }
                }
#endif
                if ((PIPE || RING) && tab_parity >= 0) {          // block staged in LDS together with the previous layer's fc_last image
                    int ht = h;                                   // (re-derived here: hoisted out of the layer loop, the lane's table address is a
                    if constexpr (LEAN != 1) asm volatile("" : "+v"(ht));      // register the 128-register conditional instantiations spill)
                    affine16_table_apply_pair(lds + args.tab_off + AFF_TABLE_LDS_STRIDE * tab_parity, ht, R, ldj);
                    tab_parity = -1;
                } else {                                          // scalar loads (one ~2 us round trip per layer: 104 floats do not fit the SGPRs at once)
                    affine16_table_apply(params + (DIR ? AFF_TABLE_INV : AFF_TABLE_FWD), R, ldj);
                }
                RNF_STAMP(6)                                      // 6: unconditional affine layer
                continue;
            }
            if (EXT && kind_is_side(kind)) {                      // the caller built this layer's matrix per sample (param offset = side slot)
                const float *m = args.side + ((size_t)d.y * args.side_n + args.sample_base + (valid ? sample_now() : 0)) * 16;
                if (kind == RNF_KIND_SIDE9) {                     // calculate_9 with a per-sample M (flow/squeezetrans.py:264-277)
                    float M9[9];
#pragma unroll
                    for (int i = 0; i < 9; ++i) M9[i] = m[i];
                    if (DIR) { float Mi[9]; inv3(M9, Mi); gs9_apply(Mi, R, ldj); }
                    else gs9_apply(M9, R, ldj);
                } else {
                    float M[16], Mi[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) M[i] = m[i];
                    if (kind == RNF_KIND_SIDE16_ROT) {            // ConditionRot (flow/rottrans.py:37-66): orthogonal, log-det 0, inverse = transpose
                        if (DIR) {
#pragma unroll
                            for (int i = 0; i < 4; ++i)
#pragma unroll
                                for (int jj = 0; jj < 4; ++jj) Mi[4 * i + jj] = M[4 * jj + i];
                            affine16_apply(Mi, 0.f, R, ldj, true);
                        } else {
                            affine16_apply(M, 0.f, R, ldj, true);
                        }
                    } else {                                      // Condition16TransLU (flow/squeezetrans.py:134-144)
                        const float det = inv4(M, Mi);
                        if (DIR) affine16_apply(Mi, -logf(fabsf(det)), R, ldj);
                        else affine16_apply(M, logf(fabsf(det)), R, ldj);
                    }
                }
                continue;
            }
            if (!LEAN && kind == RNF_KIND_GS9) {                  // Uncondition9Trans: the inverse pass uses M^-1 (squeezetrans.py:259-261)
                gs9_apply(params + (DIR ? 9 : 0), R, ldj);
                continue;
            }
            if (!LEAN && kind == RNF_KIND_GS36) {                 // Uncondition36Trans (squeezetrans.py:355-361)
                gs36_apply(params + (DIR ? 36 : 0), R, ldj);
                continue;
            }

            // ---- layers with a conditioner MLP ----
            typedef typename std::conditional<LEAN == 1, NoG, typename std::conditional<ROWS, GFragRows, GFrag<EXT>>::type>::type GF;
            GF gfrag{};
            if constexpr (ROWS) {
                gfrag.p = slot >= 0 ? args.G + (size_t)slot * (size_t)args.g_rows * 64 : nullptr;
                gfrag.row0 = g_row0; gfrag.rem0 = g_rem0; gfrag.gdiv = (int)args.g_div; gfrag.last = (int)args.g_rows - 1;
            } else if constexpr (LEAN != 1) {
                gfrag.p = nullptr;
                gfrag.rows = EXT && args.g_div > 0;
                if (FUSED) {
                    gfrag.p = stash;
                } else if (slot >= 0) {
                    if (EXT && args.g_div > 0) {
                        long long row = (args.sample_base + (valid ? sample_now() : 0)) / args.g_div;
                        if (row >= args.g_rows) row = args.g_rows - 1;
                        gfrag.p = args.G + ((size_t)slot * args.g_rows + row) * 64;
                    } else {
                        gfrag.p = args.G + ((size_t)slot * args.g_groups + group) * G_FLOATS_PER_GROUP;
                    }
                }
            }
            // where the NEXT image comes from (DMA mode): next MLP layer of this tile, else the first one of the next tile
            int nxt_off = -1, nxt_kind = 0, nxt_q = -1;
            if (PIPE) {
                int q = ((d.x >> 16) & 1023) - 1;                  // next layer with an MLP image (host-filled, rnf_api.hip run_flow)
                if (q < 0 && more_tiles) q = first_mlp;
                nxt_q = q;
                if (q >= 0) { const int2 dn = args.layers[layer_at(q)]; nxt_off = dn.y; nxt_kind = dn.x & 15; }
            } else if constexpr (!RING) {
                __syncthreads();                                   // everyone is done with the previous image
                const int tiles_now = min(kind_last_tiles(kind, KT), Lay<PREC>::MAX_TILES_IN_LDS);
                stage_floats(lds, params, Lay<PREC>::HEAD_FLOATS + tiles_now * Lay<PREC>::LAST_TILE_FLOATS, tid, NT);
                __syncthreads();
            }
            RNF_STAMP(0)                                          // 0: G load + synchronous staging (SYNC mode)

            MobiusCtx ctx;
            typename Mlp<PREC>::Act tt;
            int hh = h;                                           // (see `ht` above: the bias offsets of the hidden layers are re-derived per layer)
            if constexpr (LEAN == 2 && NW == 16) asm volatile("" : "+v"(hh));
            if constexpr (RING) {
                if (kind == RNF_KIND_MOBIUS) {
                    mobius_begin<DIR, DIR == 0>(R, perm_row, ctx);
                    Mlp<2>::template head_ring<GF, KEEP_X0>(ring, lane, hh, ctx.y.x, ctx.y.y, ctx.y.z, gfrag, tt, bad);
                } else {
                    Mlp<2>::template head_ring<GF, KEEP_X0>(ring, lane, hh, 0.f, 0.f, 0.f, gfrag, tt, bad);
                }
            } else if (LEAN == 1 || kind == RNF_KIND_MOBIUS) {
                mobius_begin<DIR, DIR == 0 && PREC != 0>(R, perm_row, ctx);
                Mlp<PREC>::template head<GF, KEEP_X0, FairT, DEEP_H>(lds, lane, hh, ctx.y.x, ctx.y.y, ctx.y.z, gfrag, tt, fair, bad, have_gpre ? gpre : nullptr);
            } else {
                Mlp<PREC>::template head<GF, KEEP_X0, FairT, DEEP_H>(lds, lane, hh, 0.f, 0.f, 0.f, gfrag, tt, fair, bad, have_gpre ? gpre : nullptr);
            }
            have_gpre = false;
            RNF_STAMP(1)                                          // 1: frame + hidden layers (H part)
            if (PIPE) {       // B1: every wave is past the H part and this layer's L part has landed
                dma_wait_all();
                RNF_STAMP_FINE(8)                                 // (-DRNF_STAMPS_FINE) 8: B1's wait for this wave's own DMA pieces
                RNF_LAYER_BARRIER();
                if (nxt_off >= 0) dma_floats(lds, args.blob + nxt_off, Lay<PREC>::HEAD_FLOATS, wave, lane, NW);
            }
            RNF_STAMP(2)                                          // 2: barrier B1 (+ DMA issue)
            const int fq1 = FUSED ? next_in_tile(pos) : -1;       // FUSED: the next MLP layer of this tile, whose G is produced during this layer
            if constexpr (FUSED) {
                if (fq1 >= 0) {                                   // out tile 0 (its weights landed during this layer's hidden phase; B1 waited for them)
                    fused_project_half(lds + args.pa_off, fused_ns, lane, h, ff, stash, bad);
                    RNF_STAMP(8)                                  // 8: FUSED projection, out tile 0
                    __syncthreads();                              // every wave is done with the buffer
                    fused_dma(slot_at(fq1), 1);                   // out tile 1 lands during the fc_last phase
                }
                RNF_STAMP(11)                                     // 11: FUSED: the two extra barriers (+ DMA issue)
            }

            // B2 (DMA mode): every wave is past the L part and the next layer's H part has landed; then the next L part is requested
            auto b2_sync = [&]() {
                RNF_STAMP(3)                                      // 3: fc_last tiles + segment math (L part)
                if (PIPE) {
                    dma_wait_all();
                    RNF_STAMP_FINE(9)                             // 9: B2's wait for this wave's own DMA pieces
                    RNF_LAYER_BARRIER();
                    RNF_STAMP_FINE(10)                            // 10: B2's barrier itself (4 is then what follows it: root finder + DMA issue)
                }
            };
            auto b2_issue = [&]() {
                if constexpr (RING) {                             // (the block itself was requested with the layer's last fc_last unit, ring_issue)
                    tab_parity = has_table(pos) ? (seq & 1) : -1;
                    ++seq;
                }
                if (PIPE) {
                    if (nxt_off >= 0) dma_floats(lds + Lay<PREC>::LAST, args.blob + nxt_off + Lay<PREC>::LAST, l_floats(nxt_kind), wave, lane, NW);
                    tab_parity = has_table(pos) ? (seq & 1) : -1;
                    ++seq;
                    stage_table(nxt_q, seq & 1);
                }
                RNF_STAMP(4)                                      // 4: barrier B2 (+ DMA issue)
            };
            auto fused_p1 = [&]() {                               // FUSED, behind B2: out tile 1 of the next layer's G, then out tile 0 of the one after
                if constexpr (FUSED) {
                    if (fq1 >= 0) {
                        fused_project_half(lds + args.pa_off, fused_ns, lane, h, ff, stash + G_FLOATS_PER_GROUP / 2, bad);
                        RNF_STAMP(9)                              // 9: FUSED projection, out tile 1
                        __syncthreads();
                        int q2 = next_in_tile(fq1);
                        if (q2 < 0 && more_tiles) q2 = first_mlp;
                        if (q2 >= 0) fused_dma(slot_at(q2), 0);
                    }
                    RNF_STAMP(11)
                }
            };
            auto barrier2 = [&]() { b2_sync(); b2_issue(); fused_p1(); };
            if (LEAN == 1 || kind == RNF_KIND_MOBIUS) {
                if constexpr (DIR != 0) {
                    InvSegs<KTI> sg;
                    float S = 0.f;
                    float4 *const istash = (KTI == 16 && args.inv_stash) ? args.inv_stash + ((size_t)blockIdx.x * NW + wave) * (size_t)(4 * max(KT - KTI, 0)) * 64 : nullptr;
                    // a guarded split-precision call (the launcher runs the exact-fp32 kernels behind it when the guard fires) may use the
                    // lean softplus: the branch is wave uniform, each side its own copy of the tile loop
                    const bool fastsp = PREC == 1 && args.guard_mode == 1;
#ifndef RNF_INV_TILEPIPE
#define RNF_INV_TILEPIPE 1
#endif
                    bool piped = false;
                    if constexpr (RNF_INV_TILEPIPE && PREC == 1 && PIPE && KTI <= MOB_MAX_TILES_IN_LDS) {
                        if (KT == KTI) {                              // wave uniform: the layer fills the instantiation's tiles (K = 64 on <1,8,...>: every BASELINE config)
                            piped = true;
                            if (fastsp) mobius_inv_tiles_pipe<KTI, true>(lds, args.K, lane, h, tt, ctx, sg, S);
                            else mobius_inv_tiles_pipe<KTI, false>(lds, args.K, lane, h, tt, ctx, sg, S);
                        }
                    }
                    if constexpr (RING) {
                        mobius_inv_tiles<KTI, PREC, false, decltype(ring)>(lds, params, KT, args.K, lane, h, tt, ctx, sg, S, tid, NT, istash, &ring);
                    } else if (!piped) {
                        if (fastsp) mobius_inv_tiles<KTI, PREC, PREC == 1>(lds, params, KT, args.K, lane, h, tt, ctx, sg, S, tid, NT, istash);
                        else mobius_inv_tiles<KTI, PREC, false>(lds, params, KT, args.K, lane, h, tt, ctx, sg, S, tid, NT, istash);
                    }
                    // the barrier right behind the tiles (the root finder does not touch LDS, and its pass count differs from wave to wave:
                    // a barrier behind it was 19 % of the wave time), the DMA request of the next fc_last image behind the root finder:
                    // in front of it the DMA address arithmetic would sit on top of the 96 live segment registers (36 spills at K = 64);
                    // the image still has the whole hidden-layer phase of the next layer to land
                    b2_sync();
                    mobius_inv_finish<KTI>(ctx, sg, S, R, ldj, istash, KTI == 16 ? 4 * max(KT - KTI, 0) : 0, lane,
                                           args.min_wsum, fastsp, bad, args.rf_first4 != 0);
                    b2_issue();
                } else {
                    float S = 0.f, A = 0.f, J = 0.f;
                    if constexpr (PREC == 2) mobius_fwd_tiles_b3(ring, KT, args.K, lane, h, tt, ctx, S, A, J);
                    else if (PIPE || KT <= MOB_MAX_TILES_IN_LDS) mobius_fwd_tiles<PREC, LEAN == 1, LEAN != 0, FairT>(lds, KT, args.K, lane, h, tt, ctx, S, A, J, fair);
                    else mobius_fwd_tiles_restage<PREC>(lds, params, KT, args.K, lane, h, tt, ctx, S, A, J, tid, NT);
                    barrier2();
                    mobius_fwd_finish<PREC != 0, LEAN != 0 && PREC == 1>(ctx, S, A, J, R, ldj, bad, args.min_wsum);
                }
            } else {
                const float *lrec = lds + Lay<PREC>::LAST;
                if constexpr (RING) lrec = ring.begin_unit();
                const f32x16 o16 = Mlp<PREC>::last(lrec, lane, h, tt);
                if (EXT && kind == RNF_KIND_COND36) {
                    const f32x16 o16b = Mlp<PREC>::last(lrec + Lay<PREC>::LAST_TILE_FLOATS, lane, h, tt);
                    barrier2();
                    cond36_finish<DIR != 0>(o16, o16b, h, R, ldj);
                } else {
                    barrier2();
                    if (EXT && kind != RNF_KIND_COND16) cond9_finish<DIR != 0>(kind, o16, h, R, ldj);
#ifdef RNF_EXP_NOCOND16
                    else if (!FUSED) cond16_finish<DIR != 0>(o16, h, R, ldj); else ldj += o16[0];
#else
                    else cond16_finish<DIR != 0>(o16, h, R, ldj);
#endif
                }
            }
            RNF_STAMP(5)                                          // 5: layer finish (bisection for the inverse)
        }

        if (PREC == 1 && args.guard_mode == 1 && valid) {            // see FlowArgs::guard
            const float chk = ldj + ((R.c0.x + R.c0.y + R.c0.z) + (R.c1.x + R.c1.y + R.c1.z) + (R.c2.x + R.c2.y + R.c2.z));
#ifndef RNF_KO_BARRIER
            if (bad || !(fabsf(chk) <= 3.0e38f)) atomicOr(args.guard, 1);
#endif
        }
        // Exact-fp32 kernels (set_precision("fp32"), and the re-run of a guarded split-precision call, which a NaN always triggers): a NaN that
        // entered a conditioner (NaN feature row / rotation) makes the sample's outputs NaN, as in the reference (its ReLU propagates NaN;
        // fmaxf would launder it into finite garbage).  `bad` has no other source in these kernels (mlp_head).
        if (PREC != 1 && bad) {
            ldj = __builtin_nanf("");
            R.c0.x = ldj;
        }
        // epilogue: outputs + fused base density + NLL partial (utils/fisher.py:217-232, agent.py:55-65)
        double lp_d = 0.0;
        if (valid && h == 0) {
            const long long sample = sample_now();
            if (args.rot_out) {
                float *dst = args.rot_out + sample * 9;
                dst[0] = R.c0.x; dst[1] = R.c1.x; dst[2] = R.c2.x;
                dst[3] = R.c0.y; dst[4] = R.c1.y; dst[5] = R.c2.y;
                dst[6] = R.c0.z; dst[7] = R.c1.z; dst[8] = R.c2.z;
            }
            if (args.ldj_out) args.ldj_out[sample] = ldj;
            if (args.logp_out || args.partials) {
                float lp = ldj;
                if (args.fisher_A) {
                    const long long row = (args.sample_base + sample) / args.fisher_div;
                    const float *A = args.fisher_A + row * 9;
                    float tr = R.c0.x * A[0];
                    tr = fmaf(R.c1.x, A[1], tr); tr = fmaf(R.c2.x, A[2], tr);
                    tr = fmaf(R.c0.y, A[3], tr); tr = fmaf(R.c1.y, A[4], tr); tr = fmaf(R.c2.y, A[5], tr);
                    tr = fmaf(R.c0.z, A[6], tr); tr = fmaf(R.c1.z, A[7], tr); tr = fmaf(R.c2.z, A[8], tr);
                    lp += tr - args.fisher_c[row];
                }
                if (args.logp_out) args.logp_out[sample] = lp;
                lp_d = (double)lp;
            }
        }
        if (args.partials) {  // fixed-order wave sum per tile; the running sum is wave uniform (a per-lane double across the whole
                              // layer stack costs a register pair, which the 128-register instantiation would keep in scratch)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) lp_d += __shfl_xor(lp_d, off, 64);
            const double t = dsum + lp_d;
            dsum = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(t)), __builtin_amdgcn_readfirstlane(__double2loint(t)));
        }
    }

    if (args.partials) {      // deterministic block partial: wave shuffle tree -> LDS -> thread 0
        __syncthreads();
        double *red = reinterpret_cast<double *>(lds);
        if (lane == 0) red[wave] = dsum;
        __syncthreads();
        if (tid == 0) {
            double s = 0.0;
            for (int w = 0; w < NW; ++w) s += red[w];
            args.partials[blockIdx.x] = s;
        }
    }
    RNF_STAMP_FLUSH
}

// fixed-order final reduction of the block partials -> out[0] += sum, out[1] += count.  guard != nullptr and guard[0] != 0: the call was
// re-run on the exact-fp32 kernels, whose partials are the ones to add.
__global__ void nll_finalize_kernel(const double *partials, int nparts, double count, double *out, int accumulate,
                                    const double *partials_fb = nullptr, int nparts_fb = 0, const int *guard = nullptr) {
    __shared__ double red[256];
    if (guard && guard[0]) { partials = partials_fb; nparts = nparts_fb; }
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) s += partials[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[0] = (accumulate ? out[0] : 0.0) + red[0];
        out[1] = (accumulate ? out[1] : 0.0) + count;
    }
}

}  // namespace rnf
