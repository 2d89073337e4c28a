// train_kernels.h -- backward pass of Flow.forward (training: agent.py:75-92 differentiates loss = mean(-ldj) through every layer).
//
// ONE launch for the whole reverse sweep.  A workgroup of 4 waves owns 64 rotations.  The forward stack kernel saved the rotation
// entering every layer (FlowArgs::states); walking the layers backwards, the workgroup
//   1. recomputes the layer's conditioner MLP in exact fp32 on the matrix cores (v_mfma_f32_32x32x2_f32), activations kept in LDS
//      as [feature][sample],
//   2. runs the reverse-mode formulas of so3_grad.h with one rotation per lane (every wave carries all 64 rotations; the K segments
//      of a Moebius layer are split over the 4 waves and their partial sums meet in LDS),
//   3. back-propagates through the MLP with three MFMA product shapes -- W.act (rows), W^T.g (columns), g.act^T over the 64
//      samples (weight gradients) -- and adds its weight-gradient tiles to the global gradient blob with coalesced float atomics.
// Weights are read straight from the "plain" parameter blob in L2 (each tile is touched once per 64 rotations).
// Training batches are 128-1024 rotations (settings/*.yml), so this path is latency-, not throughput-critical.
//
// Plain blob layout per layer (floats, torch.nn.Linear order [out][in], offsets in the layer table):
//   Moebius (NI = 3 + F, NO = 4K) / Condition16Trans (NI = F, NO = 16) conditioner MLP (flow/condition.py):
//       W0 [64][NI] | b0 [64] | W1 [64][64] | b1 | W3 | b3 | W5 | b5 | WL [NO][64] | bL [NO]
//   Uncondition16Trans:              M [16]
//   Uncondition9Trans (+LU):         M [9] + 3 floats of padding (every layer starts 16-byte aligned)
// The gradient blob has the same layout.
#pragma once
#include <hip/hip_runtime.h>

#include "flow_kernels.h"
#include "layout.h"
#include "so3_grad.h"

namespace rnf {

constexpr int TR_MAX_LAYERS = 200;
constexpr int TR_WAVES = 4;

struct TrainArgs {
    const float *states;      // [n_layers][n][9] rotation at the input of layer (iteration position) p
    const float *feature;     // [n][F] or nullptr
    const float *plain;       // plain parameter blob
    float *grads;             // gradient blob (same layout), zeroed by the caller; nullptr: input gradients only (pose refinement,
                              // eval.py:464-478), every weight-gradient product is skipped
    const float *g_rot_out;   // [n][9] dL/dR_out or nullptr (zeros)
    const float *g_ldj;       // [n]   dL/dldj
    float *g_rot_in;          // [n][9]
    float *g_feature;         // [n][F] or nullptr
    float *g_ldj_sum;         // [n_layers] sum over the batch of dL/dldj (for the log|det M| term), zeroed by the caller
    long long n;
    int n_layers, K, F;
    // dir = 1: backward of Flow.inverse (flow/flow.py:74-92; MobiusFlow.inverse with BinFind.backward, flow/mobiusflow.py:247-273).  The
    // layer table is then in ITERATION order of the inverse pass (last flow layer first), states[p] is the rotation entering iteration
    // position p and rot_final [n][9] the output of the whole pass: a Moebius layer reads its root theta back from its own output.
    int dir;
    const float *rot_final;
    unsigned long long *stamps;   // diagnostic builds only (RNF_STAMPS): per-phase cycle sums of wave 0, else nullptr
    // side layers (RNF_KIND_SIDE*: Condition16TransLU, Condition9TransLU, ConditionRot): the caller's per-sample matrices
    // side [n_side][n][16] and, out, dL/d(matrix) side_grad [n_side][n][16]; the layer's slot sits in bits 16..23 of x
    const float *side;
    float *side_grad;
    // RNF_KIND_MLP_ONLY: dL/d(outputs) of the one conditioner MLP of the call, [n][NO] row-major, NO in bits 16..23 of x
    const float *g_out_ext;
    // 16-rotation sweep only (train_block16.h): the conditioner activations the training forward saved, one slot of act_rows x 16 floats
    // per (conditioner layer, 16-rotation block), or nullptr = recompute them; mlp_base = conditioner layers below this chunk of the table
    const float *acts;
    int act_rows, mlp_base;
    // per layer: x = kind | perm_row << 4 | orthogonal << 8 | (side slot or NO) << 16, y = plain offset
    int2 layers[TR_MAX_LAYERS];
};

struct __attribute__((packed, aligned(4))) Float4U { float x, y, z, w; };      // 16-byte global load from a 4-byte aligned address

// LDS matrix [rows][64 samples], rows padded to 65 floats: conflict free both for "lanes = 32 consecutive samples of one row"
// and for "lanes = 32 consecutive rows at one sample" (the two MFMA operand patterns), and -- unlike an XOR swizzle -- every
// unrolled access is base register + immediate offset, so the reads of a product issue back to back
constexpr int LROW = 65;
constexpr int TR_LDS_HEAD_FLOATS = 5 * 64 * LROW + 4 * LROW;   // X0, H1, H2, H3, GA, YL
struct LMat {
    float *p;
    __device__ __forceinline__ float &at(int row, int s) const { return p[row * LROW + s]; }
};

// one lane's column of an LMat as a conditioner-output row (so3_grad.h accessor)
struct LaneRow {
    LMat m;
    int lane;
    __device__ __forceinline__ float get(int row) const { return m.at(row, lane); }
    __device__ __forceinline__ void put(int row, float v) const { m.at(row, lane) = v; }
};

// D-tile register r of lane (j, h) is row rho(r, h), column j (v_mfma_f32_32x32x2_f32; rho: layout.h)

__device__ __forceinline__ f32x16 zero16() {
    f32x16 c;
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    return c;
}

// ---- A operands, loaded ahead of the phase that multiplies with them (the loads then overlap the previous phase and its barrier).
// No load is followed by a select: a select right behind each load makes the compiler reuse one destination register and
// serialise the loads.  Out-of-range rows / columns are handled by clamping the ADDRESS (finite values) and by making the other
// operand zero (zeroed pad rows in LDS) or by discarding the output rows / columns they feed.
// 32 rows of a row-major weight matrix with 64 columns (16-byte aligned rows): this lane holds W[row][32h .. 32h + 31] and the
// row's bias, which enters the product as one more K step against a column of ones (1 register instead of a 16-register image).
// Rows >= n_rows repeat the last row: their output rows are never stored.
struct RowsA { float4 w[8]; float bias; };
__device__ __forceinline__ RowsA load_rows64(const float *__restrict__ W, const float *__restrict__ bias, int row, int n_rows, int h) {
    RowsA a;
    const int rc = row < n_rows ? row : n_rows - 1;
    a.bias = bias[rc];
    const float4 *src = reinterpret_cast<const float4 *>(W + (size_t)rc * 64 + 32 * h);
#pragma unroll
    for (int q = 0; q < 8; ++q) a.w[q] = src[q];
    return a;
}
// D[rho][j] = bias[row] + sum_{k<64} W[row][k] * b(k), k = 32h + m.  b(k): B element (k, this lane's sample).  The B reads of half
// a product are issued before its first MFMA (sched_barrier), so that the dependent MFMA chain runs back to back instead of paying
// one LDS latency per step.
// B elements are READ first (raw, 16 in flight) and transformed (ReLU) only after a sched_barrier: an op right behind each LDS read
// makes the compiler reuse one destination register and serialise the reads, exactly as with the global loads.
struct Identity { __device__ __forceinline__ float operator()(float v) const { return v; } };
struct Relu { __device__ __forceinline__ float operator()(float v) const { return __builtin_amdgcn_fmed3f(v, 0.f, __builtin_inff()); } };

template <class BFn, class Post = Identity>
__device__ __forceinline__ f32x16 mfma_rows64(const RowsA &a, int h, BFn b, Post post = Post()) {
    f32x16 acc = RNF_MFMA(a.bias, h ? 0.f : 1.f, zero16());
#pragma unroll
    for (int half = 0; half < 2; ++half) {                 // 16 B reads in flight, then 16 MFMAs
        float bv[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) bv[m] = b(32 * h + 16 * half + m);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 16; ++m) bv[m] = post(bv[m]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc = RNF_MFMA(a.w[4 * half + q].x, bv[4 * q], acc);
            acc = RNF_MFMA(a.w[4 * half + q].y, bv[4 * q + 1], acc);
            acc = RNF_MFMA(a.w[4 * half + q].z, bv[4 * q + 2], acc);
            acc = RNF_MFMA(a.w[4 * half + q].w, bv[4 * q + 3], acc);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    return acc;
}

// 32 columns of a row-major matrix (the transposed product): this lane holds W[k0 + 32h + m][col], m < 32.  Rows >= n_rows repeat
// the last row (their B partners must be zero); columns >= n_cols repeat the last column (their output rows are discarded).
struct ColsA { float a[32]; };
__device__ __forceinline__ ColsA load_cols(const float *__restrict__ W, int ldw, int k0, int n_rows, int col, int n_cols, int h) {
    ColsA c;
    const int kb = k0 + 32 * h;
    const float *p = W + (size_t)(kb < n_rows ? kb : n_rows - 1) * ldw + (col < n_cols ? col : n_cols - 1);   // stepped pointer
#pragma unroll
    for (int m = 0; m < 32; ++m) {
        c.a[m] = *p;
        if (kb + m + 1 < n_rows) p += ldw;
    }
    return c;
}
// acc[rho][j] += sum_{k in [k0, k0 + 64)} W[k][col] * b(k)
template <class BFn>
__device__ __forceinline__ f32x16 mfma_cols(const ColsA &c, int k0, int h, BFn b, f32x16 acc) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        float bv[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) bv[m] = b(k0 + 32 * h + 16 * half + m);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 16; ++m) acc = RNF_MFMA(c.a[16 * half + m], bv[m], acc);
        __builtin_amdgcn_sched_barrier(0);
    }
    return acc;
}

// acc[rho][j] = sum_{s<64} a(s) * b(s): the product over the block's samples (weight gradients), s = 32h + m
template <class AFn, class BFn, class Post = Identity>
__device__ __forceinline__ f32x16 mfma_samples(int h, AFn a, BFn b, Post post = Post()) {
    f32x16 acc = zero16();
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        float av[16], bv[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) { av[m] = a(32 * h + 16 * half + m); bv[m] = b(32 * h + 16 * half + m); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 16; ++m) bv[m] = post(bv[m]);
#pragma unroll
        for (int m = 0; m < 16; ++m) acc = RNF_MFMA(av[m], bv[m], acc);
        __builtin_amdgcn_sched_barrier(0);
    }
    return acc;
}

// gW[o0 + rho][col] += acc[rho][j]
__device__ __forceinline__ void scatter_add(float *gW, int ldw, int o0, int n_out, int col, bool col_ok, int h, const f32x16 &acc) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int o = o0 + rho(r, h);
        if (col_ok && o < n_out) atomicAdd(gW + (size_t)o * ldw + col, acc[r]);
    }
}

__device__ __forceinline__ void store_tile(const LMat &M, int o0, int n_out, int s, int h, const f32x16 &acc) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int o = o0 + rho(r, h);
        if (o < n_out) M.at(o, s) = acc[r];
    }
}

// sum over the 4 lanes of a quad (DPP, no LDS)
__device__ __forceinline__ float quad_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // lanes 1,0,3,2
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // lanes 2,3,0,1
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// gb[o] += sum_s G[o][s] for o < n_out: four threads per row (16 samples each), 64 rows per pass of the workgroup
__device__ __forceinline__ void bias_grad(const LMat &G, int n_out, float *gb, int tid) {
    const int q = tid & 3;
    for (int o = tid >> 2; o < n_out; o += TR_WAVES * 16) {
        float v[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = G.at(o, 16 * q + i);
        __builtin_amdgcn_sched_barrier(0);                 // all 16 reads in flight before the first add
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc += v[i];
        acc = quad_sum(acc);
        if (q == 0) atomicAdd(gb + o, acc);
    }
}

// Workgroup barrier for data exchanged through LDS only: waits for this wave's LDS traffic, NOT for its outstanding global loads
// (__syncthreads() also drains vmcnt, which would serialise the A-operand loads issued one phase ahead with every barrier)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// phase stamps of the diagnostic build (tools/phase_stamps_train.py), same mechanism as flow_kernels.h
#ifdef RNF_STAMPS
#define RNF_TSTAMP(i) { __builtin_amdgcn_sched_barrier(0); unsigned long long now_ = clock64(); tst_acc[i] += now_ - tst_t; tst_t = now_; __builtin_amdgcn_sched_barrier(0); }
#else
#define RNF_TSTAMP(i)
#endif

// Rarely used layer kinds (side layers, 6x6 Gram-Schmidt, conditional 3x3).  Inlined: keeping them out of line (-DRNF_RARE_OOL) to shield the
// common loop from their register needs measured SLOWER (graphed iteration, one box: C2 batch 1024 1.655 ms inlined / 1.691 ms out of
// line / 1.667 ms with these kinds compiled out; C4 batch 128 2.509 / 2.526 / 2.462 ms), so the shipped build inlines them.
#ifdef RNF_RARE_OOL
#define RNF_RARE_FN __device__ __attribute__((noinline))
#else
#define RNF_RARE_FN __device__ __forceinline__
#endif
RNF_RARE_FN void rare_side(int kind, int dir, const float *m, const Rot *Rin, const Rot *gR, float g_ldj, float *gM_out, Rot *gRin_p) {
    Rot gRin_v;
    Rot &gRin_ref = gRin_v;
    float gM[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) gM[i] = 0.f;
    {
        Rot &gRin = gRin_ref;
        const Rot &RinV = *Rin;
        const Rot &gRV = *gR;
                if (kind == RNF_KIND_SIDE9) {             // Condition9TransLU (squeezetrans.py:264-277): calculate_9, inverse pass M^-1
                    float M9[9], g9[9];
#pragma unroll
                    for (int i = 0; i < 9; ++i) { M9[i] = m[i]; g9[i] = 0.f; }
                    if (dir) {
                        float Mi[9], gMi[9];
                        inv3(M9, Mi);
#pragma unroll
                        for (int i = 0; i < 9; ++i) gMi[i] = 0.f;
                        gs9_backward(Mi, RinV, gRV, g_ldj, gMi, gRin);
                        inverse_matrix_grad<3>(Mi, gMi, g9);
                    } else {
                        gs9_backward(M9, RinV, gRV, g_ldj, g9, gRin);
                    }
#pragma unroll
                    for (int i = 0; i < 9; ++i) gM[i] = g9[i];
                } else {
                    float M[16], Mi[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) M[i] = m[i];
                    const bool orth = kind == RNF_KIND_SIDE16_ROT;      // ConditionRot (rottrans.py:37-66): ldj = 0, inverse pass M^T
                    Rot Rout;
                    AffineSaved sv;
                    float l;
                    if (orth) {
                        if (dir) {
                            float Mt[16], gMt[16];
#pragma unroll
                            for (int i = 0; i < 4; ++i)
#pragma unroll
                                for (int jj = 0; jj < 4; ++jj) { Mt[4 * i + jj] = M[4 * jj + i]; gMt[4 * i + jj] = 0.f; }
                            affine16_forward_saved(Mt, 0.f, RinV, Rout, l, sv);
                            affine16_backward(Mt, sv, gRV, g_ldj, true, gMt, gRin);
#pragma unroll
                            for (int i = 0; i < 4; ++i)
#pragma unroll
                                for (int jj = 0; jj < 4; ++jj) gM[4 * i + jj] = gMt[4 * jj + i];
                        } else {
                            affine16_forward_saved(M, 0.f, RinV, Rout, l, sv);
                            affine16_backward(M, sv, gRV, g_ldj, true, gM, gRin);
                        }
                    } else {                                // Condition16TransLU (squeezetrans.py:134-144): as Condition16Trans, M given
                        inv4(M, Mi);
                        if (dir) {
                            float gMi[16];
#pragma unroll
                            for (int i = 0; i < 16; ++i) gMi[i] = 0.f;
                            affine16_forward_saved(Mi, 0.f, RinV, Rout, l, sv);
                            affine16_backward(Mi, sv, gRV, g_ldj, false, gMi, gRin);
                            inverse_matrix_grad<4>(Mi, gMi, gM);
                        } else {
                            affine16_forward_saved(M, 0.f, RinV, Rout, l, sv);
                            affine16_backward(M, sv, gRV, g_ldj, false, gM, gRin);
                        }
                        const float gl = dir ? -g_ldj : g_ldj;      // d log|det M| / dM = M^-T
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj) gM[4 * i + jj] += gl * Mi[4 * jj + i];
                    }
                }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) gM_out[i] = gM[i];
    *gRin_p = gRin_v;
}
// calculate_36 with M[i] = src[i * stride] (+ diag on the diagonal); the inverse pass goes through M^-1
RNF_RARE_FN void rare_gs36(const float *src, int stride, float diag, int dir, const Rot *Rin, const Rot *gR, float g_ldj, float *gM,
                                                   Rot *gRin) {
    float M[36];
#pragma unroll
    for (int i = 0; i < 36; ++i) { M[i] = src[i * stride] + ((i % 7) == 0 ? diag : 0.f); gM[i] = 0.f; }
    Rot g;
    if (dir) {
        float Mi[36], gMi[36];
#pragma unroll
        for (int i = 0; i < 36; ++i) gMi[i] = 0.f;
        inv6(M, Mi);
        gs36_backward(Mi, *Rin, *gR, g_ldj, gMi, g);
        inverse_matrix_grad<6>(Mi, gMi, gM);
    } else {
        gs36_backward(M, *Rin, *gR, g_ldj, gM, g);
    }
    *gRin = g;
}
// the conditional 3x3 kinds with M = I + (src[i * stride])
RNF_RARE_FN void rare_cond9(int kind, int dir, const float *src, int stride, const Rot *Rin, const Rot *gR, float g_ldj, float *gM,
                                                    Rot *gRin) {
    float M[9], g9[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) M[i] = src[i * stride] + ((i % 4) == 0 ? 1.f : 0.f);
    Rot g;
    cond9_backward(kind, dir != 0, M, *Rin, *gR, g_ldj, g9, g);
#pragma unroll
    for (int i = 0; i < 9; ++i) gM[i] = g9[i];
    *gRin = g;
}

template <bool HAS_FEATURE>
__global__ __launch_bounds__(TR_WAVES * 64) void flow_train_backward_kernel(const TrainArgs args) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef RNF_STAMPS
    unsigned long long tst_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tst_t = clock64();
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int ta = wave >> 1, tb = wave & 1;              // this wave's 32x32 tile of a 64 x 64 product
    const int K = args.K, F = HAS_FEATURE ? args.F : 0;
    const bool want_w = args.grads != nullptr;            // wave-uniform
    // LDS: X0, H1, H2, H3 (pre-activations, later reused for gradients), GA (gradient / reduction scratch), C (conditioner output,
    // then its gradient)
    const LMat X0{lds}, H1{lds + 64 * LROW}, H2{lds + 2 * 64 * LROW}, H3{lds + 3 * 64 * LROW}, GA{lds + 4 * 64 * LROW},
        YL{lds + 5 * 64 * LROW}, Cm{lds + TR_LDS_HEAD_FLOATS};     // YL: the conditioning column y of the 64 rotations, [3][64]
    float *red = GA.p;                                    // [wave][value][lane] while GA is not in use

    const long long nblocks = (args.n + 63) / 64;
    for (long long blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const long long sample = blk * 64 + lane;         // the rotation this lane carries through the per-sample math
        const bool valid = sample < args.n;
        const long long bsample = blk * 64 + 32 * tb + j; // the rotation of this lane's MFMA column
        const bool bvalid = bsample < args.n;
        Rot gR;
        gR.c0 = v3f{0.f, 0.f, 0.f}; gR.c1 = gR.c0; gR.c2 = gR.c0;
        float g_ldj = 0.f;
        if (valid) {
            g_ldj = args.g_ldj ? args.g_ldj[sample] : 0.f;
            if (args.g_rot_out) {
                const float *g = args.g_rot_out + sample * 9;
                gR.c0 = v3f{g[0], g[3], g[6]}; gR.c1 = v3f{g[1], g[4], g[7]}; gR.c2 = v3f{g[2], g[5], g[8]};
            }
        }
        for (int pos = args.n_layers - 1; pos >= 0; --pos) {
            const int2 d = args.layers[pos];
            const int kind = d.x & 15, perm_row = (d.x >> 4) & 15;
            const float *P = args.plain + d.y;
            float *Gp = args.grads + d.y;
            Rot Rin;
            Rin.c0 = v3f{1.f, 0.f, 0.f}; Rin.c1 = v3f{0.f, 1.f, 0.f}; Rin.c2 = v3f{0.f, 0.f, 1.f};
            if (valid && args.states) {
                const float *s = args.states + ((size_t)pos * args.n + sample) * 9;
                Rin.c0 = v3f{s[0], s[3], s[6]}; Rin.c1 = v3f{s[1], s[4], s[7]}; Rin.c2 = v3f{s[2], s[5], s[8]};
            }
            if (kind_is_side(kind)) {                     // per-sample matrix from the caller; its gradient goes back to the caller
                const int slot = (d.x >> 16) & 255;
                const float *m = args.side + ((size_t)slot * args.n + (valid ? sample : 0)) * 16;
                float gM[16];
                Rot gRin;
                rare_side(kind, args.dir, m, &Rin, &gR, g_ldj, gM, &gRin);
                if (wave == 0 && valid && args.side_grad) {
                    float *o = args.side_grad + ((size_t)slot * args.n + sample) * 16;
#pragma unroll
                    for (int i = 0; i < 16; ++i) o[i] = gM[i];
                }
                gR = gRin;
                RNF_TSTAMP(8)
                continue;
            }
            if (kind == RNF_KIND_AFFINE16) {              // every wave carries the chain; wave 0 adds the parameter gradient
                float M[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) M[i] = P[i];
                const bool orth = (d.x >> 8) & 1;         // UnconditionRot: ldj = 0 (flow/rottrans.py:21)
                float Mp[16];                             // the parameter matrix; the inverse pass applies M^-1 (squeezetrans.py:171-174), or
                if (args.dir) {                           // M^T for the orthogonal UnconditionRot (rottrans.py:26-28)
#pragma unroll
                    for (int i = 0; i < 16; ++i) Mp[i] = M[i];
                    if (orth) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj) M[4 * i + jj] = Mp[4 * jj + i];
                    } else {
                        inv4(Mp, M);
                    }
                }
                Rot Rout, gRin;
                AffineSaved sv;
                float l;
                affine16_forward_saved(M, 0.f, Rin, Rout, l, sv);
                float gM[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) gM[i] = 0.f;
                affine16_backward(M, sv, gR, g_ldj, orth, gM, gRin);
                if (args.dir) {                           // dL/dM from dL/d(applied matrix)
                    float gA[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) gA[i] = gM[i];
                    if (orth) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj) gM[4 * i + jj] = gA[4 * jj + i];
                    } else {
                        inverse_matrix_grad<4>(M, gA, gM);
                    }
                }
                if (wave == 0 && want_w) {                // batch sums through LDS: lane 4v + q adds 16 rotations of entry v
#pragma unroll
                    for (int i = 0; i < 16; ++i) GA.at(i, lane) = valid ? gM[i] : 0.f;
                    const int v = lane >> 2, q = lane & 3;
                    float tot = 0.f;
#pragma unroll
                    for (int i = 0; i < 16; ++i) tot += GA.at(v, 16 * q + i);
                    tot = quad_sum(tot);
                    if (q == 0) atomicAdd(Gp + v, tot);
                    const float gl = wave_sum(valid && !orth ? g_ldj : 0.f);      // log|det M^-1| = -log|det M| on the inverse pass
                    if (lane == 0) atomicAdd(args.g_ldj_sum + pos, args.dir ? -gl : gl);
                }
                gR = gRin;
                RNF_TSTAMP(8)
                continue;
            }
            if (kind == RNF_KIND_GS9) {                   // Uncondition9Trans / 9TransLU: M [9] (+3 pad) in the plain blob
                float M[9], gM[9];
#pragma unroll
                for (int i = 0; i < 9; ++i) { M[i] = P[i]; gM[i] = 0.f; }
                Rot gRin;
                if (args.dir) {                           // the inverse pass applies M^-1 (squeezetrans.py:259-261)
                    float Mi[9], gMi[9];
                    inv3(M, Mi);
#pragma unroll
                    for (int i = 0; i < 9; ++i) gMi[i] = 0.f;
                    gs9_backward(Mi, Rin, gR, g_ldj, gMi, gRin);
                    inverse_matrix_grad<3>(Mi, gMi, gM);
                } else {
                    gs9_backward(M, Rin, gR, g_ldj, gM, gRin);
                }
                if (wave == 0 && want_w) {
#pragma unroll
                    for (int i = 0; i < 9; ++i) GA.at(i, lane) = valid ? gM[i] : 0.f;
#pragma unroll
                    for (int i = 9; i < 16; ++i) GA.at(i, lane) = 0.f;
                    const int v = lane >> 2, q = lane & 3;
                    float tot = 0.f;
#pragma unroll
                    for (int i = 0; i < 16; ++i) tot += GA.at(v, 16 * q + i);
                    tot = quad_sum(tot);
                    if (q == 0 && v < 9) atomicAdd(Gp + v, tot);
                }
                gR = gRin;
                RNF_TSTAMP(8)
                continue;
            }
            if (kind == RNF_KIND_GS36) {                  // Uncondition36Trans: M [36] in the plain blob (squeezetrans.py:350-361)
                float gM[36];
                Rot gRin;
                rare_gs36(P, 1, 0.f, args.dir, &Rin, &gR, g_ldj, gM, &gRin);     // the inverse pass applies M^-1 (squeezetrans.py:359-361)
                if (want_w) {                             // batch sums through LDS, 16 entries per round (wave 0), three rounds
#pragma unroll
                    for (int base = 0; base < 48; base += 16) {
                        lds_barrier();
                        if (wave == 0) {
#pragma unroll
                            for (int i = 0; i < 16; ++i) GA.at(i, lane) = (valid && base + i < 36) ? gM[base + i < 36 ? base + i : 35] : 0.f;
                            const int v = lane >> 2, q = lane & 3;
                            float tot = 0.f;
#pragma unroll
                            for (int i = 0; i < 16; ++i) tot += GA.at(v, 16 * q + i);
                            tot = quad_sum(tot);
                            if (q == 0 && base + v < 36) atomicAdd(Gp + base + v, tot);
                        }
                    }
                    lds_barrier();
                }
                gR = gRin;
                RNF_TSTAMP(8)
                continue;
            }
            // ---- layers with a conditioner MLP (Moebius: input y (+) feature, 4K outputs; Condition16Trans: feature, 16 outputs) ----
            const bool mob = kind == RNF_KIND_MOBIUS;
            const int yo = mob ? 3 : 0, NI = yo + F, NO = mob ? 4 * K : (kind == RNF_KIND_MLP_ONLY ? ((d.x >> 16) & 255) : (kind_is_cond9(kind) ? 9 : (kind == RNF_KIND_COND36 ? 36 : 16)));
            const float *W0 = P, *b0 = W0 + 64 * NI, *W1 = b0 + 64, *b1 = W1 + 4096, *W3 = b1 + 64, *b3 = W3 + 4096, *W5 = b3 + 64,
                        *b5 = W5 + 4096, *WL = b5 + 64, *bL = WL + (size_t)NO * 64;
            float *gW0 = Gp, *gb0 = gW0 + 64 * NI, *gW1 = gb0 + 64, *gb1 = gW1 + 4096, *gW3 = gb1 + 64, *gb3 = gW3 + 4096, *gW5 = gb3 + 64,
                  *gb5 = gW5 + 4096, *gWL = gb5 + 64, *gbL = gWL + (size_t)NO * 64;
            const int p1 = (perm_row + 1) % 3;
            const v3f y = get_col(Rin, p1);
            const int ntiles = (NO + 31) / 32;            // row tiles of fc_last

            // ================= forward recompute =================
            // (A operands are loaded one phase ahead: RowsA / ColsA, so their latency hides behind the previous phase and barrier)
            RowsA wnext = load_rows64(W1, b1, 32 * ta + j, 64, h);
            if (mob && wave == 0) { YL.at(0, lane) = y.x; YL.at(1, lane) = y.y; YL.at(2, lane) = y.z; }
            for (int o = NO + wave; o < ((NO + 63) & ~63); o += TR_WAVES) Cm.at(o, lane) = 0.f;      // pad rows of C: B side of WL^T slabs
            const int bs = 32 * tb + j;                   // this lane's B column (sample) in [0, 64)
            // x0 = b0 + W0[:, yo:] f  (K dimension = F in chunks of 64) + W0[:, :3] y (two K = 2 steps)
            {
                f32x16 acc = RNF_MFMA(b0[32 * ta + j], h ? 0.f : 1.f, zero16());
                const float *wrow = W0 + (size_t)(32 * ta + j) * NI;
                if (HAS_FEATURE) {
                    // chunks of 16 K steps (k = 64 (ch / 2) + 32 h + 16 (ch & 1) + u); each lane's 16 operands are one 64-byte run of its row
                    const float *frow = args.feature + (bvalid ? bsample : 0) * F;
                    const int nch = (F + 63) / 64 * 2;
                    auto kof = [&](int ch, int u) { return 64 * (ch >> 1) + 32 * h + 16 * (ch & 1) + u; };
                    auto load_chunk = [&](int ch, float (&a)[16], float (&b)[16]) {
                        const int k0 = kof(ch, 0);
                        if (k0 + 16 <= F) {                       // wave-uniform per half: 4 x 16-byte loads per operand (4-byte aligned)
                            const Float4U *pa = reinterpret_cast<const Float4U *>(wrow + yo + k0);
                            const Float4U *pb = reinterpret_cast<const Float4U *>(frow + k0);
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const Float4U va = pa[q], vb = pb[q];
                                a[4 * q] = va.x; a[4 * q + 1] = va.y; a[4 * q + 2] = va.z; a[4 * q + 3] = va.w;
                                b[4 * q] = vb.x; b[4 * q + 1] = vb.y; b[4 * q + 2] = vb.z; b[4 * q + 3] = vb.w;
                            }
                        } else {
#pragma unroll
                            for (int u = 0; u < 16; ++u) {
                                const int k = k0 + u, kc = k < F ? k : F - 1;
                                a[u] = wrow[yo + kc];
                                b[u] = frow[kc];
                            }
                        }
                    };
                    auto mul_chunk = [&](int ch, const float (&a)[16], const float (&b)[16]) {
#pragma unroll
                        for (int u = 0; u < 16; ++u) acc = RNF_MFMA(kof(ch, u) < F ? a[u] : 0.f, bvalid ? b[u] : 0.f, acc);
                    };
                    for (int ch = 0; ch < nch; ++ch) {
                        float av[16], bv[16];
                        load_chunk(ch, av, bv);
                        __builtin_amdgcn_sched_barrier(0);
                        mul_chunk(ch, av, bv);
                    }
                }
                if (mob) {
                    const float a0 = wrow[h], a2 = wrow[2];
                    lds_barrier();                        // YL
                    acc = RNF_MFMA(a0, YL.at(h, bs), acc);
                    acc = RNF_MFMA(h ? 0.f : a2, h ? 0.f : YL.at(2, bs), acc);
                }
                store_tile(X0, 32 * ta, 64, bs, h, acc);
            }
            lds_barrier();
            RNF_TSTAMP(0)
            // H1, H2 hold relu(h1), relu(h2); H3 holds t = relu(x0 + h3) (the ReLU masks only need the sign)
            {
                const RowsA w1 = wnext;
                wnext = load_rows64(W3, b3, 32 * ta + j, 64, h);
                f32x16 acc = mfma_rows64(w1, h, [&](int k) { return X0.at(k, bs); }, Relu());
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = Relu()(acc[r]);
                store_tile(H1, 32 * ta, 64, bs, h, acc);
            }
            lds_barrier();
            {
                const RowsA w3 = wnext;
                wnext = load_rows64(W5, b5, 32 * ta + j, 64, h);
                f32x16 acc = mfma_rows64(w3, h, [&](int k) { return H1.at(k, bs); });
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = Relu()(acc[r]);
                store_tile(H2, 32 * ta, 64, bs, h, acc);
            }
            lds_barrier();
            {
                const RowsA w5 = wnext;
                wnext = load_rows64(WL, bL, 32 * ta + j, NO, h);
                f32x16 acc = mfma_rows64(w5, h, [&](int k) { return H2.at(k, bs); });
                {
                    float xv[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) xv[r] = X0.at(32 * ta + rho(r, h), bs);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = Relu()(acc[r] + xv[r]);
                }
                store_tile(H3, 32 * ta, 64, bs, h, acc);
            }
            const LMat &T = H3;
            lds_barrier();
            RNF_TSTAMP(1)
            for (int rt = ta; rt < ntiles; rt += 2) {     // fc_last: C = WL t + bL
                const RowsA wl = wnext;
                if (rt + 2 < ntiles) wnext = load_rows64(WL, bL, 32 * (rt + 2) + j, NO, h);
                store_tile(Cm, 32 * rt, NO, bs, h, mfma_rows64(wl, h, [&](int k) { return T.at(k, bs); }));
            }
            ColsA cnext = load_cols(WL, 64, 0, NO, 32 * ta + j, 64, h);        // first slab of WL^T, needed after the layer math
            lds_barrier();
            RNF_TSTAMP(2)
            // ================= layer math: forward sums + backward; dL/dC overwrites C =================
            Rot gRin;
            if (mob && args.dir) {
                // ---- MobiusFlow.inverse: the root theta comes back from the layer's own output (column p0 of the next state) ----
                Rot Rout;
                Rout.c0 = v3f{1.f, 0.f, 0.f}; Rout.c1 = v3f{0.f, 1.f, 0.f}; Rout.c2 = v3f{0.f, 0.f, 1.f};
                if (valid) {
                    const float *s = pos + 1 < args.n_layers ? args.states + ((size_t)(pos + 1) * args.n + sample) * 9 : args.rot_final + sample * 9;
                    Rout.c0 = v3f{s[0], s[3], s[6]}; Rout.c1 = v3f{s[1], s[4], s[7]}; Rout.c2 = v3f{s[2], s[5], s[8]};
                } else {
                    Rout = Rin;                             // padding lanes: x = tx, i.e. theta = pi; finite everywhere, gradients exactly 0
                }
                MobiusInvSaved sv;
                mobius_inv_frame(Rin, perm_row, Rout, sv);
                const LaneRow crow{Cm, lane};
                const int k0 = wave * K / TR_WAVES, k1 = (wave + 1) * K / TR_WAVES;
                float S = 0.f, A = 0.f, J = 0.f, Cth = 0.f;
                mobius_inv_segments_sums(sv, crow, K, k0, k1, S, A, J, Cth);
                red[(wave * 4 + 0) * 64 + lane] = S;
                red[(wave * 4 + 1) * 64 + lane] = A;
                red[(wave * 4 + 2) * 64 + lane] = J;
                red[(wave * 4 + 3) * 64 + lane] = Cth;
                lds_barrier();
                S = A = J = Cth = 0.f;
#pragma unroll
                for (int w = 0; w < TR_WAVES; ++w) {
                    S += red[(w * 4 + 0) * 64 + lane]; A += red[(w * 4 + 1) * 64 + lane];
                    J += red[(w * 4 + 2) * 64 + lane]; Cth += red[(w * 4 + 3) * 64 + lane];
                }
                MobiusGrad mg;
                mobius_inv_backward_head(sv, S, A, J, Cth, gR, g_ldj, mg);
                v3f pr = v3f{0.f, 0.f, 0.f}, pv = pr;
                mobius_segments_backward_range_at(sv.b.f, sv.cs, sv.sn, sv.theta, crow, K, k0, k1, mg, crow, pr, pv);
                float *red2 = red + TR_WAVES * 4 * 64;
                red2[(wave * 6 + 0) * 64 + lane] = pr.x; red2[(wave * 6 + 1) * 64 + lane] = pr.y; red2[(wave * 6 + 2) * 64 + lane] = pr.z;
                red2[(wave * 6 + 3) * 64 + lane] = pv.x; red2[(wave * 6 + 4) * 64 + lane] = pv.y; red2[(wave * 6 + 5) * 64 + lane] = pv.z;
                lds_barrier();
#pragma unroll
                for (int w = 0; w < TR_WAVES; ++w) {
                    mg.g_r = mg.g_r + v3f{red2[(w * 6 + 0) * 64 + lane], red2[(w * 6 + 1) * 64 + lane], red2[(w * 6 + 2) * 64 + lane]};
                    mg.g_v = mg.g_v + v3f{red2[(w * 6 + 3) * 64 + lane], red2[(w * 6 + 4) * 64 + lane], red2[(w * 6 + 5) * 64 + lane]};
                }
                mobius_backward_tail(sv.b, mg, gRin);
            } else if (mob) {
                MobiusSaved sv;
                mobius_frame(Rin, perm_row, sv);
                const LaneRow crow{Cm, lane};
                const int k0 = wave * K / TR_WAVES, k1 = (wave + 1) * K / TR_WAVES;
                float S = 0.f, A = 0.f, J = 0.f;
                mobius_segments_sums(sv, crow, K, k0, k1, S, A, J);
                red[(wave * 3 + 0) * 64 + lane] = S;
                red[(wave * 3 + 1) * 64 + lane] = A;
                red[(wave * 3 + 2) * 64 + lane] = J;
                lds_barrier();
                S = A = J = 0.f;
#pragma unroll
                for (int w = 0; w < TR_WAVES; ++w) {
                    S += red[(w * 3 + 0) * 64 + lane]; A += red[(w * 3 + 1) * 64 + lane]; J += red[(w * 3 + 2) * 64 + lane];
                }
                mobius_combine(sv, S, A, J);
                MobiusGrad mg;
                mobius_backward_head(sv, gR, g_ldj, mg);
                v3f pr = v3f{0.f, 0.f, 0.f}, pv = pr;
                mobius_segments_backward_range(sv, crow, K, k0, k1, mg, crow, pr, pv);
                float *red2 = red + TR_WAVES * 3 * 64;
                red2[(wave * 6 + 0) * 64 + lane] = pr.x; red2[(wave * 6 + 1) * 64 + lane] = pr.y; red2[(wave * 6 + 2) * 64 + lane] = pr.z;
                red2[(wave * 6 + 3) * 64 + lane] = pv.x; red2[(wave * 6 + 4) * 64 + lane] = pv.y; red2[(wave * 6 + 5) * 64 + lane] = pv.z;
                lds_barrier();
#pragma unroll
                for (int w = 0; w < TR_WAVES; ++w) {
                    mg.g_r = mg.g_r + v3f{red2[(w * 6 + 0) * 64 + lane], red2[(w * 6 + 1) * 64 + lane], red2[(w * 6 + 2) * 64 + lane]};
                    mg.g_v = mg.g_v + v3f{red2[(w * 6 + 3) * 64 + lane], red2[(w * 6 + 4) * 64 + lane], red2[(w * 6 + 5) * 64 + lane]};
                }
                mobius_backward_tail(sv, mg, gRin);
            } else if (kind == RNF_KIND_MLP_ONLY) {
                // the conditioner on its own (networks of the side layers): dL/d(outputs) comes from the caller
                gRin = gR;
                lds_barrier();                          // every wave is past the fc_last stores into C
                if (wave == 0) {
                    for (int i = 0; i < NO; ++i) Cm.at(i, lane) = valid ? args.g_out_ext[sample * NO + i] : 0.f;
                }
                lds_barrier();
            } else if (kind == RNF_KIND_COND36) {
                // Condition36Trans (squeezetrans.py:334-347): M = I + reshape(net(f), 6, 6) per sample; the inverse pass applies M^-1
                float gM[36];
                rare_gs36(Cm.p + lane, LROW, 1.f, args.dir, &Rin, &gR, g_ldj, gM, &gRin);
                lds_barrier();                          // every wave has read C
                if (wave == 0) {
#pragma unroll
                    for (int i = 0; i < 36; ++i) Cm.at(i, lane) = gM[i];
                }
                lds_barrier();
            } else if (kind_is_cond9(kind)) {
                // Condition9Trans / 9RotL / 9RotR / 9RotRSmith (squeezetrans.py:234-247, rottrans.py:108-181): M = I + reshape(net(f), 3, 3)
                float gM[9];
                rare_cond9(kind, args.dir, Cm.p + lane, LROW, &Rin, &gR, g_ldj, gM, &gRin);
                lds_barrier();                          // every wave has read C
                if (wave == 0) {
#pragma unroll
                    for (int i = 0; i < 9; ++i) Cm.at(i, lane) = gM[i];
                }
                lds_barrier();
            } else {
                // Condition16Trans (flow/squeezetrans.py:41-50): M = I + reshape(net(f), 4, 4), ldj = log|det M| - 2 log|M q|^2
                float M[16], Mi[16], gM[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) { M[i] = Cm.at(i, lane) + ((i % 5) == 0 ? 1.f : 0.f); gM[i] = 0.f; }
                inv4(M, Mi);
                Rot Rout;
                AffineSaved sv;
                float l;
                if (args.dir) {                         // inverse pass: the layer applies M^-1 (squeezetrans.py:51-55), log|det M^-1| = -log|det M|
                    float gMi[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) gMi[i] = 0.f;
                    affine16_forward_saved(Mi, 0.f, Rin, Rout, l, sv);
                    affine16_backward(Mi, sv, gR, g_ldj, false, gMi, gRin);
                    inverse_matrix_grad<4>(Mi, gMi, gM);
                } else {
                    affine16_forward_saved(M, 0.f, Rin, Rout, l, sv);
                    affine16_backward(M, sv, gR, g_ldj, false, gM, gRin);
                }
                const float gl = args.dir ? -g_ldj : g_ldj;
                lds_barrier();                          // every wave has read C
                if (wave == 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) Cm.at(4 * i + jj, lane) = gM[4 * i + jj] + gl * Mi[4 * jj + i];
                }
                lds_barrier();
            }
            // (padding lanes carry gR = 0 and g_ldj = 0, so their dL/dC and everything derived from it is exactly 0)

            RNF_TSTAMP(3)
            // ================= conditioner backward =================
            // fc_last: gWL += g_c t^T, gbL += rowsum(g_c), g_t = WL^T g_c
            for (int rt = ta; want_w && rt < ntiles; rt += 2) {
                const f32x16 acc = mfma_samples(h, [&](int s) { return Cm.at(32 * rt + j, s); }, [&](int s) { return T.at(32 * tb + j, s); });
                scatter_add(gWL, 64, 32 * rt, NO, 32 * tb + j, true, h, acc);
            }
            if (want_w) bias_grad(Cm, NO, gbL, tid);
            RNF_TSTAMP(4)
            {
                f32x16 acc = zero16();
                for (int k0 = 0; k0 < NO; k0 += 64) {     // 64-row slabs of WL^T; the next slab (or W5^T for the first hidden step) loads ahead
                    const ColsA cur = cnext;
                    if (k0 + 64 < NO) cnext = load_cols(WL, 64, k0 + 64, NO, 32 * ta + j, 64, h);
                    else cnext = load_cols(W5, 64, 0, 64, 32 * ta + j, 64, h);
                    acc = mfma_cols(cur, k0, h, [&](int k) { return Cm.at(k, bs); }, acc);
                }
                {                                         // through t = relu(x0 + h3): this is dL/dh3 and the residual part of dL/dx0
                    float tv[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) tv[r] = T.at(32 * ta + rho(r, h), bs);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = tv[r] > 0.f ? acc[r] : 0.f;
                }
                lds_barrier();                            // `red` (inside GA) is no longer read
                store_tile(GA, 32 * ta, 64, bs, h, acc);
            }
            lds_barrier();
            RNF_TSTAMP(5)
            // hidden layers, last to first.  (g_out, act_in) -> gW, gb, g_in masked by the ReLU of its pre-activation.
            // cnext holds the transposed weights of this step; Wnext: the matrix of the FOLLOWING step (loaded ahead), or nullptr
            auto hidden_backward = [&](float *gW, float *gb, const LMat &Gout, const LMat &PreIn, const LMat &Gin, const float *Wnext) {
                const ColsA cur = cnext;
                if (Wnext) cnext = load_cols(Wnext, 64, 0, 64, 32 * ta + j, 64, h);
                if (want_w) {
                    const f32x16 wg = mfma_samples(h, [&](int s) { return Gout.at(32 * ta + j, s); }, [&](int s) { return PreIn.at(32 * tb + j, s); }, Relu());
                    scatter_add(gW, 64, 32 * ta, 64, 32 * tb + j, true, h, wg);
                    bias_grad(Gout, 64, gb, tid);
                }
                f32x16 acc = mfma_cols(cur, 0, h, [&](int k) { return Gout.at(k, bs); }, zero16());
                {
                    float pv[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) pv[r] = PreIn.at(32 * ta + rho(r, h), bs);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = pv[r] > 0.f ? acc[r] : 0.f;
                }
                lds_barrier();                            // Gin may alias a buffer other waves were still reading
                store_tile(Gin, 32 * ta, 64, bs, h, acc);
                lds_barrier();
            };
            // (each activation buffer is free once its ReLU mask has been applied, and takes the next gradient)
            hidden_backward(gW5, gb5, GA, H2, H3, W3);    // g_h3 (GA) -> g_h2 (H3's storage)
            hidden_backward(gW3, gb3, H3, H1, H2, W1);    // g_h2      -> g_h1 (H2's storage)
            hidden_backward(gW1, gb1, H2, X0, H1, nullptr);   // g_h1  -> chain part of g_x0 (H1's storage)
            RNF_TSTAMP(6)
            const LMat GB = H3;                           // total dL/dx0 = chain + residual
            {
                float c1[16], c2[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) { c1[i] = H1.at(16 * wave + i, lane); c2[i] = GA.at(16 * wave + i, lane); }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 16; ++i) GB.at(16 * wave + i, lane) = c1[i] + c2[i];
            }
            lds_barrier();
            // fc_first: x0 = W0 (y (+) f) + b0
            if (want_w) bias_grad(GB, 64, gb0, tid);
            if (mob) {
                if (wave < 2 && want_w) {                 // gW0[:, 0:3] += g y^T: two row tiles, columns 0..2 of a 32-column tile
                    const f32x16 acc = mfma_samples(h, [&](int s) { return GB.at(32 * wave + j, s); }, [&](int s) { return j < 3 ? YL.at(j, s) : 0.f; });
                    scatter_add(gW0, NI, 32 * wave, 64, j, j < 3, h, acc);
                }
                float gy0 = 0.f, gy1 = 0.f, gy2 = 0.f;                      // the conditioner-input path of dL/dy (every wave, all rows)
                for (int o0 = 0; o0 < 64; o0 += 16) {
                    float gv[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) gv[i] = GB.at(o0 + i, lane);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float *wr = W0 + (size_t)(o0 + i) * NI;
                        gy0 = fmaf(wr[0], gv[i], gy0);
                        gy1 = fmaf(wr[1], gv[i], gy1);
                        gy2 = fmaf(wr[2], gv[i], gy2);
                    }
                }
                set_col(gRin, p1, get_col(gRin, p1) + v3f{gy0, gy1, gy2});
            }
            if (HAS_FEATURE) {
                // gW0[o][yo + c] += sum_s g[o][s] f[s][c]: tiles 2 (rows) x ceil(F/32) (columns), 4 waves
                const int ctiles = (F + 31) / 32;
                for (int t = wave; want_w && t < 2 * ctiles; t += TR_WAVES) {
                    const int rt = t & 1, ct = t >> 1;
                    const int c = 32 * ct + j;
                    f32x16 acc = zero16();
                    const long long r0 = blk * 64 + 32 * h;
                    const float *fp = args.feature + (r0 < args.n ? r0 : args.n - 1) * F + (c < F ? c : F - 1);   // row 32h + m of the block, column c
#pragma unroll
                    for (int half = 0; half < 2; ++half) {                   // (rows of padding rotations: their g is exactly 0)
                        float b[16], gq[16];
#pragma unroll
                        for (int u = 0; u < 16; ++u) {
                            b[u] = *fp;
                            if (blk * 64 + 32 * h + 16 * half + u + 1 < args.n) fp += F;
                            gq[u] = GB.at(32 * rt + j, 32 * h + 16 * half + u);
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int u = 0; u < 16; ++u) acc = RNF_MFMA(gq[u], b[u], acc);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    scatter_add(gW0 + yo, NI, 32 * rt, 64, c, c < F, h, acc);
                }
                asm volatile("" ::: "memory");       // keeps the two tile loops apart (hoisting the second loop's loads over the first spills)
                if (args.g_feature) {
                    // g_f[s][c] = sum_o g[o][s] W0[o][yo + c]: tiles 2 (sample rows) x ceil(F/32) (feature columns); A = g^T from LDS,
                    // B = rows of W0 (coalesced over the feature index), and the tile lands feature-contiguous for the read-modify-write
                    for (int t = wave; t < 2 * ctiles; t += TR_WAVES) {
                        const int st = t & 1, ct = t >> 1;
                        const int c = 32 * ct + j, cc = c < F ? c : F - 1;
                        f32x16 acc = zero16();
#pragma unroll
                        for (int half = 0; half < 2; ++half) {
                            float av[16], bv[16];
#pragma unroll
                            for (int m = 0; m < 16; ++m) {
                                const int o = 32 * h + 16 * half + m;
                                av[m] = GB.at(o, 32 * st + j);
                                bv[m] = W0[(size_t)o * NI + yo + cc];
                            }
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int m = 0; m < 16; ++m) acc = RNF_MFMA(av[m], bv[m], acc);
                            __builtin_amdgcn_sched_barrier(0);
                        }
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const long long smp = blk * 64 + 32 * st + rho(r, h);
                            if (c < F && smp < args.n) args.g_feature[smp * F + c] += acc[r];
                        }
                    }
                }
            }
            lds_barrier();
            gR = gRin;
            RNF_TSTAMP(7)
        }
        lds_barrier();                                    // g_rot_in may BE g_rot_out (chunked sweeps, rnf_api.hip): every wave has read its copy
        if (valid && wave == 0 && args.g_rot_in) {
            float *o = args.g_rot_in + sample * 9;
            o[0] = gR.c0.x; o[1] = gR.c1.x; o[2] = gR.c2.x; o[3] = gR.c0.y; o[4] = gR.c1.y; o[5] = gR.c2.y; o[6] = gR.c0.z; o[7] = gR.c1.z; o[8] = gR.c2.z;
        }
    }
#ifdef RNF_STAMPS
    RNF_TSTAMP(9)
    if (args.stamps && tid == 0) for (int i_ = 0; i_ < 10; ++i_) atomicAdd(args.stamps + i_, tst_acc[i_]);
#endif
}

// d log|det M| / dM = M^-T, weighted by the batch sum of dL/dldj (Uncondition16Trans, flow/squeezetrans.py:33-38,57-66)
__global__ void affine_logdet_grad_kernel(const TrainArgs args) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= args.n_layers || !args.grads) return;
    const int2 d = args.layers[l];
    if ((d.x & 15) != RNF_KIND_AFFINE16 || ((d.x >> 8) & 1)) return;
    float M[16], Mi[16];
    for (int i = 0; i < 16; ++i) M[i] = args.plain[d.y + i];
    inv4(M, Mi);
    const float g = args.g_ldj_sum[l];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) args.grads[d.y + 4 * i + j] += g * Mi[4 * j + i];
}

}  // namespace rnf
