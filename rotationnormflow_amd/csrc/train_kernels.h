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
    float *grads;             // gradient blob (same layout), zeroed by the caller
    const float *g_rot_out;   // [n][9] dL/dR_out or nullptr (zeros)
    const float *g_ldj;       // [n]   dL/dldj
    float *g_rot_in;          // [n][9]
    float *g_feature;         // [n][F] or nullptr
    float *g_ldj_sum;         // [n_layers] sum over the batch of dL/dldj (for the log|det M| term), zeroed by the caller
    long long n;
    int n_layers, K, F;
    // per layer: x = kind | perm_row << 4 | orthogonal << 8, y = plain offset
    int2 layers[TR_MAX_LAYERS];
};

// LDS matrix [rows][64 samples] with an XOR swizzle: conflict free both for "lanes = 32 consecutive samples of one row" and for
// "lanes = 32 consecutive rows at one sample" (the two MFMA operand patterns)
struct LMat {
    float *p;
    __device__ __forceinline__ float &at(int row, int s) const { return p[row * 64 + (s ^ (row & 63))]; }
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

// acc[rho][j] += sum_{k<64} W[row][k] * b(k):  A = 32 rows of a row-major weight matrix (16-byte aligned rows, 64 columns),
// this lane supplies row `row` (masked by row_ok), k = 32h + m.  b(k): B element (k, this lane's sample).
template <class BFn>
__device__ __forceinline__ f32x16 mfma_rows64(const float *__restrict__ W, int row, bool row_ok, int h, BFn b, f32x16 acc) {
    float4 w[8];
    const float4 *src = reinterpret_cast<const float4 *>(W + (size_t)row * 64 + 32 * h);
#pragma unroll
    for (int q = 0; q < 8; ++q) w[q] = row_ok ? src[q] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int k = 32 * h + 4 * q;
        acc = RNF_MFMA(w[q].x, b(k), acc);
        acc = RNF_MFMA(w[q].y, b(k + 1), acc);
        acc = RNF_MFMA(w[q].z, b(k + 2), acc);
        acc = RNF_MFMA(w[q].w, b(k + 3), acc);
    }
    return acc;
}

// acc[rho][j] += sum_{k<nk} W[k][col] * b(k): A = 32 columns of a row-major matrix with leading dimension ldw (the transposed
// product); this lane supplies column `col` (masked by col_ok), k = (nk/2) h + m.  nk even.
template <class BFn>
__device__ __forceinline__ f32x16 mfma_cols(const float *__restrict__ W, int ldw, int nk, int col, bool col_ok, int h, BFn b, f32x16 acc) {
    const int half = nk >> 1;
    for (int m0 = 0; m0 < half; m0 += 16) {
        float a[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int k = half * h + m0 + u;
            a[u] = (col_ok && m0 + u < half) ? W[(size_t)k * ldw + col] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int k = half * h + m0 + u;
            acc = RNF_MFMA(a[u], (m0 + u < half) ? b(k) : 0.f, acc);
        }
    }
    return acc;
}

// acc[rho][j] = sum_{s<64} a(s) * b(s): the product over the block's samples (weight gradients), s = 32h + m
template <class AFn, class BFn>
__device__ __forceinline__ f32x16 mfma_samples(int h, AFn a, BFn b) {
    f32x16 acc = zero16();
#pragma unroll 8
    for (int m = 0; m < 32; ++m) acc = RNF_MFMA(a(32 * h + m), b(32 * h + m), acc);
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

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// gb[o] += sum_s G[o][s] for o < n_out (thread t handles rows t, t + 256, ...)
__device__ __forceinline__ void bias_grad(const LMat &G, int n_out, float *gb, int tid) {
    for (int o = tid; o < n_out; o += TR_WAVES * 64) {
        float acc = 0.f;
#pragma unroll 8
        for (int s = 0; s < 64; ++s) acc += G.at(o, s);
        atomicAdd(gb + o, acc);
    }
}

// hidden layer forward: OUT = W relu?(IN) + b, one 32x32 tile per wave
template <class BFn>
__device__ __forceinline__ void hidden_forward(const float *W, const float *bias, BFn in, const LMat &OUT, int ta, int tb, int j, int h) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bias[32 * ta + rho(r, h)];
    acc = mfma_rows64(W, 32 * ta + j, true, h, in, acc);
    store_tile(OUT, 32 * ta, 64, 32 * tb + j, h, acc);
}

__global__ __launch_bounds__(TR_WAVES * 64) void flow_train_backward_kernel(const TrainArgs args) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int ta = wave >> 1, tb = wave & 1;              // this wave's 32x32 tile of a 64 x 64 product
    const int K = args.K, F = args.F;
    // LDS: X0, H1, H2, H3 (pre-activations), GA, GB (gradients / reduction scratch), C (conditioner output, then its gradient)
    const LMat X0{lds}, H1{lds + 4096}, H2{lds + 8192}, H3{lds + 12288}, GA{lds + 16384}, GB{lds + 20480}, Cm{lds + 24576};
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
            g_ldj = args.g_ldj[sample];
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
            if (valid) {
                const float *s = args.states + ((size_t)pos * args.n + sample) * 9;
                Rin.c0 = v3f{s[0], s[3], s[6]}; Rin.c1 = v3f{s[1], s[4], s[7]}; Rin.c2 = v3f{s[2], s[5], s[8]};
            }
            if (kind == RNF_KIND_AFFINE16) {              // every wave carries the chain; wave 0 adds the parameter gradient
                float M[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) M[i] = P[i];
                Rot Rout, gRin;
                AffineSaved sv;
                float l;
                affine16_forward_saved(M, 0.f, Rin, Rout, l, sv);
                float gM[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) gM[i] = 0.f;
                const bool orth = (d.x >> 8) & 1;         // UnconditionRot: ldj = 0 (flow/rottrans.py:21)
                affine16_backward(M, sv, gR, g_ldj, orth, gM, gRin);
                if (wave == 0) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float tot = wave_sum(valid ? gM[i] : 0.f);
                        if (lane == 0) atomicAdd(Gp + i, tot);
                    }
                    const float gl = wave_sum(valid && !orth ? g_ldj : 0.f);
                    if (lane == 0) atomicAdd(args.g_ldj_sum + pos, gl);
                }
                gR = gRin;
                continue;
            }
            // ---- layers with a conditioner MLP (Moebius: input y (+) feature, 4K outputs; Condition16Trans: feature, 16 outputs) ----
            const bool mob = kind == RNF_KIND_MOBIUS;
            const int yo = mob ? 3 : 0, NI = yo + F, NO = mob ? 4 * K : 16;
            const float *W0 = P, *b0 = W0 + 64 * NI, *W1 = b0 + 64, *b1 = W1 + 4096, *W3 = b1 + 64, *b3 = W3 + 4096, *W5 = b3 + 64,
                        *b5 = W5 + 4096, *WL = b5 + 64, *bL = WL + (size_t)NO * 64;
            float *gW0 = Gp, *gb0 = gW0 + 64 * NI, *gW1 = gb0 + 64, *gb1 = gW1 + 4096, *gW3 = gb1 + 64, *gb3 = gW3 + 4096, *gW5 = gb3 + 64,
                  *gb5 = gW5 + 4096, *gWL = gb5 + 64, *gbL = gWL + (size_t)NO * 64;
            const int p1 = (perm_row + 1) % 3;
            const v3f y = get_col(Rin, p1);
            const int ntiles = (NO + 31) / 32;            // row tiles of fc_last

            // ================= forward recompute =================
            // x0 = b0 + W0[:, yo:] f  (matrix cores, K dimension = F in chunks of 64)  + W0[:, :3] y (per lane, below)
            {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = b0[32 * ta + rho(r, h)];
                for (int c0 = 0; c0 < F; c0 += 64) {
                    for (int m0 = 0; m0 < 32; m0 += 16) {
                        float a[16], b[16];
#pragma unroll
                        for (int u = 0; u < 16; ++u) {
                            const int k = c0 + 32 * h + m0 + u;
                            a[u] = k < F ? W0[(size_t)(32 * ta + j) * NI + yo + k] : 0.f;
                            b[u] = (k < F && bvalid) ? args.feature[bsample * F + k] : 0.f;
                        }
#pragma unroll
                        for (int u = 0; u < 16; ++u) acc = RNF_MFMA(a[u], b[u], acc);
                    }
                }
                store_tile(X0, 32 * ta, 64, 32 * tb + j, h, acc);
            }
            __syncthreads();
            if (mob) {
                for (int o = 16 * wave; o < 16 * wave + 16; ++o)
                    X0.at(o, lane) += fmaf(W0[(size_t)o * NI + 2], y.z, fmaf(W0[(size_t)o * NI + 1], y.y, W0[(size_t)o * NI] * y.x));
                __syncthreads();
            }
            const int bs = 32 * tb + j;                   // this lane's B column (sample) in [0, 64)
            hidden_forward(W1, b1, [&](int k) { return fmaxf(X0.at(k, bs), 0.f); }, H1, ta, tb, j, h);
            __syncthreads();
            hidden_forward(W3, b3, [&](int k) { return fmaxf(H1.at(k, bs), 0.f); }, H2, ta, tb, j, h);
            __syncthreads();
            hidden_forward(W5, b5, [&](int k) { return fmaxf(H2.at(k, bs), 0.f); }, H3, ta, tb, j, h);
            __syncthreads();
            auto t_at = [&](int k, int s) { return fmaxf(X0.at(k, s) + H3.at(k, s), 0.f); };      // t = relu(x0 + h3)
            for (int rt = ta; rt < ntiles; rt += 2) {     // fc_last: C = WL t + bL
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = 32 * rt + rho(r, h);
                    acc[r] = o < NO ? bL[o] : 0.f;
                }
                acc = mfma_rows64(WL, 32 * rt + j, 32 * rt + j < NO, h, [&](int k) { return t_at(k, bs); }, acc);
                store_tile(Cm, 32 * rt, NO, bs, h, acc);
            }
            __syncthreads();

            // ================= layer math: forward sums + backward; dL/dC overwrites C =================
            Rot gRin;
            if (mob) {
                MobiusSaved sv;
                mobius_frame(Rin, perm_row, sv);
                const LaneRow crow{Cm, lane};
                const int k0 = wave * K / TR_WAVES, k1 = (wave + 1) * K / TR_WAVES;
                float S = 0.f, A = 0.f, J = 0.f;
                mobius_segments_sums(sv, crow, K, k0, k1, S, A, J);
                red[(wave * 3 + 0) * 64 + lane] = S;
                red[(wave * 3 + 1) * 64 + lane] = A;
                red[(wave * 3 + 2) * 64 + lane] = J;
                __syncthreads();
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
                __syncthreads();
#pragma unroll
                for (int w = 0; w < TR_WAVES; ++w) {
                    mg.g_r = mg.g_r + v3f{red2[(w * 6 + 0) * 64 + lane], red2[(w * 6 + 1) * 64 + lane], red2[(w * 6 + 2) * 64 + lane]};
                    mg.g_v = mg.g_v + v3f{red2[(w * 6 + 3) * 64 + lane], red2[(w * 6 + 4) * 64 + lane], red2[(w * 6 + 5) * 64 + lane]};
                }
                mobius_backward_tail(sv, mg, gRin);
            } else {
                // Condition16Trans (flow/squeezetrans.py:41-50): M = I + reshape(net(f), 4, 4), ldj = log|det M| - 2 log|M q|^2
                float M[16], Mi[16], gM[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) { M[i] = Cm.at(i, lane) + ((i % 5) == 0 ? 1.f : 0.f); gM[i] = 0.f; }
                inv4(M, Mi);
                Rot Rout;
                AffineSaved sv;
                float l;
                affine16_forward_saved(M, 0.f, Rin, Rout, l, sv);
                affine16_backward(M, sv, gR, g_ldj, false, gM, gRin);
                __syncthreads();                          // every wave has read C
                if (wave == 0) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) Cm.at(4 * i + jj, lane) = gM[4 * i + jj] + g_ldj * Mi[4 * jj + i];
                }
                __syncthreads();
            }
            // (padding lanes carry gR = 0 and g_ldj = 0, so their dL/dC and everything derived from it is exactly 0)

            // ================= conditioner backward =================
            // fc_last: gWL += g_c t^T, gbL += rowsum(g_c), g_t = WL^T g_c
            for (int rt = ta; rt < ntiles; rt += 2) {
                const int o = 32 * rt + j;
                const f32x16 acc = mfma_samples(h, [&](int s) { return o < NO ? Cm.at(o, s) : 0.f; }, [&](int s) { return t_at(32 * tb + j, s); });
                scatter_add(gWL, 64, 32 * rt, NO, 32 * tb + j, true, h, acc);
            }
            bias_grad(Cm, NO, gbL, tid);
            {
                f32x16 acc = mfma_cols(WL, 64, NO, 32 * ta + j, true, h, [&](int k) { return Cm.at(k, bs); }, zero16());
#pragma unroll
                for (int r = 0; r < 16; ++r) {            // through t = relu(x0 + h3): this is dL/dh3 and the residual part of dL/dx0
                    const int i = 32 * ta + rho(r, h);
                    acc[r] = (X0.at(i, bs) + H3.at(i, bs)) > 0.f ? acc[r] : 0.f;
                }
                __syncthreads();                          // `red` (inside GA) is no longer read
                store_tile(GA, 32 * ta, 64, bs, h, acc);
            }
            __syncthreads();
            // hidden layers, last to first.  (g_out, act_in) -> gW, gb, g_in masked by the ReLU of its pre-activation
            auto hidden_backward = [&](const float *W, float *gW, float *gb, const LMat &Gout, const LMat &PreIn, const LMat &Gin) {
                const f32x16 wg = mfma_samples(h, [&](int s) { return Gout.at(32 * ta + j, s); }, [&](int s) { return fmaxf(PreIn.at(32 * tb + j, s), 0.f); });
                scatter_add(gW, 64, 32 * ta, 64, 32 * tb + j, true, h, wg);
                bias_grad(Gout, 64, gb, tid);
                f32x16 acc = mfma_cols(W, 64, 64, 32 * ta + j, true, h, [&](int k) { return Gout.at(k, bs); }, zero16());
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = PreIn.at(32 * ta + rho(r, h), bs) > 0.f ? acc[r] : 0.f;
                __syncthreads();                          // Gin may alias a buffer other waves were still reading
                store_tile(Gin, 32 * ta, 64, bs, h, acc);
                __syncthreads();
            };
            hidden_backward(W5, gW5, gb5, GA, H2, GB);    // g_h3 (GA) -> g_h2 (GB)
            hidden_backward(W3, gW3, gb3, GB, H1, H3);    // g_h2 (GB) -> g_h1 (H3's storage: h3 is no longer needed)
            hidden_backward(W1, gW1, gb1, H3, X0, H2);    // g_h1      -> chain part of g_x0 (H2's storage)
            // total dL/dx0 = chain + residual, into GB
            for (int o = 16 * wave; o < 16 * wave + 16; ++o) GB.at(o, lane) = H2.at(o, lane) + GA.at(o, lane);
            __syncthreads();
            // fc_first: x0 = W0 (y (+) f) + b0
            bias_grad(GB, 64, gb0, tid);
            if (mob) {
                for (int o = 16 * wave; o < 16 * wave + 16; ++o) {          // gW0[:, 0:3]: this wave's 16 rows
                    const float go = GB.at(o, lane);
                    const float c0 = wave_sum(go * y.x), c1 = wave_sum(go * y.y), c2 = wave_sum(go * y.z);
                    if (lane == 0) {
                        atomicAdd(gW0 + (size_t)o * NI, c0);
                        atomicAdd(gW0 + (size_t)o * NI + 1, c1);
                        atomicAdd(gW0 + (size_t)o * NI + 2, c2);
                    }
                }
                float gy0 = 0.f, gy1 = 0.f, gy2 = 0.f;                      // the conditioner-input path of dL/dy (every wave, all rows)
                for (int o = 0; o < 64; ++o) {
                    const float go = GB.at(o, lane);
                    gy0 = fmaf(W0[(size_t)o * NI], go, gy0);
                    gy1 = fmaf(W0[(size_t)o * NI + 1], go, gy1);
                    gy2 = fmaf(W0[(size_t)o * NI + 2], go, gy2);
                }
                set_col(gRin, p1, get_col(gRin, p1) + v3f{gy0, gy1, gy2});
            }
            if (F) {
                // gW0[o][yo + c] += sum_s g[o][s] f[s][c]: tiles 2 (rows) x ceil(F/32) (columns), 4 waves
                const int ctiles = (F + 31) / 32;
                for (int t = wave; t < 2 * ctiles; t += TR_WAVES) {
                    const int rt = t & 1, ct = t >> 1;
                    const int c = 32 * ct + j;
                    const f32x16 acc = mfma_samples(h, [&](int s) { return GB.at(32 * rt + j, s); },
                                                    [&](int s) { return (c < F && blk * 64 + s < args.n) ? args.feature[(blk * 64 + s) * F + c] : 0.f; });
                    scatter_add(gW0 + yo, NI, 32 * rt, 64, c, c < F, h, acc);
                }
                if (args.g_feature) {
                    // g_f[c][s] = sum_o W0[o][yo + c] g[o][s]: tiles ceil(F/32) (feature rows) x 2 (sample columns)
                    for (int t = wave; t < 2 * ctiles; t += TR_WAVES) {
                        const int st = t & 1, ct = t >> 1;
                        const int c = 32 * ct + j, s = 32 * st + j;
                        const f32x16 acc = mfma_cols(W0 + yo, NI, 64, c, c < F, h, [&](int k) { return GB.at(k, s); }, zero16());
                        const long long smp = blk * 64 + s;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int cc = 32 * ct + rho(r, h);
                            if (cc < F && smp < args.n) args.g_feature[smp * F + cc] += acc[r];
                        }
                    }
                }
            }
            __syncthreads();
            gR = gRin;
        }
        if (valid && wave == 0) {
            float *o = args.g_rot_in + sample * 9;
            o[0] = gR.c0.x; o[1] = gR.c1.x; o[2] = gR.c2.x; o[3] = gR.c0.y; o[4] = gR.c1.y; o[5] = gR.c2.y; o[6] = gR.c0.z; o[7] = gR.c1.z; o[8] = gR.c2.z;
        }
    }
}

// d log|det M| / dM = M^-T, weighted by the batch sum of dL/dldj (Uncondition16Trans, flow/squeezetrans.py:33-38,57-66)
__global__ void affine_logdet_grad_kernel(const TrainArgs args) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= args.n_layers) return;
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
