// train_block16.h -- the backward sweep of train_kernels.h on 16-ROTATION workgroups (round 3).
//
// Same phases, same reverse-mode formulas (so3_grad.h), same plain / gradient blob layout as flow_train_backward_kernel; what changes is
// the shape of the work a workgroup owns.  The 64-rotation kernel keeps 150 KB of LDS per workgroup (one per CU), gives every wave a
// 32 x 32 x 64 product (32 dependent v_mfma_f32_32x32x2_f32, 64 cycles each) per matrix and fills 16 CUs at the reference's batch of 1024
// (settings/*.yml).  Here a workgroup of 8 waves owns 16 rotations:
//   * activations live in LDS as [feature][16 samples] (rows of 17 floats): 22 KB + 68 B per conditioner-output row, so the segment
//     count is bounded by LDS at K <= 512 instead of 64;
//   * every product is tiled 16 x 16 on v_mfma_f32_16x16x4_f32 (32 cycles): a wave owns 16 output rows of a 64-row product
//     (16 matrix instructions + the bias step), and the weight-gradient products sum over the 16 samples in 4 steps;
//   * WAVE SPECIALISATION: waves 0..3 ("chain") carry the dependent chain -- forward recompute, per-rotation layer math, data gradients
//     (W^T g) -- and waves 4..7 ("gradient") run every weight-gradient product (g act^T, bias sums) and its float atomics in the SAME
//     barrier window in which the chain runs the matching data-gradient product, plus half of the fc_last tiles of the forward
//     recompute.  Both groups execute the same sequence of workgroup barriers (gfx950 has one barrier per workgroup); the data a
//     gradient wave reads in a window is exactly what the chain leaves untouched until the window's closing barrier.  The fc_last weight
//     gradient (the largest: 4K x 64) is split: half beside the chain's WL^T product, half after the layer's last barrier, beside the
//     chain's constant layer and the next layer's x0 (its inputs -- dL/dC in LDS, t in registers -- stay valid until the next fc_last);
//   * the per-rotation layer math runs with 16 threads per rotation (chain waves): the K segments of a Moebius layer are split 16 ways
//     (4 lane groups x 4 waves), partial sums meet through two lane shuffles and one LDS exchange;
//   * bias gradients are one more matrix instruction against a column of ones instead of an LDS reduction loop.
// A batch of 1024 rotations is 64 workgroups, each with a quarter of the dependent matrix chain.  The price: every workgroup adds a
// full-size weight gradient with float atomics (executed at the memory side at ~1.3 TB/s chip-wide, MI355X_MICROARCH.md) and streams
// every weight matrix from L2, four times as much as with 64-rotation blocks -- rnf_api.hip picks the block size by batch
// (RNF_TRAIN_BLOCK=16|64 / rnf_set_train_block override).  Measured: profiles/README.md "Training".
//
// Lane (c, q) = (lane & 15, lane >> 4).  v_mfma_f32_16x16x4_f32: A[i = c][k = q], B[k = q][n = c], D register r = D[4q + r][c].
// K steps of a 64-deep product are ordered k = 16q + m (m = 0..15) so that a lane's A operands are 16 consecutive floats of a weight
// row (4 x 16-byte loads) and its B operands 16 LDS rows 16q + m at one sample: with rows of 17 floats the four lane groups read banks
// 16q + c -- conflict free.
#pragma once
#include "train_kernels.h"

namespace rnf {
namespace b16 {

constexpr int SB = 16;                  // rotations per workgroup
constexpr int LR = 17;                  // LDS row stride (floats)
constexpr int CHAIN = 4;                 // waves 0..3: the dependent chain (forward recompute, layer math, data gradients)
constexpr int WAVES = 8;                 // waves 4..7: the weight-gradient products and their atomics, beside the chain
constexpr int HEAD_FLOATS = 5 * 64 * LR + 4 * LR;        // X0, H1, H2, H3, GA, YL
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define RNF_MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

struct Mat {
    float *p;
    __device__ __forceinline__ float &at(int row, int s) const { return p[row * LR + s]; }
};
// one sample's column of a Mat as a conditioner-output row (so3_grad.h accessor)
struct SampleCol {
    Mat m;
    int c;
    __device__ __forceinline__ float get(int row) const { return m.at(row, c); }
    __device__ __forceinline__ void put(int row, float v) const { m.at(row, c) = v; }
};

__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }

// ---- A operands, loaded one phase ahead (train_kernels.h explains why no load is followed by a select) ----
// 16 rows of a row-major matrix with 64 columns: this lane holds W[row][16q .. 16q + 15] and the row's bias
struct RowsA { float4 w[4]; float bias; };
__device__ __forceinline__ RowsA load_rows(const float *__restrict__ W, const float *__restrict__ bias, int row, int n_rows, int q) {
    RowsA a;
    const int rc = row < n_rows ? row : n_rows - 1;
    a.bias = bias[rc];
    const float4 *src = reinterpret_cast<const float4 *>(W + (size_t)rc * 64 + 16 * q);
#pragma unroll
    for (int i = 0; i < 4; ++i) a.w[i] = src[i];
    return a;
}
// the 16 B operands of a 64-deep product: element m = M[16q + m][this lane's sample]; raw reads first, then the transform
template <class Post = Identity>
__device__ __forceinline__ void read_b(const Mat &M, int row0, int q, int c, float (&bv)[16], Post post = Post()) {
#pragma unroll
    for (int m = 0; m < 16; ++m) bv[m] = M.at(row0 + 16 * q + m, c);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < 16; ++m) bv[m] = post(bv[m]);
}
// D[4q + r][c] = bias[row] + sum_k W[row][k] b(k)
__device__ __forceinline__ f32x4 mfma_rows(const RowsA &a, int q, const float (&bv)[16]) {
    f32x4 acc = RNF_MFMA4(a.bias, q ? 0.f : 1.f, zero4());
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        acc = RNF_MFMA4(a.w[i].x, bv[4 * i], acc);
        acc = RNF_MFMA4(a.w[i].y, bv[4 * i + 1], acc);
        acc = RNF_MFMA4(a.w[i].z, bv[4 * i + 2], acc);
        acc = RNF_MFMA4(a.w[i].w, bv[4 * i + 3], acc);
    }
    return acc;
}

// 16 columns of a row-major matrix (the transposed product): this lane holds W[k0 + 16q + m][col], m < 16.  Rows >= n_rows repeat the
// last row (their B partners are zero); columns >= n_cols repeat the last column (their output rows are discarded).
struct ColsA { float a[16]; };
__device__ __forceinline__ ColsA load_cols(const float *__restrict__ W, int ldw, int k0, int n_rows, int col, int n_cols, int q) {
    ColsA cA;
    const int kb = k0 + 16 * q;
    const float *p = W + (size_t)(kb < n_rows ? kb : n_rows - 1) * ldw + (col < n_cols ? col : n_cols - 1);
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        cA.a[m] = *p;
        if (kb + m + 1 < n_rows) p += ldw;
    }
    return cA;
}
__device__ __forceinline__ f32x4 mfma_cols(const ColsA &cA, const float (&bv)[16], f32x4 acc) {
#pragma unroll
    for (int m = 0; m < 16; ++m) acc = RNF_MFMA4(cA.a[m], bv[m], acc);
    return acc;
}

// products over the block's 16 samples (weight gradients): step m covers sample s = 4m + q
__device__ __forceinline__ void read_s(const Mat &M, int row, int q, float (&v)[4]) {
#pragma unroll
    for (int m = 0; m < 4; ++m) v[m] = M.at(row, 4 * m + q);
}
__device__ __forceinline__ f32x4 mfma_s(const float (&a)[4], const float (&b)[4]) {
    f32x4 acc = zero4();
#pragma unroll
    for (int m = 0; m < 4; ++m) acc = RNF_MFMA4(a[m], b[m], acc);
    return acc;
}
// row sums over the samples (bias gradients): the product against a matrix of ones; every column of the tile holds them
__device__ __forceinline__ f32x4 mfma_s1(const float (&a)[4]) {
    f32x4 acc = zero4();
#pragma unroll
    for (int m = 0; m < 4; ++m) acc = RNF_MFMA4(a[m], 1.0f, acc);
    return acc;
}

// gW[o0 + 4q + r][col] += acc[r]
__device__ __forceinline__ void scatter_add(float *gW, int ldw, int o0, int n_out, int col, bool col_ok, int q, const f32x4 &acc) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int o = o0 + 4 * q + r;
        if (col_ok && o < n_out) atomicAdd(gW + (size_t)o * ldw + col, acc[r]);
    }
}
// gb[o0 + 4q + r] += rowsum[r] (lanes c == 0)
__device__ __forceinline__ void bias_add(float *gb, int o0, int n_out, int q, int c, const f32x4 &acc) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int o = o0 + 4 * q + r;
        if (c == 0 && o < n_out) atomicAdd(gb + o, acc[r]);
    }
}
__device__ __forceinline__ void store_tile(const Mat &M, int o0, int n_out, int q, int c, const f32x4 &acc) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int o = o0 + 4 * q + r;
        if (o < n_out) M.at(o, c) = acc[r];
    }
}

// Sum of per-thread partial values over the 16 threads that share a rotation (4 lane groups x 4 chain waves).  `red` needs CHAIN * N * 16
// floats that no wave is still reading; ends with every thread holding the totals.
template <int N>
__device__ __forceinline__ void block_sum(float (&v)[N], float *red, int wave, int q, int c) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        v[i] += __shfl_xor(v[i], 16, 64);
        v[i] += __shfl_xor(v[i], 32, 64);
    }
    if (q == 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) red[(wave * N + i) * 16 + c] = v[i];
    }
    lds_barrier();
#pragma unroll
    for (int i = 0; i < N; ++i) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < CHAIN; ++w) t += red[(w * N + i) * 16 + c];
        v[i] = t;
    }
}

// batch sums of up to 16 per-rotation values (parameter gradients of the constant layers), wave 0 only: lanes q == 0 park their values
// in `S` [16][LR]; lane 4v + q2 adds four rotations of entry v.  Same wave: its LDS operations complete in order.
__device__ __forceinline__ float batch_sum16(const Mat &S, const float (&val)[16], int lane) {
    const int c = lane & 15, q = lane >> 4;
    if (q == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) S.at(i, c) = val[i];
    }
    const int v = lane >> 2, q2 = lane & 3;
    float tot = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) tot += S.at(v, 4 * q2 + i);
    return quad_sum(tot);                                 // valid in every lane of the quad; entry v = lane >> 2
}

// RARE: the instantiation that also carries the side / Gram-Schmidt / conditional 3x3 layer kinds.  Inlined beside the common kinds they
// cost the common path 49 spilled registers (C2, batch 1024, graphed iteration: 1.126 ms against 0.993 ms without them), so the launcher
// picks the instantiation from the layer table.
// Saved activations of one conditioner layer and one 16-rotation block: rows 0..255 = X0 (raw), H1, H2 (after ReLU), T = relu(x0 + h3), then
// the conditioner outputs C, 16 floats per row -- the LDS matrices without their padding column.  Written by the training forward
// (flow_train_forward16_kernel), read back by the backward sweep instead of recomputing the conditioner; all 512 threads move 16 bytes a time.
constexpr int ACT_HEAD_ROWS = 256;
// rows [row_begin, row_end) by `nthreads` threads (t = 0 .. nthreads - 1).  In the backward sweep the GRADIENT waves do this: they may
// still be reading the previous layer's dL/dC (their deferred fc_last tiles) and overwrite it only behind that, in their own program order,
// while the chain waves, busy with a constant layer, touch none of these buffers.
__device__ __forceinline__ void acts_to_lds(const float *__restrict__ slot, int row_begin, int row_end, float *lds, float *cm, int t, int nthreads) {
    const float4 *src = reinterpret_cast<const float4 *>(slot);
    for (int f = 4 * row_begin + t; f < row_end * 4; f += nthreads) {
        const float4 v = src[f];
        const int row = f >> 2, col = 4 * (f & 3);
        float *dst = row < ACT_HEAD_ROWS ? lds + row * LR + col : cm + (row - ACT_HEAD_ROWS) * LR + col;
        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
    }
}
__device__ __forceinline__ void lds_to_acts(float *__restrict__ slot, int rows, const float *lds, const float *cm, int tid) {
    float4 *dst = reinterpret_cast<float4 *>(slot);
    for (int f = tid; f < rows * 4; f += WAVES * 64) {
        const int row = f >> 2, col = 4 * (f & 3);
        const float *src = row < ACT_HEAD_ROWS ? lds + row * LR + col : cm + (row - ACT_HEAD_ROWS) * LR + col;
        dst[f] = float4{src[0], src[1], src[2], src[3]};
    }
}

template <bool HAS_FEATURE, bool RARE>
__global__ __launch_bounds__(WAVES * 64, 2) void flow_train_backward16_kernel(const TrainArgs args) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
#ifdef RNF_STAMPS
    unsigned long long tst_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tst_t = clock64();
#endif
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q = lane >> 4;
    const bool gw = wave >= CHAIN;                        // gradient wave (wave-uniform): no per-rotation state, weight-gradient products only
    const int w4 = wave & 3;                              // row tile of a 64-row product (chain wave w4 and gradient wave w4 + 4 share it)
    const int grp = 4 * w4 + q;                           // which sixteenth of a rotation's segments this (chain) thread owns
    const int K = args.K, F = HAS_FEATURE ? args.F : 0;
    const bool want_w = args.grads != nullptr;            // wave-uniform
    // LDS: X0, H1, H2, H3 (pre-activations, later reused for gradients), GA (gradient / reduction scratch), YL (the conditioning column y
    // of the 16 rotations, [3][16]), C (conditioner output, then its gradient)
    const Mat X0{lds}, H1{lds + 64 * LR}, H2{lds + 2 * 64 * LR}, H3{lds + 3 * 64 * LR}, GA{lds + 4 * 64 * LR}, YL{lds + 5 * 64 * LR},
        Cm{lds + HEAD_FLOATS};
    float *red = GA.p;                                    // reduction scratch while GA is not in use

    const long long nblocks = (args.n + SB - 1) / SB;
    for (long long blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const long long sample = blk * SB + c;            // the rotation this thread carries (16 threads per rotation)
        const bool valid = sample < args.n;
        const bool writer = wave == 0 && q == 0;          // the one thread per rotation that stores per-rotation results
        int mlp_idx = args.mlp_base;                      // saved activations: slot index of the conditioner layer being processed
        if (args.acts) {
            for (int l = 0; l < args.n_layers; ++l) {
                const int kd = args.layers[l].x & 15;
                mlp_idx += (kd == RNF_KIND_MOBIUS || kd == RNF_KIND_COND16) ? 1 : 0;
            }
        }
        Rot gR;
        gR.c0 = v3f{0.f, 0.f, 0.f}; gR.c1 = gR.c0; gR.c2 = gR.c0;
        float g_ldj = 0.f;
        if (valid && !gw) {
            g_ldj = args.g_ldj ? args.g_ldj[sample] : 0.f;
            if (args.g_rot_out) {
                const float *g = args.g_rot_out + sample * 9;
                gR.c0 = v3f{g[0], g[3], g[6]}; gR.c1 = v3f{g[1], g[4], g[7]}; gR.c2 = v3f{g[2], g[5], g[8]};
            }
        }
        // Two loops over the layers, one per wave group (the groups share nothing but LDS and the barrier sequence; one loop with a branch
        // inside would keep the gradient waves' prefetch registers alive across the chain's code).
        if (gw) {
            for (int pos = args.n_layers - 1; pos >= 0; --pos) {
                const int2 d = args.layers[pos];
                const int kind = d.x & 15;
                const float *P = args.plain + d.y;
                float *Gp = args.grads + d.y;
                // ---- gradient waves: the barriers of every layer, the fc_last tiles 4..7 (mod 8) of the forward recompute, and every
                // weight-gradient product, each in the window in which the chain waves run the matching data-gradient product ----
                if (RARE && kind == RNF_KIND_GS36 && want_w) { lds_barrier(); lds_barrier(); lds_barrier(); lds_barrier(); }
                if (!(kind == RNF_KIND_MOBIUS || kind == RNF_KIND_COND16 || kind == RNF_KIND_MLP_ONLY || (RARE && (kind == RNF_KIND_COND36 || kind_is_cond9(kind))))) continue;
                const bool mob = kind == RNF_KIND_MOBIUS;
                const int yo = mob ? 3 : 0, NI = yo + F, NO = mob ? 4 * K : (kind == RNF_KIND_MLP_ONLY ? ((d.x >> 16) & 255) : (kind_is_cond9(kind) ? 9 : (kind == RNF_KIND_COND36 ? 36 : 16)));
                const float *WL = P + 64 * NI + 64 + 3 * (4096 + 64), *bL = WL + (size_t)NO * 64;
                float *gW0 = Gp, *gb0 = gW0 + 64 * NI, *gW1 = gb0 + 64, *gb1 = gW1 + 4096, *gW3 = gb1 + 64, *gb3 = gW3 + 4096, *gW5 = gb3 + 64,
                      *gb5 = gW5 + 4096, *gWL = gb5 + 64, *gbL = gWL + (size_t)NO * 64;
                const int ntiles = (NO + 15) / 16, row0 = 16 * w4;
                const Mat &T = H3;
                float tb[4][4];                           // t over the samples, B side of gWL: read now, used after the layer math
                if (args.acts) {                          // the forward's activations come back from memory: one fill, one barrier
                    --mlp_idx;
                    acts_to_lds(args.acts + ((size_t)mlp_idx * nblocks + blk) * args.act_rows * 16, 0, ACT_HEAD_ROWS + NO, lds, Cm.p,
                                tid - CHAIN * 64, CHAIN * 64);
                    lds_barrier();
                    if (want_w) {
#pragma unroll
                        for (int ct = 0; ct < 4; ++ct) read_s(T, 16 * ct + c, q, tb[ct]);
                    }
                } else {
                    RowsA wnext = load_rows(WL, bL, 16 * wave + c, NO, q);
                    if (mob) lds_barrier();               // YL
                    lds_barrier();                        // x0
                    lds_barrier(); lds_barrier(); lds_barrier();      // h1, h2, t
                    {
                        float tv[16];
                        read_b(T, 0, q, c, tv);
                        for (int rt = wave; rt < ntiles; rt += WAVES) {
                            const RowsA wl = wnext;
                            if (rt + WAVES < ntiles) wnext = load_rows(WL, bL, 16 * (rt + WAVES) + c, NO, q);
                            store_tile(Cm, 16 * rt, NO, q, c, mfma_rows(wl, q, tv));
                        }
                    }
                    if (want_w) {
#pragma unroll
                        for (int ct = 0; ct < 4; ++ct) read_s(T, 16 * ct + c, q, tb[ct]);
                    }
                    lds_barrier();                        // C
                }
                lds_barrier(); lds_barrier();             // the two exchanges of the layer math; dL/dC is in place
                // fc_last: gWL += g_c t^T, gbL += rowsum(g_c).  dL/dC stays in place until the NEXT conditioner layer's fc_last (behind four
                // barriers these waves join), and t is in registers: half of this wave's row tiles now, beside the chain's WL^T product; the
                // other half after the layer's last barrier, beside the chain's constant layer and the next x0 (where these waves would idle).
                // (What the chain writes into C before those four barriers is the pad rows [NO', pad64(NO')) of the NEXT conditioner layer:
                // rows below 64 when NO' < 64, rows that are padding for this layer too otherwise -- every Moebius layer of a flow has the
                // same K -- while a deferred tile is never this wave's first one, i.e. starts at row 64 or above.)
                auto last_wgrad = [&](int rt_begin, int rt_end) {
                    for (int rt = rt_begin; rt < rt_end; rt += CHAIN) {
                        float av[4];
                        read_s(Cm, 16 * rt + c, q, av);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int ct = 0; ct < 4; ++ct) scatter_add(gWL, 64, 16 * rt, NO, 16 * ct + c, true, q, mfma_s(av, tb[ct]));
                        bias_add(gbL, 16 * rt, NO, q, c, mfma_s1(av));
                    }
                };
                const int my_tiles = ntiles > w4 ? (ntiles - w4 + CHAIN - 1) / CHAIN : 0;      // row tiles w4, w4 + 4, ...
                const int rt_split = w4 + CHAIN * ((my_tiles + 1) / 2);
                if (want_w) last_wgrad(w4, rt_split < ntiles ? rt_split : ntiles);
                lds_barrier(); lds_barrier();             // g_t stored
                auto hidden_wgrad = [&](float *gW, float *gb, const Mat &Gout, const Mat &PreIn) {
                    if (want_w) {
                        float av[4], bv[4][4];
                        read_s(Gout, row0 + c, q, av);
#pragma unroll
                        for (int ct = 0; ct < 4; ++ct) read_s(PreIn, 16 * ct + c, q, bv[ct]);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int ct = 0; ct < 4; ++ct) {
#pragma unroll
                            for (int m = 0; m < 4; ++m) bv[ct][m] = Relu()(bv[ct][m]);
                            scatter_add(gW, 64, row0, 64, 16 * ct + c, true, q, mfma_s(av, bv[ct]));
                        }
                        bias_add(gb, row0, 64, q, c, mfma_s1(av));
                    }
                    lds_barrier(); lds_barrier();         // the chain's store of the next gradient
                };
                hidden_wgrad(gW5, gb5, GA, H2);
                hidden_wgrad(gW3, gb3, H3, H1);
                hidden_wgrad(gW1, gb1, H2, X0);
                lds_barrier();                            // total dL/dx0 (H3's storage)
                if (want_w) {                             // fc_first: gb0, gW0[:, 0:3] += g y^T, gW0[:, yo:] += g f^T
                    const Mat GB = H3;
                    float ga[4];
                    read_s(GB, row0 + c, q, ga);
                    bias_add(gb0, row0, 64, q, c, mfma_s1(ga));
                    if (mob) {
                        float yb[4];
#pragma unroll
                        for (int m = 0; m < 4; ++m) yb[m] = YL.at(c < 3 ? c : 0, 4 * m + q);
                        scatter_add(gW0, NI, row0, 64, c, c < 3, q, mfma_s(ga, yb));
                    }
                    if (HAS_FEATURE) {                    // this wave's row tile x ceil(F/16) column tiles, four tiles' loads in flight
                        const int ctiles = (F + 15) / 16;
                        for (int ct0 = 0; ct0 < ctiles; ct0 += 4) {
                            float fb[4][4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const int cc = 16 * (ct0 + u) + c;
#pragma unroll
                                for (int m = 0; m < 4; ++m) {
                                    const long long smp = blk * SB + 4 * m + q;
                                    fb[u][m] = args.feature[(smp < args.n ? smp : args.n - 1) * F + (cc < F ? cc : F - 1)];
                                }
                            }
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int u = 0; u < 4; ++u) {     // (rows of padding rotations: their g is exactly 0)
                                const int cc = 16 * (ct0 + u) + c;
                                scatter_add(gW0 + yo, NI, row0, 64, cc, cc < F, q, mfma_s(ga, fb[u]));
                            }
                        }
                    }
                }
                lds_barrier();
                if (want_w) last_wgrad(rt_split, ntiles);
                continue;
            }
        } else {
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
            if (RARE && kind_is_side(kind)) {                     // per-sample matrix from the caller; its gradient goes back to the caller
                const int slot = (d.x >> 16) & 255;
                const float *m = args.side + ((size_t)slot * args.n + (valid ? sample : 0)) * 16;
                float gM[16];
                Rot gRin;
                rare_side(kind, args.dir, m, &Rin, &gR, g_ldj, gM, &gRin);
                if (writer && valid && args.side_grad) {
                    float *o = args.side_grad + ((size_t)slot * args.n + sample) * 16;
#pragma unroll
                    for (int i = 0; i < 16; ++i) o[i] = gM[i];
                }
                gR = gRin;
                RNF_TSTAMP(8)
                continue;
            }
            if (kind == RNF_KIND_AFFINE16) {              // every thread carries the chain; wave 0 adds the parameter gradient
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
                if (wave == 0 && want_w) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) gM[i] = valid ? gM[i] : 0.f;
                    const float tot = batch_sum16(GA, gM, lane);
                    if ((lane & 3) == 0) atomicAdd(Gp + (lane >> 2), tot);
                    const float gl = wave_sum(valid && !orth && q == 0 ? g_ldj : 0.f);   // log|det M^-1| = -log|det M| on the inverse pass
                    if (lane == 0) atomicAdd(args.g_ldj_sum + pos, args.dir ? -gl : gl);
                }
                gR = gRin;
                RNF_TSTAMP(8)
                continue;
            }
            if (RARE && kind == RNF_KIND_GS9) {                   // Uncondition9Trans / 9TransLU: M [9] (+3 pad) in the plain blob
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
                    float val[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) val[i] = (valid && i < 9) ? gM[i < 9 ? i : 8] : 0.f;
                    const float tot = batch_sum16(GA, val, lane);
                    if ((lane & 3) == 0 && (lane >> 2) < 9) atomicAdd(Gp + (lane >> 2), tot);
                }
                gR = gRin;
                RNF_TSTAMP(8)
                continue;
            }
            if (RARE && kind == RNF_KIND_GS36) {                  // Uncondition36Trans: M [36] in the plain blob (squeezetrans.py:350-361)
                float gM[36];
                Rot gRin;
                rare_gs36(P, 1, 0.f, args.dir, &Rin, &gR, g_ldj, gM, &gRin);     // the inverse pass applies M^-1 (squeezetrans.py:359-361)
                if (want_w) {                             // batch sums through LDS, 16 entries per round (wave 0), three rounds
#pragma unroll
                    for (int base = 0; base < 48; base += 16) {
                        lds_barrier();
                        if (wave == 0) {
                            float val[16];
#pragma unroll
                            for (int i = 0; i < 16; ++i) val[i] = (valid && base + i < 36) ? gM[base + i < 36 ? base + i : 35] : 0.f;
                            const float tot = batch_sum16(GA, val, lane);
                            if ((lane & 3) == 0 && base + (lane >> 2) < 36) atomicAdd(Gp + base + (lane >> 2), tot);
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
            const int ntiles = (NO + 15) / 16;            // 16-row tiles of fc_last
            const int row0 = 16 * w4;                     // this wave's rows of a 64-row product

            // ================= the conditioner's activations: read back what the forward saved, or recompute them =================
            if (mob && writer) { YL.at(0, c) = y.x; YL.at(1, c) = y.y; YL.at(2, c) = y.z; }
            for (int o = NO + grp; o < ((NO + 63) & ~63); o += 16) Cm.at(o, c) = 0.f;      // pad rows of C: B side of the WL^T slabs
            const Mat &T = H3;
            if (args.acts) {
                --mlp_idx;                                // (the gradient waves bring the slot in)
            } else {
                RowsA wnext = load_rows(W1, b1, row0 + c, 64, q);
                // x0 = b0 + W0[:, yo:] f (K dimension = F in chunks of 64) + W0[:, :3] y (one K = 4 step)
                {
                    f32x4 acc = RNF_MFMA4(b0[row0 + c], q ? 0.f : 1.f, zero4());
                    const float *wrow = W0 + (size_t)(row0 + c) * NI;
                    if (HAS_FEATURE) {
                        // chunk ch covers k = 64 ch + 16 q + u: each lane's 16 operands are one 64-byte run of its weight row / feature row
                        const float *frow = args.feature + (valid ? sample : 0) * F;
                        const int nch = (F + 63) / 64;
                        for (int ch = 0; ch < nch; ++ch) {
                            float av[16], bv[16];
                            const int k0 = 64 * ch + 16 * q;
                            if (k0 + 16 <= F) {               // 4 x 16-byte loads per operand (4-byte aligned)
                                const Float4U *pa = reinterpret_cast<const Float4U *>(wrow + yo + k0);
                                const Float4U *pb = reinterpret_cast<const Float4U *>(frow + k0);
    #pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    const Float4U va = pa[i], vb = pb[i];
                                    av[4 * i] = va.x; av[4 * i + 1] = va.y; av[4 * i + 2] = va.z; av[4 * i + 3] = va.w;
                                    bv[4 * i] = vb.x; bv[4 * i + 1] = vb.y; bv[4 * i + 2] = vb.z; bv[4 * i + 3] = vb.w;
                                }
                            } else {
    #pragma unroll
                                for (int u = 0; u < 16; ++u) {
                                    const int k = k0 + u, kc = k < F ? k : F - 1;
                                    av[u] = wrow[yo + kc];
                                    bv[u] = frow[kc];
                                }
                            }
                            __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
                            for (int u = 0; u < 16; ++u) acc = RNF_MFMA4(k0 + u < F ? av[u] : 0.f, valid ? bv[u] : 0.f, acc);
                        }
                    }
                    if (mob) {
                        const float a = wrow[q < 3 ? q : 2];
                        lds_barrier();                        // YL
                        acc = RNF_MFMA4(q < 3 ? a : 0.f, YL.at(q < 3 ? q : 0, c), acc);
                    }
                    store_tile(X0, row0, 64, q, c, acc);
                }
                lds_barrier();
                RNF_TSTAMP(0)
                // H1, H2 hold relu(h1), relu(h2); H3 holds t = relu(x0 + h3) (the ReLU masks only need the sign)
                {
                    const RowsA w1 = wnext;
                    wnext = load_rows(W3, b3, row0 + c, 64, q);
                    float bv[16];
                    read_b(X0, 0, q, c, bv, Relu());
                    f32x4 acc = mfma_rows(w1, q, bv);
    #pragma unroll
                    for (int r = 0; r < 4; ++r) acc[r] = Relu()(acc[r]);
                    store_tile(H1, row0, 64, q, c, acc);
                }
                lds_barrier();
                {
                    const RowsA w3 = wnext;
                    wnext = load_rows(W5, b5, row0 + c, 64, q);
                    float bv[16];
                    read_b(H1, 0, q, c, bv);
                    f32x4 acc = mfma_rows(w3, q, bv);
    #pragma unroll
                    for (int r = 0; r < 4; ++r) acc[r] = Relu()(acc[r]);
                    store_tile(H2, row0, 64, q, c, acc);
                }
                lds_barrier();
                {
                    const RowsA w5 = wnext;
                    wnext = load_rows(WL, bL, row0 + c, NO, q);
                    float bv[16], xv[4];
                    read_b(H2, 0, q, c, bv);
                    f32x4 acc = mfma_rows(w5, q, bv);
    #pragma unroll
                    for (int r = 0; r < 4; ++r) xv[r] = X0.at(row0 + 4 * q + r, c);
                    __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
                    for (int r = 0; r < 4; ++r) acc[r] = Relu()(acc[r] + xv[r]);
                    store_tile(H3, row0, 64, q, c, acc);
                }
                lds_barrier();
                RNF_TSTAMP(1)
                {                                             // fc_last: C = WL t + bL; the B operands (t) are read once for all row tiles
                    float tv[16];
                    read_b(T, 0, q, c, tv);
                    for (int rt = wave; rt < ntiles; rt += WAVES) {
                        const RowsA wl = wnext;
                        if (rt + WAVES < ntiles) wnext = load_rows(WL, bL, 16 * (rt + WAVES) + c, NO, q);
                        store_tile(Cm, 16 * rt, NO, q, c, mfma_rows(wl, q, tv));
                    }
                }
            }
            ColsA cnext = load_cols(WL, 64, 0, NO, row0 + c, 64, q);         // first slab of WL^T, needed after the layer math
            lds_barrier();
            RNF_TSTAMP(2)
            // ================= layer math: forward sums + backward; dL/dC overwrites C =================
            Rot gRin;
            const SampleCol crow{Cm, c};
            const int k0s = grp * K / 16, k1s = (grp + 1) * K / 16;           // this thread's segments
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
                float sm[4] = {0.f, 0.f, 0.f, 0.f};
                mobius_inv_segments_sums(sv, crow, K, k0s, k1s, sm[0], sm[1], sm[2], sm[3]);
                block_sum<4>(sm, red, wave, q, c);
                MobiusGrad mg;
                mobius_inv_backward_head(sv, sm[0], sm[1], sm[2], sm[3], gR, g_ldj, mg);
                v3f pr = v3f{0.f, 0.f, 0.f}, pv = pr;
                mobius_segments_backward_range_at(sv.b.f, sv.cs, sv.sn, sv.theta, crow, K, k0s, k1s, mg, crow, pr, pv);
                float pp[6] = {pr.x, pr.y, pr.z, pv.x, pv.y, pv.z};
                block_sum<6>(pp, red + CHAIN * 4 * 16, wave, q, c);
                mg.g_r = mg.g_r + v3f{pp[0], pp[1], pp[2]};
                mg.g_v = mg.g_v + v3f{pp[3], pp[4], pp[5]};
                mobius_backward_tail(sv.b, mg, gRin);
            } else if (mob) {
                MobiusSaved sv;
                mobius_frame(Rin, perm_row, sv);
                float sm[3] = {0.f, 0.f, 0.f};
                mobius_segments_sums(sv, crow, K, k0s, k1s, sm[0], sm[1], sm[2]);
                block_sum<3>(sm, red, wave, q, c);
                mobius_combine(sv, sm[0], sm[1], sm[2]);
                MobiusGrad mg;
                mobius_backward_head(sv, gR, g_ldj, mg);
                v3f pr = v3f{0.f, 0.f, 0.f}, pv = pr;
                mobius_segments_backward_range(sv, crow, K, k0s, k1s, mg, crow, pr, pv);
                float pp[6] = {pr.x, pr.y, pr.z, pv.x, pv.y, pv.z};
                block_sum<6>(pp, red + CHAIN * 4 * 16, wave, q, c);
                mg.g_r = mg.g_r + v3f{pp[0], pp[1], pp[2]};
                mg.g_v = mg.g_v + v3f{pp[3], pp[4], pp[5]};
                mobius_backward_tail(sv, mg, gRin);
            } else if (kind == RNF_KIND_MLP_ONLY) {
                // the conditioner on its own (networks of the side layers): dL/d(outputs) comes from the caller
                gRin = gR;
                lds_barrier();                          // every wave is past the fc_last stores into C
                for (int i = grp; i < NO; i += 16) Cm.at(i, c) = valid ? args.g_out_ext[sample * NO + i] : 0.f;
                lds_barrier();
            } else if (RARE && kind == RNF_KIND_COND36) {
                // Condition36Trans (squeezetrans.py:334-347): M = I + reshape(net(f), 6, 6) per sample; the inverse pass applies M^-1
                float gM[36];
                rare_gs36(Cm.p + c, LR, 1.f, args.dir, &Rin, &gR, g_ldj, gM, &gRin);
                lds_barrier();                          // every thread has read C
                if (writer) {
#pragma unroll
                    for (int i = 0; i < 36; ++i) Cm.at(i, c) = gM[i];
                }
                lds_barrier();
            } else if (RARE && kind_is_cond9(kind)) {
                // Condition9Trans / 9RotL / 9RotR / 9RotRSmith (squeezetrans.py:234-247, rottrans.py:108-181): M = I + reshape(net(f), 3, 3)
                float gM[9];
                rare_cond9(kind, args.dir, Cm.p + c, LR, &Rin, &gR, g_ldj, gM, &gRin);
                lds_barrier();                          // every thread has read C
                if (writer) {
#pragma unroll
                    for (int i = 0; i < 9; ++i) Cm.at(i, c) = gM[i];
                }
                lds_barrier();
            } else {
                // Condition16Trans (flow/squeezetrans.py:41-50): M = I + reshape(net(f), 4, 4), ldj = log|det M| - 2 log|M q|^2
                float M[16], Mi[16], gM[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) { M[i] = Cm.at(i, c) + ((i % 5) == 0 ? 1.f : 0.f); gM[i] = 0.f; }
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
                lds_barrier();                          // every thread has read C
                if (writer) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) Cm.at(4 * i + jj, c) = gM[4 * i + jj] + gl * Mi[4 * jj + i];
                }
                lds_barrier();
            }
            // (padding rotations carry gR = 0 and g_ldj = 0, so their dL/dC and everything derived from it is exactly 0)

            RNF_TSTAMP(3)
            // ================= conditioner backward =================
            // fc_last: g_t = WL^T g_c (the gradient waves add gWL, gbL meanwhile)
            RNF_TSTAMP(4)
            {
                f32x4 acc = zero4();
                for (int k0 = 0; k0 < NO; k0 += 64) {     // 64-row slabs of WL^T; the next slab (or W5^T for the first hidden step) loads ahead
                    const ColsA cur = cnext;
                    if (k0 + 64 < NO) cnext = load_cols(WL, 64, k0 + 64, NO, row0 + c, 64, q);
                    else cnext = load_cols(W5, 64, 0, 64, row0 + c, 64, q);
                    float bv[16];
                    read_b(Cm, k0, q, c, bv);
                    acc = mfma_cols(cur, bv, acc);
                }
                {                                         // through t = relu(x0 + h3): this is dL/dh3 and the residual part of dL/dx0
                    float tv[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) tv[r] = T.at(row0 + 4 * q + r, c);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[r] = tv[r] > 0.f ? acc[r] : 0.f;
                }
                lds_barrier();                            // `red` (inside GA) is no longer read
                store_tile(GA, row0, 64, q, c, acc);
            }
            lds_barrier();
            RNF_TSTAMP(5)
            // hidden layers, last to first.  (g_out, act_in) -> gW, gb, g_in masked by the ReLU of its pre-activation.
            // cnext holds the transposed weights of this step; Wnext: the matrix of the FOLLOWING step (loaded ahead), or nullptr
            auto hidden_backward = [&](const Mat &Gout, const Mat &PreIn, const Mat &Gin, const float *Wnext) {
                const ColsA cur = cnext;
                if (Wnext) cnext = load_cols(Wnext, 64, 0, 64, row0 + c, 64, q);
                float gv[16];
                read_b(Gout, 0, q, c, gv);
                f32x4 acc = mfma_cols(cur, gv, zero4());
                {
                    float pv[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) pv[r] = PreIn.at(row0 + 4 * q + r, c);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[r] = pv[r] > 0.f ? acc[r] : 0.f;
                }
                lds_barrier();                            // Gin may alias a buffer other waves were still reading
                store_tile(Gin, row0, 64, q, c, acc);
                lds_barrier();
            };
            // (each activation buffer is free once its ReLU mask has been applied, and takes the next gradient)
            hidden_backward(GA, H2, H3, W3);    // g_h3 (GA) -> g_h2 (H3's storage)
            hidden_backward(H3, H1, H2, W1);    // g_h2      -> g_h1 (H2's storage)
            hidden_backward(H2, X0, H1, nullptr);   // g_h1  -> chain part of g_x0 (H1's storage)
            RNF_TSTAMP(6)
            const Mat GB = H3;                            // total dL/dx0 = chain + residual
            {
                float c1[4], c2[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { c1[i] = H1.at(row0 + 4 * q + i, c); c2[i] = GA.at(row0 + 4 * q + i, c); }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 4; ++i) GB.at(row0 + 4 * q + i, c) = c1[i] + c2[i];
            }
            lds_barrier();
            // fc_first: x0 = W0 (y (+) f) + b0
            float gv[16];                                 // dL/dx0[16q + m][this rotation]: B side of W0^T g, A side of g^T W0
            if (mob || HAS_FEATURE) read_b(GB, 0, q, c, gv);
            if (mob) {
                // the conditioner-input path of dL/dy: D[n][rotation] = sum_o W0[o][n] g[o][rotation], n < 3 (rows 3..15 of the tile repeat
                // column 2 and are discarded); every wave computes it, rows 0..2 land in lanes q == 0
                float wa[16];
                const float *wp = W0 + (size_t)(16 * q) * NI + (c < 3 ? c : 2);
#pragma unroll
                for (int m = 0; m < 16; ++m) wa[m] = wp[(size_t)m * NI];
                __builtin_amdgcn_sched_barrier(0);
                f32x4 acc = zero4();
#pragma unroll
                for (int m = 0; m < 16; ++m) acc = RNF_MFMA4(wa[m], gv[m], acc);
                const float gy0 = __shfl(acc[0], c, 64), gy1 = __shfl(acc[1], c, 64), gy2 = __shfl(acc[2], c, 64);
                set_col(gRin, p1, get_col(gRin, p1) + v3f{gy0, gy1, gy2});
            }
            if (HAS_FEATURE) {
                const int ctiles = (F + 15) / 16;
                if (args.g_feature) {
                    // g_f[s][cc] = sum_o g[o][s] W0[o][yo + cc]: D[rotation][feature column]; A = g^T (gv), B = rows of W0 (coalesced over the
                    // feature index); the tile lands feature-contiguous for the read-modify-write
                    for (int ct = w4; ct < ctiles; ct += CHAIN) {
                        const int cc = 16 * ct + c, ccl = cc < F ? cc : F - 1;
                        float bw[16];
                        const float *wp = W0 + (size_t)(16 * q) * NI + yo + ccl;
#pragma unroll
                        for (int m = 0; m < 16; ++m) bw[m] = wp[(size_t)m * NI];
                        __builtin_amdgcn_sched_barrier(0);
                        f32x4 acc = zero4();
#pragma unroll
                        for (int m = 0; m < 16; ++m) acc = RNF_MFMA4(gv[m], bw[m], acc);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const long long smp = blk * SB + 4 * q + r;
                            if (cc < F && smp < args.n) args.g_feature[smp * F + cc] += acc[r];
                        }
                    }
                }
            }
            lds_barrier();
            gR = gRin;
            RNF_TSTAMP(7)
        }
        }
        lds_barrier();                                    // g_rot_in may BE g_rot_out (chunked sweeps, rnf_api.hip): every wave has read its copy
        if (valid && writer && args.g_rot_in) {
            float *o = args.g_rot_in + sample * 9;
            o[0] = gR.c0.x; o[1] = gR.c1.x; o[2] = gR.c2.x; o[3] = gR.c0.y; o[4] = gR.c1.y; o[5] = gR.c2.y; o[6] = gR.c0.z; o[7] = gR.c1.z; o[8] = gR.c2.z;
        }
    }
#ifdef RNF_STAMPS
    RNF_TSTAMP(9)
    if (args.stamps && tid == 0) for (int i_ = 0; i_ < 10; ++i_) atomicAdd(args.stamps + i_, tst_acc[i_]);
#endif
}


// ---------------------------------------------------------------------------------------------------------------------------------
// The training FORWARD on the same 16-rotation workgroups: Flow.forward from the PLAIN parameter blob (no packed blob, no pack launch),
// exact fp32 on v_mfma_f32_16x16x4_f32 -- arithmetic identical to the forward recompute of the backward sweep above, product by product.
// For the reference's batches (128-1024 rotations) the fused stack kernel of flow_kernels.h is a latency chain of one wave per 32 rotations
// (0.27 ms for 24 layer pairs at batch 1024); here 4 waves share the conditioner of 16 rotations and the K segments are split 16 ways.
// Layer kinds: Moebius, Uncondition16Trans / UnconditionRot, Condition16Trans (forward direction); the launcher keeps every other flow on
// the stack kernel.  Saves the rotation entering every layer (args.states), as rnf_flow_forward_train does.
struct FwdArgs {
    const float *rot;         // [n][9]
    const float *feature;     // [n][F] (unpadded) or nullptr
    const float *plain;       // plain parameter blob
    float *rot_out;           // [n][9]
    float *ldj_out;           // [n]
    float *states;            // [n_layers][n][9]
    float *acts;              // conditioner activations for the backward sweep (lds_to_acts), or nullptr: the sweep recomputes them
    long long n;
    int n_layers, K, F, act_rows;
    int2 layers[TR_MAX_LAYERS];   // x = kind | perm_row << 4 | orthogonal << 8, y = plain offset (the backward's table)
};

template <bool HAS_FEATURE>
__global__ __launch_bounds__(WAVES * 64, 2) void flow_train_forward16_kernel(const FwdArgs args) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, q = lane >> 4;
    const bool helper = wave >= CHAIN;                    // waves 4..7: half of the fc_last tiles, nothing else
    const int w4 = wave & 3, grp = 4 * w4 + q, row0 = 16 * w4;
    const int K = args.K, F = HAS_FEATURE ? args.F : 0;
    const Mat X0{lds}, H1{lds + 64 * LR}, H2{lds + 2 * 64 * LR}, H3{lds + 3 * 64 * LR}, GA{lds + 4 * 64 * LR}, YL{lds + 5 * 64 * LR},
        Cm{lds + HEAD_FLOATS};
    float *red = GA.p;
    const long long nblocks = (args.n + SB - 1) / SB;
    for (long long blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const long long sample = blk * SB + c;
        const bool valid = sample < args.n;
        const bool writer = wave == 0 && q == 0;
        Rot R;
        R.c0 = v3f{1.f, 0.f, 0.f}; R.c1 = v3f{0.f, 1.f, 0.f}; R.c2 = v3f{0.f, 0.f, 1.f};
        float ldj = 0.f;
        int mlp_idx = 0;
        if (valid && !helper) {
            const float *s = args.rot + sample * 9;
            R.c0 = v3f{s[0], s[3], s[6]}; R.c1 = v3f{s[1], s[4], s[7]}; R.c2 = v3f{s[2], s[5], s[8]};
        }
        for (int pos = 0; pos < args.n_layers; ++pos) {
            const int2 d = args.layers[pos];
            const int kind = d.x & 15, perm_row = (d.x >> 4) & 15;
            const float *P = args.plain + d.y;
            const bool mlp = kind != RNF_KIND_AFFINE16;
            const bool mob = kind == RNF_KIND_MOBIUS;
            const int yo = mob ? 3 : 0, NI = yo + F, NO = mob ? 4 * K : 16;
            const float *W0 = P, *b0 = W0 + 64 * NI, *W1 = b0 + 64, *b1 = W1 + 4096, *W3 = b1 + 64, *b3 = W3 + 4096, *W5 = b3 + 64,
                        *b5 = W5 + 4096, *WL = b5 + 64, *bL = WL + (size_t)NO * 64;
            const int ntiles = (NO + 15) / 16;
            if (helper) {
                if (!mlp) continue;
                RowsA wnext = load_rows(WL, bL, 16 * wave + c, NO, q);
                if (mob) lds_barrier();
                lds_barrier(); lds_barrier(); lds_barrier(); lds_barrier();
                float tv[16];
                read_b(H3, 0, q, c, tv);
                for (int rt = wave; rt < ntiles; rt += WAVES) {
                    const RowsA wl = wnext;
                    if (rt + WAVES < ntiles) wnext = load_rows(WL, bL, 16 * (rt + WAVES) + c, NO, q);
                    store_tile(Cm, 16 * rt, NO, q, c, mfma_rows(wl, q, tv));
                }
                lds_barrier();                            // C
                if (args.acts) lds_to_acts(args.acts + ((size_t)mlp_idx * nblocks + blk) * args.act_rows * 16, ACT_HEAD_ROWS + NO, lds, Cm.p, tid);
                ++mlp_idx;
                if (mob) lds_barrier();                   // the exchange of the segment sums
                lds_barrier();                            // end of the layer
                continue;
            }
            if (valid && writer) {                        // the rotation entering the layer, for the backward sweep
                float *o = args.states + ((size_t)pos * args.n + sample) * 9;
                o[0] = R.c0.x; o[1] = R.c1.x; o[2] = R.c2.x; o[3] = R.c0.y; o[4] = R.c1.y; o[5] = R.c2.y; o[6] = R.c0.z; o[7] = R.c1.z; o[8] = R.c2.z;
            }
            if (!mlp) {                                   // Uncondition16Trans (squeezetrans.py:57-66) / UnconditionRot (rottrans.py:8-23)
                float M[16], Mi[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) M[i] = P[i];
                const bool orth = (d.x >> 8) & 1;
                const float det = orth ? 1.f : inv4(M, Mi);
                affine16_apply(M, logf(fabsf(det)), R, ldj, orth);
                continue;
            }
            const int p1 = (perm_row + 1) % 3;
            const v3f y = get_col(R, p1);
            RowsA wnext = load_rows(W1, b1, row0 + c, 64, q);
            if (mob && writer) { YL.at(0, c) = y.x; YL.at(1, c) = y.y; YL.at(2, c) = y.z; }
            {
                f32x4 acc = RNF_MFMA4(b0[row0 + c], q ? 0.f : 1.f, zero4());
                const float *wrow = W0 + (size_t)(row0 + c) * NI;
                if (HAS_FEATURE) {
                    const float *frow = args.feature + (valid ? sample : 0) * F;
                    const int nch = (F + 63) / 64;
                    for (int ch = 0; ch < nch; ++ch) {
                        float av[16], bv[16];
                        const int k0 = 64 * ch + 16 * q;
                        if (k0 + 16 <= F) {
                            const Float4U *pa = reinterpret_cast<const Float4U *>(wrow + yo + k0);
                            const Float4U *pb = reinterpret_cast<const Float4U *>(frow + k0);
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const Float4U va = pa[i], vb = pb[i];
                                av[4 * i] = va.x; av[4 * i + 1] = va.y; av[4 * i + 2] = va.z; av[4 * i + 3] = va.w;
                                bv[4 * i] = vb.x; bv[4 * i + 1] = vb.y; bv[4 * i + 2] = vb.z; bv[4 * i + 3] = vb.w;
                            }
                        } else {
#pragma unroll
                            for (int u = 0; u < 16; ++u) {
                                const int k = k0 + u, kc = k < F ? k : F - 1;
                                av[u] = wrow[yo + kc];
                                bv[u] = frow[kc];
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int u = 0; u < 16; ++u) acc = RNF_MFMA4(k0 + u < F ? av[u] : 0.f, valid ? bv[u] : 0.f, acc);
                    }
                }
                if (mob) {
                    const float a = wrow[q < 3 ? q : 2];
                    lds_barrier();                        // YL
                    acc = RNF_MFMA4(q < 3 ? a : 0.f, YL.at(q < 3 ? q : 0, c), acc);
                }
                store_tile(X0, row0, 64, q, c, acc);
            }
            lds_barrier();
            {
                const RowsA w1 = wnext;
                wnext = load_rows(W3, b3, row0 + c, 64, q);
                float bv[16];
                read_b(X0, 0, q, c, bv, Relu());
                f32x4 acc = mfma_rows(w1, q, bv);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = Relu()(acc[r]);
                store_tile(H1, row0, 64, q, c, acc);
            }
            lds_barrier();
            {
                const RowsA w3 = wnext;
                wnext = load_rows(W5, b5, row0 + c, 64, q);
                float bv[16];
                read_b(H1, 0, q, c, bv);
                f32x4 acc = mfma_rows(w3, q, bv);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = Relu()(acc[r]);
                store_tile(H2, row0, 64, q, c, acc);
            }
            lds_barrier();
            {
                const RowsA w5 = wnext;
                wnext = load_rows(WL, bL, row0 + c, NO, q);
                float bv[16], xv[4];
                read_b(H2, 0, q, c, bv);
                f32x4 acc = mfma_rows(w5, q, bv);
#pragma unroll
                for (int r = 0; r < 4; ++r) xv[r] = X0.at(row0 + 4 * q + r, c);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = Relu()(acc[r] + xv[r]);
                store_tile(H3, row0, 64, q, c, acc);
            }
            lds_barrier();
            {
                float tv[16];
                read_b(H3, 0, q, c, tv);
                for (int rt = wave; rt < ntiles; rt += WAVES) {
                    const RowsA wl = wnext;
                    if (rt + WAVES < ntiles) wnext = load_rows(WL, bL, 16 * (rt + WAVES) + c, NO, q);
                    store_tile(Cm, 16 * rt, NO, q, c, mfma_rows(wl, q, tv));
                }
            }
            lds_barrier();                                // C
            if (args.acts) lds_to_acts(args.acts + ((size_t)mlp_idx * nblocks + blk) * args.act_rows * 16, ACT_HEAD_ROWS + NO, lds, Cm.p, tid);
            ++mlp_idx;
            if (mob) {
                MobiusSaved sv;
                mobius_frame(R, perm_row, sv);
                const SampleCol crow{Cm, c};
                float sm[3] = {0.f, 0.f, 0.f};
                mobius_segments_sums(sv, crow, K, grp * K / 16, (grp + 1) * K / 16, sm[0], sm[1], sm[2]);
                block_sum<3>(sm, red, wave, q, c);
                mobius_combine(sv, sm[0], sm[1], sm[2]);
                set_col(R, sv.p0, sv.tx);
                set_col(R, sv.p2, sv.tzu * sv.inv_tzu);
                ldj += logf(sm[2] / sm[0]);
            } else {                                      // Condition16Trans (squeezetrans.py:41-50): M = I + reshape(net(f), 4, 4)
                float M[16], Mi[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) M[i] = Cm.at(i, c) + ((i % 5) == 0 ? 1.f : 0.f);
                const float det = inv4(M, Mi);
                affine16_apply(M, logf(fabsf(det)), R, ldj, false);
            }
            lds_barrier();                                // C and the reduction scratch are free again
        }
        if (valid && writer) {
            float *o = args.rot_out + sample * 9;
            o[0] = R.c0.x; o[1] = R.c1.x; o[2] = R.c2.x; o[3] = R.c0.y; o[4] = R.c1.y; o[5] = R.c2.y; o[6] = R.c0.z; o[7] = R.c1.z; o[8] = R.c2.z;
            args.ldj_out[sample] = ldj;
        }
    }
}

}  // namespace b16
}  // namespace rnf
