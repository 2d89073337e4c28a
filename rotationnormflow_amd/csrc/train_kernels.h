// train_kernels.h -- backward pass of Flow.forward (training: agent.py:75-92 differentiates loss = mean(-ldj) through every layer).
//
// Design (round 1, correctness first): one rotation per lane, one wave (64 rotations) per workgroup, ONE launch for the whole
// reverse sweep.  The forward stack kernel saves the rotation at the input of every layer (FlowArgs::states); this kernel walks
// the layers backwards and, per layer, recomputes the conditioner MLP in fp32 with plain per-lane loops (weights arrive through
// wave-uniform scalar loads from the "plain" parameter blob, activations live in LDS as [feature][sample]), applies the
// reverse-mode formulas of so3_grad.h, and accumulates the parameter gradients of its 64 rotations into the global gradient
// blob with coalesced float atomics.  Training batches are 128-1024 rotations (settings/*.yml), so this path is latency-, not
// throughput-critical; the inference kernels are untouched.
//
// Plain blob layout per layer (floats, torch.nn.Linear order [out][in], offsets in the layer table):
//   Moebius (NI = 3 + F, NO = 4K) / Condition16Trans (NI = F, NO = 16) conditioner MLP (flow/condition.py):
//       W0 [64][NI] | b0 [64] | W1 [64][64] | b1 | W3 | b3 | W5 | b5 | WL [NO][64] | bL [NO]
//   Uncondition16Trans:              M [16]
// The gradient blob has the same layout.
#pragma once
#include <hip/hip_runtime.h>

#include "layout.h"
#include "so3_grad.h"

namespace rnf {

constexpr int TR_MAX_LAYERS = 200;

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

// LDS matrix [rows][64 samples] with an XOR swizzle so that both "lane = sample" and "lane = row" accesses are conflict free
struct LMat {
    float *p;
    __device__ __forceinline__ float &at(int row, int s) const { return p[row * 64 + (s ^ (row & 63))]; }
};

// one lane's column of an LMat as a conditioner-output row (so3_grad.h accessor)
struct LaneRow {
    LMat m;
    int lane;
    float mask;
    __device__ __forceinline__ float get(int row) const { return m.at(row, lane); }
    __device__ __forceinline__ void put(int row, float v) const { m.at(row, lane) = v * mask; }
};

// out[o][lane] = (bias[o]) + sum_i W[o][i] in_reg[i]   (NI <= 64 inputs held in registers)
template <int NI>
__device__ __forceinline__ void matvec_rows(const float *W, int ldw, const float *bias, int n_out, const float (&in)[NI], const LMat &out,
                                            int lane) {
    for (int o = 0; o < n_out; ++o) {
        float acc = bias ? bias[o] : 0.f;
        const float *w = W + (size_t)o * ldw;
#pragma unroll
        for (int i = 0; i < NI; ++i) acc = fmaf(w[i], in[i], acc);
        out.at(o, lane) = acc;
    }
}

// g_in[i] = sum_o W[o][i] g_out[o][lane]   (64 inputs)
__device__ __forceinline__ void matvec_cols64(const float *W, int ldw, int n_out, const LMat &gout, int lane, float (&gin)[64]) {
#pragma unroll
    for (int i = 0; i < 64; ++i) gin[i] = 0.f;
    for (int o = 0; o < n_out; ++o) {
        const float g = gout.at(o, lane);
        const float *w = W + (size_t)o * ldw;
#pragma unroll
        for (int i = 0; i < 64; ++i) gin[i] = fmaf(w[i], g, gin[i]);
    }
}

// gW[o][i] += sum_s G[o][s] A[i][s] for i = lane (one 64-wide block of inputs), gb[o] += sum_s G[o][s] (if gb and lane row)
__device__ __forceinline__ void wgrad64(const LMat &G, int n_out, const LMat &A, float *gW, int ldw, float *gb, int lane, int nvalid) {
    float a[64];
#pragma unroll
    for (int s = 0; s < 64; ++s) a[s] = s < nvalid ? A.at(lane, s) : 0.f;
    for (int o = 0; o < n_out; ++o) {
        float acc = 0.f;
#pragma unroll
        for (int s = 0; s < 64; ++s) acc = fmaf(a[s], G.at(o, s), acc);
        atomicAdd(gW + (size_t)o * ldw + lane, acc);
    }
    if (gb) {
        for (int o0 = 0; o0 < n_out; o0 += 64) {
            const int o = o0 + lane;
            if (o < n_out) {
                float acc = 0.f;
                for (int s = 0; s < nvalid; ++s) acc += G.at(o, s);
                atomicAdd(gb + o, acc);
            }
        }
    }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return __shfl(v, 0, 64);
}

__global__ __launch_bounds__(64) void flow_train_backward_kernel(const TrainArgs args) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x;
    const int K = args.K, F = args.F;
    const int NO = 4 * K;
    // LDS: X0, H1, H2, H3 (pre-activations, 64 rows each), GA, GB (gradient ping-pong, 64 rows), C (NO rows)
    LMat X0{lds}, H1{lds + 4096}, H2{lds + 8192}, H3{lds + 12288}, GA{lds + 16384}, GB{lds + 20480}, Cm{lds + 24576};

    const long long nblocks = (args.n + 63) / 64;
    for (long long blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const long long sample = blk * 64 + lane;
        const bool valid = sample < args.n;
        const int nvalid = (int)((args.n - blk * 64) < 64 ? (args.n - blk * 64) : 64);
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
            if (kind == RNF_KIND_AFFINE16) {
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
                const bool orth = (d.x >> 8) & 1;                // UnconditionRot: ldj = 0 (flow/rottrans.py:21)
                affine16_backward(M, sv, gR, g_ldj, orth, gM, gRin);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float tot = wave_sum(valid ? gM[i] : 0.f);
                    if (lane == 0) atomicAdd(Gp + i, tot);
                }
                const float gl = wave_sum(valid && !orth ? g_ldj : 0.f);
                if (lane == 0) atomicAdd(args.g_ldj_sum + pos, gl);
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
            // x0 = W0 (y (+) f) + b0
            for (int o = 0; o < 64; ++o) {
                float acc = b0[o];
                if (mob) acc = fmaf(W0[(size_t)o * NI + 2], y.z, fmaf(W0[(size_t)o * NI + 1], y.y, fmaf(W0[(size_t)o * NI], y.x, acc)));
                X0.at(o, lane) = acc;
            }
            if (F) {
                const float *f = args.feature + (valid ? sample : 0) * F;
                for (int j = 0; j < F; ++j) {
                    const float fj = valid ? f[j] : 0.f;
                    for (int o = 0; o < 64; ++o) X0.at(o, lane) = fmaf(W0[(size_t)o * NI + yo + j], fj, X0.at(o, lane));
                }
            }
            float act[64];
#pragma unroll
            for (int i = 0; i < 64; ++i) act[i] = fmaxf(X0.at(i, lane), 0.f);
            matvec_rows<64>(W1, 64, b1, 64, act, H1, lane);
#pragma unroll
            for (int i = 0; i < 64; ++i) act[i] = fmaxf(H1.at(i, lane), 0.f);
            matvec_rows<64>(W3, 64, b3, 64, act, H2, lane);
#pragma unroll
            for (int i = 0; i < 64; ++i) act[i] = fmaxf(H2.at(i, lane), 0.f);
            matvec_rows<64>(W5, 64, b5, 64, act, H3, lane);
#pragma unroll
            for (int i = 0; i < 64; ++i) act[i] = fmaxf(X0.at(i, lane) + H3.at(i, lane), 0.f);          // t
            matvec_rows<64>(WL, 64, bL, NO, act, Cm, lane);
            // layer math forward + backward; the gradient w.r.t. the conditioner output overwrites C in place
            Rot gRin;
            const float pmask = valid ? 1.f : 0.f;                                        // padding lanes: no parameter gradient
            if (mob) {
                Rot Rout;
                MobiusSaved sv;
                float l;
                const LaneRow crow{Cm, lane, 1.f}, grow{Cm, lane, pmask};
                mobius_segments_forward(Rin, perm_row, crow, K, Rout, l, sv);
                mobius_segments_backward(sv, crow, K, gR, g_ldj, grow, gRin);
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
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) Cm.at(4 * i + j, lane) = (gM[4 * i + j] + g_ldj * Mi[4 * j + i]) * pmask;
            }
            // ---- conditioner backward.  C now holds g_c [NO][64] ----
            __syncthreads();
            // fc_last: t recomputed into GA (activation, [64][64]); gWL, gbL; g_t -> registers
#pragma unroll
            for (int i = 0; i < 64; ++i) GA.at(i, lane) = act[i];                                       // act still holds t
            __syncthreads();
            wgrad64(Cm, NO, GA, gWL, 64, gbL, lane, nvalid);
            float g[64];
            matvec_cols64(WL, 64, NO, Cm, lane, g);
            // t = relu(x0 + h3)
            float gx0[64];
#pragma unroll
            for (int i = 0; i < 64; ++i) { g[i] = (X0.at(i, lane) + H3.at(i, lane)) > 0.f ? g[i] : 0.f; gx0[i] = g[i]; }
            // L5: h3 = W5 relu(h2) + b5
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 64; ++i) { GB.at(i, lane) = valid ? g[i] : 0.f; GA.at(i, lane) = fmaxf(H2.at(i, lane), 0.f); }
            __syncthreads();
            wgrad64(GB, 64, GA, gW5, 64, gb5, lane, nvalid);
            matvec_cols64(W5, 64, 64, GB, lane, g);
#pragma unroll
            for (int i = 0; i < 64; ++i) g[i] = H2.at(i, lane) > 0.f ? g[i] : 0.f;
            // L3
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 64; ++i) { GB.at(i, lane) = valid ? g[i] : 0.f; GA.at(i, lane) = fmaxf(H1.at(i, lane), 0.f); }
            __syncthreads();
            wgrad64(GB, 64, GA, gW3, 64, gb3, lane, nvalid);
            matvec_cols64(W3, 64, 64, GB, lane, g);
#pragma unroll
            for (int i = 0; i < 64; ++i) g[i] = H1.at(i, lane) > 0.f ? g[i] : 0.f;
            // L1
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 64; ++i) { GB.at(i, lane) = valid ? g[i] : 0.f; GA.at(i, lane) = fmaxf(X0.at(i, lane), 0.f); }
            __syncthreads();
            wgrad64(GB, 64, GA, gW1, 64, gb1, lane, nvalid);
            matvec_cols64(W1, 64, 64, GB, lane, g);
#pragma unroll
            for (int i = 0; i < 64; ++i) g[i] = (X0.at(i, lane) > 0.f ? g[i] : 0.f) + gx0[i];           // total dL/dx0
            // fc_first: x0 = W0 (y (+) f) + b0
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 64; ++i) GB.at(i, lane) = valid ? g[i] : 0.f;
            __syncthreads();
            {
                {   // gb0
                    float acc = 0.f;
                    for (int s = 0; s < nvalid; ++s) acc += GB.at(lane, s);
                    atomicAdd(gb0 + lane, acc);
                }
                if (mob) {   // gW0[:, 0:3] and the conditioner-input path of dL/dy
                    float gy[3] = {0.f, 0.f, 0.f};
                    for (int o = 0; o < 64; ++o) {
                        const float go = GB.at(o, lane);                                  // padding lanes hold 0
                        gy[0] = fmaf(W0[(size_t)o * NI], go, gy[0]);
                        gy[1] = fmaf(W0[(size_t)o * NI + 1], go, gy[1]);
                        gy[2] = fmaf(W0[(size_t)o * NI + 2], go, gy[2]);
                        const float c0 = wave_sum(go * y.x), c1 = wave_sum(go * y.y), c2 = wave_sum(go * y.z);
                        if (lane == 0) {
                            atomicAdd(gW0 + (size_t)o * NI, c0);
                            atomicAdd(gW0 + (size_t)o * NI + 1, c1);
                            atomicAdd(gW0 + (size_t)o * NI + 2, c2);
                        }
                    }
                    set_col(gRin, p1, get_col(gRin, p1) + v3f{gy[0], gy[1], gy[2]});
                }
                if (F) {
                    // gW0[o][yo+j] += sum_s g[o][s] f[s][j]  (lane = j);  g_feature[s][j] += sum_o W0[o][yo+j] g[o][s]  (lane = s)
                    for (int j0 = 0; j0 < F; j0 += 64) {
                        const int j = j0 + lane;
                        if (j < F) {
                            for (int o = 0; o < 64; ++o) {
                                float acc = 0.f;
                                for (int s = 0; s < nvalid; ++s) acc = fmaf(GB.at(o, s), args.feature[(blk * 64 + s) * F + j], acc);
                                atomicAdd(gW0 + (size_t)o * NI + yo + j, acc);
                            }
                        }
                    }
                    if (args.g_feature && valid) {
                        float *gf = args.g_feature + sample * F;
                        for (int j = 0; j < F; ++j) {
                            float acc = 0.f;
                            for (int o = 0; o < 64; ++o) acc = fmaf(W0[(size_t)o * NI + yo + j], GB.at(o, lane), acc);
                            gf[j] += acc;
                        }
                    }
                }
            }
            __syncthreads();
            gR = gRin;
        }
        if (valid) {
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
