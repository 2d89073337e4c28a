// featproj_kernel.h -- feature projection pre-pass for conditional flows.
//
// Every layer that consumes the per-sample feature vector f (conditional MobiusFlow: flow/mobiusflow.py:53-57 with
// fc_first.weight[:, 3:]; Condition16Trans: flow/squeezetrans.py:47 with the whole fc_first) needs
//     G_l = W_l f + b_l      (64 values per sample per layer)
// which does not depend on the rotation state, so it is hoisted out of the layer stack: one fp32-MFMA GEMM
// [64 x F] x [F x 32 samples] per (layer, out tile, wave), written to scratch in accumulator-fragment order so the
// stack kernel can load it straight into the fc_first accumulator (coalesced 1 KiB per wave instruction).
#pragma once
#include "flow_kernels.h"

namespace rnf {

constexpr int MAX_SLOTS = 224;
constexpr int FP_KCHUNK = 256;               // features held in registers at a time (128 VGPRs of B fragments)

struct FeatProjArgs {
    const float *feat;       // [n, F] row-major, F % 8 == 0
    const float *blob;
    float *G;                // [n_slots][g_groups][2][4][64] float4
    long long n;
    long long g_groups;
    int F;
    int n_slots;
    int feat_off[MAX_SLOTS]; // blob offset (floats) of each slot's featproj record
};

template <int NW>
__global__ __launch_bounds__(NW * 64) void featproj_kernel(const FeatProjArgs args) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    constexpr int NT = NW * 64;
    constexpr int TILE = NW * TILE_SAMPLES;
    const long long ntiles = (args.n + TILE - 1) / TILE;
    const int F = args.F;
    const int ngroups_k = F / 8;                                   // float4 k-groups per out row

    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long group = tile * NW + wave;
        const long long sample = group * TILE_SAMPLES + j;
        const bool valid = sample < args.n;
        for (int kc = 0; kc < F; kc += FP_KCHUNK) {
            const int nu = min(FP_KCHUNK, F - kc) / 8;
            float4 bf[FP_KCHUNK / 8];
#pragma unroll
            for (int u = 0; u < FP_KCHUNK / 8; ++u) {
                bf[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (u < nu && valid)
                    bf[u] = *reinterpret_cast<const float4 *>(args.feat + sample * F + kc + 8 * u + 4 * h);
            }
            for (int slot = 0; slot < args.n_slots; ++slot) {
                const float *rec = args.blob + args.feat_off[slot];
                for (int ot = 0; ot < 2; ++ot) {
                    __syncthreads();
                    stage_floats(lds, rec + ((size_t)ot * ngroups_k + kc / 8) * 256, nu * 256, tid, NT);
                    __syncthreads();
                    float *g = args.G + (((size_t)slot * args.g_groups + group) * 2 + ot) * (4 * 64 * 4);
                    f32x16 acc;
                    if (kc == 0) {
                        const float *bias = rec + (size_t)2 * ngroups_k * 256 + (ot * 2 + h) * 16;
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[r] = bias[r];
                    } else {
                        acc = load_g16(g, lane);
                    }
#pragma unroll
                    for (int u = 0; u < FP_KCHUNK / 8; ++u) {
                        if (u < nu) {
                            float4 a = lds_f4(lds, u * 64 + lane);
                            acc = RNF_MFMA(a.x, bf[u].x, acc);
                            acc = RNF_MFMA(a.y, bf[u].y, acc);
                            acc = RNF_MFMA(a.z, bf[u].z, acc);
                            acc = RNF_MFMA(a.w, bf[u].w, acc);
                        }
                    }
                    float4 *g4 = reinterpret_cast<float4 *>(g);
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        g4[q * 64 + lane] = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
                }
            }
        }
    }
}

}  // namespace rnf
