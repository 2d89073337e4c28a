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
#ifdef RNF_FP_BURST
constexpr bool FP_SPREAD = false;            // A/B switch: every global access of a tile in one burst behind its barrier
#else
constexpr bool FP_SPREAD = true;
#endif

struct FeatProjArgs {
    const float *feat;       // [n, F] row-major, F % 8 == 0
    const float *blob;
    float *G;                // [n_slots][g_groups][2][4][64] float4
    long long n;
    long long g_groups;
    int F;
    int n_slots;
    int row_mode;            // 1: the "samples" are shared feature rows; G holds a 64-float record per (slot, row), see flow_kernels.h GFrag
    const int *only_if;      // != nullptr: run only when *only_if != 0 (exact-fp32 re-run of a guarded split-precision call)
    int feat_off[MAX_SLOTS]; // blob offset (floats) of each slot's featproj record
};

// where one lane's 16 accumulator registers of (slot, out tile) live in G: fragment order per 32-sample group (coalesced 1 KiB wave
// transactions), or -- row mode -- inside the 64-float record of the lane's feature row
struct GOut {
    float4 *p;
    int stride;      // float4 between consecutive register quads
    bool ok;
    __device__ __forceinline__ GOut(const FeatProjArgs &a, int slot, long long group, long long sample, bool valid, int ot, int lane, int h) {
        if (a.row_mode) {
            p = reinterpret_cast<float4 *>(a.G + ((size_t)slot * a.n + (valid ? sample : 0)) * 64 + (ot * 2 + h) * 16);
            stride = 1;
            ok = valid;
        } else {
            p = reinterpret_cast<float4 *>(a.G + (((size_t)slot * a.g_groups + group) * 2 + ot) * (4 * 64 * 4)) + lane;
            stride = 64;
            ok = true;
        }
    }
    // non-temporal stores: the scratch is written once and read by another kernel after 0.3 - 3 GB more of it have gone by -- there is
    // nothing to gain from keeping it in L2, and the weight tiles every workgroup streams from there stay resident (C5 -4.3 %, C4 -1.3 %)
    __device__ __forceinline__ void store(int q, float4 v) const {
        typedef float f4v __attribute__((ext_vector_type(4)));
        if (ok) __builtin_nontemporal_store(f4v{v.x, v.y, v.z, v.w}, reinterpret_cast<f4v *>(p + q * stride));
    }
    __device__ __forceinline__ f32x16 load() const {
        const float4 b0 = p[0], b1 = p[stride], b2 = p[2 * stride], b3 = p[3 * stride];
        f32x16 c;
        c[0] = b0.x; c[1] = b0.y; c[2] = b0.z; c[3] = b0.w;
        c[4] = b1.x; c[5] = b1.y; c[6] = b1.z; c[7] = b1.w;
        c[8] = b2.x; c[9] = b2.y; c[10] = b2.z; c[11] = b2.w;
        c[12] = b3.x; c[13] = b3.y; c[14] = b3.z; c[15] = b3.w;
        return c;
    }
};

// asynchronous LDS read of one 16-byte operand fragment: the data is valid only behind an s_waitcnt lgkmcnt(0) that takes `d` as an operand
template <int OFF>
__device__ __forceinline__ void fp_lds_read(h8 &d, unsigned lds_addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(lds_addr), "n"(OFF));
}
__device__ __forceinline__ void fp_lds_read_at(h8 &d, unsigned lds_addr, int off) {       // `off` is a compile-time constant after unrolling
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(lds_addr), "n"(off));
}

// MULTI (split precision): F > FP_KCHUNK features go through several K-chunks whose partial sums travel through the scratch (only F > 512
// reaches this kernel that way: 256 < F <= 512 runs featproj_ksplit_kernel).  The single-chunk instantiation has no accumulator start to
// fetch at all (round 5: the bias moved into the stack kernel's fc_first image): 16 registers fewer live across the tile loop, no spills.
template <int NW, int PREC, bool MULTI = false>
__global__ __launch_bounds__(NW * 64) void featproj_kernel(const FeatProjArgs args) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (args.only_if && __hip_atomic_load(args.only_if, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    constexpr int NT = NW * 64;
    constexpr int TILE = NW * TILE_SAMPLES;
    const long long ntiles = (args.n + TILE - 1) / TILE;
    const int F = args.F;

    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long group = tile * NW + wave;
        const long long sample = group * TILE_SAMPLES + j;
        const bool valid = sample < args.n;
        for (int kc = 0; kc < F; kc += FP_KCHUNK) {
            if constexpr (PREC == 0) {
                // ---- exact fp32: v_mfma_f32_32x32x2_f32, image [ot][F/8][64] float4 ----
                const int ngroups_k = F / 8;
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
                        const GOut gout(args, slot, group, sample, valid, ot, lane, h);
                        f32x16 acc;
                        if (kc == 0) {                                 // (round 5: no bias here -- it sits in the stack kernel's fc_first image)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                        } else {
                            acc = gout.load();
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
#pragma unroll
                        for (int q = 0; q < 4; ++q) gout.store(q, make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]));
                    }
                }
            } else {
                // ---- split precision: v_mfma_f32_32x32x16_f16 x3, image [ot][ceil(F/16)][hi,lo][64] 8 x fp16 ----
                // k-step s covers features 16s .. 16s+15; lane-half h supplies 16s + 8h + 0..7 straight from its row.
                const int nsteps_all = (F + 15) / 16;
                const int ns = (min(FP_KCHUNK, F - kc) + 15) / 16;
                h8 bh[FP_KCHUNK / 16], bl[FP_KCHUNK / 16];
#pragma unroll
                for (int s = 0; s < FP_KCHUNK / 16; ++s) {
                    f2 v[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = f2{0.f, 0.f};
                    const int k0 = kc + 16 * s + 8 * h;
                    if (s < ns && valid && k0 < F) {          // F % 8 == 0: an 8-group is entirely inside or outside
                        const float4 p0 = *reinterpret_cast<const float4 *>(args.feat + sample * F + k0);
                        const float4 p1 = *reinterpret_cast<const float4 *>(args.feat + sample * F + k0 + 4);
                        v[0] = f2{p0.x, p0.y}; v[1] = f2{p0.z, p0.w}; v[2] = f2{p1.x, p1.y}; v[3] = f2{p1.z, p1.w};
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const h2 ph = __builtin_convertvector(v[q], h2);
                        const h2 pl = __builtin_convertvector((v[q] - __builtin_convertvector(ph, f2)) * FEAT_LO_SCALE, h2);
                        bh[s][2 * q] = ph[0]; bh[s][2 * q + 1] = ph[1];
                        bl[s][2 * q] = pl[0]; bl[s][2 * q + 1] = pl[1];
                    }
                }
                // weight tiles (slot, out tile) stream through two LDS buffers by LDS-DMA: tile t + 1 lands while tile t multiplies,
                // one barrier per tile (the synchronous copy with two barriers per tile was 5x the MFMA time of the tile)
                const int n_t = 2 * args.n_slots;
                constexpr int BUF = (FP_KCHUNK / 16) * 512;
                auto tile_src = [&](int t) {
                    return args.blob + args.feat_off[t >> 1] + ((size_t)(t & 1) * nsteps_all + kc / 16) * 512;
                };
                __syncthreads();                                   // every wave has finished with both buffers (previous chunk / tile)
                dma_floats(lds, tile_src(0), ns * 512, wave, lane, NW);
                // Every global access of a tile is issued at its TOP, behind the barrier: the DMA of the next tile's weights, the loads of
                // the next tile's accumulator start (bias, or the partial sums of the previous K-chunk) and the stores of the previous
                // tile's result.  vmcnt counts loads, LDS-DMA and stores alike, so a load or store issued later in the tile makes the
                // compiler's (or the next tile's) s_waitcnt vmcnt(0) wait for the DMA just issued as well: with the bias loaded right in
                // front of the matrix instructions and the result stored behind them, DMA latency + matrix time + store latency ADDED UP
                // (0.27 + 0.37 + 0.2 ms per launch instead of their maximum).
                const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                auto start_of = [&](int t) {                       // accumulator start of tile t: zero, or (MULTI, later chunks) the partial sums
                    if (!MULTI || kc == 0) return zero16;
                    return GOut(args, t >> 1, group, sample, valid, t & 1, lane, h).load();
                };
                f32x16 pend, cur0 = start_of(0), nxt0 = cur0;      // (single chunk: cur0 / nxt0 are the constant zero -- no registers)
                int pend_t = -1;
                auto flush = [&]() {
                    if (pend_t < 0) return;
                    const GOut gout(args, pend_t >> 1, group, sample, valid, pend_t & 1, lane, h);
#pragma unroll
                    for (int q = 0; q < 4; ++q) gout.store(q, make_float4(pend[4 * q], pend[4 * q + 1], pend[4 * q + 2], pend[4 * q + 3]));
                };
                // SPREAD (full chunks, 8 waves): the tile's 12 vector-memory instructions per wave -- 4 DMA pieces of the next weight tile, 4
                // pieces of the next accumulator start, 4 stores of the previous result -- ride ONE PER K-STEP behind the matrix
                // instructions instead of going out in one burst behind the barrier, where all eight waves queued 96 KB at the texture
                // unit while the matrix pipe sat idle.
                const bool spread = FP_SPREAD && NW == 8 && ns == FP_KCHUNK / 16 && !args.row_mode;
                for (int t = 0; t < n_t; ++t) {
                    const float *wl = lds + (t & 1) * BUF;
#ifndef RNF_FPKO_BAR
                    dma_wait_all();
                    __syncthreads();                               // tile t is complete; nobody still reads the other buffer
#endif
                    if (!spread) {
                        if (t + 1 < n_t) {
                            dma_floats(lds + ((t + 1) & 1) * BUF, tile_src(t + 1), ns * 512, wave, lane, NW);
                            nxt0 = start_of(t + 1);
                        }
                        flush();                                   // tile t - 1
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    // (hi.lo and lo.hi add into ONE accumulator: a third one is 3.5 % faster on the bare loop, tools/micro/featproj_lds_bound.hip,
                    // and nothing in the kernel -- C4 7.87 -> 7.98 ms same-box A/B, round 5)
                    f32x16 acc1 = cur0, acc2;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
                    if (spread) {
                        const bool more = t + 1 < n_t;
                        const float *nsrc = tile_src(more ? t + 1 : t);
                        float *ndst = lds + ((t + 1) & 1) * BUF;
                        const float4 *nstart = nullptr;            // MULTI, later chunks: the four 16-byte pieces of the next tile's partial sums
                        int nstride = 0;
                        if (MULTI && kc != 0) {
                            const int tn = more ? t + 1 : t;
                            const GOut g(args, tn >> 1, group, sample, valid, tn & 1, lane, h);
                            nstart = g.p;
                            nstride = g.stride;
                        }
                        const GOut pout(args, (pend_t < 0 ? 0 : pend_t) >> 1, group, sample, valid, (pend_t < 0 ? 0 : pend_t) & 1, lane, h);
                        // one k-step of operand look-ahead: the A fragments of step s + 1 are requested BEFORE the matrix instructions of step
                        // s.  Left to the compiler every step was "multiply, then read the next operands into the same registers, wait": the
                        // LDS latency (eight waves ask for 2 KiB each per step) in series with the three matrix instructions, at two waves per
                        // SIMD -- matrix pipe 41 % busy.  The compiler sinks plain LDS loads back to their use, so the reads are issued by
                        // inline asm and consumed behind an explicit lgkmcnt(0) through which their registers pass as operands.
                        const unsigned wl_lds = (unsigned)(size_t)(const __attribute__((address_space(3))) float *)wl + 16u * lane;
                        h8 ah, al;
                        fp_lds_read<0>(ah, wl_lds);
                        fp_lds_read<1024>(al, wl_lds);
#pragma unroll
                        for (int s = 0; s < FP_KCHUNK / 16; ++s) {
                            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah), "+v"(al) :: "memory");
                            h8 nah = ah, nal = al;
#ifndef RNF_FPKO_LDS
                            if (s + 1 < FP_KCHUNK / 16) {
                                fp_lds_read_at(nah, wl_lds, (s + 1) * 2048);
                                fp_lds_read_at(nal, wl_lds, (s + 1) * 2048 + 1024);
                            }
#endif
                            __builtin_amdgcn_sched_barrier(0);
#ifndef RNF_FPKO_MFMA
                            acc1 = RNF_MFMA_H(ah, bh[s], acc1);
                            acc2 = RNF_MFMA_H(ah, bl[s], acc2);
                            acc2 = RNF_MFMA_H(al, bh[s], acc2);
#else
                            acc1[s] += (float)ah[0] + (float)al[1];
#endif
                            ah = nah; al = nal;
                            __builtin_amdgcn_sched_barrier(0);
#ifdef RNF_FPKO_HALFDMA
                            if (s < 2) {
#else
                            if (s < 4) {
#endif
#ifndef RNF_FPKO_DMA
                                if (more) {
                                    const int base = wave * 64 + s * (NW * 64);       // float4 index, wave uniform: 1 KiB per instruction
                                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(nsrc + 4 * (size_t)(base + lane)),
                                                                     (__attribute__((address_space(3))) void *)(ndst + 4 * base), 16, 0, 0);
                                }
#endif
                            } else if (s < 8) {
                                if (MULTI && kc != 0 && more) {
                                    const float4 v = nstart[(s - 4) * nstride];
                                    nxt0[4 * (s - 4)] = v.x; nxt0[4 * (s - 4) + 1] = v.y; nxt0[4 * (s - 4) + 2] = v.z; nxt0[4 * (s - 4) + 3] = v.w;
                                }
                            } else if (s < 12) {
#ifndef RNF_FPKO_STORE
                                if (pend_t >= 0)
                                    pout.store(s - 8, make_float4(pend[4 * (s - 8)], pend[4 * (s - 8) + 1], pend[4 * (s - 8) + 2], pend[4 * (s - 8) + 3]));
#endif
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    } else if (ns == FP_KCHUNK / 16) {             // full chunk: branch-free
#pragma unroll
                        for (int s = 0; s < FP_KCHUNK / 16; ++s) {
                            const h8 ah = lds_h8(wl, (s * 2 + 0) * 64 + lane);
                            const h8 al = lds_h8(wl, (s * 2 + 1) * 64 + lane);
                            acc1 = RNF_MFMA_H(ah, bh[s], acc1);
                            acc2 = RNF_MFMA_H(ah, bl[s], acc2);
                            acc2 = RNF_MFMA_H(al, bh[s], acc2);
                        }
                    } else {
#pragma unroll
                        for (int s = 0; s < FP_KCHUNK / 16; ++s) {
                            if (s < ns) {
                                const h8 ah = lds_h8(wl, (s * 2 + 0) * 64 + lane);
                                const h8 al = lds_h8(wl, (s * 2 + 1) * 64 + lane);
                                acc1 = RNF_MFMA_H(ah, bh[s], acc1);
                                acc2 = RNF_MFMA_H(ah, bl[s], acc2);
                                acc2 = RNF_MFMA_H(al, bh[s], acc2);
                            }
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) pend[r] = fmaf(acc2[r], (1.0f / FEAT_LO_SCALE), acc1[r]);
                    pend_t = t;
                    cur0 = nxt0;
                }
                dma_wait_all();
                flush();
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// feature_dim in (256, 512] (BASELINE configs[4], F = 512), split precision: the K dimension is split over PAIRS OF WAVES instead of over
// two passes whose partial sums travel through the scratch (round 2: G written twice and re-read once per layer, 10 of C5's 27 ms).
// Workgroup = 8 waves = 4 pairs; pair p = wave & 3 owns one 32-sample group, wave p holds features [0, 256) of it as B fragments, wave p + 4
// features [256, 512).  A weight tile (slot, out tile) is the whole K range (64 KB) in one of two LDS buffers; every wave multiplies its
// half; wave p + 4 hands its 32 x 32 partial sums to wave p through a double-buffered 4 KB LDS slot (written behind the tile's matrix
// instructions, read behind the NEXT tile's barrier: no extra synchronisation) and wave p writes the finished tile.  LDS: 2 x 64 KB
// weights + 2 x 4 x 4 KB exchange = 160 KB.  Every global access of a tile rides one per k-step behind the matrix instructions, as above.
// ------------------------------------------------------------------------------------------------------------
constexpr int FP2_BUF = 2 * (FP_KCHUNK / 16) * 512;       // floats: one weight tile, K = 512
constexpr int FP2_X = 4 * 64 * 16;                        // floats: exchange slots of the 4 pairs for one tile
constexpr size_t FP2_LDS_BYTES = sizeof(float) * (2 * FP2_BUF + 2 * FP2_X);

__global__ __launch_bounds__(8 * 64) void featproj_ksplit_kernel(const FeatProjArgs args) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (args.only_if && __hip_atomic_load(args.only_if, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int pair = wave & 3, khalf = wave >> 2;
    const int F = args.F;
    const int nsteps_all = (F + 15) / 16;                      // k-steps in a record row block (both halves, <= 32)
    const int ns_lo = FP_KCHUNK / 16;                          // k-steps of the first half: always full
    const int ns_mine = khalf ? nsteps_all - ns_lo : ns_lo;    // second half may be ragged
    const long long ntiles = (args.n + 127) / 128;
    float *xbuf = lds + 2 * FP2_BUF;
    const int n_t = 2 * args.n_slots;
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long group = tile * 4 + pair;
        const long long sample = group * TILE_SAMPLES + j;
        const bool valid = sample < args.n;
        h8 bh[FP_KCHUNK / 16], bl[FP_KCHUNK / 16];
#pragma unroll
        for (int s = 0; s < FP_KCHUNK / 16; ++s) {
            f2 v[4] = {f2{0.f, 0.f}, f2{0.f, 0.f}, f2{0.f, 0.f}, f2{0.f, 0.f}};
            const int k0 = khalf * FP_KCHUNK + 16 * s + 8 * h;
            if (valid && k0 < F) {
                const float4 p0 = *reinterpret_cast<const float4 *>(args.feat + sample * F + k0);
                const float4 p1 = *reinterpret_cast<const float4 *>(args.feat + sample * F + k0 + 4);
                v[0] = f2{p0.x, p0.y}; v[1] = f2{p0.z, p0.w}; v[2] = f2{p1.x, p1.y}; v[3] = f2{p1.z, p1.w};
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const h2 ph = __builtin_convertvector(v[q], h2);
                const h2 pl = __builtin_convertvector((v[q] - __builtin_convertvector(ph, f2)) * FEAT_LO_SCALE, h2);
                bh[s][2 * q] = ph[0]; bh[s][2 * q + 1] = ph[1];
                bl[s][2 * q] = pl[0]; bl[s][2 * q + 1] = pl[1];
            }
        }
        auto tile_src = [&](int t) { return args.blob + args.feat_off[t >> 1] + (size_t)(t & 1) * nsteps_all * 512; };
        __syncthreads();                                       // every wave has finished with both buffers (previous sample tile)
        dma_floats(lds, tile_src(0), nsteps_all * 512, wave, lane, 8);
        f32x16 pend;                                           // (round 5: accumulators start at zero; the bias sits in the stack kernel's fc_first image)
        int pend_t = -1;
        for (int t = 0; t < n_t; ++t) {
            const float *wl = lds + (t & 1) * FP2_BUF + khalf * (ns_lo * 512);
            dma_wait_all();
            __syncthreads();                                   // tile t has landed; nobody still reads the other buffer; exchange slot (t - 1) & 1 is written
            const bool more = t + 1 < n_t;
            if (pend_t >= 0 && khalf == 0) {                   // finish tile t - 1: add the partner's half
                const float4 *x = reinterpret_cast<const float4 *>(xbuf + (pend_t & 1) * FP2_X + pair * (64 * 16)) + lane;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 o = x[q * 64];
                    pend[4 * q] += o.x; pend[4 * q + 1] += o.y; pend[4 * q + 2] += o.z; pend[4 * q + 3] += o.w;
                }
            }
            const float *nsrc = tile_src(more ? t + 1 : t);
            float *ndst = lds + ((t + 1) & 1) * FP2_BUF;
            const GOut pout(args, (pend_t < 0 ? 0 : pend_t) >> 1, group, sample, valid, (pend_t < 0 ? 0 : pend_t) & 1, lane, h);
            const int n_dma = (nsteps_all * 128 + 511) / 512;  // 1 KiB pieces per wave of the next tile (<= 8)
            f32x16 acc1, acc2;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[r] = acc2[r] = 0.f;
            const unsigned wl_lds = (unsigned)(size_t)(const __attribute__((address_space(3))) float *)wl + 16u * lane;
            h8 ah, al;
            fp_lds_read<0>(ah, wl_lds);
            fp_lds_read<1024>(al, wl_lds);
#pragma unroll
            for (int s = 0; s < FP_KCHUNK / 16; ++s) {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah), "+v"(al) :: "memory");
                h8 nah = ah, nal = al;
                if (s + 1 < FP_KCHUNK / 16) {
                    fp_lds_read_at(nah, wl_lds, (s + 1) * 2048);
                    fp_lds_read_at(nal, wl_lds, (s + 1) * 2048 + 1024);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (s < ns_mine) {                              // wave uniform (a ragged second half: zero operands beyond F anyway)
                    acc1 = RNF_MFMA_H(ah, bh[s], acc1);
                    acc2 = RNF_MFMA_H(ah, bl[s], acc2);
                    acc2 = RNF_MFMA_H(al, bh[s], acc2);
                }
                ah = nah; al = nal;
                __builtin_amdgcn_sched_barrier(0);
                if (s < 8) {                                   // the next weight tile: up to 8 pieces of 1 KiB per wave
                    if (more && s < n_dma) {
                        const int base = wave * 64 + s * 512;  // float4 index, wave uniform
                        if (base < nsteps_all * 128)
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(nsrc + 4 * (size_t)(base + lane)),
                                                             (__attribute__((address_space(3))) void *)(ndst + 4 * base), 16, 0, 0);
                    }
                } else if (s >= 12) {                          // the previous tile's finished result
                    if (pend_t >= 0 && khalf == 0)
                        pout.store(s - 12, make_float4(pend[4 * (s - 12)], pend[4 * (s - 12) + 1], pend[4 * (s - 12) + 2], pend[4 * (s - 12) + 3]));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) pend[r] = fmaf(acc2[r], (1.0f / FEAT_LO_SCALE), acc1[r]);
            if (khalf) {                                       // hand the upper half's partial sums to the partner (read behind the next barrier)
                float4 *x = reinterpret_cast<float4 *>(xbuf + (t & 1) * FP2_X + pair * (64 * 16)) + lane;
#pragma unroll
                for (int q = 0; q < 4; ++q) x[q * 64] = make_float4(pend[4 * q], pend[4 * q + 1], pend[4 * q + 2], pend[4 * q + 3]);
            }
            pend_t = t;
        }
        dma_wait_all();
        __syncthreads();                                       // the last tile's exchange slot is written
        if (khalf == 0) {
            const float4 *x = reinterpret_cast<const float4 *>(xbuf + (pend_t & 1) * FP2_X + pair * (64 * 16)) + lane;
            const GOut gout(args, pend_t >> 1, group, sample, valid, pend_t & 1, lane, h);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 o = x[q * 64];
                gout.store(q, make_float4(pend[4 * q] + o.x, pend[4 * q + 1] + o.y, pend[4 * q + 2] + o.z, pend[4 * q + 3] + o.w));
            }
        }
    }
}

}  // namespace rnf
