// Microbenchmark: do v_mfma_f32_32x32x16_f16 (matrix cores) and plain VALU work overlap on one gfx950 SIMD?
//  (a) wave-specialised: in a workgroup of 8 waves (2 per SIMD) waves 0-3 issue only MFMAs, waves 4-7 only VALU; the time of the
//      pair is compared with each half running alone (the other half idles at the barrier).
//  (b) same wave: NM independent MFMAs followed by NV independent v_fma per step, 1 / 2 / 4 waves per SIMD.
// Build+run:  hipcc --offload-arch=gfx950 -O3 -o /tmp/ovl tools/micro/mfma_f16_valu_overlap.hip && /tmp/ovl
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#define MFMA_H(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

// ROLE bit 0: the MFMA waves work; bit 1: the VALU waves work; bit 2: VALU waves at s_setprio 3, MFMA waves at 0;
// bit 3: an s_nop 7 + s_nop 7 pad behind every MFMA; bit 4: the VALU waves are the OLDER ones (waves 0-3)
template <int ROLE>
__global__ void specialised(float *out, unsigned long long *cyc, int iters, float c1, float c2) {
    const int wave = (ROLE & 16) ? 7 - (int)(threadIdx.x >> 6) : (int)(threadIdx.x >> 6);
    f32x16 acc[4];
    for (int q = 0; q < 4; ++q)
        for (int r = 0; r < 16; ++r) acc[q][r] = threadIdx.x * 0.001f + r + q;
    h8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.01f * j + threadIdx.x * 1e-4f); b[j] = (_Float16)(0.5f - 0.01f * j); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
    __syncthreads();
    unsigned long long t0 = clock64();
    if (wave < 4) {
        if (ROLE & 4) __builtin_amdgcn_s_setprio(0);
        if (ROLE & 1)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        acc[q] = MFMA_H(a, b, acc[q]);      // 16 MFMAs, 4 independent chains
                        if (ROLE & 8) { asm volatile("s_nop 7\n\ts_nop 7"); __builtin_amdgcn_sched_barrier(0); }
                    }
            }
    } else {
        if (ROLE & 4) __builtin_amdgcn_s_setprio(3);
        if (ROLE & 2)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 32; ++u)
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = fmaf(v[i], c1, c2);             // 256 VALU, 8 independent chains
            }
    }
    unsigned long long t1 = clock64();
    float s = 0;
    for (int q = 0; q < 4; ++q)
        for (int r = 0; r < 16; ++r) s += acc[q][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) atomicMax(cyc + (wave < 4 ? 0 : 1), t1 - t0);
}

template <int NM, int NV>
__global__ void same_wave(float *out, unsigned long long *cyc, int iters, float c1, float c2) {
    f32x16 acc[4];
    for (int q = 0; q < 4; ++q)
        for (int r = 0; r < 16; ++r) acc[q][r] = threadIdx.x * 0.001f + r + q;
    h8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.01f * j + threadIdx.x * 1e-4f); b[j] = (_Float16)(0.5f - 0.01f * j); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
    unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < NM; ++q) acc[q & 3] = MFMA_H(a, b, acc[q & 3]);
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i & 7] = fmaf(v[i & 7], c1, c2);
        __builtin_amdgcn_sched_barrier(0);
    }
    unsigned long long t1 = clock64();
    float s = 0;
    for (int q = 0; q < 4; ++q)
        for (int r = 0; r < 16; ++r) s += acc[q][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) atomicAdd(cyc, t1 - t0);
}

template <int ROLE>
static void run_spec(const char *label) {
    const int blocks = 256, threads = 512, iters = 1000;
    float *out;
    unsigned long long *cyc, h[2] = {0, 0};
    hipMalloc(&out, sizeof(float) * blocks * threads);
    hipMalloc(&cyc, 16);
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(cyc, 0, 16);
        hipLaunchKernelGGL((specialised<ROLE>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 0.999f, 0.001f);
        hipDeviceSynchronize();
    }
    hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    printf("%-58s MFMA waves: %7.1f cycles/iter (16 MFMA)   VALU waves: %7.1f cycles/iter (256 v_fma)\n", label, (double)h[0] / iters,
           (double)h[1] / iters);
    hipFree(out);
    hipFree(cyc);
}

template <int NM, int NV>
static void run_same(int threads) {
    const int blocks = 256, iters = 4000;
    float *out;
    unsigned long long *cyc, h = 0;
    hipMalloc(&out, sizeof(float) * blocks * threads);
    hipMalloc(&cyc, 8);
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(cyc, 0, 8);
        hipLaunchKernelGGL((same_wave<NM, NV>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 0.999f, 0.001f);
        hipDeviceSynchronize();
    }
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    double waves = (double)blocks * threads / 64;
    printf("same wave: %d MFMA + %2d v_fma per step, waves/SIMD=%d : %7.1f cycles/step per wave  (%.1f per SIMD-step)\n", NM, NV, threads / 256,
           (double)h / waves / iters, (double)h / waves / iters);
    hipFree(out);
    hipFree(cyc);
}

int main() {
    run_spec<1>("specialised, only the MFMA waves work");
    run_spec<2>("specialised, only the VALU waves work");
    run_spec<3>("specialised, both work (overlap => max, no overlap => sum)");
    run_spec<3 + 4>("both work, VALU waves prio 3 / MFMA waves prio 0");
    run_spec<3 + 8>("both work, s_nop pads behind every MFMA");
    run_spec<1 + 8>("only MFMA waves, s_nop pads");
    run_spec<3 + 4 + 8>("both work, prio + pads");
    run_spec<3 + 16>("both work, VALU waves are the older ones");
    run_spec<3 + 16 + 8>("both work, VALU waves older + pads");
    for (int t = 1024; t <= 1024; t *= 2) {
        run_same<4, 0>(t);
        run_same<0, 32>(t);
        run_same<4, 16>(t);
        run_same<4, 32>(t);
        run_same<4, 64>(t);
    }
    return 0;
}
