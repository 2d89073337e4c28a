// Microbenchmark: how many VALU instructions hide behind one v_mfma_f32_32x32x2_f32 (64-cycle fp32 matrix instruction) on
// gfx950, for one or two waves per SIMD, independent or dependent VALU, plain or transcendental.  Prints cycles per step
// (s_memtime) where a step = 1 dependent MFMA + NV VALU instructions, fenced with sched_barrier(0) like the tile loop of
// csrc/flow_kernels.h.   Build+run:  hipcc --offload-arch=gfx950 -O3 -o /tmp/coissue tools/micro/mfma_valu_coissue.hip && /tmp/coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// MODE 0: independent plain VALU (8 rotating chains); 1: one dependent plain chain; 2: independent transcendental (v_exp);
// 3: no MFMA at all (VALU only, independent); 4: two independent MFMA chains per wave (head-phase shape) + NV VALU
template <int NV, int MODE>
__global__ void k(float *out, unsigned long long *cyc, int iters, float c1, float c2) {
    f32x16 acc, acc2;
    for (int r = 0; r < 16; ++r) { acc[r] = threadIdx.x * 0.001f + r; acc2[r] = r * 0.5f; }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
    float a = 1.0f + threadIdx.x * 1e-6f, b = 0.5f;
    unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE != 3) acc = MFMA(a, b, acc);
            if (MODE == 4) acc2 = MFMA(b, a, acc2);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                if (MODE == 1) v[0] = fmaf(v[0], c1, c2);
                else if (MODE == 2) v[i & 7] = __builtin_amdgcn_exp2f(v[i & 7]);
                else v[i & 7] = fmaf(v[i & 7], c1, c2);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = clock64();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc[r] + acc2[r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) atomicAdd(cyc, t1 - t0);
}

template <int NV, int MODE>
static void run(int threads, const char *label) {
    const int blocks = 256, iters = 2000;
    float *out;
    unsigned long long *cyc, h = 0;
    hipMalloc(&out, sizeof(float) * blocks * threads);
    hipMalloc(&cyc, 8);
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(cyc, 0, 8);
        hipLaunchKernelGGL((k<NV, MODE>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 0.999f, 0.001f);
        hipDeviceSynchronize();
    }
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    double waves = (double)blocks * threads / 64;
    printf("%-44s waves/SIMD=%d NV=%2d : %7.1f cycles/step\n", label, threads / 256, NV, (double)h / waves / (iters * 8.0));
    hipFree(out);
    hipFree(cyc);
}

#define SWEEP(MODE, label)                                                          \
    for (int t = 256; t <= 512; t += 256) {                                         \
        run<0, MODE>(t, label); run<4, MODE>(t, label); run<8, MODE>(t, label);     \
        run<10, MODE>(t, label); run<12, MODE>(t, label); run<16, MODE>(t, label);  \
        run<24, MODE>(t, label);                                                    \
    }

int main() {
    SWEEP(0, "1 MFMA + NV independent v_fma")
    SWEEP(1, "1 MFMA + NV dependent v_fma")
    SWEEP(2, "1 MFMA + NV independent v_exp")
    SWEEP(3, "no MFMA, NV independent v_fma")
    SWEEP(4, "2 independent MFMA + NV independent v_fma")
    return 0;
}
