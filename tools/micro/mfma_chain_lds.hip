// Microbenchmark for csrc/train_kernels.h: cycles of one 32x32 tile product = 33 dependent v_mfma_f32_32x32x2_f32 whose B operands
// come from LDS in two batches of 16 reads (MODE 0), from registers (MODE 1), or with the chain split over two accumulators
// (MODE 2, B from LDS), or with a ReLU on every B element right before its MFMA (MODE 3).  4 waves per workgroup (one per SIMD), like the backward kernel.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/chain tools/micro/mfma_chain_lds.hip && /tmp/chain
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

template <int MODE, int LIVE = 0>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, int iters, int lds_off) {
    extern __shared__ float lds_all[];
    float *lds = lds_all + lds_off;
    float live[LIVE + 1];                                   // LIVE extra values kept in registers across the timed loop
#pragma unroll
    for (int i = 0; i < LIVE; ++i) live[i] = out[(threadIdx.x + 7 * i) & 1023];
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    for (int i = threadIdx.x; i < 64 * 65; i += 256) lds[i] = 0.001f * i;
    __syncthreads();
    float a[32];
    for (int i = 0; i < 32; ++i) a[i] = 1.0f + 1e-3f * (i + lane);
    f32x16 acc, acc2;
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
    unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float bv[16];
#pragma unroll
            for (int m = 0; m < 16; ++m) bv[m] = MODE == 1 ? a[(m + it) & 31] : lds[(32 * h + 16 * half + m) * 65 + j + (it & 31)];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                if (MODE == 3) bv[m] = fmaxf(bv[m], 0.f);                 // ReLU between the read and the MFMA (2 VALU per step)
                if (MODE == 5) {                                          // pure chain, accumulator forced into AGPRs
                    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a[16 * half + m]), "v"(bv[m]));
                } else if (MODE == 6) {                                   // pure chain, accumulator forced into VGPRs
                    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a[16 * half + m]), "v"(bv[m]));
                } else if (MODE == 4) {                                   // the same, but every ReLU result lands in ONE register
                    float t;
                    asm volatile("v_max_f32 %0, %2, %2\n\tv_max_f32 %0, 0, %0\n\ts_nop 1\n\tv_mfma_f32_32x32x2_f32 %1, %3, %0, %1"
                                 : "=&v"(t), "+a"(acc) : "v"(bv[m]), "v"(a[16 * half + m]));
                } else if (MODE == 2 && (m & 1)) acc2 = MFMA(a[16 * half + m], bv[m], acc2);
                else acc = MFMA(a[16 * half + m], bv[m], acc);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = clock64();
    float s = 0;
#pragma unroll
    for (int i = 0; i < LIVE; ++i) s += live[i];
    for (int r = 0; r < 16; ++r) s += acc[r] + acc2[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) atomicAdd(cyc, t1 - t0);
}

template <int MODE, int LIVE = 0>
static void run(const char *label, size_t lds_bytes = 64 * 65 * 4, int lds_off = 0) {
    const int blocks = 16, iters = 1000;
    float *out;
    unsigned long long *cyc, hc = 0;
    hipMalloc(&out, sizeof(float) * blocks * 256);
    hipMemset(out, 0, sizeof(float) * blocks * 256);
    hipMalloc(&cyc, 8);
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(cyc, 0, 8);
        hipFuncSetAttribute(reinterpret_cast<const void *>(k<MODE, LIVE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        hipLaunchKernelGGL((k<MODE, LIVE>), dim3(blocks), dim3(256), lds_bytes, 0, out, cyc, iters, lds_off);
        hipDeviceSynchronize();
    }
    hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-52s : %7.1f cycles per 32-MFMA product (%5.1f per MFMA)\n", label, (double)hc / (blocks * 4) / iters, (double)hc / (blocks * 4) / iters / 32);
    hipFree(out);
    hipFree(cyc);
}

int main() {
    run<0>("B from LDS, 2 x (16 reads, 16 dependent MFMAs)");
    run<1>("B from registers");
    run<2>("B from LDS, two accumulators interleaved");
    run<3>("B from LDS, ReLU (2 VALU) before every MFMA");
    run<4>("  same, ReLU result always in one VGPR (asm)");
    run<3>("ReLU variant, 150 KB of LDS allocated, data at offset 0", 150 * 1024, 0);
    run<3>("ReLU variant, 150 KB of LDS allocated, data at 120 KB", 150 * 1024, 120 * 256);
    run<5>("pure chain, accumulator in AGPRs (asm)");
    run<6>("pure chain, accumulator in VGPRs (asm)");
    run<1, 150>("B from registers, +150 live VGPRs");
    run<1, 300>("B from registers, +300 live registers (VGPR + AGPR)");
    run<1, 420>("B from registers, +420 live registers (full 512 file)");
    return 0;
}
