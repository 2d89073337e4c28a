// Microbenchmark behind DESIGN.md section 3.2 "what bounds the projection kernels": the inner loop of featproj_kernel<8,1> with everything
// but its LDS operand reads and matrix instructions removed (no global loads, no DMA, no stores, no barrier).
//   Per k-step a wave reads two 1 KiB A fragments (weights hi, lo: ds_read_b128 x 2) from LDS and issues three v_mfma_f32_32x32x16_f16
//   against B fragments that live in its registers (the features of its 32 samples), with one k-step of operand look-ahead as in the kernel.
//   8 waves per workgroup (2 per SIMD), one workgroup per CU, every CU busy.
// MODE 0: the kernel's loop (2 LDS reads + 3 MFMAs per step)        -> what the LDS port allows the matrix pipe to reach
// MODE 1: operands from registers (no LDS reads)                    -> the matrix pipe alone (should be ~100 %)
// MODE 2: one LDS read + 3 MFMAs per step                           -> half the LDS traffic
// MODE 3: 2 LDS reads + 6 MFMAs per step (a fragment used for two sample tiles; needs twice the feature registers) -> the variant DESIGN names
// MODE 4: the kernel's loop with the three products in THREE accumulators (the kernel adds hi.lo and lo.hi into one: a dependent pair)
// Output: cycles per k-step per wave, and matrix-pipe utilisation = (MFMAs per step x 32 cycles x 2 waves per SIMD) / (cycles per step).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fpb tools/micro/featproj_lds_bound.hip && /tmp/fpb
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
#define MFMA_H(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
constexpr int KSTEPS = 16;

__device__ __forceinline__ void lds_read(h8 &d, unsigned addr, int off) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(off));
}

// (the look-ahead reads need compile-time offsets: the k-steps are unrolled by template recursion)
template <int MODE, int S>
__device__ __forceinline__ void step(unsigned wl, h8 &ah, h8 &al, const h8 (&bh)[KSTEPS], const h8 (&bl)[KSTEPS], f32x16 &acc1, f32x16 &acc2,
                                     f32x16 &acc3, f32x16 &acc4) {
    if (MODE != 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah), "+v"(al) :: "memory");
    h8 nah = ah, nal = al;
    if constexpr (MODE != 1 && S + 1 < KSTEPS) {
        lds_read(nah, wl, (S + 1) * 2048);
        if (MODE != 2) lds_read(nal, wl, (S + 1) * 2048 + 1024); else nal = nah;
    }
    __builtin_amdgcn_sched_barrier(0);
    acc1 = MFMA_H(ah, bh[S], acc1);
    acc2 = MFMA_H(ah, bl[S], acc2);
    if (MODE == 4) acc3 = MFMA_H(al, bh[S], acc3);           // third accumulator: no matrix instruction waits for the one in front of it
    else acc2 = MFMA_H(al, bh[S], acc2);
    if (MODE == 3) {
        acc3 = MFMA_H(ah, bl[S], acc3);
        acc4 = MFMA_H(ah, bh[S], acc4);
        acc4 = MFMA_H(al, bl[S], acc4);
    }
    ah = nah; al = nal;
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (S + 1 < KSTEPS) step<MODE, S + 1>(wl, ah, al, bh, bl, acc1, acc2, acc3, acc4);
}

template <int MODE>
__global__ __launch_bounds__(512) void loop_kernel(float *out, unsigned long long *cyc, int tiles) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 2 * KSTEPS * 512; i += 512) lds[i] = 1e-3f * (float)(i & 1023);
    __syncthreads();
    h8 bh[KSTEPS], bl[KSTEPS];
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) { bh[s][j] = (_Float16)(0.01f * (s + j) + 1e-3f * lane); bl[s][j] = (_Float16)(1e-3f * (s - j)); }
    f32x16 acc1, acc2, acc3, acc4;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[r] = acc2[r] = acc3[r] = acc4[r] = 0.f;
    const unsigned base = (unsigned)(size_t)(const __attribute__((address_space(3))) float *)lds + 16u * lane;
    __syncthreads();
    const unsigned long long t0 = clock64();
    for (int t = 0; t < tiles; ++t) {
        const unsigned wl = base + (t & 1) * (KSTEPS * 2048);
        h8 ah, al;
        if (MODE != 1) { lds_read(ah, wl, 0); if (MODE != 2) lds_read(al, wl, 1024); else al = ah; }
        else { ah = bh[t & 15]; al = bl[(t + 3) & 15]; }
        step<MODE, 0>(wl, ah, al, bh, bl, acc1, acc2, acc3, acc4);
    }
    const unsigned long long t1 = clock64();
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) sum += acc1[r] + acc2[r] + acc3[r] + acc4[r];
    out[blockIdx.x * 512 + threadIdx.x] = sum;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char *what, int mfmas_per_step) {
    int dev = 0;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, dev);
    const int grid = prop.multiProcessorCount, tiles = 2000;
    float *out;
    unsigned long long *cyc;
    hipMalloc(&out, sizeof(float) * grid * 512);
    hipMalloc(&cyc, sizeof(unsigned long long) * grid);
    const size_t lds_bytes = sizeof(float) * 2 * KSTEPS * 512;          // two 32 KiB tile buffers, as the kernel has
    hipFuncSetAttribute(reinterpret_cast<const void *>(loop_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL(loop_kernel<MODE>, dim3(grid), dim3(512), lds_bytes, 0, out, cyc, tiles);      // warm-up
    hipEventRecord(a);
    hipLaunchKernelGGL(loop_kernel<MODE>, dim3(grid), dim3(512), lds_bytes, 0, out, cyc, tiles);
    hipEventRecord(b);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    unsigned long long h[1024];
    hipMemcpy(h, cyc, sizeof(unsigned long long) * grid, hipMemcpyDeviceToHost);
    double mean = 0.0;
    for (int i = 0; i < grid; ++i) mean += (double)h[i];
    mean /= grid;
    // clock64() counter ticks per k-step (mean over the workgroups' wave 0) beside the wall time: if the counter runs at the shader clock,
    // a SIMD whose two waves keep the matrix pipe saturated needs 2 x mfmas_per_step x 32 ticks per step
    const double us_per_step = ms * 1e3 / ((double)tiles * KSTEPS);
    const double ticks = mean / ((double)tiles * KSTEPS);
    printf("%-80s %7.3f ms  %7.2f ns / k-step  %7.1f clock64 ticks / k-step  (matrix-bound: %d)  implied %.2f GHz\n", what, ms, us_per_step * 1e3,
           ticks, 2 * mfmas_per_step * 32, ticks / (us_per_step * 1e3));
    hipFree(out);
    hipFree(cyc);
}

int main() {
    run<1>("MODE 1: matrix instructions only, operands in registers", 3);
    run<0>("MODE 0: the kernel's loop: 2 LDS fragment reads + 3 MFMAs", 3);
    run<2>("MODE 2: 1 LDS fragment read + 3 MFMAs", 3);
    run<3>("MODE 3: 2 LDS fragment reads + 6 MFMAs (fragment shared by two sample tiles)", 6);
    run<4>("MODE 4: 2 LDS fragment reads + 3 MFMAs into three accumulators", 3);
    printf("matrix-pipe share of MODE 0 = t(MODE 1) / t(MODE 0); of MODE 3 = 2 t(MODE 1) / t(MODE 3)\n");
    return 0;
}
