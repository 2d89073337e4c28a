// Microbenchmark (round 6): issue cost of the VALU forms the inverse root finder is made of, one and two waves per SIMD:
// v_fma_f32, v_pk_fma_f32 (two fp32 FMAs per lane), v_rcp_f32, and the root finder's own mix (23 packed + 4 v_rcp per segment pair).
// clock64 runs at 100 MHz on gfx950, so the kernel is timed with HIP events and converted with the shader clock the run reports:
// prints ns per instruction per wave and the implied cycles at 2.4 GHz.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rates tools/micro/valu_issue_rates.hip && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));

// MODE 0: 16 independent v_fma_f32 chains; 1: 16 independent v_pk_fma_f32 chains; 2: 16 independent v_rcp_f32; 3: 8 v_pk_fma + 8 v_fma interleaved
template <int MODE>
__global__ __launch_bounds__(512) void k(float *out, int iters, float c1, float c2) {
    f2 v[16];
    for (int i = 0; i < 16; ++i) v[i] = f2{threadIdx.x * 0.01f + i, threadIdx.x * 0.02f + i};
    const f2 a = {c1, c1}, b = {c2, c2};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (MODE == 0) v[i].x = __builtin_fmaf(v[i].x, c1, c2);
                else if (MODE == 1) v[i] = __builtin_elementwise_fma(v[i], a, b);
                else if (MODE == 2) v[i].x = __builtin_amdgcn_rcpf(v[i].x);
                else { if (i & 1) v[i] = __builtin_elementwise_fma(v[i], a, b); else v[i].x = __builtin_fmaf(v[i].x, c1, c2); }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += v[i].x + v[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static void run(int threads, const char *label) {
    const int blocks = 256, iters = 20000;
    float *out;
    hipMalloc(&out, sizeof(float) * blocks * threads);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(threads), 0, 0, out, iters, 0.999f, 0.001f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double insts = (double)iters * 64;                    // per wave
    const int wps = threads / 256;
    const double ns_per_inst_simd = ms * 1e6 / (insts * wps);   // SIMD time per instruction (waves of a SIMD share the issue port)
    printf("%-34s waves/SIMD=%d : %6.3f ms  %5.2f ns per instruction per SIMD  = %4.1f cycles at 2.4 GHz\n", label, wps, ms, ns_per_inst_simd, ns_per_inst_simd * 2.4);
    hipFree(out);
}

int main() {
    for (int t = 256; t <= 512; t += 256) {
        run<0>(t, "v_fma_f32 (16 chains)");
        run<1>(t, "v_pk_fma_f32 (16 chains)");
        run<2>(t, "v_rcp_f32 (16 chains)");
        run<3>(t, "v_pk_fma / v_fma alternating");
    }
    return 0;
}
