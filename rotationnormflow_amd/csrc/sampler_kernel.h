// sampler_kernel.h -- matrix-Fisher sampler on the device (utils/fisher.py:117-207,234-243).
//
// R ~ MF(A), A = U diag(S) V^T (proper SVD):  q ~ Bingham(diag(0, 2(S1+S2), 2(S0+S2), 2(S0+S1))) on S^3 by rejection from
// the angular-central-Gaussian envelope ACG(Omega = I + 2 Lambda / b), b = 1.5, bound M* = e^{-(4-b)/2} (4/b)^2
// (fisher.py:117-207), then R = U R(q) V^T.  The reference draws 8n candidates per batch with torch's generator and keeps
// the first n accepted; rejection sampling is exact whatever the batching, so here every output sample runs its own
// accept/reject loop on a counter-based Philox4x32-10 stream (counter = sample index, row, attempt; key = seed).
// Parity with the reference is therefore STATISTICAL, checked in tests/test_gpu_sampler.py.
#pragma once
#include <hip/hip_runtime.h>

#include "so3_math.h"

namespace rnf {

struct Philox {
    unsigned k0, k1;
    __device__ __forceinline__ void round(unsigned (&c)[4], unsigned a, unsigned b) const {
        const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ a, n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c[3] ^ b, n3 = (unsigned)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    }
    __device__ __forceinline__ void operator()(unsigned (&c)[4]) const {
        unsigned a = k0, b = k1;
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            round(c, a, b);
            a += 0x9E3779B9u;
            b += 0xBB67AE85u;
        }
    }
};

__device__ __forceinline__ float u01(unsigned x) { return ((x >> 8) + 0.5f) * (1.0f / 16777216.0f); }   // (0, 1)

__global__ void fisher_sample_kernel(const float *U, const float *V, const float *lam, long long B, long long n,
                                     unsigned long long seed, float *out, int *fail_flag) {
    const long long total = B * n;
    const Philox rng{(unsigned)seed, (unsigned)(seed >> 32)};
    const float b = 1.5f, m_star = expf(-(4.0f - b) * 0.5f) * (4.0f / b) * (4.0f / b);
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const long long row = idx / n, i = idx - row * n;
        float L[4], om[4], sd[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            L[k] = lam[row * 4 + k];
            om[k] = 1.0f + 2.0f * L[k] / b;
            sd[k] = 1.0f / sqrtf(om[k]);
        }
        float q[4] = {1.f, 0.f, 0.f, 0.f};
        bool ok = false;
        for (unsigned attempt = 0; attempt < 4096u && !ok; ++attempt) {
            unsigned c0[4] = {(unsigned)i, (unsigned)(i >> 32), (unsigned)row, attempt * 2u};
            unsigned c1[4] = {(unsigned)i, (unsigned)(i >> 32), (unsigned)row, attempt * 2u + 1u};
            rng(c0);
            rng(c1);
            // four standard normals (Box-Muller) and one uniform
            const float r0 = sqrtf(-2.0f * logf(u01(c0[0]))), r1 = sqrtf(-2.0f * logf(u01(c0[2])));
            float s0, k0, s1, k1;
            sincosf(kTwoPi * u01(c0[1]), &s0, &k0);
            sincosf(kTwoPi * u01(c0[3]), &s1, &k1);
            float y[4] = {sd[0] * r0 * k0, sd[1] * r0 * s0, sd[2] * r1 * k1, sd[3] * r1 * s1};
            const float inv = 1.0f / sqrtf(y[0] * y[0] + y[1] * y[1] + y[2] * y[2] + y[3] * y[3]);
            float eb = 0.f, ea = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                y[k] *= inv;
                eb += L[k] * y[k] * y[k];
                ea += om[k] * y[k] * y[k];
            }
            const float p_bing = expf(-eb), p_acg = 1.0f / (ea * ea);
            if (u01(c1[0]) < p_bing / (m_star * p_acg)) {
                ok = true;
#pragma unroll
                for (int k = 0; k < 4; ++k) q[k] = y[k];
            }
        }
        if (!ok) atomicExch(fail_flag, 1);
        // R(q) (unit q, real part first: fisher.py:14-50), then U R V^T
        Rot Rq;
        quat_to_rot(q, 1.0f, Rq);
        const float rq[9] = {Rq.c0.x, Rq.c1.x, Rq.c2.x, Rq.c0.y, Rq.c1.y, Rq.c2.y, Rq.c0.z, Rq.c1.z, Rq.c2.z};
        const float *u = U + row * 9, *v = V + row * 9;
        float t[9];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int c = 0; c < 3; ++c) t[3 * a + c] = u[3 * a] * rq[c] + u[3 * a + 1] * rq[3 + c] + u[3 * a + 2] * rq[6 + c];
        float *o = out + idx * 9;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int c = 0; c < 3; ++c) o[3 * a + c] = t[3 * a] * v[3 * c] + t[3 * a + 1] * v[3 * c + 1] + t[3 * a + 2] * v[3 * c + 2];
    }
}

}  // namespace rnf
