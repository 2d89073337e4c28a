// so3_math.h -- per-sample device math of the flow layers (gfx950, one rotation per lane pair).
// Citations are to the reference tree (flow/..., utils/...) and to SURVEY.md Appendix A.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#define RNF_HD __host__ __device__ __forceinline__

namespace rnf {

constexpr float kPi = 3.14159265358979323846f;
constexpr float kTwoPi = 6.28318530717958647692f;

// ---- 1-ulp hardware transcendentals (v_rcp_f32 / v_rsq_f32 / v_sqrt_f32 / v_exp_f32 / v_log_f32) --------------
// The host definitions exist only so that tests can exercise the surrounding algebra on the CPU.
#if defined(__HIP_DEVICE_COMPILE__)
RNF_HD float hw_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
RNF_HD float hw_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
RNF_HD float hw_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
RNF_HD float hw_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
RNF_HD float hw_log2(float x) { return __builtin_amdgcn_logf(x); }
#else
RNF_HD float hw_rcp(float x) { return 1.0f / x; }
RNF_HD float hw_rsq(float x) { return 1.0f / sqrtf(x); }
RNF_HD float hw_sqrt(float x) { return sqrtf(x); }
RNF_HD float hw_exp2(float x) { return exp2f(x); }
RNF_HD float hw_log2(float x) { return log2f(x); }
#endif

// torch.nn.functional.softplus(beta=1, threshold=20) (flow/mobiusflow.py:69), branch free:
//   softplus(x) = max(x, 0) + log1p(exp(-|x|));  e = exp(-|x|) in (0, 1];  u = fl(1 + e);
//   log1p(e) = log(u) + (e - (u - 1)) / u   (the second term restores the bits of e lost in forming u).
// For x > 20 the log1p term is < 2.1e-9 < ulp(x)/2, so the result rounds to x exactly like the reference's threshold.
RNF_HD float softplus(float x) {
    const float ax = fabsf(x);
    const float p = ax * 1.44269502162933349609375f;                    // log2(e) split hi + lo: keeps exp(-|x|) at ~1 ulp
    const float r = fmaf(ax, 1.92596299112661746e-8f, fmaf(ax, 1.44269502162933349609375f, -p));
    float e = hw_exp2(-p);
    e = fmaf(-0.693147180559945309f * e, r, e);
    const float u = 1.0f + e;
    // (e - (u - 1)) <= 2^-24 is the rounding residue of u; 1/u in [1/2, 1] only needs ~10% accuracy there: 1 - e/2
    const float l = fmaf(hw_log2(u), 0.693147180559945309f, (e - (u - 1.0f)) * fmaf(-0.5f, e, 1.0f));
    return fmaxf(x, 0.0f) + l;
}

// max(x, 0) as an integer max on the bit pattern (negative floats are negative integers): ONE instruction on the device, where
// fmaxf(x, 0) on a matrix-instruction result costs two (the compiler first quiets a possible signalling NaN with v_max x, x)
RNF_HD float relu_bits(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __int_as_float(max(__float_as_int(x), 0));
#else
    return x > 0.0f ? x : 0.0f;
#endif
}

// softplus without the argument split and the log1p residue: those terms are < 1e-8 absolute on a weight that is then divided by the sum
// of K such weights (the split-precision kernels use this form in both directions, so forward and inverse see the same weights)
RNF_HD float softplus_lean(float x) {
    const float e = hw_exp2(-1.44269504088896341f * fabsf(x));
    return fmaf(hw_log2(1.0f + e), 0.693147180559945309f, relu_bits(x));
}

// atan2(y, x) mapped to [0, 2pi) (the wrap of flow/mobiusflow.py:98-99 folded in).  Octant reduction to a in [0,1],
// then the classic single-precision arctangent: |t| <= tan(pi/8) via t = (a-1)/(a+1), odd degree-9 minimax (~2 ulp).
// first half: octant + tan(pi/8) reduction -> t with |t| <= tan(pi/8) and the "added pi/4" flag
RNF_HD void atan_reduce(float y, float x, float &t, bool &big) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float a = mn * hw_rcp(mx);
    big = a > 0.414213562373095f;
    t = big ? (a - 1.0f) * hw_rcp(a + 1.0f) : a;
}
// second half: polynomial + octant/quadrant reconstruction into [0, 2pi)
RNF_HD float atan_finish(float y, float x, float t, bool big) {
    const float z = t * t;
    float p = fmaf(fmaf(fmaf(8.05374449538e-2f, z, -1.38776856032e-1f), z, 1.99777106478e-1f), z, -3.33329491539e-1f);
    p = fmaf(p * z, t, t);
    p += big ? 0.785398163397448310f : 0.0f;
    p = fabsf(y) > fabsf(x) ? 1.57079632679489662f - p : p;
    p = x < 0.0f ? 3.14159265358979324f - p : p;
    p = y < 0.0f ? 6.28318530717958648f - p : p;
    return p;
}
RNF_HD float angle_0_2pi(float y, float x) {
    float t;
    bool big;
    atan_reduce(y, x, t, big);
    return atan_finish(y, x, t, big);
}

// sin and cos for |x| <= ~16 (the layer only needs [0, 2pi)): quadrant reduction with a 3-term Cody-Waite split of
// pi/2 and the classic single-precision minimax polynomials on [-pi/4, pi/4] (~1 ulp).  Replaces libm sincosf, whose
// huge-argument path costs a private-memory (scratch) table on gfx950.
RNF_HD void sincos_small(float x, float &sn, float &cs) {
    const float k = rintf(x * 0.636619772367581343f);            // x * 2/pi
    float r = fmaf(-k, 1.5703125f, x);
    r = fmaf(-k, 4.837512969970703125e-4f, r);
    r = fmaf(-k, 7.54978995489188216e-8f, r);
    const float r2 = r * r;
    float ps = fmaf(fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f);
    ps = fmaf(ps * r2, r, r);
    float pc = fmaf(fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2, 4.166664568298827e-2f);
    pc = fmaf(pc * r2, r2, fmaf(-0.5f, r2, 1.0f));
    const int q = (int)k;
    const float s1 = (q & 1) ? pc : ps;
    const float c1 = (q & 1) ? ps : pc;
    sn = (q & 2) ? -s1 : s1;
    cs = ((q + 1) & 2) ? -c1 : c1;
}

// sin and cos of theta in [pi/2, 3 pi/2] -- the only band the inverse pass's root finder visits (BinFind's bracket, flow/mobiusflow.py:196-224)
// -- without any quadrant logic (round 6): x = theta - pi in [-pi/2, pi/2] (pi as hi + lo: the subtraction of hi is exact there), the Taylor
// polynomials of degree 13 / 14 on x itself (truncation < 1e-9), sin theta = -sin x, cos theta = -cos x.  Max error 1.2e-7 (sincos_small:
// 0.9e-7); 17 VALU instead of 25, three evaluations per Moebius layer.
RNF_HD void sincos_pi_band(float theta, float &sn, float &cs) {
    const float x = (theta - 3.14159274101257324f) + 8.74227765734758577e-8f;
    const float z = x * x;
    float p = fmaf(1.60590438e-10f, z, -2.50521084e-8f);
    p = fmaf(p, z, 2.75573192e-6f);
    p = fmaf(p, z, -1.98412698e-4f);
    p = fmaf(p, z, 8.33333333e-3f);
    p = fmaf(p, z, -1.66666667e-1f);
    sn = -fmaf(p * z, x, x);
    float q = fmaf(-1.14707456e-11f, z, 2.08767570e-9f);
    q = fmaf(q, z, -2.75573192e-7f);
    q = fmaf(q, z, 2.48015873e-5f);
    q = fmaf(q, z, -1.38888889e-3f);
    q = fmaf(q, z, 4.16666667e-2f);
    q = fmaf(q, z, -0.5f);
    cs = -fmaf(q, z, 1.0f);
}

// sin and cos of 2 h for |h| <= pi/4 without any quadrant logic: the minimax polynomials above on h itself, then the double-angle
// formulas.  The forward Moebius layer's transformed angle is pi + 2 h with h = sum wt_k atan(t_k), |atan(t_k)| <= atan(0.98) = 0.775 < pi/4
// and the weights summing to 1, so the reduction of sincos_small (rint, three Cody-Waite steps, four selects) is dead weight there.
RNF_HD void sincos_twice(float h, float &sn, float &cs) {
    const float r2 = h * h;
    float ps = fmaf(fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f);
    ps = fmaf(ps * r2, h, h);                                    // sin h
    float pc = fmaf(fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2, 4.166664568298827e-2f);
    pc = fmaf(pc * r2, r2, fmaf(-0.5f, r2, 1.0f));               // cos h
    sn = (ps + ps) * pc;
    cs = fmaf(-(ps + ps), ps, 1.0f);
}

// The rotation state: three column vectors kept as SSA vector values (never an indexable array: hipcc turns a
// select chain over array elements into a scratch-memory lookup).  Columns, not rows, are what the coupling layers
// read and write (flow/mobiusflow.py:50-51,80-83).
typedef float v3f __attribute__((ext_vector_type(3)));
struct Rot {
    v3f c0, c1, c2;
};
// Operands BY VALUE on purpose: with a pointer/reference argument the optimizer (which simplifies this function before
// it is inlined) folds the three conditional loads into ONE load at a computed address, and the whole rotation state
// is demoted from registers to scratch memory.  The same happens with `p == 0 ? R.c0 : ...` (an lvalue conditional).
RNF_HD v3f pick_col(v3f c0, v3f c1, v3f c2, int p) {
    v3f out = c2;
    if (p == 0) out = c0;
    if (p == 1) out = c1;
    return out;
}
RNF_HD v3f get_col(const Rot &R, int p) { return pick_col(R.c0, R.c1, R.c2, p); }
RNF_HD void set_col(Rot &R, int p, v3f c) {
    if (p == 0) R.c0 = c;
    if (p == 1) R.c1 = c;
    if (p == 2) R.c2 = c;
}
RNF_HD float dot3(v3f a, v3f b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
RNF_HD v3f cross3(v3f a, v3f b) {
    v3f c;
    c.x = a.y * b.z - a.z * b.y;
    c.y = a.z * b.x - a.x * b.z;
    c.z = a.x * b.y - a.y * b.x;
    return c;
}
RNF_HD v3f normalize3(v3f a) { return a * hw_rsq(dot3(a, a)); }

// In-plane frame of a Moebius layer (flow/mobiusflow.py:64-67,148-151): r = -x/|x|, v = (y x r)/|y x r|.
// Everything the layer does lives in span(r, v) (the plane orthogonal to y), so segments are handled in 2-D
// coordinates (a.r, a.v); the projection (I - y y^T) w of mobiusflow.py:62-63 is implied by taking w.r and w.v.
struct Frame {
    v3f r, v;
};
RNF_HD Frame make_frame(v3f x, v3f y) {
    Frame f;
    f.r = x * (-hw_rsq(dot3(x, x)));
    f.v = normalize3(cross3(y, f.r));
    return f;
}

// One segment's squashed centre in frame coordinates: w <- 0.7/(1+|w|) w (flow/mobiusflow.py:72).
RNF_HD void squash_center(float w0, float w1, float w2, const Frame &f, float &ur, float &uv) {
    float wr = fmaf(w2, f.r.z, fmaf(w1, f.r.y, w0 * f.r.x));
    float wv = fmaf(w2, f.v.z, fmaf(w1, f.v.y, w0 * f.v.x));
    float sc = 0.7f * hw_rcp(1.0f + hw_sqrt(fmaf(wv, wv, wr * wr)));
    ur = wr * sc;
    uv = wv * sc;
}

// arctangent on |t| <= 1 with no range reduction: t * P(t^2), degree-8 minimax fit (max abs error 1e-7 in fp32 Horner).
RNF_HD float atan_unit(float t) {
    const float z = t * t;
    float p = fmaf(2.456724578e-03f, z, -1.440135792e-02f);
    p = fmaf(p, z, 3.978122362e-02f);
    p = fmaf(p, z, -7.234857378e-02f);
    p = fmaf(p, z, 1.049894609e-01f);
    p = fmaf(p, z, -1.416122920e-01f);
    p = fmaf(p, z, 1.998590677e-01f);
    p = fmaf(p, z, -3.333259703e-01f);
    p = fmaf(p, z, 9.999998864e-01f);
    return p * t;
}

// Moebius map of the unit point z = (cs, sn) = e^{i theta} about centre u = (ur, uv), |u| < 0.7 (flow/mobiusflow.py:17-24):
// in complex form h(z) = (1-|u|^2)(z-u)/|z-u|^2 - u = conj(z) (z-u)/conj(z-u), hence
//     arg h = 2 arg(z - u) - theta = theta + 2 atan(-b / (1 - a)),   (a, b) = u conj(z),   1 - a > 0.3,
// and |t| = |b / (1 - a)| <= tan(asin 0.7) < 1: one quadrant, no octant logic, no wrap (theta in [pi/2, 3pi/2] => phi in
// (0, 2pi), which is the range of the reference's wrapped atan2, mobiusflow.py:94-99).  c = (1-|u|^2)/|z-u|^2 is the
// segment's |dh/dtheta| (identity checked in tests/test_oracle_golden.py).
RNF_HD void mobius_angle(float cs, float sn, float theta, float ur, float uv, float &phi, float &c) {
    const float a = fmaf(uv, sn, ur * cs);
    const float b = fmaf(uv, cs, -ur * sn);
    const float e1 = 1.0f - a;
    phi = fmaf(2.0f, atan_unit(-b * hw_rcp(e1)), theta);
    const float u2 = fmaf(uv, uv, ur * ur);
    c = (1.0f - u2) * hw_rcp(fmaf(b, b, e1 * e1));
}

// ---- one forward segment, cut into 8 slices of ~10 VALU issue slots --------------------------------------------------
// Same arithmetic as squash_center -> mobius_angle -> softplus -> accumulate, but exposed stage by stage so
// the forward kernel can hand-interleave ONE slice behind every MFMA of the next fc_last tile (a 64-cycle matrix
// instruction covers ~10 VALU issue slots of the two waves sharing a SIMD; unbalanced slices leave the pipe idle).
struct SegState {
    float wr, wv, sc, ur, uv, b, e1, u2, d2, c, t, z, p, pp, r, e, u, lg;
};
// zc, zs, ztheta: the layer's input point z = (cos, sin) in frame coordinates and its angle (== pi up to rounding)
template <int STAGE>
RNF_HD void seg_stage(SegState &g, float s_raw, float w0, float w1, float w2, const Frame &f, float zc, float zs, float ztheta,
                      float &S, float &A, float &J) {
    if constexpr (STAGE == 0) {
        g.wr = fmaf(w2, f.r.z, fmaf(w1, f.r.y, w0 * f.r.x));
        g.wv = fmaf(w2, f.v.z, fmaf(w1, f.v.y, w0 * f.v.x));
        g.sc = fmaf(g.wv, g.wv, g.wr * g.wr);
    } else if constexpr (STAGE == 1) {
        g.sc = 0.7f * hw_rcp(1.0f + hw_sqrt(g.sc));
        g.ur = g.wr * g.sc;
        g.uv = g.wv * g.sc;
    } else if constexpr (STAGE == 2) {
        const float a = fmaf(g.uv, zs, g.ur * zc);
        g.b = fmaf(g.uv, zc, -g.ur * zs);
        g.e1 = 1.0f - a;
        g.u2 = fmaf(g.uv, g.uv, g.ur * g.ur);
        g.d2 = fmaf(g.b, g.b, g.e1 * g.e1);
    } else if constexpr (STAGE == 3) {
        g.c = (1.0f - g.u2) * hw_rcp(g.d2);
        g.t = -g.b * hw_rcp(g.e1);
        g.z = g.t * g.t;
        g.p = fmaf(2.456724578e-03f, g.z, -1.440135792e-02f);
    } else if constexpr (STAGE == 4) {
        float p = fmaf(g.p, g.z, 3.978122362e-02f);
        p = fmaf(p, g.z, -7.234857378e-02f);
        p = fmaf(p, g.z, 1.049894609e-01f);
        p = fmaf(p, g.z, -1.416122920e-01f);
        p = fmaf(p, g.z, 1.998590677e-01f);
        p = fmaf(p, g.z, -3.333259703e-01f);
        p = fmaf(p, g.z, 9.999998864e-01f);
        g.p = fmaf(2.0f, p * g.t, ztheta);                                   // phi
    } else if constexpr (STAGE == 5) {
        const float ax = fabsf(s_raw);
        g.pp = ax * 1.44269502162933349609375f;
        g.r = fmaf(ax, 1.92596299112661746e-8f, fmaf(ax, 1.44269502162933349609375f, -g.pp));
        float e = hw_exp2(-g.pp);
        g.e = fmaf(-0.693147180559945309f * e, g.r, e);
    } else if constexpr (STAGE == 6) {
        g.u = 1.0f + g.e;
        g.lg = hw_log2(g.u);
        g.r = (g.e - (g.u - 1.0f)) * fmaf(-0.5f, g.e, 1.0f);
    } else {
        const float sp = fmaxf(s_raw, 0.0f) + fmaf(g.lg, 0.693147180559945309f, g.r);
        S += sp;
        A = fmaf(sp, g.p, A);
        J = fmaf(sp, g.c, J);
    }
}
RNF_HD void segment_full(float s_raw, float w0, float w1, float w2, const Frame &f, float zc, float zs, float ztheta, float &S,
                         float &A, float &J) {
    SegState g;
    seg_stage<0>(g, s_raw, w0, w1, w2, f, zc, zs, ztheta, S, A, J);
    seg_stage<1>(g, s_raw, w0, w1, w2, f, zc, zs, ztheta, S, A, J);
    seg_stage<2>(g, s_raw, w0, w1, w2, f, zc, zs, ztheta, S, A, J);
    seg_stage<3>(g, s_raw, w0, w1, w2, f, zc, zs, ztheta, S, A, J);
    seg_stage<4>(g, s_raw, w0, w1, w2, f, zc, zs, ztheta, S, A, J);
    seg_stage<5>(g, s_raw, w0, w1, w2, f, zc, zs, ztheta, S, A, J);
    seg_stage<6>(g, s_raw, w0, w1, w2, f, zc, zs, ztheta, S, A, J);
    seg_stage<7>(g, s_raw, w0, w1, w2, f, zc, zs, ztheta, S, A, J);
}

// The forward segment with the layer's input point fixed at z = (-1, 0), theta = pi: x expressed in its own frame IS (-1, 0)
// (r = -x/|x|, v orthogonal to x; the reference's atan2(x.v, x.r) returns pi up to rounding, mobiusflow.py:104-108), so
// a = -ur, b = -uv and phi = pi + 2 atan(uv / (1 + ur)).  softplus without the argument-splitting / log1p-residue terms of
// softplus(): their contribution is below 1e-8 absolute on a term that is then divided by the sum of K such terms.
// 44 VALU instructions (6 transcendental) instead of 57; used by the split-precision forward kernel, which is VALU-issue bound.
// HALF: A accumulates sp * atan(t) only -- the caller adds the constant part once per layer (sum sp * phi = pi * S + 2 * that).
template <bool HALF = false>
RNF_HD void segment_fwd_pi(float s_raw, float w0, float w1, float w2, const Frame &f, float &S, float &A, float &J) {
    const float wr = fmaf(w2, f.r.z, fmaf(w1, f.r.y, w0 * f.r.x));
    const float wv = fmaf(w2, f.v.z, fmaf(w1, f.v.y, w0 * f.v.x));
    const float sc = 0.7f * hw_rcp(1.0f + hw_sqrt(fmaf(wv, wv, wr * wr)));
    const float ur = wr * sc, uv = wv * sc;
    const float e1 = 1.0f + ur;
    const float t = uv * hw_rcp(e1);
    const float c = (1.0f - fmaf(uv, uv, ur * ur)) * hw_rcp(fmaf(uv, uv, e1 * e1));
    const float at = atan_unit(t);
    const float phi = HALF ? at : fmaf(2.0f, at, kPi);
    // the full softplus (log1p residue kept): these are the exact-fp32 kernels a guarded split-precision launch is re-run on when its
    // weights were too small for fl(1 + e) -- the lean form would round them just the same
    const float sp = softplus(s_raw);
    S += sp;
    A = fmaf(sp, phi, A);
    J = fmaf(sp, c, J);
}

// segment_fwd_pi<true> cut into three slices of ~15 VALU instructions, one per matrix instruction of the next fc_last tile
// (flow_kernels.h tile_pipe_h; whether the slices really sit between the matrix instructions is a build switch there).
struct SegPi {
    float ur, uv, t, c, p, z;
};
template <int STAGE>
RNF_HD void seg_pi_stage(SegPi &g, float s_raw, float w0, float w1, float w2, const Frame &f, float &S, float &A, float &J) {
    if constexpr (STAGE == 0) {
        const float wr = fmaf(w2, f.r.z, fmaf(w1, f.r.y, w0 * f.r.x));
        const float wv = fmaf(w2, f.v.z, fmaf(w1, f.v.y, w0 * f.v.x));
        const float sc = 0.7f * hw_rcp(1.0f + hw_sqrt(fmaf(wv, wv, wr * wr)));
        g.ur = wr * sc;
        g.uv = wv * sc;
    } else if constexpr (STAGE == 1) {
        const float e1 = 1.0f + g.ur;
        g.t = g.uv * hw_rcp(e1);
        g.c = (1.0f - fmaf(g.uv, g.uv, g.ur * g.ur)) * hw_rcp(fmaf(g.uv, g.uv, e1 * e1));
        g.z = g.t * g.t;
        float p = fmaf(2.456724578e-03f, g.z, -1.440135792e-02f);
        p = fmaf(p, g.z, 3.978122362e-02f);
        p = fmaf(p, g.z, -7.234857378e-02f);
        g.p = fmaf(p, g.z, 1.049894609e-01f);
    } else {
        float p = fmaf(g.p, g.z, -1.416122920e-01f);
        p = fmaf(p, g.z, 1.998590677e-01f);
        p = fmaf(p, g.z, -3.333259703e-01f);
        p = fmaf(p, g.z, 9.999998864e-01f);
        const float e = hw_exp2(-1.44269504088896341f * fabsf(s_raw));
        const float sp = fmaf(hw_log2(1.0f + e), 0.693147180559945309f, relu_bits(s_raw));
        S += sp;
        A = fmaf(sp, p * g.t, A);                  // HALF convention: sp * atan(t)
        J = fmaf(sp, g.c, J);
    }
}

// ---- the forward segment WITHOUT the squash reciprocal (round 2) --------------------------------------------------------------
// With a = 0.7 w.r, b = 0.7 w.v (the frame handed in is PRE-SCALED by 0.7: Frame7), n2 = a^2 + b^2 = (0.7 |w|)^2 and D = 1 + |w|
// the squashed centre of flow/mobiusflow.py:72 is u = (a, b) / D, and everything the segment needs is a ratio in which D cancels:
//     t = uv / (1 + ur)                       = b / (D + a)
//     c = (1 - |u|^2) / (uv^2 + (1 + ur)^2)   = (D^2 - n2) / ((D + a)^2 + b^2)
// so 0.7 / (1 + |w|) is never formed: 33 VALU instructions with 5 transcendentals (sqrt, 2 rcp, exp2, log2) instead of 44 with 6,
// and D + a >= 1 + 0.3 |w| >= 1 is better conditioned than 1 + ur >= 0.3.  HALF convention (A accumulates sp * atan(t)).
constexpr float kSquash = 0.7f;                         // flow/mobiusflow.py:72
constexpr float kInvSquash = 1.0f / 0.7f;
RNF_HD Frame scale_frame(const Frame &f, float k) { return Frame{f.r * k, f.v * k}; }
// make_frame with both vectors of length k, the factor folded into the two normalisations (4 instructions fewer than scale_frame(make_frame))
RNF_HD Frame make_frame_scaled(v3f x, v3f y, float k) {
    Frame f;
    f.r = x * (-k * hw_rsq(dot3(x, x)));
    const v3f cr = cross3(y, f.r);                                // |cr| = k |y x r^|
    f.v = cr * (k * hw_rsq(dot3(cr, cr)));
    return f;
}

struct SegS7 {
    float e, b, bb, num, t, c, z, p;     // four values cross each stage boundary (the pipelined tile keeps four segments in flight)
};
// log2(1 + 2^x) that is finite for every finite x and keeps its RELATIVE accuracy for x << 0 (the kernels that cannot rely on the range
// guard's exact-fp32 re-run: training passes, device-packed blobs, RNF_GUARD=0, C-ABI callers without fallback images): the
// exponent is clamped and the result floored at x (the reference's softplus is the identity there), and below 2^x = 2^-6 -- where
// fl(1 + e) would round e to a few bits -- the series log2 e * (e - e^2/2 + e^3/3) takes over (truncation 2^-20 relative at the switch).
RNF_HD float softplus2_safe(float x) {
    const float e = hw_exp2(fminf(x, 126.0f));
    const float big = fmaxf(x, hw_log2(1.0f + e));              // x >= 126: the clamped form returns 126, the function is x (log2(1 + 2^x) > x always)
    const float small = 1.44269504088896341f * e * fmaf(e, fmaf(e, 0.333333333f, -0.5f), 1.0f);
    return e < 0.015625f ? small : big;
}

// The one-piece form of the lean kernels, overflow-proof in ONE more instruction (round 4: weights the reference's own training produced
// reach s = 250, s log2 e = 360 -- tests/golden/trained_cond4 -- where 2^x is inf and round 3 re-ran the whole launch on the exact-fp32
// kernels): t = log2(1 + 2^x) is >= x everywhere, equals x (to fp32 rounding) from x ~ 25 on and is +inf from x = 128 on, so the median
// of (t, x, 127) is t below 127 and x above.  exp2 + add + log2 + med3 = 4 VALU.  (What remains for the launch guard is the all-weights-
// tiny corner: flow_kernels.h kMinWeightSum.)
RNF_HD float softplus2_lean(float x) {
    const float t = hw_log2(1.0f + hw_exp2(x));
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_fmed3f(t, x, 127.0f);
#else
    return x >= 127.0f ? x : t;
#endif
}

template <int STAGE, bool SAFE = true>
RNF_HD void seg_s7_stage(SegS7 &g, float s_raw, float w0, float w1, float w2, const Frame &f7, float &S, float &A, float &J) {
    if constexpr (STAGE == 0) {
        const float a = fmaf(w2, f7.r.z, fmaf(w1, f7.r.y, w0 * f7.r.x));
        g.b = fmaf(w2, f7.v.z, fmaf(w1, f7.v.y, w0 * f7.v.x));
        g.bb = g.b * g.b;                                           // shared by n2 and by the denominator of c (one instruction less)
        const float n2 = fmaf(a, a, g.bb);
        const float D = fmaf(hw_sqrt(n2), kInvSquash, 1.0f);
        g.e = D + a;
        g.num = fmaf(D, D, -n2);
    } else if constexpr (STAGE == 1) {
        g.t = g.b * hw_rcp(g.e);
        g.c = g.num * hw_rcp(fmaf(g.e, g.e, g.bb));
        g.z = g.t * g.t;
        float p = fmaf(2.456724578e-03f, g.z, -1.440135792e-02f);
        p = fmaf(p, g.z, 3.978122362e-02f);
        p = fmaf(p, g.z, -7.234857378e-02f);
        g.p = fmaf(p, g.z, 1.049894609e-01f);
    } else {
        float p = fmaf(g.p, g.z, -1.416122920e-01f);
        p = fmaf(p, g.z, 1.998590677e-01f);
        p = fmaf(p, g.z, -3.333259703e-01f);
        p = fmaf(p, g.z, 9.999998864e-01f);
#ifdef RNF_SP_SPLIT
        const float e = hw_exp2(-1.44269504088896341f * fabsf(s_raw));
        const float sp = fmaf(hw_log2(1.0f + e), 0.693147180559945309f, relu_bits(s_raw));
#else
        // softplus(s) / ln 2 = log2(1 + 2^(s log2 e)) in one piece (the packers fold log2 e into the weights that produce s): S, A and J all carry the common factor 1 / ln 2 and the layer only uses
        // their ratios (theta' = pi + 2 A / S, ldj = log(J / S)).  No argument split: for s < -17 the weight rounds to 0 exactly as before
        // (< 1e-7 absolute on a term divided by the sum of K such terms); for s > 88 the exponential overflows to inf and the median picks
        // s itself (softplus2_lean).  That form (SAFE = false) is only instantiated by kernels the launcher runs GUARDED (flow_kernels.h:
        // LEAN / FUSED): their layer finish flags a weight sum so small that fl(1 + e) has rounded the weights themselves
        // (mobius_fwd_finish: kMinWeightSum) and the launch is re-run on the exact-fp32 kernels, whose softplus is the full form.
        const float sp = SAFE ? softplus2_safe(s_raw) : softplus2_lean(s_raw);                   // s_raw arrives multiplied by log2 e (layout.h S_PRESCALE)
#endif
        S += sp;
        A = fmaf(sp, p * g.t, A);
        J = fmaf(sp, g.c, J);
    }
}
// The same segment (segment_fwd_s7<true>: the overflow-safe softplus) cut into SIX slices of 6 - 10 VALU instructions, one behind every
// matrix instruction of a bf16x3 fc_last tile (flow_kernels.h last_slot_b3: 24 instructions per tile = 4 segments x 6 slices).  Same
// operations in the same order per value: bit-identical to the three-slice form.
struct SegS6 {
    float a, b, bb, e, num, t, c, z, p, ex, lg;
};
template <int STAGE>
RNF_HD void seg_s7_stage6(SegS6 &g, float s_raw, float w0, float w1, float w2, const Frame &f7, float &S, float &A, float &J) {
    if constexpr (STAGE == 0) {
        g.a = fmaf(w2, f7.r.z, fmaf(w1, f7.r.y, w0 * f7.r.x));
        g.b = fmaf(w2, f7.v.z, fmaf(w1, f7.v.y, w0 * f7.v.x));
    } else if constexpr (STAGE == 1) {
        g.bb = g.b * g.b;
        const float n2 = fmaf(g.a, g.a, g.bb);
        const float D = fmaf(hw_sqrt(n2), kInvSquash, 1.0f);
        g.e = D + g.a;
        g.num = fmaf(D, D, -n2);
    } else if constexpr (STAGE == 2) {
        g.t = g.b * hw_rcp(g.e);
        g.c = g.num * hw_rcp(fmaf(g.e, g.e, g.bb));
        g.z = g.t * g.t;
        g.ex = hw_exp2(fminf(s_raw, 126.0f));                      // softplus2_safe, first piece
    } else if constexpr (STAGE == 3) {
        float p = fmaf(2.456724578e-03f, g.z, -1.440135792e-02f);
        p = fmaf(p, g.z, 3.978122362e-02f);
        p = fmaf(p, g.z, -7.234857378e-02f);
        g.p = fmaf(p, g.z, 1.049894609e-01f);
        g.lg = hw_log2(1.0f + g.ex);
    } else if constexpr (STAGE == 4) {
        float p = fmaf(g.p, g.z, -1.416122920e-01f);
        p = fmaf(p, g.z, 1.998590677e-01f);
        p = fmaf(p, g.z, -3.333259703e-01f);
        p = fmaf(p, g.z, 9.999998864e-01f);
        g.p = p * g.t;
    } else {
        const float big = fmaxf(s_raw, g.lg);
        const float small = 1.44269504088896341f * g.ex * fmaf(g.ex, fmaf(g.ex, 0.333333333f, -0.5f), 1.0f);
        const float sp = g.ex < 0.015625f ? small : big;
        S += sp;
        A = fmaf(sp, g.p, A);
        J = fmaf(sp, g.c, J);
    }
}

template <bool SAFE = true>
RNF_HD void segment_fwd_s7(float s_raw, float w0, float w1, float w2, const Frame &f7, float &S, float &A, float &J) {
    SegS7 g;
    seg_s7_stage<0, SAFE>(g, s_raw, w0, w1, w2, f7, S, A, J);
    seg_s7_stage<1, SAFE>(g, s_raw, w0, w1, w2, f7, S, A, J);
    seg_s7_stage<2, SAFE>(g, s_raw, w0, w1, w2, f7, S, A, J);
}

// pytorch3d.transforms.matrix_to_quaternion (published 0.7.5 rule; call site flow/squeezetrans.py:34):
// four candidates from sqrt(max(0, 1 +- m00 +- m11 +- m22)), keep the one with the largest |q_i| (first on ties),
// denominators floored at 0.1.  Real part first.
RNF_HD void rot_to_quat(const Rot &R, float (&q)[4]) {
    const float m00 = R.c0.x, m01 = R.c1.x, m02 = R.c2.x, m10 = R.c0.y, m11 = R.c1.y, m12 = R.c2.y, m20 = R.c0.z, m21 = R.c1.z, m22 = R.c2.z;
    float a0 = hw_sqrt(fmaxf(1.0f + m00 + m11 + m22, 0.0f));
    float a1 = hw_sqrt(fmaxf(1.0f + m00 - m11 - m22, 0.0f));
    float a2 = hw_sqrt(fmaxf(1.0f - m00 + m11 - m22, 0.0f));
    float a3 = hw_sqrt(fmaxf(1.0f - m00 - m11 + m22, 0.0f));
    int best = 0;
    float ab = a0;
    if (a1 > ab) { ab = a1; best = 1; }
    if (a2 > ab) { ab = a2; best = 2; }
    if (a3 > ab) { ab = a3; best = 3; }
    float s01 = m21 - m12, s02 = m02 - m20, s03 = m10 - m01;   // row 0 off-diagonals
    float p12 = m10 + m01, p13 = m02 + m20, p23 = m12 + m21;
    float c0, c1, c2, c3;
    if (best == 0)      { c0 = a0 * a0; c1 = s01;     c2 = s02;     c3 = s03; }
    else if (best == 1) { c0 = s01;     c1 = a1 * a1; c2 = p12;     c3 = p13; }
    else if (best == 2) { c0 = s02;     c1 = p12;     c2 = a2 * a2; c3 = p23; }
    else                { c0 = s03;     c1 = p13;     c2 = p23;     c3 = a3 * a3; }
    float inv = 0.5f * hw_rcp(fmaxf(ab, 0.1f));
    q[0] = c0 * inv; q[1] = c1 * inv; q[2] = c2 * inv; q[3] = c3 * inv;
}

// pytorch3d.transforms.quaternion_to_matrix (scale invariant: two_s = 2/|q|^2), call site flow/squeezetrans.py:37
RNF_HD void quat_to_rot(const float (&q)[4], float l2, Rot &R) {
    const float w = q[0], x = q[1], y = q[2], z = q[3];
    const float s2 = 2.0f * hw_rcp(l2);
    R.c0.x = 1.0f - s2 * (y * y + z * z); R.c1.x = s2 * (x * y - z * w);        R.c2.x = s2 * (x * z + y * w);
    R.c0.y = s2 * (x * y + z * w);        R.c1.y = 1.0f - s2 * (x * x + z * z); R.c2.y = s2 * (y * z - x * w);
    R.c0.z = s2 * (x * z - y * w);        R.c1.z = s2 * (y * z + x * w);        R.c2.z = 1.0f - s2 * (x * x + y * y);
}

// calculate_16 (flow/squeezetrans.py:33-38) with a given 4x4 M (row-major) and log|det M|:
// q' = M q(R); R' = R(q'/|q'|); ldj = log|det M| - 4 log|q'| = log|det M| - 2 log|q'|^2
// `orthogonal`: M is a 4-D rotation (UnconditionRot, flow/rottrans.py:8-23): |q'| = 1 and det M = +-1, the reference
// returns a log-det of exactly 0, so nothing is added.
RNF_HD void affine16_apply(const float (&M)[16], float logabsdet, Rot &R, float &ldj, bool orthogonal = false) {
    float q[4], t[4];
    rot_to_quat(R, q);
#pragma unroll
    for (int i = 0; i < 4; ++i)
        t[i] = fmaf(M[4 * i + 3], q[3], fmaf(M[4 * i + 2], q[2], fmaf(M[4 * i + 1], q[1], M[4 * i] * q[0])));
    float l2 = fmaf(t[3], t[3], fmaf(t[2], t[2], fmaf(t[1], t[1], t[0] * t[0])));
    quat_to_rot(t, l2, R);
    if (!orthogonal) ldj += logabsdet - 2.0f * logf(l2);
}

// calculate_16 for a CONSTANT matrix without the quaternion detour.  For a unit quaternion every product q_i q_j is an affine
// function of the rotation entries (w^2 = (1 + R00 + R11 + R22) / 4, w x = (R21 - R12) / 4, x y = (R01 + R10) / 4, ...), so
// q' q'^T = M (q q^T) M^T is affine in R too, and both |q'|^2 and |q'|^2 R(q'/|q'|) are linear in q' q'^T: the whole layer is
//     [ |q'|^2 ; |q'|^2 R' (row-major) ] = T . [ 1 ; R (row-major) ]       with one 10x10 table T(M) built by the packer (in double),
// 90 fma + 1 rcp + 9 mul + log instead of the 4-candidate square-root selection, matrix-vector product and re-expansion (~330 VALU).
// Same result as affine16_apply up to fp32 rounding (the quaternion's sign and the candidate choice never mattered).
// Storage: two halves of 52 floats -- half hh holds rows 5 hh .. 5 hh + 4 of the table (T[52 hh + 10 i' + k]) followed by log|det M| and
// the orthogonal flag -- so that the two lanes of a pair can each evaluate five of the ten rows from 13 aligned float4 of their own
// (affine16_table_apply_pair) and exchange the results.
constexpr int AFF_HALF = 52;
RNF_HD int aff_idx(int row, int k) { return AFF_HALF * (row / 5) + 10 * (row % 5) + k; }
RNF_HD void affine16_table(const double *M /* 4x4 row-major */, float logabsdet, bool orthogonal, float *T /* AFF_TABLE_FLOATS */) {
#pragma clang fp contract(off)
    T[50] = T[AFF_HALF + 50] = logabsdet;
    T[51] = T[AFF_HALF + 51] = orthogonal ? 1.0f : 0.0f;
    for (int k = 0; k < 10; ++k) {
        double in[10];
        for (int i = 0; i < 10; ++i) in[i] = (i == k) ? 1.0 : 0.0;
        const double one = in[0];
        const double *r = in + 1;
        double Q[4][4];
        Q[0][0] = 0.25 * (one + r[0] + r[4] + r[8]);
        Q[1][1] = 0.25 * (one + r[0] - r[4] - r[8]);
        Q[2][2] = 0.25 * (one - r[0] + r[4] - r[8]);
        Q[3][3] = 0.25 * (one - r[0] - r[4] + r[8]);
        Q[0][1] = Q[1][0] = 0.25 * (r[7] - r[5]);
        Q[0][2] = Q[2][0] = 0.25 * (r[2] - r[6]);
        Q[0][3] = Q[3][0] = 0.25 * (r[3] - r[1]);
        Q[1][2] = Q[2][1] = 0.25 * (r[1] + r[3]);
        Q[1][3] = Q[3][1] = 0.25 * (r[2] + r[6]);
        Q[2][3] = Q[3][2] = 0.25 * (r[5] + r[7]);
        double MQ[4][4], P[4][4];
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                double a = 0.0;
                for (int l = 0; l < 4; ++l) a += M[4 * i + l] * Q[l][j];
                MQ[i][j] = a;
            }
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                double a = 0.0;
                for (int l = 0; l < 4; ++l) a += MQ[i][l] * M[4 * j + l];
                P[i][j] = a;
            }
        double o[10];
        o[0] = P[0][0] + P[1][1] + P[2][2] + P[3][3];
        o[1] = P[0][0] + P[1][1] - P[2][2] - P[3][3];
        o[2] = 2.0 * (P[1][2] - P[0][3]);
        o[3] = 2.0 * (P[1][3] + P[0][2]);
        o[4] = 2.0 * (P[1][2] + P[0][3]);
        o[5] = P[0][0] - P[1][1] + P[2][2] - P[3][3];
        o[6] = 2.0 * (P[2][3] - P[0][1]);
        o[7] = 2.0 * (P[1][3] - P[0][2]);
        o[8] = 2.0 * (P[2][3] + P[0][1]);
        o[9] = P[0][0] - P[1][1] - P[2][2] + P[3][3];
        for (int i = 0; i < 10; ++i) T[aff_idx(i, k)] = (float)o[i];
    }
}

RNF_HD void affine16_finish(const float (&o)[10], float logabsdet, float orthogonal, Rot &R, float &ldj) {
    const float inv = hw_rcp(o[0]);
    R.c0.x = o[1] * inv; R.c1.x = o[2] * inv; R.c2.x = o[3] * inv;
    R.c0.y = o[4] * inv; R.c1.y = o[5] * inv; R.c2.y = o[6] * inv;
    R.c0.z = o[7] * inv; R.c1.z = o[8] * inv; R.c2.z = o[9] * inv;
    if (orthogonal == 0.0f) ldj += fmaf(-2.0f * 0.693147180559945309f, hw_log2(o[0]), logabsdet);
}
RNF_HD void affine16_table_apply(const float *T, Rot &R, float &ldj) {
    const float r[9] = {R.c0.x, R.c1.x, R.c2.x, R.c0.y, R.c1.y, R.c2.y, R.c0.z, R.c1.z, R.c2.z};
    float o[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        float a = T[aff_idx(i, 0)];
#pragma unroll
        for (int k = 0; k < 9; ++k) a = fmaf(T[aff_idx(i, 1 + k)], r[k], a);
        o[i] = a;
    }
    affine16_finish(o, T[50], T[51], R, ldj);
}
#if defined(__HIPCC__)
// The same layer with the ten rows split over the two lanes (j, j + 32) that hold one rotation: lane half h evaluates rows 5h .. 5h + 4
// (45 FMAs instead of 90) and one v_permlane32_swap per row hands both halves to both lanes (v, v -> {row of half 0, row of half 1}).
__device__ __forceinline__ void affine16_table_apply_pair(const float *T, int h, Rot &R, float &ldj) {
    const float r[9] = {R.c0.x, R.c1.x, R.c2.x, R.c0.y, R.c1.y, R.c2.y, R.c0.z, R.c1.z, R.c2.z};
    const float *Th = T + AFF_HALF * h;
    float o[10];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        float a = Th[10 * i];
#pragma unroll
        for (int k = 0; k < 9; ++k) a = fmaf(Th[10 * i + 1 + k], r[k], a);
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(a), false, false);
        o[i] = __uint_as_float(sw[0]);
        o[5 + i] = __uint_as_float(sw[1]);
    }
    affine16_finish(o, Th[50], Th[51], R, ldj);
}
#endif

// ---- 3x3 / 6x6 ablation layers: Gram-Schmidt of two transformed columns, log-det from three tangent directions ----------------
// (calculate_9 / calculate_36, flow/squeezetrans.py:176-231, 293-331).  a0, a1: the two columns; da0[k], da1[k]: their derivatives
// along tangent direction k.  Forward mode through normalise / project / normalise / cross; the three tangent images
// vee(dR' R'^T) form a 3x3 matrix whose |det| is the volume change.
RNF_HD float det3v(v3f a, v3f b, v3f c) { return dot3(a, cross3(b, c)); }

RNF_HD void gram_schmidt_tangent(v3f a0, v3f a1, const v3f (&da0)[3], const v3f (&da1)[3], Rot &R, float &ldj) {
    const float i0 = hw_rsq(dot3(a0, a0));
    const v3f t0 = a0 * i0;
    const float dot = dot3(t0, a1);
    const v3f b1 = a1 - t0 * dot;
    const float i1 = hw_rsq(dot3(b1, b1));
    const v3f t1 = b1 * i1;
    const v3f t2 = cross3(t0, t1);
    v3f vec[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const v3f dt0 = (da0[k] - t0 * dot3(t0, da0[k])) * i0;
        const float ddot = dot3(dt0, a1) + dot3(t0, da1[k]);
        const v3f db1 = da1[k] - t0 * ddot - dt0 * dot;
        const v3f dt1 = (db1 - t1 * dot3(t1, db1)) * i1;
        const v3f dt2 = cross3(dt0, t1) + cross3(t0, dt1);
        // delta = dR' R'^T with R' = [t0 t1 t2] (columns): delta[a][b] = sum_c dR'[a][c] R'[b][c]
        vec[k] = v3f{dt0.x * t0.y + dt1.x * t1.y + dt2.x * t2.y,       // delta[0][1]
                     dt0.x * t0.z + dt1.x * t1.z + dt2.x * t2.z,       // delta[0][2]
                     dt0.y * t0.z + dt1.y * t1.z + dt2.y * t2.z};      // delta[1][2]
    }
    R.c0 = t0; R.c1 = t1; R.c2 = t2;
    ldj += 0.693147180559945309f * hw_log2(fabsf(det3v(vec[0], vec[1], vec[2])));
}

RNF_HD v3f mat3_mul(const float *M, v3f a) {          // row-major 3x3 times vector
    return v3f{fmaf(M[2], a.z, fmaf(M[1], a.y, M[0] * a.x)), fmaf(M[5], a.z, fmaf(M[4], a.y, M[3] * a.x)),
               fmaf(M[8], a.z, fmaf(M[7], a.y, M[6] * a.x))};
}

// calculate_9 (squeezetrans.py:199-231): Gram-Schmidt of X = M R.  With X = Q U (U upper triangular, u00 = |x0|, u22 = q2 . x2) the
// reference's tangent-space determinant (three directions R G_k pushed through normalise / project / normalise / cross) is, in closed
// form,  ldj = 2 log|u22| - 2 log u00  (checked against the forward-mode restatement in oracle.gs9 to 1e-14 in fp64,
// tests/test_oracle_golden.py); only the 6x6 variant needs the tangent machinery above.
RNF_HD void gs9_apply(const float *M, Rot &R, float &ldj) {
    const v3f x0 = mat3_mul(M, R.c0), x1 = mat3_mul(M, R.c1), x2 = mat3_mul(M, R.c2);
    const float n0 = dot3(x0, x0);
    const v3f q0 = x0 * hw_rsq(n0);
    const v3f b1 = x1 - q0 * dot3(q0, x1);
    const v3f q1 = b1 * hw_rsq(dot3(b1, b1));
    const v3f q2 = cross3(q0, q1);
    const float u22 = dot3(q2, x2);
    R.c0 = q0; R.c1 = q1; R.c2 = q2;
    ldj += 0.693147180559945309f * hw_log2(u22 * u22 * hw_rcp(n0));
}

// calculate_36: (r0 (+) r1) as a 6-vector times M [6][6]; tangent directions G_k R (left multiplication):
// G_0 c = (c.y, -c.x, 0), G_1 c = (c.z, 0, -c.x), G_2 c = (0, c.z, -c.y)
RNF_HD void mat6_mul(const float *M, v3f u, v3f w, v3f &a0, v3f &a1) {
    float o[6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
        o[i] = fmaf(M[6 * i + 5], w.z, fmaf(M[6 * i + 4], w.y, fmaf(M[6 * i + 3], w.x, fmaf(M[6 * i + 2], u.z, fmaf(M[6 * i + 1], u.y, M[6 * i] * u.x)))));
    a0 = v3f{o[0], o[1], o[2]};
    a1 = v3f{o[3], o[4], o[5]};
}
RNF_HD void gs36_apply(const float *M, Rot &R, float &ldj) {
    const v3f r0 = R.c0, r1 = R.c1;
    v3f a0, a1, da0[3], da1[3];
    mat6_mul(M, r0, r1, a0, a1);
    mat6_mul(M, v3f{r0.y, -r0.x, 0.f}, v3f{r1.y, -r1.x, 0.f}, da0[0], da1[0]);
    mat6_mul(M, v3f{r0.z, 0.f, -r0.x}, v3f{r1.z, 0.f, -r1.x}, da0[1], da1[1]);
    mat6_mul(M, v3f{0.f, r0.z, -r0.y}, v3f{0.f, r1.z, -r1.y}, da0[2], da1[2]);
    gram_schmidt_tangent(a0, a1, da0, da1, R, ldj);
}

// ---- conditional 3x3 layers: per-sample M (row-major) ---------------------------------------------------------------------------
// inverse by cofactors (torch.linalg.inv of squeezetrans.py:245)
RNF_HD void inv3(const float (&m)[9], float (&o)[9]) {
    const float c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
    const float id = 1.0f / (m[0] * c00 + m[1] * c01 + m[2] * c02);
    o[0] = c00 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = c01 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = c02 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

// Gram-Schmidt rotation of the COLUMNS of M (calculate_9_r_smith, rottrans.py:85-91), returned as columns q0, q1, q2
RNF_HD void smith3(const float (&m)[9], v3f &q0, v3f &q1, v3f &q2) {
    const v3f a0 = v3f{m[0], m[3], m[6]}, a1 = v3f{m[1], m[4], m[7]};
    q0 = a0 * hw_rsq(dot3(a0, a0));
    const v3f b1 = a1 - q0 * dot3(q0, a1);
    q1 = b1 * hw_rsq(dot3(b1, b1));
    q2 = cross3(q0, q1);
}

// Orthogonal polar factor U V^T of M (calculate_9_l / calculate_9_r, rottrans.py:72-82 take it from a batched SVD) by the Newton
// iteration X <- (X + X^-T) / 2, X^-T = cof(X) / det(X): quadratically convergent, M = I + MLP output is well conditioned.
// Rows of the result in p0, p1, p2.
RNF_HD void polar3(const float (&m)[9], v3f &p0, v3f &p1, v3f &p2) {
    p0 = v3f{m[0], m[1], m[2]}; p1 = v3f{m[3], m[4], m[5]}; p2 = v3f{m[6], m[7], m[8]};
#pragma unroll 1
    for (int it = 0; it < 10; ++it) {
        const v3f c0 = cross3(p1, p2), c1 = cross3(p2, p0), c2 = cross3(p0, p1);      // rows of the cofactor matrix
        const float hid = 0.5f / dot3(p0, c0);
        p0 = p0 * 0.5f + c0 * hid;
        p1 = p1 * 0.5f + c1 * hid;
        p2 = p2 * 0.5f + c2 * hid;
    }
}

// R <- P R for a matrix given by its rows; R <- R Q for a matrix given by its columns (or rows = Q^T when transposed)
RNF_HD void left_mul_rows(v3f p0, v3f p1, v3f p2, Rot &R) {
    R.c0 = v3f{dot3(p0, R.c0), dot3(p1, R.c0), dot3(p2, R.c0)};
    R.c1 = v3f{dot3(p0, R.c1), dot3(p1, R.c1), dot3(p2, R.c1)};
    R.c2 = v3f{dot3(p0, R.c2), dot3(p1, R.c2), dot3(p2, R.c2)};
}
RNF_HD void right_mul_cols(v3f q0, v3f q1, v3f q2, Rot &R) {          // (R Q)[:, c] = sum_k R[:, k] Q[k][c], q_c = column c of Q
    const v3f r0 = R.c0, r1 = R.c1, r2 = R.c2;
    R.c0 = r0 * q0.x + r1 * q0.y + r2 * q0.z;
    R.c1 = r0 * q1.x + r1 * q1.y + r2 * q1.z;
    R.c2 = r0 * q2.x + r1 * q2.y + r2 * q2.z;
}

// 6x6 inverse (torch.linalg.inv of squeezetrans.py:345) by Gauss-Jordan with partial pivoting, everything in registers: the pivot row is
// brought into place by compare-and-swap of whole rows (selects, no dynamic indexing -- a dynamically indexed register array would go to
// scratch memory).  About 1.3 k VALU instructions per matrix; only the inverse pass of Condition36Trans pays it.
RNF_HD void inv6(float (&a)[36], float (&b)[36]) {
#pragma unroll
    for (int i = 0; i < 36; ++i) b[i] = (i % 7 == 0) ? 1.0f : 0.0f;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
#pragma unroll
        for (int r = c + 1; r < 6; ++r) {                 // bubble the largest |a[r][c]| up to row c
            const bool sw = fabsf(a[6 * r + c]) > fabsf(a[6 * c + c]);
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const float ac = a[6 * c + k], ar = a[6 * r + k], bc = b[6 * c + k], br = b[6 * r + k];
                a[6 * c + k] = sw ? ar : ac; a[6 * r + k] = sw ? ac : ar;
                b[6 * c + k] = sw ? br : bc; b[6 * r + k] = sw ? bc : br;
            }
        }
        const float ip = 1.0f / a[6 * c + c];
#pragma unroll
        for (int k = 0; k < 6; ++k) { a[6 * c + k] *= ip; b[6 * c + k] *= ip; }
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            if (r == c) continue;
            const float f = a[6 * r + c];
#pragma unroll
            for (int k = 0; k < 6; ++k) { a[6 * r + k] = fmaf(-f, a[6 * c + k], a[6 * r + k]); b[6 * r + k] = fmaf(-f, b[6 * c + k], b[6 * r + k]); }
        }
    }
}

// 4x4 inverse and determinant by cofactors (Condition16Trans.inverse: torch.linalg.inv, flow/squeezetrans.py:51-55;
// my_det_4_4: squeezetrans.py:17-22).  Returns det(M); Minv = adj(M)/det.
RNF_HD float inv4(const float (&m)[16], float (&o)[16]) {
    float s0 = m[0] * m[5] - m[4] * m[1],  s1 = m[0] * m[6] - m[4] * m[2],  s2 = m[0] * m[7] - m[4] * m[3];
    float s3 = m[1] * m[6] - m[5] * m[2],  s4 = m[1] * m[7] - m[5] * m[3],  s5 = m[2] * m[7] - m[6] * m[3];
    float c5 = m[10] * m[15] - m[14] * m[11], c4 = m[9] * m[15] - m[13] * m[11], c3 = m[9] * m[14] - m[13] * m[10];
    float c2 = m[8] * m[15] - m[12] * m[11],  c1 = m[8] * m[14] - m[12] * m[10], c0 = m[8] * m[13] - m[12] * m[9];
    float det = s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0;
    float id = 1.0f / det;
    o[0]  = ( m[5] * c5 - m[6] * c4 + m[7] * c3) * id;
    o[1]  = (-m[1] * c5 + m[2] * c4 - m[3] * c3) * id;
    o[2]  = ( m[13] * s5 - m[14] * s4 + m[15] * s3) * id;
    o[3]  = (-m[9] * s5 + m[10] * s4 - m[11] * s3) * id;
    o[4]  = (-m[4] * c5 + m[6] * c2 - m[7] * c1) * id;
    o[5]  = ( m[0] * c5 - m[2] * c2 + m[3] * c1) * id;
    o[6]  = (-m[12] * s5 + m[14] * s2 - m[15] * s1) * id;
    o[7]  = ( m[8] * s5 - m[10] * s2 + m[11] * s1) * id;
    o[8]  = ( m[4] * c4 - m[5] * c2 + m[7] * c0) * id;
    o[9]  = (-m[0] * c4 + m[1] * c2 - m[3] * c0) * id;
    o[10] = ( m[12] * s4 - m[13] * s2 + m[15] * s0) * id;
    o[11] = (-m[8] * s4 + m[9] * s2 - m[11] * s0) * id;
    o[12] = (-m[4] * c3 + m[5] * c1 - m[6] * c0) * id;
    o[13] = ( m[0] * c3 - m[1] * c1 + m[2] * c0) * id;
    o[14] = (-m[12] * s3 + m[13] * s1 - m[14] * s0) * id;
    o[15] = ( m[8] * s3 - m[9] * s1 + m[10] * s0) * id;
    return det;
}

}  // namespace rnf
