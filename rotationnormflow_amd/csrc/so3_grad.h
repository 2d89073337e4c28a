// so3_grad.h -- reverse-mode derivatives of the per-sample layer math (training path).
//
// Every function differentiates the SAME smooth formulation the forward kernels evaluate (so3_math.h): in-plane Moebius
// segments, quaternion affine.  The reference differentiates its 3-D formulation with autograd (agent.py:79-90); the two
// agree on every gradient that training uses, because they are the same function on the rotation manifold and all layer
// outputs are rotations: ambient (off-manifold) components of dL/dR may differ, parameter gradients do not.
// Checked on the CPU against torch autograd of the oracle (tests/test_host_grad.py).
#pragma once
#include "so3_math.h"
#include "layout.h"

namespace rnf {

RNF_HD v3f scale3(v3f a, float s) { return a * s; }

// y = a / |a|:  g_a = (g_y - y (y.g_y)) / |a|
RNF_HD v3f normalize_bwd(v3f y, float inv_norm, v3f gy) { return (gy - y * dot3(y, gy)) * inv_norm; }

// ---------------------------------------------------------------------------------------------------------------------
// Moebius layer, forward part that depends on the conditioner output `cond` (reference row order: [K weights | K x 3 centres],
// flow/mobiusflow.py:58-61), written per sample with plain loops.  Used by the training kernels (one rotation per lane).
// ---------------------------------------------------------------------------------------------------------------------
struct MobiusSaved {
    Frame f;
    v3f x, y;
    float inv_x, inv_cr;      // 1/|x|, 1/|y x r|
    float S, A, J, Phi, sn, cs;
    v3f tx, tzu;              // transformed column, un-normalised third column
    float inv_tzu;
    bool cyc;
    int p0, p1, p2;
};

// Accessors for one sample's conditioner outputs / their gradients: `float get(int row)`, `void put(int row, float)`.
// Strided: element `row` at p[row * stride] (stride 1 for a contiguous row).
struct StridedRow {
    float *p;
    long long stride;
    RNF_HD float get(int row) const { return p[row * stride]; }
    RNF_HD void put(int row, float v) const { p[row * stride] = v; }
};

// --- forward, in three pieces so that the segments can be split over several waves (train_kernels.h) -------------------------
RNF_HD void mobius_frame(const Rot &Rin, int perm_row, MobiusSaved &sv) {
    sv.p0 = perm_row % 3; sv.p1 = (perm_row + 1) % 3; sv.p2 = (perm_row + 2) % 3;
    sv.cyc = (sv.p1 - sv.p0 == 1) || (sv.p1 - sv.p0 == -2);
    sv.x = get_col(Rin, sv.p0);
    sv.y = get_col(Rin, sv.p1);
    sv.inv_x = 1.0f / sqrtf(dot3(sv.x, sv.x));
    sv.f.r = sv.x * (-sv.inv_x);
    const v3f cr = cross3(sv.y, sv.f.r);
    sv.inv_cr = 1.0f / sqrtf(dot3(cr, cr));
    sv.f.v = cr * sv.inv_cr;
}

// partial sums over segments [k0, k1): S = sum softplus, A = sum softplus * phi, J = sum softplus * c
template <class CondRow>
RNF_HD void mobius_segments_sums(const MobiusSaved &sv, const CondRow &cond, int K, int k0, int k1, float &S, float &A, float &J) {
    for (int k = k0; k < k1; ++k) {
        float ur, uv, phi, c;
        squash_center(cond.get(K + 3 * k), cond.get(K + 3 * k + 1), cond.get(K + 3 * k + 2), sv.f, ur, uv);
        mobius_angle(-1.0f, 0.0f, kPi, ur, uv, phi, c);                 // z = x expressed in its own frame is (-1, 0): theta = pi
        const float sp = softplus(cond.get(k));
        S += sp; A = fmaf(sp, phi, A); J = fmaf(sp, c, J);
    }
}

RNF_HD void mobius_combine(MobiusSaved &sv, float S, float A, float J) {
    sv.S = S; sv.A = A; sv.J = J;
    sv.Phi = A / S;
    sincos_small(sv.Phi, sv.sn, sv.cs);
    sv.tx = sv.f.v * sv.sn + sv.f.r * sv.cs;
    sv.tzu = sv.cyc ? cross3(sv.tx, sv.y) : cross3(sv.y, sv.tx);
    sv.inv_tzu = 1.0f / sqrtf(dot3(sv.tzu, sv.tzu));
}

template <class CondRow>
RNF_HD void mobius_segments_forward(const Rot &Rin, int perm_row, const CondRow &cond, int K, Rot &Rout, float &ldj, MobiusSaved &sv) {
    mobius_frame(Rin, perm_row, sv);
    float S = 0.f, A = 0.f, J = 0.f;
    mobius_segments_sums(sv, cond, K, 0, K, S, A, J);
    mobius_combine(sv, S, A, J);
    Rout = Rin;
    set_col(Rout, sv.p0, sv.tx);
    set_col(Rout, sv.p2, sv.tzu * sv.inv_tzu);
    ldj = logf(J / S);
}

// --- backward, same three pieces ---------------------------------------------------------------------------------------------
struct MobiusGrad {
    float g_A, g_J, g_S;      // d/d(sums)
    v3f g_r, g_v, g_y;        // running gradients of the frame vectors and the conditioning column
};

RNF_HD void mobius_backward_head(const MobiusSaved &sv, const Rot &gRout, float g_ldj, MobiusGrad &mg) {
    const v3f tz = sv.tzu * sv.inv_tzu;
    v3f g_tx = get_col(gRout, sv.p0);
    mg.g_y = get_col(gRout, sv.p1);
    const v3f g_tzu = normalize_bwd(tz, sv.inv_tzu, get_col(gRout, sv.p2));
    if (sv.cyc) {            // tzu = tx x y:  g_tx += y x g,  g_y += g x tx
        g_tx = g_tx + cross3(sv.y, g_tzu);
        mg.g_y = mg.g_y + cross3(g_tzu, sv.tx);
    } else {                 // tzu = y x tx:  g_y += tx x g,  g_tx += g x y
        mg.g_y = mg.g_y + cross3(sv.tx, g_tzu);
        g_tx = g_tx + cross3(g_tzu, sv.y);
    }
    // tx = r cos(Phi) + v sin(Phi)
    mg.g_r = g_tx * sv.cs;
    mg.g_v = g_tx * sv.sn;
    const float g_Phi = dot3(g_tx, sv.f.v * sv.cs - sv.f.r * sv.sn);
    // Phi = A / S,  ldj = log J - log S
    const float invS = 1.0f / sv.S;
    mg.g_A = g_Phi * invS;
    mg.g_J = g_ldj / sv.J;
    mg.g_S = -g_Phi * sv.A * invS * invS - g_ldj * invS;
}

// segments [k0, k1): writes their conditioner-output gradients, ADDS their contribution to g_r / g_v (pass zero-initialised
// accumulators when the range is a partial one)
template <class CondRow, class GradRow>
RNF_HD void mobius_segments_backward_range(const MobiusSaved &sv, const CondRow &cond, int K, int k0, int k1, const MobiusGrad &mg,
                                           const GradRow &g_cond, v3f &g_r, v3f &g_v) {
    for (int k = k0; k < k1; ++k) {
        const float s_raw = cond.get(k);
        const float w0 = cond.get(K + 3 * k), w1 = cond.get(K + 3 * k + 1), w2 = cond.get(K + 3 * k + 2);
        // recompute the segment (cheaper than storing 6 values x K per sample)
        const float wr = fmaf(w2, sv.f.r.z, fmaf(w1, sv.f.r.y, w0 * sv.f.r.x));
        const float wv = fmaf(w2, sv.f.v.z, fmaf(w1, sv.f.v.y, w0 * sv.f.v.x));
        const float n = hw_sqrt(fmaf(wv, wv, wr * wr));
        const float inv1n = hw_rcp(1.0f + n);
        const float sc = 0.7f * inv1n;
        const float ur = wr * sc, uv = wv * sc;
        const float e1 = 1.0f + ur;                       // z = (-1, 0): a = -ur, b = -uv
        const float inv_e1 = hw_rcp(e1);
        const float t = uv * inv_e1;
        const float phi = fmaf(2.0f, atan_unit(t), kPi);
        const float u2 = fmaf(uv, uv, ur * ur), d2 = fmaf(uv, uv, e1 * e1);
        const float inv_d2 = hw_rcp(d2);
        const float c = (1.0f - u2) * inv_d2;
        const float sp = softplus(s_raw);
        // d/d(sp, phi, c)
        const float g_sp = mg.g_S + mg.g_A * phi + mg.g_J * c;
        const float g_phi = mg.g_A * sp, g_c = mg.g_J * sp;
        // sp = softplus(s): sigmoid
        g_cond.put(k, g_sp * hw_rcp(1.0f + hw_exp2(-1.44269504088896341f * s_raw)));
        // phi = pi + 2 atan(t), t = uv / e1
        const float g_t = g_phi * 2.0f * hw_rcp(fmaf(t, t, 1.0f));
        float g_uv = g_t * inv_e1, g_ur = -g_t * t * inv_e1;
        // c = (1 - u2) / d2
        const float g_u2 = -g_c * inv_d2, g_d2 = -g_c * c * inv_d2;
        g_ur += 2.0f * ur * g_u2 + 2.0f * e1 * g_d2;
        g_uv += 2.0f * uv * g_u2 + 2.0f * uv * g_d2;
        // ur = wr sc, uv = wv sc, sc = 0.7 / (1 + n), n = |(wr, wv)|
        const float g_sc = g_ur * wr + g_uv * wv;
        float g_wr = g_ur * sc, g_wv = g_uv * sc;
        const float g_n = -g_sc * sc * inv1n;
        if (n > 0.f) { const float gn = g_n * hw_rcp(n); g_wr = fmaf(gn, wr, g_wr); g_wv = fmaf(gn, wv, g_wv); }
        // wr = w.r, wv = w.v
        g_cond.put(K + 3 * k, g_wr * sv.f.r.x + g_wv * sv.f.v.x);
        g_cond.put(K + 3 * k + 1, g_wr * sv.f.r.y + g_wv * sv.f.v.y);
        g_cond.put(K + 3 * k + 2, g_wr * sv.f.r.z + g_wv * sv.f.v.z);
        g_r = g_r + v3f{w0, w1, w2} * g_wr;
        g_v = g_v + v3f{w0, w1, w2} * g_wv;
    }
}

RNF_HD void mobius_backward_tail(const MobiusSaved &sv, const MobiusGrad &mg, Rot &gRin) {
    // v = cr / |cr|, cr = y x r
    const v3f g_cr = normalize_bwd(sv.f.v, sv.inv_cr, mg.g_v);
    const v3f g_y = mg.g_y + cross3(sv.f.r, g_cr);
    const v3f g_r = mg.g_r + cross3(g_cr, sv.y);
    // r = -x / |x|
    const v3f g_x = normalize_bwd(sv.f.r, sv.inv_x, g_r) * -1.0f;
    gRin.c0 = v3f{0.f, 0.f, 0.f}; gRin.c1 = gRin.c0; gRin.c2 = gRin.c0;
    set_col(gRin, sv.p0, g_x);
    set_col(gRin, sv.p1, g_y);
}

// Given dL/dRout (columns) and dL/dldj: gradient w.r.t. the conditioner output (g_cond, same order as cond) and w.r.t.
// the input columns x (p0) and y (p1) EXCLUDING the path through the conditioner's input (the caller adds W0^T g there).
// g_cond may alias cond (each segment's four values are read before their gradients are written).
template <class CondRow, class GradRow>
RNF_HD void mobius_segments_backward(const MobiusSaved &sv, const CondRow &cond, int K, const Rot &gRout, float g_ldj, const GradRow &g_cond,
                                     Rot &gRin) {
    MobiusGrad mg;
    mobius_backward_head(sv, gRout, g_ldj, mg);
    mobius_segments_backward_range(sv, cond, K, 0, K, mg, g_cond, mg.g_r, mg.g_v);
    mobius_backward_tail(sv, mg, gRin);
}

// ---------------------------------------------------------------------------------------------------------------------
// MobiusFlow.inverse backward (flow/mobiusflow.py:127-183 with BinFind.backward, :247-273).
//
// Forward of the inverse layer, per sample: frame (r, v) from the GIVEN column tx and the conditioning column ty; theta = the root of
//     F(theta) = sum_k sp_k phi_k(theta; u_k) / S = T           (T = angle of tx in its own frame == pi, a constant)
// on the bisection grid; x = r cos(theta) + v sin(theta); ldj = -log(J / S), J = sum_k sp_k c_k(theta; u_k); third column from x and ty.
// BinFind.backward is the implicit-function gradient of the root AT THE RETURNED ITERATE:  d theta = -(sum_p F_p dp) / F_theta with
// F_theta = sum_k sp_k c_k / S = J / S  (d phi_k / d theta = c_k), so with
//     g_theta = g_x . (v cos - r sin)  +  g_J * sum_k sp_k dc_k/dtheta          (the second term: ldj is evaluated at theta too)
// the root contributes the adjoint  g_F = -g_theta * S / J  to F = A / S (A = sum_k sp_k phi_k), and every segment gets
//     g_sp_k = g_S + g_A phi_k + g_J c_k,   g_phi_k = g_A sp_k,   g_c_k = g_J sp_k
// exactly as in the forward layer -- only the values of (g_S, g_A, g_J) and the evaluation point (cos, sin, theta) instead of
// (-1, 0, pi) differ.  The layer's output state gives theta back without a second root search: x = column p0 of the layer output.
// ---------------------------------------------------------------------------------------------------------------------
struct MobiusInvSaved {
    MobiusSaved b;            // frame of (tx, ty): b.x = tx, b.y = ty, b.f, b.inv_x, b.inv_cr, b.cyc, b.p0..p2
    float cs, sn, theta;      // evaluation point: x = r cs + v sn
    float S, A, J, Cth;       // sums at theta: sp, sp phi, sp c, sp dc/dtheta
    v3f x, zu;                // restored column and un-normalised third column
    float inv_zu;
};

RNF_HD void mobius_inv_frame(const Rot &Rin, int perm_row, const Rot &Rout, MobiusInvSaved &sv) {
    mobius_frame(Rin, perm_row, sv.b);
    sv.x = get_col(Rout, sv.b.p0);
    const float c0 = dot3(sv.x, sv.b.f.r), s0 = dot3(sv.x, sv.b.f.v);
    const float inv = 1.0f / sqrtf(fmaf(s0, s0, c0 * c0));
    sv.cs = c0 * inv;
    sv.sn = s0 * inv;
    sv.theta = angle_0_2pi(s0, c0);
    sv.zu = sv.b.cyc ? cross3(sv.x, sv.b.y) : cross3(sv.b.y, sv.x);
    sv.inv_zu = 1.0f / sqrtf(dot3(sv.zu, sv.zu));
}

// one segment at a general point of the circle: phi, c and dc/dtheta
RNF_HD void mobius_segment_at(float ur, float uv, float cs, float sn, float theta, float &phi, float &c, float &dc) {
    const float a = fmaf(uv, sn, ur * cs), b = fmaf(uv, cs, -ur * sn);
    const float e1 = 1.0f - a;
    phi = fmaf(2.0f, atan_unit(-b * hw_rcp(e1)), theta);
    const float inv_d2 = hw_rcp(fmaf(b, b, e1 * e1));
    c = (1.0f - fmaf(uv, uv, ur * ur)) * inv_d2;
    dc = 2.0f * b * c * inv_d2;                           // d(d2)/dtheta = -2 b
}

template <class CondRow>
RNF_HD void mobius_inv_segments_sums(const MobiusInvSaved &sv, const CondRow &cond, int K, int k0, int k1, float &S, float &A, float &J, float &Cth) {
    for (int k = k0; k < k1; ++k) {
        float ur, uv, phi, c, dc;
        squash_center(cond.get(K + 3 * k), cond.get(K + 3 * k + 1), cond.get(K + 3 * k + 2), sv.b.f, ur, uv);
        mobius_segment_at(ur, uv, sv.cs, sv.sn, sv.theta, phi, c, dc);
        const float sp = softplus(cond.get(k));
        S += sp; A = fmaf(sp, phi, A); J = fmaf(sp, c, J); Cth = fmaf(sp, dc, Cth);
    }
}

RNF_HD void mobius_inv_backward_head(MobiusInvSaved &sv, float S, float A, float J, float Cth, const Rot &gRout, float g_ldj, MobiusGrad &mg) {
    sv.S = S; sv.A = A; sv.J = J; sv.Cth = Cth;
    const v3f z3 = sv.zu * sv.inv_zu;
    v3f g_x = get_col(gRout, sv.b.p0);
    mg.g_y = get_col(gRout, sv.b.p1);
    const v3f g_zu = normalize_bwd(z3, sv.inv_zu, get_col(gRout, sv.b.p2));
    if (sv.b.cyc) {          // zu = x x ty
        g_x = g_x + cross3(sv.b.y, g_zu);
        mg.g_y = mg.g_y + cross3(g_zu, sv.x);
    } else {                 // zu = ty x x
        mg.g_y = mg.g_y + cross3(sv.x, g_zu);
        g_x = g_x + cross3(g_zu, sv.b.y);
    }
    // x = r cos(theta) + v sin(theta)
    mg.g_r = g_x * sv.cs;
    mg.g_v = g_x * sv.sn;
    // ldj = -(log J - log S)
    mg.g_J = -g_ldj / J;
    const float g_theta = dot3(g_x, sv.b.f.v * sv.cs - sv.b.f.r * sv.sn) + mg.g_J * Cth;
    // root of A / S = T:  adjoint of F is -g_theta / F_theta, F_theta = J / S
    const float invS = 1.0f / S;
    const float g_F = -g_theta * S / J;
    mg.g_A = g_F * invS;
    mg.g_S = -g_F * A * invS * invS + g_ldj * invS;
}

// segments [k0, k1) of a layer evaluated at (cs, sn, theta): conditioner-output gradients and the frame-vector contributions
template <class CondRow, class GradRow>
RNF_HD void mobius_segments_backward_range_at(const Frame &f, float cs, float sn, float theta, const CondRow &cond, int K, int k0, int k1,
                                              const MobiusGrad &mg, const GradRow &g_cond, v3f &g_r, v3f &g_v) {
    for (int k = k0; k < k1; ++k) {
        const float s_raw = cond.get(k);
        const float w0 = cond.get(K + 3 * k), w1 = cond.get(K + 3 * k + 1), w2 = cond.get(K + 3 * k + 2);
        const float wr = fmaf(w2, f.r.z, fmaf(w1, f.r.y, w0 * f.r.x));
        const float wv = fmaf(w2, f.v.z, fmaf(w1, f.v.y, w0 * f.v.x));
        const float n = hw_sqrt(fmaf(wv, wv, wr * wr));
        const float inv1n = hw_rcp(1.0f + n);
        const float sc = 0.7f * inv1n;
        const float ur = wr * sc, uv = wv * sc;
        const float a = fmaf(uv, sn, ur * cs), b = fmaf(uv, cs, -ur * sn);
        const float e1 = 1.0f - a;
        const float inv_e1 = hw_rcp(e1);
        const float t = -b * inv_e1;
        const float phi = fmaf(2.0f, atan_unit(t), theta);
        const float u2 = fmaf(uv, uv, ur * ur), d2 = fmaf(b, b, e1 * e1);
        const float inv_d2 = hw_rcp(d2);
        const float c = (1.0f - u2) * inv_d2;
        const float sp = softplus(s_raw);
        const float g_sp = mg.g_S + mg.g_A * phi + mg.g_J * c;
        const float g_phi = mg.g_A * sp, g_c = mg.g_J * sp;
        g_cond.put(k, g_sp * hw_rcp(1.0f + hw_exp2(-1.44269504088896341f * s_raw)));
        // phi = theta + 2 atan(t), t = -b / e1, e1 = 1 - a
        const float g_t = g_phi * 2.0f * hw_rcp(fmaf(t, t, 1.0f));
        float g_b = -g_t * inv_e1, g_e1 = -g_t * t * inv_e1;
        // c = (1 - u2) / d2, d2 = b^2 + e1^2
        const float g_u2 = -g_c * inv_d2, g_d2 = -g_c * c * inv_d2;
        g_b += 2.0f * b * g_d2;
        g_e1 += 2.0f * e1 * g_d2;
        const float g_a = -g_e1;
        // (a, b) = u conj(z)
        const float g_ur = fmaf(g_a, cs, -g_b * sn) + 2.0f * ur * g_u2;
        const float g_uv = fmaf(g_a, sn, g_b * cs) + 2.0f * uv * g_u2;
        const float g_sc = g_ur * wr + g_uv * wv;
        float g_wr = g_ur * sc, g_wv = g_uv * sc;
        const float g_n = -g_sc * sc * inv1n;
        if (n > 0.f) { const float gn = g_n * hw_rcp(n); g_wr = fmaf(gn, wr, g_wr); g_wv = fmaf(gn, wv, g_wv); }
        g_cond.put(K + 3 * k, g_wr * f.r.x + g_wv * f.v.x);
        g_cond.put(K + 3 * k + 1, g_wr * f.r.y + g_wv * f.v.y);
        g_cond.put(K + 3 * k + 2, g_wr * f.r.z + g_wv * f.v.z);
        g_r = g_r + v3f{w0, w1, w2} * g_wr;
        g_v = g_v + v3f{w0, w1, w2} * g_wv;
    }
}

// whole layer (host test entry): dL/d(cond) and dL/dRin (excluding the conditioner-input path of ty) from dL/dRout, dL/dldj
template <class CondRow, class GradRow>
RNF_HD void mobius_inverse_backward(const Rot &Rin, int perm_row, const Rot &Rout, const CondRow &cond, int K, const Rot &gRout, float g_ldj,
                                    const GradRow &g_cond, Rot &gRin) {
    MobiusInvSaved sv;
    mobius_inv_frame(Rin, perm_row, Rout, sv);
    float S = 0.f, A = 0.f, J = 0.f, Cth = 0.f;
    mobius_inv_segments_sums(sv, cond, K, 0, K, S, A, J, Cth);
    MobiusGrad mg;
    mobius_inv_backward_head(sv, S, A, J, Cth, gRout, g_ldj, mg);
    mobius_segments_backward_range_at(sv.b.f, sv.cs, sv.sn, sv.theta, cond, K, 0, K, mg, g_cond, mg.g_r, mg.g_v);
    mobius_backward_tail(sv.b, mg, gRin);
}

// M^-1 enters the inverse pass of the affine layers (flow/squeezetrans.py:51-55,171-174: torch.linalg.inv): dL/dM = -M^-T (dL/dM^-1) M^-T
template <int N>
RNF_HD void inverse_matrix_grad(const float *Minv, const float *gMinv, float *gM) {
    float t[N * N];
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) {           // t = Minv^T gMinv
            float a = 0.f;
            for (int l = 0; l < N; ++l) a += Minv[l * N + i] * gMinv[l * N + j];
            t[i * N + j] = a;
        }
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) {           // gM = -t Minv^T
            float a = 0.f;
            for (int l = 0; l < N; ++l) a += t[i * N + l] * Minv[j * N + l];
            gM[i * N + j] = -a;
        }
}

// ---------------------------------------------------------------------------------------------------------------------
// calculate_16 (flow/squeezetrans.py:33-38) backward.  Forward: q = quat(R) (candidate `best`), t = M q, l2 = |t|^2,
// R' = rot(t) with two_s = 2 / l2, ldj = log|det M| - 2 log l2.
// ---------------------------------------------------------------------------------------------------------------------
struct AffineSaved {
    float q[4], t[4], l2;
    int best;
    float ab;                 // the chosen sqrt(max(0, 1 +- m00 +- m11 +- m22))
};

RNF_HD void affine16_forward_saved(const float (&M)[16], float logabsdet, const Rot &Rin, Rot &Rout, float &ldj, AffineSaved &sv) {
    const float m00 = Rin.c0.x, m01 = Rin.c1.x, m02 = Rin.c2.x, m10 = Rin.c0.y, m11 = Rin.c1.y, m12 = Rin.c2.y, m20 = Rin.c0.z, m21 = Rin.c1.z, m22 = Rin.c2.z;
    const float a[4] = {sqrtf(fmaxf(1.0f + m00 + m11 + m22, 0.0f)), sqrtf(fmaxf(1.0f + m00 - m11 - m22, 0.0f)),
                        sqrtf(fmaxf(1.0f - m00 + m11 - m22, 0.0f)), sqrtf(fmaxf(1.0f - m00 - m11 + m22, 0.0f))};
    int best = 0;
    for (int i = 1; i < 4; ++i) if (a[i] > a[best]) best = i;
    sv.best = best;
    sv.ab = a[best];
    rot_to_quat(Rin, sv.q);
    for (int i = 0; i < 4; ++i) sv.t[i] = M[4 * i] * sv.q[0] + M[4 * i + 1] * sv.q[1] + M[4 * i + 2] * sv.q[2] + M[4 * i + 3] * sv.q[3];
    sv.l2 = sv.t[0] * sv.t[0] + sv.t[1] * sv.t[1] + sv.t[2] * sv.t[2] + sv.t[3] * sv.t[3];
    quat_to_rot(sv.t, sv.l2, Rout);
    ldj = logabsdet - 2.0f * logf(sv.l2);
}

// gM (16, accumulated into by the caller over the batch) gets g_t (x) q; the log|det M| term (sum_b g_ldj) M^-T is added by the
// caller once per batch.  Returns dL/dRin.
RNF_HD void affine16_backward(const float (&M)[16], const AffineSaved &sv, const Rot &gRout, float g_ldj, bool orthogonal, float (&gM)[16],
                              Rot &gRin) {
    const float w = sv.t[0], x = sv.t[1], y = sv.t[2], z = sv.t[3];
    const float s2 = 2.0f / sv.l2;
    // R' entries E_ij = delta_ij - s2 * P_ij(t) (diag) or s2 * P_ij (off-diag); collect g wrt s2 and wrt the quadratic forms
    const float g00 = gRout.c0.x, g01 = gRout.c1.x, g02 = gRout.c2.x, g10 = gRout.c0.y, g11 = gRout.c1.y, g12 = gRout.c2.y, g20 = gRout.c0.z, g21 = gRout.c1.z, g22 = gRout.c2.z;
    // R' = I + s2 * Q,  Q00 = -(yy+zz), Q01 = xy - zw, Q02 = xz + yw, Q10 = xy + zw, Q11 = -(xx+zz), Q12 = yz - xw, Q20 = xz - yw, Q21 = yz + xw, Q22 = -(xx+yy)
    const float Q00 = -(y * y + z * z), Q01 = x * y - z * w, Q02 = x * z + y * w, Q10 = x * y + z * w, Q11 = -(x * x + z * z), Q12 = y * z - x * w,
                Q20 = x * z - y * w, Q21 = y * z + x * w, Q22 = -(x * x + y * y);
    const float g_s2 = g00 * Q00 + g01 * Q01 + g02 * Q02 + g10 * Q10 + g11 * Q11 + g12 * Q12 + g20 * Q20 + g21 * Q21 + g22 * Q22;
    // dQ/dt
    float gt[4];
    gt[0] = s2 * (-g01 * z + g02 * y + g10 * z - g12 * x - g20 * y + g21 * x);
    gt[1] = s2 * (g01 * y + g02 * z + g10 * y - 2.0f * g11 * x - g12 * w + g20 * z + g21 * w - 2.0f * g22 * x);
    gt[2] = s2 * (-2.0f * g00 * y + g01 * x + g02 * w + g10 * x + g12 * z - g20 * w + g21 * z - 2.0f * g22 * y);
    gt[3] = s2 * (-2.0f * g00 * z - g01 * w + g02 * x + g10 * w - 2.0f * g11 * z + g12 * y + g20 * x + g21 * y);
    // s2 = 2 / l2, ldj = ... - 2 log l2,  l2 = |t|^2
    const float g_l2 = -g_s2 * s2 / sv.l2 - (orthogonal ? 0.f : 2.0f * g_ldj / sv.l2);
    for (int i = 0; i < 4; ++i) gt[i] += 2.0f * sv.t[i] * g_l2;
    // t = M q
    float gq[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            gM[4 * i + j] += gt[i] * sv.q[j];
            gq[j] += M[4 * i + j] * gt[i];
        }
    // q = cand / (2 max(ab, 0.1)), cand depends on `best` (so3_math.h rot_to_quat); ab = sqrt(1 +- m00 +- m11 +- m22)
    const float den = 2.0f * fmaxf(sv.ab, 0.1f);
    const float inv = 1.0f / den;
    float gc[4] = {gq[0] * inv, gq[1] * inv, gq[2] * inv, gq[3] * inv};       // gradient wrt the candidate row
    // the diagonal candidate entry is ab^2; d(q)/d(ab) through the denominator: q_i = cand_i / (2 ab) when ab > 0.1
    float g_ab = 0.f;
    if (sv.ab > 0.1f) g_ab = -(gq[0] * sv.q[0] + gq[1] * sv.q[1] + gq[2] * sv.q[2] + gq[3] * sv.q[3]) / sv.ab;
    g_ab += gc[sv.best] * 2.0f * sv.ab;
    // ab = sqrt(u), u = 1 + s0 m00 + s1 m11 + s2 m22
    const float g_u = sv.ab > 0.f ? g_ab / (2.0f * sv.ab) : 0.f;
    const float sg[4][3] = {{1, 1, 1}, {1, -1, -1}, {-1, 1, -1}, {-1, -1, 1}};
    float r00 = g_u * sg[sv.best][0], r11 = g_u * sg[sv.best][1], r22 = g_u * sg[sv.best][2];
    float r01 = 0.f, r02 = 0.f, r10 = 0.f, r12 = 0.f, r20 = 0.f, r21 = 0.f;
    // off-diagonal candidate entries (rot_to_quat): s01 = m21 - m12, s02 = m02 - m20, s03 = m10 - m01, p12 = m10 + m01, p13 = m02 + m20, p23 = m12 + m21
    float g_s01 = 0.f, g_s02 = 0.f, g_s03 = 0.f, g_p12 = 0.f, g_p13 = 0.f, g_p23 = 0.f;
    if (sv.best == 0)      { g_s01 = gc[1]; g_s02 = gc[2]; g_s03 = gc[3]; }
    else if (sv.best == 1) { g_s01 = gc[0]; g_p12 = gc[2]; g_p13 = gc[3]; }
    else if (sv.best == 2) { g_s02 = gc[0]; g_p12 = gc[1]; g_p23 = gc[3]; }
    else                   { g_s03 = gc[0]; g_p13 = gc[1]; g_p23 = gc[2]; }
    r21 += g_s01 + g_p23; r12 += -g_s01 + g_p23;
    r02 += g_s02 + g_p13; r20 += -g_s02 + g_p13;
    r10 += g_s03 + g_p12; r01 += -g_s03 + g_p12;
    gRin.c0 = v3f{r00, r10, r20};
    gRin.c1 = v3f{r01, r11, r21};
    gRin.c2 = v3f{r02, r12, r22};
}

// ---------------------------------------------------------------------------------------------------------------------
// calculate_9 (flow/squeezetrans.py:199-231) backward, on the closed form of so3_math.h gs9_apply: X = M R, q0 = x0/|x0|,
// b1 = x1 - (q0.x1) q0, q1 = b1/|b1|, q2 = q0 x q1, u22 = q2.x2, ldj = 2 log|u22| - 2 log|x0|.
// gM (9, accumulated into) and dL/dR (columns).
// ---------------------------------------------------------------------------------------------------------------------
RNF_HD void gs9_backward(const float (&M)[9], const Rot &R, const Rot &gRout, float g_ldj, float (&gM)[9], Rot &gRin) {
    const v3f x0 = mat3_mul(M, R.c0), x1 = mat3_mul(M, R.c1), x2 = mat3_mul(M, R.c2);
    const float iu00 = 1.0f / sqrtf(dot3(x0, x0));
    const v3f q0 = x0 * iu00;
    const float d = dot3(q0, x1);
    const v3f b1 = x1 - q0 * d;
    const float iu11 = 1.0f / sqrtf(dot3(b1, b1));
    const v3f q1 = b1 * iu11;
    const v3f q2 = cross3(q0, q1);
    const float u22 = dot3(q2, x2);
    // ldj = 2 log|u22| - 2 log u00
    const float g_u22 = 2.0f * g_ldj / u22;
    v3f g_q2 = gRout.c2 + x2 * g_u22;
    const v3f g_x2 = q2 * g_u22;
    // q2 = q0 x q1
    v3f g_q0 = gRout.c0 + cross3(q1, g_q2);
    const v3f g_q1 = gRout.c1 + cross3(g_q2, q0);
    // q1 = b1 / |b1|, b1 = x1 - d q0, d = q0 . x1
    const v3f g_b1 = normalize_bwd(q1, iu11, g_q1);
    const float g_d = -dot3(g_b1, q0);
    g_q0 = g_q0 - g_b1 * d + x1 * g_d;
    const v3f g_x1 = g_b1 + q0 * g_d;
    // q0 = x0 / |x0|;  -2 log u00
    const v3f g_x0 = normalize_bwd(q0, iu00, g_q0) - q0 * (2.0f * g_ldj * iu00);
    // X = M R: gM[i][j] += sum_c gX[i][c] R[j][c];  gR[:, c] = M^T gX[:, c]
    const v3f gx[3] = {g_x0, g_x1, g_x2};
    const v3f rc[3] = {R.c0, R.c1, R.c2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        gM[0] += gx[c].x * rc[c].x; gM[1] += gx[c].x * rc[c].y; gM[2] += gx[c].x * rc[c].z;
        gM[3] += gx[c].y * rc[c].x; gM[4] += gx[c].y * rc[c].y; gM[5] += gx[c].y * rc[c].z;
        gM[6] += gx[c].z * rc[c].x; gM[7] += gx[c].z * rc[c].y; gM[8] += gx[c].z * rc[c].z;
    }
    auto mt = [&](v3f g) { return v3f{M[0] * g.x + M[3] * g.y + M[6] * g.z, M[1] * g.x + M[4] * g.y + M[7] * g.z, M[2] * g.x + M[5] * g.y + M[8] * g.z}; };
    gRin.c0 = mt(g_x0); gRin.c1 = mt(g_x1); gRin.c2 = mt(g_x2);
}

// ---------------------------------------------------------------------------------------------------------------------
// calculate_36 (flow/squeezetrans.py:293-331) backward.  Forward: (a0; a1) = M (r0; r1) with M [6][6] in 3x3 blocks M_bc, Gram-Schmidt
// t0 = a0/|a0|, b1 = a1 - (t0.a1) t0, t1 = b1/|b1|, t2 = t0 x t1, R' = [t0 t1 t2].  The reference's log-determinant of the three tangent
// images (directions G_k R pushed through normalise / project / normalise / cross) has the closed form
//     ldj = log|u . (p x q)| - 2 log|a0| - log|b1|,
//     p = r0 x (M00^T t2) + r1 x (M01^T t2),   q = r0 x (M00^T t1) + r1 x (M01^T t1),   u = r0 x (M10^T t2) + r1 x (M11^T t2)
// (rows t2^T B0, t1^T B0, t2^T B1 of the angular-velocity Jacobian, B_b = -(M_b0 hat(r0) + M_b1 hat(r1)); equal to the forward-mode
// restatement oracle.gs36 to 4e-15 in fp64, tests/test_host_grad.py).  This function differentiates that closed form by hand.
// gM (36, accumulated into) and dL/dR (the third column of R does not enter calculate_36: its gradient is 0).
// ---------------------------------------------------------------------------------------------------------------------
RNF_HD v3f blk_mv(const float *M, int bi, int bj, v3f v) {           // M_{bi,bj} v
    const float *m = M + 18 * bi + 3 * bj;
    return v3f{m[0] * v.x + m[1] * v.y + m[2] * v.z, m[6] * v.x + m[7] * v.y + m[8] * v.z, m[12] * v.x + m[13] * v.y + m[14] * v.z};
}
RNF_HD v3f blk_mtv(const float *M, int bi, int bj, v3f v) {          // M_{bi,bj}^T v
    const float *m = M + 18 * bi + 3 * bj;
    return v3f{m[0] * v.x + m[6] * v.y + m[12] * v.z, m[1] * v.x + m[7] * v.y + m[13] * v.z, m[2] * v.x + m[8] * v.y + m[14] * v.z};
}
RNF_HD void blk_outer_add(float *gM, int bi, int bj, v3f a, v3f b) {   // gM_{bi,bj} += a b^T
    float *g = gM + 18 * bi + 3 * bj;
    g[0] += a.x * b.x; g[1] += a.x * b.y; g[2] += a.x * b.z;
    g[6] += a.y * b.x; g[7] += a.y * b.y; g[8] += a.y * b.z;
    g[12] += a.z * b.x; g[13] += a.z * b.y; g[14] += a.z * b.z;
}
// closed-form forward (used by the host mirror test; the kernels' forward is so3_math.h gs36_apply)
RNF_HD void gs36_closed_form(const float *M, const Rot &R, Rot &Rout, float &ldj) {
    const v3f r0 = R.c0, r1 = R.c1;
    const v3f a0 = blk_mv(M, 0, 0, r0) + blk_mv(M, 0, 1, r1), a1 = blk_mv(M, 1, 0, r0) + blk_mv(M, 1, 1, r1);
    const float in0 = 1.0f / sqrtf(dot3(a0, a0));
    const v3f t0 = a0 * in0;
    const v3f b1 = a1 - t0 * dot3(t0, a1);
    const float in1 = 1.0f / sqrtf(dot3(b1, b1));
    const v3f t1 = b1 * in1, t2 = cross3(t0, t1);
    const v3f p = cross3(r0, blk_mtv(M, 0, 0, t2)) + cross3(r1, blk_mtv(M, 0, 1, t2));
    const v3f q = cross3(r0, blk_mtv(M, 0, 0, t1)) + cross3(r1, blk_mtv(M, 0, 1, t1));
    const v3f u = cross3(r0, blk_mtv(M, 1, 0, t2)) + cross3(r1, blk_mtv(M, 1, 1, t2));
    Rout.c0 = t0; Rout.c1 = t1; Rout.c2 = t2;
    ldj = logf(fabsf(dot3(u, cross3(p, q)))) + 2.0f * logf(in0) + logf(in1);
}
RNF_HD void gs36_backward(const float *M, const Rot &R, const Rot &gRout, float g_ldj, float *gM, Rot &gRin) {
    const v3f r0 = R.c0, r1 = R.c1;
    const v3f a0 = blk_mv(M, 0, 0, r0) + blk_mv(M, 0, 1, r1), a1 = blk_mv(M, 1, 0, r0) + blk_mv(M, 1, 1, r1);
    const float in0 = 1.0f / sqrtf(dot3(a0, a0));
    const v3f t0 = a0 * in0;
    const float d = dot3(t0, a1);
    const v3f b1 = a1 - t0 * d;
    const float in1 = 1.0f / sqrtf(dot3(b1, b1));
    const v3f t1 = b1 * in1, t2 = cross3(t0, t1);
    const v3f c00 = blk_mtv(M, 0, 0, t2), c01 = blk_mtv(M, 0, 1, t2), e00 = blk_mtv(M, 0, 0, t1), e01 = blk_mtv(M, 0, 1, t1),
              f10 = blk_mtv(M, 1, 0, t2), f11 = blk_mtv(M, 1, 1, t2);
    const v3f p = cross3(r0, c00) + cross3(r1, c01), q = cross3(r0, e00) + cross3(r1, e01), u = cross3(r0, f10) + cross3(r1, f11);
    const v3f pq = cross3(p, q);
    const float gD = g_ldj / dot3(u, pq);
    const v3f g_u = pq * gD, g_p = cross3(q, u) * gD, g_q = cross3(u, p) * gD;
    // x = r x c:  g_r += c x g_x,  g_c = g_x x r
    v3f g_r0 = cross3(c00, g_p) + cross3(e00, g_q) + cross3(f10, g_u);
    v3f g_r1 = cross3(c01, g_p) + cross3(e01, g_q) + cross3(f11, g_u);
    const v3f g_c00 = cross3(g_p, r0), g_c01 = cross3(g_p, r1), g_e00 = cross3(g_q, r0), g_e01 = cross3(g_q, r1),
              g_f10 = cross3(g_u, r0), g_f11 = cross3(g_u, r1);
    // c = M_b^T t:  gM_b += t g_c^T,  g_t += M_b g_c
    blk_outer_add(gM, 0, 0, t2, g_c00); blk_outer_add(gM, 0, 1, t2, g_c01);
    blk_outer_add(gM, 0, 0, t1, g_e00); blk_outer_add(gM, 0, 1, t1, g_e01);
    blk_outer_add(gM, 1, 0, t2, g_f10); blk_outer_add(gM, 1, 1, t2, g_f11);
    v3f g_t2 = gRout.c2 + blk_mv(M, 0, 0, g_c00) + blk_mv(M, 0, 1, g_c01) + blk_mv(M, 1, 0, g_f10) + blk_mv(M, 1, 1, g_f11);
    v3f g_t1 = gRout.c1 + blk_mv(M, 0, 0, g_e00) + blk_mv(M, 0, 1, g_e01);
    v3f g_t0 = gRout.c0;
    // t2 = t0 x t1
    g_t0 = g_t0 + cross3(t1, g_t2);
    g_t1 = g_t1 + cross3(g_t2, t0);
    // t1 = b1 / |b1|;  - log|b1|
    const v3f g_b1 = normalize_bwd(t1, in1, g_t1) - t1 * (g_ldj * in1);
    // b1 = a1 - d t0, d = t0 . a1
    const float g_d = -dot3(g_b1, t0);
    g_t0 = g_t0 - g_b1 * d + a1 * g_d;
    const v3f g_a1 = g_b1 + t0 * g_d;
    // t0 = a0 / |a0|;  - 2 log|a0|
    const v3f g_a0 = normalize_bwd(t0, in0, g_t0) - t0 * (2.0f * g_ldj * in0);
    blk_outer_add(gM, 0, 0, g_a0, r0); blk_outer_add(gM, 0, 1, g_a0, r1);
    blk_outer_add(gM, 1, 0, g_a1, r0); blk_outer_add(gM, 1, 1, g_a1, r1);
    g_r0 = g_r0 + blk_mtv(M, 0, 0, g_a0) + blk_mtv(M, 1, 0, g_a1);
    g_r1 = g_r1 + blk_mtv(M, 0, 1, g_a0) + blk_mtv(M, 1, 1, g_a1);
    gRin.c0 = g_r0; gRin.c1 = g_r1; gRin.c2 = v3f{0.f, 0.f, 0.f};
}

// ---------------------------------------------------------------------------------------------------------------------
// Conditional 3x3 rotation layers (flow/rottrans.py:68-91,108-181): per-sample M = I + reshape(net(feature), 3, 3).
//   Condition9RotL:      R' = Q R      Condition9RotR: R' = R Q,      Q = U V^T of svd(M R) R^T = the orthogonal polar factor of M
//                        (R is orthogonal), inverse pass: M^T, i.e. Q^T;  ldj = 0
//   Condition9RotRSmith: R' = R N,     N = Gram-Schmidt of the COLUMNS of M, inverse pass: R N^T;  ldj = 0
// The reference differentiates torch.svd; here the polar factor is differentiated directly: M = Q S (S symmetric positive definite),
// dQ = Q Omega with Omega skew and Omega S + S Omega = Q^T dM - dM^T Q, hence for G = Q^T (dL/dQ):
//     dL/dM = Q hat(z),   (tr(S) I - S) z = vee(G - G^T)          (hat(z) S + S hat(z) = hat((tr(S) I - S) z) for symmetric S)
// Q given by its rows (polar3 of so3_math.h), gQ row-major.
// ---------------------------------------------------------------------------------------------------------------------
RNF_HD void polar3_backward(const float (&M)[9], v3f p0, v3f p1, v3f p2, const float (&gQ)[9], float (&gM)[9]) {
    const float Q[9] = {p0.x, p0.y, p0.z, p1.x, p1.y, p1.z, p2.x, p2.y, p2.z};
    float S[9], G[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            S[3 * i + j] = Q[i] * M[j] + Q[3 + i] * M[3 + j] + Q[6 + i] * M[6 + j];            // Q^T M
            G[3 * i + j] = Q[i] * gQ[j] + Q[3 + i] * gQ[3 + j] + Q[6 + i] * gQ[6 + j];          // Q^T gQ
        }
    const float tr = S[0] + S[4] + S[8];
    // symmetrise S (it is symmetric up to rounding) and build T = tr(S) I - S
    const float s01 = 0.5f * (S[1] + S[3]), s02 = 0.5f * (S[2] + S[6]), s12 = 0.5f * (S[5] + S[7]);
    const float T[9] = {tr - S[0], -s01, -s02, -s01, tr - S[4], -s12, -s02, -s12, tr - S[8]};
    float Ti[9];
    inv3(T, Ti);
    const v3f w = v3f{G[7] - G[5], G[2] - G[6], G[3] - G[1]};                                 // vee(G - G^T)
    const v3f z = v3f{Ti[0] * w.x + Ti[1] * w.y + Ti[2] * w.z, Ti[3] * w.x + Ti[4] * w.y + Ti[5] * w.z, Ti[6] * w.x + Ti[7] * w.y + Ti[8] * w.z};
    const float Z[9] = {0.f, -z.z, z.y, z.z, 0.f, -z.x, -z.y, z.x, 0.f};                       // hat(z)
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) gM[3 * i + j] = Q[3 * i] * Z[j] + Q[3 * i + 1] * Z[3 + j] + Q[3 * i + 2] * Z[6 + j];
}

// One conditional 3x3 layer: forward from the saved input (so that the caller need not keep the output) and the reverse step.
// kind: RNF_KIND_COND9_*; inverse: the layer ran inside Flow.inverse.  gM is dL/d(net output) (the identity is a constant).
RNF_HD void cond9_backward(int kind, bool inverse, const float (&M)[9], const Rot &Rin, const Rot &gRout, float g_ldj, float (&gM)[9], Rot &gRin) {
    for (int i = 0; i < 9; ++i) gM[i] = 0.f;
    if (kind == RNF_KIND_COND9_GS) {                       // Condition9Trans (squeezetrans.py:234-247): inverse pass applies M^-1
        if (inverse) {
            float Mi[9], gMi[9];
            inv3(M, Mi);
            for (int i = 0; i < 9; ++i) gMi[i] = 0.f;
            gs9_backward(Mi, Rin, gRout, g_ldj, gMi, gRin);
            inverse_matrix_grad<3>(Mi, gMi, gM);
        } else {
            gs9_backward(M, Rin, gRout, g_ldj, gM, gRin);
        }
        return;
    }
    const v3f rc[3] = {Rin.c0, Rin.c1, Rin.c2}, gc[3] = {gRout.c0, gRout.c1, gRout.c2};
    auto rin = [&](int i, int j) { const v3f c = rc[j]; return i == 0 ? c.x : (i == 1 ? c.y : c.z); };      // R[i][j]
    auto gout = [&](int i, int j) { const v3f c = gc[j]; return i == 0 ? c.x : (i == 1 ? c.y : c.z); };
    float N[9], gN[9], gRi[9];                              // the rotation that multiplies R, and its gradient
    const bool left = kind == RNF_KIND_COND9_POLAR_L;
    v3f p0, p1, p2;
    if (kind == RNF_KIND_COND9_SMITH) {
        v3f q0, q1, q2;
        smith3(M, q0, q1, q2);
        const float Nf[9] = {q0.x, q1.x, q2.x, q0.y, q1.y, q2.y, q0.z, q1.z, q2.z};
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) N[3 * i + j] = inverse ? Nf[3 * j + i] : Nf[3 * i + j];
    } else {
        polar3(M, p0, p1, p2);
        const float Qf[9] = {p0.x, p0.y, p0.z, p1.x, p1.y, p1.z, p2.x, p2.y, p2.z};
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) N[3 * i + j] = inverse ? Qf[3 * j + i] : Qf[3 * i + j];
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float a = 0.f, b = 0.f;
            for (int l = 0; l < 3; ++l) {
                if (left) { a += gout(i, l) * rin(j, l); b += N[3 * l + i] * gout(l, j); }       // R' = N R: gN = gR' R^T, gR = N^T gR'
                else      { a += rin(l, i) * gout(l, j); b += gout(i, l) * N[3 * j + l]; }       // R' = R N: gN = R^T gR', gR = gR' N^T
            }
            gN[3 * i + j] = a;
            gRi[3 * i + j] = b;
        }
    gRin.c0 = v3f{gRi[0], gRi[3], gRi[6]}; gRin.c1 = v3f{gRi[1], gRi[4], gRi[7]}; gRin.c2 = v3f{gRi[2], gRi[5], gRi[8]};
    float gF[9];                                            // gradient w.r.t. the un-transposed factor
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) gF[3 * i + j] = inverse ? gN[3 * j + i] : gN[3 * i + j];
    if (kind == RNF_KIND_COND9_SMITH) {                     // N = Gram-Schmidt of M's columns = the rotation gs9 makes of M I
        Rot I, gq, dump;
        I.c0 = v3f{1.f, 0.f, 0.f}; I.c1 = v3f{0.f, 1.f, 0.f}; I.c2 = v3f{0.f, 0.f, 1.f};
        gq.c0 = v3f{gF[0], gF[3], gF[6]}; gq.c1 = v3f{gF[1], gF[4], gF[7]}; gq.c2 = v3f{gF[2], gF[5], gF[8]};
        gs9_backward(M, I, gq, 0.f, gM, dump);
    } else {
        polar3_backward(M, p0, p1, p2, gF, gM);
    }
}

}  // namespace rnf
