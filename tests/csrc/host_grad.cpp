// TEST INFRASTRUCTURE: csrc/so3_grad.h compiled for the HOST so the reverse-mode formulas of the training path can be
// checked on a CPU against torch autograd of the oracle (tests/test_host_grad.py).  Not part of librnf_hip.so.
#include "../../rotationnormflow_amd/csrc/so3_grad.h"
#include "../../rotationnormflow_amd/csrc/fisher_math.h"

using namespace rnf;

static Rot load_rot(const float *s) {
    Rot R;
    R.c0 = v3f{s[0], s[3], s[6]}; R.c1 = v3f{s[1], s[4], s[7]}; R.c2 = v3f{s[2], s[5], s[8]};
    return R;
}
static void store_rot(const Rot &R, float *d) {
    d[0] = R.c0.x; d[1] = R.c1.x; d[2] = R.c2.x; d[3] = R.c0.y; d[4] = R.c1.y; d[5] = R.c2.y; d[6] = R.c0.z; d[7] = R.c1.z; d[8] = R.c2.z;
}

extern "C" {
void hg_mobius(const float *Rin, int perm_row, const float *cond, int K, const float *gRout, const float *g_ldj, int n, float *Rout,
               float *ldj, float *g_cond, float *gRin) {
    for (int i = 0; i < n; ++i) {
        Rot R = load_rot(Rin + 9 * i), Ro, gi;
        MobiusSaved sv;
        float l;
        mobius_segments_forward(R, perm_row, StridedRow{const_cast<float *>(cond) + (size_t)4 * K * i, 1}, K, Ro, l, sv);
        store_rot(Ro, Rout + 9 * i);
        ldj[i] = l;
        mobius_segments_backward(sv, StridedRow{const_cast<float *>(cond) + (size_t)4 * K * i, 1}, K, load_rot(gRout + 9 * i), g_ldj[i],
                                 StridedRow{g_cond + (size_t)4 * K * i, 1}, gi);
        store_rot(gi, gRin + 9 * i);
    }
}
// backward of MobiusFlow.inverse given the layer's input, its output (which carries the root theta) and the conditioner output
void hg_mobius_inverse(const float *Rin, int perm_row, const float *Rout, const float *cond, int K, const float *gRout, const float *g_ldj, int n,
                       float *g_cond, float *gRin) {
    for (int i = 0; i < n; ++i) {
        Rot gi;
        mobius_inverse_backward(load_rot(Rin + 9 * i), perm_row, load_rot(Rout + 9 * i), StridedRow{const_cast<float *>(cond) + (size_t)4 * K * i, 1}, K,
                                load_rot(gRout + 9 * i), g_ldj[i], StridedRow{g_cond + (size_t)4 * K * i, 1}, gi);
        store_rot(gi, gRin + 9 * i);
    }
}
void hg_inverse_matrix_grad4(const float *Minv, const float *gMinv, float *gM) { inverse_matrix_grad<4>(Minv, gMinv, gM); }
void hg_affine(const float *M, float logabsdet, const float *Rin, const float *gRout, const float *g_ldj, int n, float *Rout, float *ldj,
               float *gM, float *gRin) {
    float m[16], gm[16];
    for (int k = 0; k < 16; ++k) { m[k] = M[k]; gm[k] = 0.f; }
    for (int i = 0; i < n; ++i) {
        Rot Ro, gi;
        AffineSaved sv;
        float l;
        affine16_forward_saved(m, logabsdet, load_rot(Rin + 9 * i), Ro, l, sv);
        store_rot(Ro, Rout + 9 * i);
        ldj[i] = l;
        affine16_backward(m, sv, load_rot(gRout + 9 * i), g_ldj[i], false, gm, gi);
        store_rot(gi, gRin + 9 * i);
    }
    for (int k = 0; k < 16; ++k) gM[k] = gm[k];
}
void hg_gs9(const float *M, const float *Rin, const float *gRout, const float *g_ldj, int n, float *Rout, float *ldj, float *gM, float *gRin) {
    float m[9], gm[9];
    for (int k = 0; k < 9; ++k) { m[k] = M[k]; gm[k] = 0.f; }
    for (int i = 0; i < n; ++i) {
        Rot R = load_rot(Rin + 9 * i), gi;
        const Rot Rin0 = R;
        float l = 0.f;
        gs9_apply(m, R, l);
        store_rot(R, Rout + 9 * i);
        ldj[i] = l;
        gs9_backward(m, Rin0, load_rot(gRout + 9 * i), g_ldj[i], gm, gi);
        store_rot(gi, gRin + 9 * i);
    }
    for (int k = 0; k < 9; ++k) gM[k] = gm[k];
}
// matrix-Fisher log-constant and its derivative for fixed Q (fisher_math.h); proper SVD factors for inspection
void hg_fisher_const(const double *A, int B, int norm_type, double Q, double *c, double *dc, double *U, double *S, double *V) {
    for (int b = 0; b < B; ++b) {
        c[b] = fisher_log_const(A + 9 * b, norm_type, Q, dc + 9 * b);
        proper_svd3(A + 9 * b, U + 9 * b, S + 3 * b, V + 9 * b);
    }
}
// conditional 3x3 layers (so3_grad.h cond9_backward): per-sample M [n][9]; returns dL/dM and dL/dR_in
void hg_cond9(int kind, int inverse, const float *M, const float *Rin, const float *gRout, const float *g_ldj, int n, float *gM, float *gRin) {
    for (int i = 0; i < n; ++i) {
        float m[9], gm[9];
        for (int k = 0; k < 9; ++k) m[k] = M[9 * i + k];
        Rot gi;
        cond9_backward(kind, inverse != 0, m, load_rot(Rin + 9 * i), load_rot(gRout + 9 * i), g_ldj[i], gm, gi);
        for (int k = 0; k < 9; ++k) gM[9 * i + k] = gm[k];
        store_rot(gi, gRin + 9 * i);
    }
}
// calculate_36 (so3_grad.h gs36_backward + its closed-form forward): one M [36] for all samples (per_sample = 0) or M [n][36]
void hg_gs36(const float *M, int per_sample, int inverse, const float *Rin, const float *gRout, const float *g_ldj, int n, float *Rout, float *ldj,
             float *gM, float *gRin) {
    for (int i = 0; i < n; ++i) {
        float m[36], gm[36], mi[36], gmi[36];
        for (int k = 0; k < 36; ++k) { m[k] = M[(per_sample ? 36 * i : 0) + k]; gm[k] = 0.f; gmi[k] = 0.f; }
        Rot Ro, gi;
        float l;
        if (inverse) {
            float tmp[36];
            for (int k = 0; k < 36; ++k) tmp[k] = m[k];
            inv6(tmp, mi);
            gs36_closed_form(mi, load_rot(Rin + 9 * i), Ro, l);
            gs36_backward(mi, load_rot(Rin + 9 * i), load_rot(gRout + 9 * i), g_ldj[i], gmi, gi);
            inverse_matrix_grad<6>(mi, gmi, gm);
        } else {
            gs36_closed_form(m, load_rot(Rin + 9 * i), Ro, l);
            gs36_backward(m, load_rot(Rin + 9 * i), load_rot(gRout + 9 * i), g_ldj[i], gm, gi);
        }
        store_rot(Ro, Rout + 9 * i);
        ldj[i] = l;
        for (int k = 0; k < 36; ++k) gM[(per_sample ? 36 * i : 0) + k] += gm[k];
        store_rot(gi, gRin + 9 * i);
    }
}
}
