// TEST INFRASTRUCTURE: compiles csrc/so3_math.h for the HOST so the per-sample algebra of the kernels can be checked on
// a CPU against float64 numpy (tests/test_host_math.py).  Not part of librnf_hip.so.
#include "../../rotationnormflow_amd/csrc/so3_math.h"

using namespace rnf;

extern "C" {
void hm_softplus(const float *x, float *y, int n) { for (int i = 0; i < n; ++i) y[i] = softplus(x[i]); }
void hm_angle(const float *y, const float *x, float *o, int n) { for (int i = 0; i < n; ++i) o[i] = angle_0_2pi(y[i], x[i]); }
void hm_sincos(const float *x, float *s, float *c, int n) { for (int i = 0; i < n; ++i) sincos_small(x[i], s[i], c[i]); }
void hm_inv4(const float *m, float *o, float *det, int n) {
    for (int i = 0; i < n; ++i) {
        float a[16], b[16];
        for (int k = 0; k < 16; ++k) a[k] = m[16 * i + k];
        det[i] = inv4(a, b);
        for (int k = 0; k < 16; ++k) o[16 * i + k] = b[k];
    }
}
// calculate_16 on row-major rotations with one shared M
void hm_affine16(const float *M, float logabsdet, const float *Rin, float *Rout, float *ldj, int n) {
    float m[16];
    for (int k = 0; k < 16; ++k) m[k] = M[k];
    for (int i = 0; i < n; ++i) {
        const float *s = Rin + 9 * i;
        Rot R;
        R.c0 = v3f{s[0], s[3], s[6]}; R.c1 = v3f{s[1], s[4], s[7]}; R.c2 = v3f{s[2], s[5], s[8]};
        float l = 0.f;
        affine16_apply(m, logabsdet, R, l);
        float *d = Rout + 9 * i;
        d[0] = R.c0.x; d[1] = R.c1.x; d[2] = R.c2.x; d[3] = R.c0.y; d[4] = R.c1.y; d[5] = R.c2.y; d[6] = R.c0.z; d[7] = R.c1.z; d[8] = R.c2.z;
        ldj[i] = l;
    }
}
// the same layer through the packer's 10x10 table (what the stack kernels run for a constant M)
void hm_affine16_table(const float *M, float logabsdet, const float *Rin, float *Rout, float *ldj, int n) {
    double m[16];
    for (int k = 0; k < 16; ++k) m[k] = M[k];
    float T[104];                                 // AFF_TABLE_FLOATS (layout.h)
    affine16_table(m, logabsdet, false, T);
    for (int i = 0; i < n; ++i) {
        const float *s = Rin + 9 * i;
        Rot R;
        R.c0 = v3f{s[0], s[3], s[6]}; R.c1 = v3f{s[1], s[4], s[7]}; R.c2 = v3f{s[2], s[5], s[8]};
        float l = 0.f;
        affine16_table_apply(T, R, l);
        float *d = Rout + 9 * i;
        d[0] = R.c0.x; d[1] = R.c1.x; d[2] = R.c2.x; d[3] = R.c0.y; d[4] = R.c1.y; d[5] = R.c2.y; d[6] = R.c0.z; d[7] = R.c1.z; d[8] = R.c2.z;
        ldj[i] = l;
    }
}
// calculate_9 (n = 3) / calculate_36 (n = 6) on row-major rotations with one shared M
void hm_gs(const float *M, int nm, const float *Rin, float *Rout, float *ldj, int n) {
    for (int i = 0; i < n; ++i) {
        const float *s = Rin + 9 * i;
        Rot R;
        R.c0 = v3f{s[0], s[3], s[6]}; R.c1 = v3f{s[1], s[4], s[7]}; R.c2 = v3f{s[2], s[5], s[8]};
        float l = 0.f;
        if (nm == 3) gs9_apply(M, R, l); else gs36_apply(M, R, l);
        float *d = Rout + 9 * i;
        d[0] = R.c0.x; d[1] = R.c1.x; d[2] = R.c2.x; d[3] = R.c0.y; d[4] = R.c1.y; d[5] = R.c2.y; d[6] = R.c0.z; d[7] = R.c1.z; d[8] = R.c2.z;
        ldj[i] = l;
    }
}
// one Moebius forward layer given the raw conditioner outputs (reference row order [K weights | K x 3 centres])
void hm_mobius_forward(const float *Rin, const float *cond, int K, int perm_row, float *Rout, float *ldj, int n) {
    for (int i = 0; i < n; ++i) {
        const float *s = Rin + 9 * i;
        Rot R;
        R.c0 = v3f{s[0], s[3], s[6]}; R.c1 = v3f{s[1], s[4], s[7]}; R.c2 = v3f{s[2], s[5], s[8]};
        const int p0 = perm_row % 3, p1 = (perm_row + 1) % 3, p2 = (perm_row + 2) % 3;
        v3f x = get_col(R, p0), y = get_col(R, p1);
        Frame f = make_frame(x, y);
        float xr = dot3(x, f.r), xv = dot3(x, f.v);
        float inv = 1.0f / sqrtf(xr * xr + xv * xv);
        float zc = xr * inv, zs = xv * inv, zth = angle_0_2pi(xv, xr);
        float S = 0, A = 0, J = 0;
        const float *c = cond + (size_t)4 * K * i;
        for (int k = 0; k < K; ++k) {
            if (k & 1) {                       // exercise both formulations of the segment math
                float ur, uv, phi, cc;
                squash_center(c[K + 3 * k], c[K + 3 * k + 1], c[K + 3 * k + 2], f, ur, uv);
                mobius_angle(zc, zs, zth, ur, uv, phi, cc);
                float sp = softplus(c[k]);
                S += sp; A += sp * phi; J += sp * cc;
            } else {
                segment_full(c[k], c[K + 3 * k], c[K + 3 * k + 1], c[K + 3 * k + 2], f, zc, zs, zth, S, A, J);
            }
        }
        float sn, cs;
        sincos_small(A / S, sn, cs);
        v3f tx = f.v * sn + f.r * cs;
        v3f tz = normalize3(cross3(tx, y));
        set_col(R, p0, tx); set_col(R, p2, tz);
        float *d = Rout + 9 * i;
        d[0] = R.c0.x; d[1] = R.c1.x; d[2] = R.c2.x; d[3] = R.c0.y; d[4] = R.c1.y; d[5] = R.c2.y; d[6] = R.c0.z; d[7] = R.c1.z; d[8] = R.c2.z;
        ldj[i] = logf(J / S);
    }
}
}
