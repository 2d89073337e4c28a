// fisher_math.h -- log-normaliser of the matrix-Fisher density and its derivative w.r.t. the parameter matrix A.
//
// Reference: utils/fisher.py:67-76 (proper_svd_N), :79-97 (matrix_fisher_norm_N, type_approx 0 and 1), :217-232 (_log_prob):
//     log p(R) = tr(A^T R) - c(A),      c = s0 + s1 + s2 + log norm(s),   s = PROPER singular values of A
// (A = U' diag(s) V'^T with U', V' in SO(3); the smallest value carries the sign of det A).  agent.py:57-65 keeps a network-predicted A
// in the autograd graph, so training needs dc/dA; for a function of the proper singular values  dc/dA = U' diag(dc/ds) V'^T.
//   type 1:  c = sum s - 1/2 log(8 pi (s0+s1)(s1+s2)(s0+s2))
//   type 0:  norm = (1 + Q/6 + s0 s1 s2 / 6) / exp(sum s)  with Q = (S**2).sum() over the WHOLE batch of matrices (the reference's call has
//            no `dim`), i.e. Q = sum_b |A_b|_F^2 and s0 s1 s2 = det A:  c_b = log(1 + Q/6 + det(A_b)/6)  -- no SVD at all.
// Plain double arithmetic, host + device (checked on the CPU against autograd of the oracle, tests/test_host_grad.py).
#pragma once
#include <cmath>

#ifndef RNF_FM_HD
#if defined(__HIPCC__)
#define RNF_FM_HD __host__ __device__ inline
#else
#define RNF_FM_HD inline
#endif
#endif

namespace rnf {

RNF_FM_HD double det3d(const double a[9]) {
    return a[0] * (a[4] * a[8] - a[5] * a[7]) - a[1] * (a[3] * a[8] - a[5] * a[6]) + a[2] * (a[3] * a[7] - a[4] * a[6]);
}

// cofactor matrix: d det(A) / dA
RNF_FM_HD void cofactor3d(const double a[9], double c[9]) {
    c[0] = a[4] * a[8] - a[5] * a[7];  c[1] = a[5] * a[6] - a[3] * a[8];  c[2] = a[3] * a[7] - a[4] * a[6];
    c[3] = a[2] * a[7] - a[1] * a[8];  c[4] = a[0] * a[8] - a[2] * a[6];  c[5] = a[1] * a[6] - a[0] * a[7];
    c[6] = a[1] * a[5] - a[2] * a[4];  c[7] = a[2] * a[3] - a[0] * a[5];  c[8] = a[0] * a[4] - a[1] * a[3];
}

// Proper SVD of a 3x3 matrix (row-major a[9]): eigen-decomposition of A^T A by cyclic Jacobi (robust for repeated values), columns
// sorted by decreasing singular value, u_i = A v_i / s_i for the two dominant ones, u2 = u0 x u1 and v2 = v0 x v1 so that both
// factors are rotations, s2 = u2^T A v2 (signed).  U, V row-major with the vectors as COLUMNS.
RNF_FM_HD void proper_svd3(const double a[9], double U[9], double s[3], double V[9]) {
    double m[3][3], v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) m[i][j] = a[i] * a[j] + a[3 + i] * a[3 + j] + a[6 + i] * a[6 + j];
    for (int sweep = 0; sweep < 12; ++sweep) {
        const double off = fabs(m[0][1]) + fabs(m[0][2]) + fabs(m[1][2]);
        if (off <= 1e-300 || off <= 1e-18 * (fabs(m[0][0]) + fabs(m[1][1]) + fabs(m[2][2]))) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (m[p][q] == 0.0) continue;
                const double theta = (m[q][q] - m[p][p]) / (2.0 * m[p][q]);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
                for (int k = 0; k < 3; ++k) {               // M <- M J,  V <- V J
                    const double mkp = m[k][p], mkq = m[k][q];
                    m[k][p] = cs * mkp - sn * mkq;
                    m[k][q] = sn * mkp + cs * mkq;
                    const double vkp = v[k][p], vkq = v[k][q];
                    v[k][p] = cs * vkp - sn * vkq;
                    v[k][q] = sn * vkp + cs * vkq;
                }
                for (int k = 0; k < 3; ++k) {               // M <- J^T M
                    const double mpk = m[p][k], mqk = m[q][k];
                    m[p][k] = cs * mpk - sn * mqk;
                    m[q][k] = sn * mpk + cs * mqk;
                }
            }
    }
    int o0 = 0, o1 = 1, o2 = 2, t;                          // order of decreasing eigenvalue
    if (m[o0][o0] < m[o1][o1]) { t = o0; o0 = o1; o1 = t; }
    if (m[o1][o1] < m[o2][o2]) { t = o1; o1 = o2; o2 = t; }
    if (m[o0][o0] < m[o1][o1]) { t = o0; o0 = o1; o1 = t; }
    double v0[3] = {v[0][o0], v[1][o0], v[2][o0]}, v1[3] = {v[0][o1], v[1][o1], v[2][o1]};
    double v2[3] = {v0[1] * v1[2] - v0[2] * v1[1], v0[2] * v1[0] - v0[0] * v1[2], v0[0] * v1[1] - v0[1] * v1[0]};
    double u0[3], u1[3], u2[3];
    for (int i = 0; i < 3; ++i) {
        u0[i] = a[3 * i] * v0[0] + a[3 * i + 1] * v0[1] + a[3 * i + 2] * v0[2];
        u1[i] = a[3 * i] * v1[0] + a[3 * i + 1] * v1[1] + a[3 * i + 2] * v1[2];
    }
    s[0] = sqrt(u0[0] * u0[0] + u0[1] * u0[1] + u0[2] * u0[2]);
    const double i0 = s[0] > 0.0 ? 1.0 / s[0] : 0.0;
    for (int i = 0; i < 3; ++i) u0[i] *= i0;
    const double d01 = u0[0] * u1[0] + u0[1] * u1[1] + u0[2] * u1[2];      // re-orthogonalise (exact up to rounding already)
    for (int i = 0; i < 3; ++i) u1[i] -= d01 * u0[i];
    s[1] = sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
    const double i1 = s[1] > 0.0 ? 1.0 / s[1] : 0.0;
    for (int i = 0; i < 3; ++i) u1[i] *= i1;
    u2[0] = u0[1] * u1[2] - u0[2] * u1[1];
    u2[1] = u0[2] * u1[0] - u0[0] * u1[2];
    u2[2] = u0[0] * u1[1] - u0[1] * u1[0];
    s[2] = 0.0;
    for (int i = 0; i < 3; ++i) s[2] += u2[i] * (a[3 * i] * v2[0] + a[3 * i + 1] * v2[1] + a[3 * i + 2] * v2[2]);
    for (int i = 0; i < 3; ++i) {
        U[3 * i] = u0[i]; U[3 * i + 1] = u1[i]; U[3 * i + 2] = u2[i];
        V[3 * i] = v0[i]; V[3 * i + 1] = v1[i]; V[3 * i + 2] = v2[i];
    }
}

// c(A) for norm_type 0 / 1; Q = sum over the batch of |A_b|_F^2 (type 0 only).  dc (optional): dc/dA_b for FIXED Q.
RNF_FM_HD double fisher_log_const(const double a[9], int norm_type, double Q, double *dc) {
    if (norm_type == 0) {
        const double D = 1.0 + Q / 6.0 + det3d(a) / 6.0;
        if (dc) {
            cofactor3d(a, dc);
            for (int k = 0; k < 9; ++k) dc[k] /= 6.0 * D;
        }
        return log(D);
    }
    double U[9], s[3], V[9];
    proper_svd3(a, U, s, V);
    const double p01 = s[0] + s[1], p12 = s[1] + s[2], p02 = s[0] + s[2];
    if (dc) {
        const double f[3] = {1.0 - 0.5 * (1.0 / p01 + 1.0 / p02), 1.0 - 0.5 * (1.0 / p01 + 1.0 / p12), 1.0 - 0.5 * (1.0 / p12 + 1.0 / p02)};
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) dc[3 * i + j] = f[0] * U[3 * i] * V[3 * j] + f[1] * U[3 * i + 1] * V[3 * j + 1] + f[2] * U[3 * i + 2] * V[3 * j + 2];
    }
    return s[0] + s[1] + s[2] - 0.5 * log(8.0 * 3.14159265358979323846 * p01 * p12 * p02);
}

}  // namespace rnf
