// equalize.h -- pack-time equalisation of the conditioner MLP for the split-precision ("f16x2") kernels.
//
// Why.  The f16x2 kernels carry every operand of the 64-wide GEMMs as x = hi + lo with lo an UNSCALED fp16 (flow_kernels.h), so a pair
// has an ABSOLUTE resolution of 2^-25: 22 significant bits only while |x| >= 2^-3.  A ReLU network is invariant under
//     x_l -> D_l x_l   (D_l a positive diagonal per hidden pre-activation vector):  W_l -> D_l W_l D_{l-1}^-1,  b_l -> D_l b_l,
// so a checkpoint may sit ANYWHERE on that orbit (layer scales of a network trained without weight decay are not balanced): with the
// hidden layers of flow/condition.py:24-30 at 2^-8 of their "natural" scale (compensated in the next layer) the unscaled pairs keep
// ~14 bits and the conditioner output is off by 1e-2 with no overflow, no NaN, no flag.
//
// What.  The packers (host: rnf_api.hip, device: pack_device.h) move every layer to ONE canonical point of its orbit before the
// fp16 split: power-of-two factors (exact in fp32) chosen per hidden unit so that the estimated root-mean-square of every hidden
// pre-activation is in (2^-2, 2^-1] -- the regime of a freshly initialised torch.nn.Linear stack, where the absolute floor of the pairs
// sits 2^-22 below the signal in every layer.  The MLP computes the same function (up to the exactness of power-of-two scaling):
//     x0' = D0 x0 (fc_first image, its bias and the feature-projection rows scaled by D0)
//     x1' = D1 x1,  W1' = D1 W1 D0^-1;   x2' = D2 x2,  W3' = D2 W3 D1^-1;   x3' = D0 x3,  W5' = D0 W5 D2^-1   (the residual x0 + x3
//     of flow/condition.py:29 forces the same factor on x0 and x3);   out = Wl' relu(x0' + x3') + bl,  Wl' = Wl D0^-1.
// Estimate of the mean squares: the conditioning column y is a unit vector (E y_c^2 = 1/3), a ReLU passes half of the second moment, and
// the mean square of a feature entry, `feat_ms`, is the ONE data-dependent number: 1 unless the caller says otherwise (rnf_set_feature_ms;
// the Python runtime measures it on the first feature batch a parameter version is packed for -- a flow evaluated on features 40x
// larger than assumed keeps its parity but loses ~5 bits in fc_last, whose columns are scaled down with the activations feeding it):
//     q0_i = |W0y_i|^2 / 3 + feat_ms |W0f_i|^2 + b0_i^2,   q_l,i = 1/2 sum_j W_l,ij^2 q_{l-1,j} + b_l,i^2,   D0 from q0 + q3, D1 from q1, D2 from q2.
// Being off by a binade or two costs nothing; beyond that the error grows in proportion to the mis-estimate (measured: features 40x larger
// than assumed -> 2.5x the mean, 7x the maximum log-det error of a 4-layer flow), which is why feat_ms is an input.  The packers
// additionally AUDIT every packed layer on probe inputs (rnf_api.hip audit_mlp) and refuse f16x2 when the packed network does not
// reproduce the exact one.
//
// Host and device evaluate the SAME inline functions in the same order in double with contraction off, so the two packers stay
// bit-identical (tests/test_gpu_grad.py::test_device_packer_matches_host_packer).
#pragma once
#include <math.h>
#if defined(__HIPCC__)
#define RNF_EQ_HD __host__ __device__ inline
#else
#define RNF_EQ_HD inline
#endif

namespace rnf {

constexpr int EQ_TARGET_EXP = -1;          // scaled rms of a hidden pre-activation lands in (2^(T-1), 2^T]
constexpr int EQ_CLAMP = 60;               // |exponent| bound: the scaled weights stay finite in fp32 for any sane checkpoint

// exponent e with 2^e sqrt(q) in (2^(T-1), 2^T];  q = 0 / non-finite -> 0 (a dead unit: any factor is exact)
RNF_EQ_HD int eq_exponent(double q) {
    if (!(q > 0.0) || !(q < 1.0e300)) return 0;
    int k = 0;
    (void)frexp(q, &k);                    // q = m 2^k, m in [0.5, 1)  =>  sqrt(q) in [2^((k-1)/2), 2^(k/2))
    const int c = k >= 0 ? (k + 1) / 2 : -((-k) / 2);      // ceil(k / 2)
    int e = EQ_TARGET_EXP - c;
    if (e > EQ_CLAMP) e = EQ_CLAMP;
    if (e < -EQ_CLAMP) e = -EQ_CLAMP;
    return e;
}

// mean square of x0_i = fc_first row i applied to (y (+) feature) + bias; `yo` = 3 (Moebius: the first three columns multiply the unit
// vector y) or 0 (Condition16Trans and relatives: feature only)
RNF_EQ_HD double eq_q_first(const float *w_row, int ni, int yo, float b, double feat_ms) {
#pragma clang fp contract(off)
    double sy = 0.0;
    for (int c = 0; c < yo; ++c) sy += (double)w_row[c] * (double)w_row[c];
    double s = sy * (1.0 / 3.0);
    double sf = 0.0;
    for (int c = yo; c < ni; ++c) sf += (double)w_row[c] * (double)w_row[c];
    s += sf * feat_ms;
    s += (double)b * (double)b;
    return s;
}

// mean square of x_l,i = W_l[i] . relu(x_{l-1}) + b_l,i given the mean squares of x_{l-1}
RNF_EQ_HD double eq_q_hidden(const float *w_row /* 64 */, const double *q_prev /* 64 */, float b) {
#pragma clang fp contract(off)
    double s = 0.0;
    for (int j = 0; j < 64; ++j) s += ((double)w_row[j] * (double)w_row[j]) * q_prev[j];
    s = 0.5 * s;
    s += (double)b * (double)b;
    return s;
}

// the three exponent vectors of one MLP (host form; the device packer runs the same per-unit functions one unit per thread)
struct EqExponents {
    int e[3][64];                          // e[0]: x0 / x3 / fc_last input, e[1]: x1, e[2]: x2
};
inline void eq_exponents_host(const float *W0, int ni, int yo, const float *b0, const float *const hw[3], const float *const hb[3], double feat_ms,
                              EqExponents &out) {
    double q0[64], q1[64], q2[64], q3[64];
    for (int i = 0; i < 64; ++i) q0[i] = eq_q_first(W0 + (size_t)i * ni, ni, yo, b0[i], feat_ms);
    for (int i = 0; i < 64; ++i) q1[i] = eq_q_hidden(hw[0] + (size_t)i * 64, q0, hb[0][i]);
    for (int i = 0; i < 64; ++i) q2[i] = eq_q_hidden(hw[1] + (size_t)i * 64, q1, hb[1][i]);
    for (int i = 0; i < 64; ++i) q3[i] = eq_q_hidden(hw[2] + (size_t)i * 64, q2, hb[2][i]);
    for (int i = 0; i < 64; ++i) {
        out.e[0][i] = eq_exponent(q0[i] + q3[i]);
        out.e[1][i] = eq_exponent(q1[i]);
        out.e[2][i] = eq_exponent(q2[i]);
    }
}

}  // namespace rnf
