// rnf_api.hip -- C ABI of librnf_hip.so (include/rnf_hip.h): host-side parameter packing + kernel launchers.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>
#include <vector>

#include "../../include/rnf_hip.h"
#include "featproj_kernel.h"
#include "flow_kernels.h"
#include "train_kernels.h"
#include "train_block16.h"
#include "pack_device.h"
#include "sampler_kernel.h"
#include "fisher_math.h"
#include "layout.h"
#include "equalize.h"
#include "svd4_lapack.h"

using namespace rnf;

static_assert(RNF_LAYER_MOBIUS == RNF_KIND_MOBIUS && RNF_LAYER_AFFINE16 == RNF_KIND_AFFINE16 &&
                  RNF_LAYER_AFFINE16_COND == RNF_KIND_COND16 && RNF_DESC_STRIDE == D_STRIDE && RNF_HIDDEN == HID &&
                  RNF_LAYER_SIDE16 == RNF_KIND_SIDE16 && RNF_LAYER_SIDE16_ROT == RNF_KIND_SIDE16_ROT && RNF_LAYER_SIDE9 == RNF_KIND_SIDE9,
              "include/rnf_hip.h and csrc/layout.h disagree");

// ------------------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}
#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) return fail("%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

extern "C" int rnf_abi_version(void) { return RNF_ABI_VERSION; }
extern "C" const char *rnf_last_error(void) { return g_err; }

// ------------------------------------------------------------------------------------------------------------
// packing (pure host code)
// ------------------------------------------------------------------------------------------------------------
extern "C" int64_t rnf_mobius_packed_floats(int32_t K) { return K > 0 ? mobius_packed_floats(K) : -1; }
extern "C" int64_t rnf_mobius_packed_floats_prec(int32_t K, int32_t prec) { return (K > 0 && prec >= 0 && prec <= 2) ? mobius_packed_floats_p(K, prec) : -1; }
extern "C" int64_t rnf_cond_packed_floats_prec(int32_t n_out, int32_t prec) { return (prec >= 0 && prec <= 2) ? cond_packed_floats_p(n_out == 36 ? 2 : 1, prec) : -1; }
extern "C" int64_t rnf_affine16_packed_floats(void) { return AFF_FLOATS; }
extern "C" int64_t rnf_cond16_packed_floats(void) { return COND16_FLOATS; }
extern "C" int64_t rnf_featproj_packed_floats(int32_t F) { return (F >= 0 && F % 8 == 0) ? featproj_packed_floats(F) : -1; }

// [OUT=32*n_ot][64] row-major weight rows `rowmap(ot, i)` -> image [ot][tg][lane] float4
template <typename RowFn>
static void pack_w64(float *img, int n_ot, RowFn row_of /* (ot, i) -> const float* row of 64 or nullptr */) {
    for (int ot = 0; ot < n_ot; ++ot)
        for (int tg = 0; tg < 8; ++tg)
            for (int lane = 0; lane < 64; ++lane) {
                const int i = lane & 31, h = lane >> 5;
                const float *row = row_of(ot, i);
                float *dst = img + (((size_t)ot * 8 + tg) * 64 + lane) * 4;
                for (int c = 0; c < 4; ++c) dst[c] = row ? row[8 * tg + 4 * h + c] : 0.f;
            }
}

// split-precision image of the same rows: [ot][s (4)][hi, lo][lane] 8 x fp16, element j of lane (i, h) of k-step
// s = 2t + s' is W[row][32t + 16s' + 8(j>>2) + 4h + (j&3)] as hi = fp16(w), lo = fp16(w - hi) (unscaled, flow_kernels.h; the feature
// projection images keep lo scaled by 2^12, layout.h)
static thread_local bool g_half_overflow = false;     // per calling thread: ctypes releases the GIL and two flows may be packed at once
template <typename RowFn>
static void pack_w64_h(float *img, int n_ot, RowFn row_of) {
    _Float16 *out = reinterpret_cast<_Float16 *>(img);
    for (int ot = 0; ot < n_ot; ++ot)
        for (int s = 0; s < 4; ++s)
            for (int lane = 0; lane < 64; ++lane) {
                const int i = lane & 31, h = lane >> 5;
                const float *row = row_of(ot, i);
                _Float16 *hi = out + ((((size_t)ot * 4 + s) * 2 + 0) * 64 + lane) * 8;
                _Float16 *lo = out + ((((size_t)ot * 4 + s) * 2 + 1) * 64 + lane) * 8;
                for (int j = 0; j < 8; ++j) {
                    const float w = row ? row[16 * s + 8 * (j >> 2) + 4 * h + (j & 3)] : 0.f;
                    if (!(std::fabs(w) < 65504.0f)) g_half_overflow = true;      // also catches NaN / inf
                    const _Float16 wh = (_Float16)w;
                    hi[j] = wh;
                    lo[j] = (_Float16)((w - (float)wh) * W_LO_SCALE);
                }
            }
}

// bf16x3 image of the same rows (layout.h Lay<2>): [ot][s (4)][hi, mid, lo][lane] 8 x bf16, element j of lane (i, h) of k-step s as in the
// fp16 image; the three terms are the TRUNCATED splits the kernels use for the activations (top 16 bits of w, of w - hi, of the rest):
// hi + mid + lo reproduces w to 2^-24 |w|, nothing to overflow or underflow short of fp32 itself.
static inline uint16_t bf16_trunc(float x) { uint32_t u; std::memcpy(&u, &x, 4); return (uint16_t)(u >> 16); }
static inline float bf16_float(uint16_t b) { uint32_t u = (uint32_t)b << 16; float x; std::memcpy(&x, &u, 4); return x; }
template <typename RowFn>
static void pack_w64_b3(float *img, int n_ot, RowFn row_of) {
    uint16_t *out = reinterpret_cast<uint16_t *>(img);
    for (int ot = 0; ot < n_ot; ++ot)
        for (int s = 0; s < 4; ++s)
            for (int lane = 0; lane < 64; ++lane) {
                const int i = lane & 31, h = lane >> 5;
                const float *row = row_of(ot, i);
                uint16_t *dst[3];
                for (int t = 0; t < 3; ++t) dst[t] = out + ((((size_t)ot * 4 + s) * 3 + t) * 64 + lane) * 8;
                for (int j = 0; j < 8; ++j) {
                    const float w = row ? row[16 * s + 8 * (j >> 2) + 4 * h + (j & 3)] : 0.f;
                    const uint16_t wh = bf16_trunc(w);
                    const float r1 = w - bf16_float(wh);
                    const uint16_t wm = bf16_trunc(r1);
                    const float r2 = r1 - bf16_float(wm);
                    dst[0][j] = wh; dst[1][j] = wm; dst[2][j] = bf16_trunc(r2);
                }
            }
}

// bias image [ot][h][16]: b[32*ot + rho(r,h)] (or the mapped row's bias)
template <typename BiasFn>
static void pack_bias(float *img, int n_ot, BiasFn bias_of /* (ot, row_in_tile) -> float */) {
    for (int ot = 0; ot < n_ot; ++ot)
        for (int h = 0; h < 2; ++h)
            for (int r = 0; r < 16; ++r) img[(ot * 2 + h) * 16 + r] = bias_of(ot, rho(r, h));
}

static void pack_hidden(float *out, const float *const w[3], const float *const b[3], int prec) {
    const int w_tile = prec == 2 ? Lay<2>::W_TILE : Lay<0>::W_TILE, hb = prec == 2 ? Lay<2>::HB : MOB_HB;
    for (int L = 0; L < 3; ++L) {
        const float *W = w[L];
        auto row_of = [&](int ot, int i) { return W + (size_t)(32 * ot + i) * 64; };
        float *dst = out + MOB_HID + (size_t)L * 2 * w_tile;
        if (prec == 2) pack_w64_b3(dst, 2, row_of);
        else if (prec) pack_w64_h(dst, 2, row_of);
        else pack_w64(dst, 2, row_of);
        const float *B = b[L];
        pack_bias(out + hb + L * 2 * 2 * 16, 2, [&](int ot, int row) { return B[32 * ot + row]; });
    }
}

// feature projection record: weights Wf [64][ldw] starting at column col0, F columns; bias b0 [64]
static void pack_featproj_h(float *out, const float *W, int ldw, int col0, int F, const float *b0) {
    const int ns = (F + 15) / 16;
    _Float16 *o16 = reinterpret_cast<_Float16 *>(out);
    for (int ot = 0; ot < 2; ++ot)
        for (int s = 0; s < ns; ++s)
            for (int lane = 0; lane < 64; ++lane) {
                const int i = lane & 31, h = lane >> 5;
                _Float16 *hi = o16 + ((((size_t)ot * ns + s) * 2 + 0) * 64 + lane) * 8;
                _Float16 *lo = o16 + ((((size_t)ot * ns + s) * 2 + 1) * 64 + lane) * 8;
                for (int j = 0; j < 8; ++j) {
                    const int k = 16 * s + 8 * h + j;
                    const float w = k < F ? W[(size_t)(32 * ot + i) * ldw + col0 + k] : 0.f;
                    if (!(std::fabs(w) < 65504.0f)) g_half_overflow = true;
                    const _Float16 wh = (_Float16)w;
                    hi[j] = wh;
                    lo[j] = (_Float16)((w - (float)wh) * FEAT_LO_SCALE);
                }
            }
    // bias image: zeros since round 5 -- b0 rides in the bias slot of the stack kernel's fc_first image (pack_first_bias), where it costs
    // nothing (that matrix step exists anyway); the projection kernels start every accumulator at zero (no bias loads, 16 fewer live registers)
    (void)b0;
    pack_bias(out + (size_t)2 * ns * 512, 2, [&](int, int) { return 0.f; });
}

static void pack_featproj(float *out, const float *W, int ldw, int col0, int F, const float *b0) {
    const int ng = F / 8;
    for (int ot = 0; ot < 2; ++ot)
        for (int tg = 0; tg < ng; ++tg)
            for (int lane = 0; lane < 64; ++lane) {
                const int i = lane & 31, h = lane >> 5;
                float *dst = out + (((size_t)ot * ng + tg) * 64 + lane) * 4;
                for (int c = 0; c < 4; ++c) dst[c] = W[(size_t)(32 * ot + i) * ldw + col0 + 8 * tg + 4 * h + c];
            }
    (void)b0;                                              // see pack_featproj_h: the bias lives in the fc_first image
    pack_bias(out + (size_t)2 * ng * 256, 2, [&](int, int) { return 0.f; });
}


// ---- the conditioner MLP on the canonical point of its ReLU-rescaling orbit (equalize.h): what the f16x2 images are split from ----
struct ScaledMlp {
    int ni = 0, yo = 0, n_out = 0;
    std::vector<float> W0, b0, hw[3], hb[3], WL;     // W0 [64][ni], hw[l] [64][64], WL [n_out][64]
    EqExponents eq;
};

// RNF_EQUALIZE=0 / rnf_set_equalize(0): split the weights as given (measurement / test switch: what the audit and the range guard
// catch without the equalisation)
static int g_equalize = -1;
static bool equalise_allowed() {
    if (g_equalize < 0) {
        const char *e = std::getenv("RNF_EQUALIZE");
        g_equalize = (e && e[0] == '0') ? 0 : 1;
    }
    return g_equalize == 1;
}
extern "C" int rnf_set_equalize(int on) {
    const int old = equalise_allowed() ? 1 : 0;
    g_equalize = on ? 1 : 0;
    return old;
}
// the audit's refusal can be switched off separately (tests of the run-time guard need an unequalised, unaudited image)
static int g_audit = 1;
extern "C" int rnf_set_pack_audit(int on) {
    const int old = g_audit;
    g_audit = on ? 1 : 0;
    return old;
}

// mean square of a feature entry assumed by the equalisation of the packers called from this thread (equalize.h); default 1
static thread_local double g_feature_ms = 1.0;
extern "C" double rnf_set_feature_ms(double ms) {
    const double old = g_feature_ms;
    g_feature_ms = (ms > 1.0e-20 && ms < 1.0e20) ? ms : 1.0;
    return old;
}

static void scale_mlp(const float *fc_first_w, int ni, int yo, const float *fc_first_b, const float *const hw[3], const float *const hb[3],
                      const float *fc_last_w, int n_out, bool equalise, ScaledMlp &m) {
    m.ni = ni; m.yo = yo; m.n_out = n_out;
    if (equalise && equalise_allowed()) eq_exponents_host(fc_first_w, ni, yo, fc_first_b, hw, hb, g_feature_ms, m.eq);
    else std::memset(&m.eq, 0, sizeof(m.eq));
    const int *e0 = m.eq.e[0], *e1 = m.eq.e[1], *e2 = m.eq.e[2];
    m.W0.resize((size_t)64 * ni);
    m.b0.resize(64);
    for (int o = 0; o < 64; ++o) {
        for (int c = 0; c < ni; ++c) m.W0[(size_t)o * ni + c] = std::ldexp(fc_first_w[(size_t)o * ni + c], e0[o]);
        m.b0[o] = std::ldexp(fc_first_b[o], e0[o]);
    }
    const int *eo[3] = {e1, e2, e0}, *ei[3] = {e0, e1, e2};      // x1 = W1 relu(x0), x2 = W3 relu(x1), x3 = W5 relu(x2) (flow/condition.py:24-29)
    for (int l = 0; l < 3; ++l) {
        m.hw[l].resize(4096);
        m.hb[l].resize(64);
        for (int o = 0; o < 64; ++o) {
            for (int j = 0; j < 64; ++j) m.hw[l][o * 64 + j] = std::ldexp(hw[l][o * 64 + j], eo[l][o] - ei[l][j]);
            m.hb[l][o] = std::ldexp(hb[l][o], eo[l][o]);
        }
    }
    m.WL.resize((size_t)n_out * 64);
    for (int o = 0; o < n_out; ++o)
        for (int j = 0; j < 64; ++j) m.WL[(size_t)o * 64 + j] = std::ldexp(fc_last_w[(size_t)o * 64 + j], -e0[j]);
}

// Pack-time AUDIT of the split-precision images: the packed network (scaled weights and every activation rounded to fp16 hi + lo pairs
// exactly as the kernels hold them, three products per term, accumulation in double so that only the REPRESENTATION error shows) against the
// exact network (the weights as given, double) on probe inputs: unit vectors y, feature rows N(0, 1) and N(0, 1/64) from a fixed
// generator.  Returns max_o |out_packed - out_exact| / max(1, sum_j |Wl_oj t_j| + |bl_o|).  A layer that the equalisation could not bring into the regime
// where the pairs carry ~22 bits (estimates far off, a degenerate checkpoint) is refused like a weight outside the fp16 range, and the
// flow runs on the exact-fp32 kernels.
static inline void split_h(float v, float lo_scale, double &hi, double &lo) {
    const _Float16 h = (_Float16)v;
    hi = (double)(float)h;
    lo = (double)(float)(_Float16)((v - (float)h) * lo_scale) / lo_scale;
}
static double audit_mlp(const ScaledMlp &m, const float *fc_first_w, const float *fc_first_b, const float *const hw[3], const float *const hb[3],
                        const float *fc_last_w, const float *fc_last_b) {
    const int ni = m.ni, yo = m.yo, n_out = m.n_out, F = ni - yo;
    constexpr int PROBES = 6;
    unsigned long long rng = 0x9E3779B97F4A7C15ull;
    auto uni = [&]() { rng = rng * 6364136223846793005ull + 1442695040888963407ull; return (double)((rng >> 11) & ((1ull << 53) - 1)) / (double)(1ull << 53); };
    auto gauss = [&]() { double s = 0.0; for (int i = 0; i < 12; ++i) s += uni(); return s - 6.0; };
    std::vector<double> whi((size_t)3 * 4096), wlo((size_t)3 * 4096), lhi((size_t)n_out * 64), llo((size_t)n_out * 64);
    for (int l = 0; l < 3; ++l)
        for (int i = 0; i < 4096; ++i) split_h(m.hw[l][i], W_LO_SCALE, whi[l * 4096 + i], wlo[l * 4096 + i]);
    for (size_t i = 0; i < (size_t)n_out * 64; ++i) split_h(m.WL[i], W_LO_SCALE, lhi[i], llo[i]);
    std::vector<double> fhi((size_t)64 * (F > 0 ? F : 1)), flo(fhi.size());
    for (int o = 0; o < 64; ++o)
        for (int k = 0; k < F; ++k) split_h(m.W0[(size_t)o * ni + yo + k], FEAT_LO_SCALE, fhi[(size_t)o * F + k], flo[(size_t)o * F + k]);
    double worst = 0.0;
    std::vector<float> in(ni > 0 ? ni : 1);
    for (int p = 0; p < PROBES; ++p) {
        double y[3] = {gauss(), gauss(), gauss()};
        const double yn = std::sqrt(y[0] * y[0] + y[1] * y[1] + y[2] * y[2]) + 1e-30;
        for (int c = 0; c < yo; ++c) in[c] = (float)(y[c] / yn);
        const double fs = ((p & 1) ? 0.125 : 1.0) * std::sqrt(g_feature_ms);
        for (int k = 0; k < F; ++k) in[yo + k] = (float)(fs * gauss());
        // exact
        double x0[64], a[64], b[64], x3[64];
        for (int o = 0; o < 64; ++o) {
            double s = fc_first_b[o];
            for (int c = 0; c < ni; ++c) s += (double)fc_first_w[(size_t)o * ni + c] * in[c];
            x0[o] = s;
            a[o] = s > 0 ? s : 0;
        }
        for (int l = 0; l < 3; ++l) {
            for (int o = 0; o < 64; ++o) {
                double s = hb[l][o];
                for (int j = 0; j < 64; ++j) s += (double)hw[l][o * 64 + j] * a[j];
                b[o] = s;
            }
            for (int o = 0; o < 64; ++o) { x3[o] = b[o]; a[o] = b[o] > 0 ? b[o] : 0; }
        }
        double t[64];
        for (int o = 0; o < 64; ++o) { const double v = x0[o] + x3[o]; t[o] = v > 0 ? v : 0; }
        // packed arithmetic
        float px0[64], pa[64], pb[64];
        std::vector<double> ph(F > 0 ? F : 1), pl(ph.size());
        for (int k = 0; k < F; ++k) split_h(in[yo + k], FEAT_LO_SCALE, ph[k], pl[k]);
        for (int o = 0; o < 64; ++o) {
            double s = m.b0[o];
            for (int c = 0; c < yo; ++c) s += (double)m.W0[(size_t)o * ni + c] * in[c];
            for (int k = 0; k < F; ++k) s += fhi[(size_t)o * F + k] * ph[k] + fhi[(size_t)o * F + k] * pl[k] + flo[(size_t)o * F + k] * ph[k];
            px0[o] = (float)s;
            pa[o] = px0[o] > 0 ? px0[o] : 0.f;
        }
        for (int l = 0; l < 3; ++l) {
            double ah[64], al[64];
            for (int j = 0; j < 64; ++j) split_h(pa[j], W_LO_SCALE, ah[j], al[j]);
            for (int o = 0; o < 64; ++o) {
                double s = m.hb[l][o];
                const double *wh = &whi[l * 4096 + o * 64], *wl = &wlo[l * 4096 + o * 64];
                for (int j = 0; j < 64; ++j) s += wh[j] * ah[j] + wh[j] * al[j] + wl[j] * ah[j];
                pb[o] = (float)s;
            }
            for (int o = 0; o < 64; ++o) pa[o] = pb[o] > 0 ? pb[o] : 0.f;
        }
        double th[64], tl[64];
        for (int j = 0; j < 64; ++j) { const float v = px0[j] + pb[j]; split_h(v > 0 ? v : 0.f, W_LO_SCALE, th[j], tl[j]); }
        for (int o = 0; o < n_out; ++o) {
            double se = fc_last_b[o], sp = fc_last_b[o], mag = std::fabs((double)fc_last_b[o]);
            for (int j = 0; j < 64; ++j) {
                se += (double)fc_last_w[(size_t)o * 64 + j] * t[j];
                mag += std::fabs((double)fc_last_w[(size_t)o * 64 + j] * t[j]);
                sp += lhi[(size_t)o * 64 + j] * th[j] + lhi[(size_t)o * 64 + j] * tl[j] + llo[(size_t)o * 64 + j] * th[j];
            }
            const double err = std::fabs(sp - se) / std::fmax(1.0, mag);       // relative to the un-cancelled magnitude of the output's sum
            if (!(err <= worst)) worst = err;          // NaN lands here too
        }
    }
    return worst;
}
// largest audited error a split-precision layer may show (conditioner outputs, relative above 1); a balanced layer sits at ~2e-7
constexpr double AUDIT_MAX_ERR = 4.0e-6;
static thread_local double g_last_audit = 0.0;     // per calling thread, like g_half_overflow
extern "C" double rnf_last_pack_audit(void) { return g_last_audit; }

extern "C" int rnf_pack_mobius(const float *fc_first_w, const float *fc_first_b, const float *l1_w, const float *l1_b,
                               const float *l3_w, const float *l3_b, const float *l5_w, const float *l5_b,
                               const float *fc_last_w, const float *fc_last_b, int32_t K, int32_t F, int32_t prec,
                               float *out, float *out_feat) {
    if (K <= 0) return fail("rnf_pack_mobius: segments=%d must be positive", K);
    if (prec != RNF_PREC_FP32 && prec != RNF_PREC_F16X2 && prec != RNF_PREC_BF16X3) return fail("rnf_pack_mobius: unknown precision %d", prec);
    g_half_overflow = false;
    g_last_audit = 0.0;
    if (F < 0 || F % 8) return fail("rnf_pack_mobius: feature_dim=%d must be a multiple of 8 (pad on the host)", F);
    const int ni = 3 + F;
    const float *hw[3] = {l1_w, l3_w, l5_w};
    const float *hb[3] = {l1_b, l3_b, l5_b};
    // split precision: the layer is first moved to the canonical point of its ReLU-rescaling orbit (equalize.h); exact fp32: packed as given
    ScaledMlp m;
    scale_mlp(fc_first_w, ni, 3, fc_first_b, hw, hb, fc_last_w, 4 * K, prec == RNF_PREC_F16X2, m);
    // fc_first: float2 per lane = (W0[o][h], h ? b0[o] : W0[o][2]); round 5: conditional layers too (the projection record has no bias)
    for (int ot = 0; ot < 2; ++ot)
        for (int lane = 0; lane < 64; ++lane) {
            const int o = 32 * ot + (lane & 31), h = lane >> 5;
            float *dst = out + MOB_FIRST + (ot * 64 + lane) * 2;
            dst[0] = m.W0[(size_t)o * ni + h];
            dst[1] = h ? m.b0[o] : m.W0[(size_t)o * ni + 2];
        }
    const float *shw[3] = {m.hw[0].data(), m.hw[1].data(), m.hw[2].data()};
    const float *shb[3] = {m.hb[0].data(), m.hb[1].data(), m.hb[2].data()};
    pack_hidden(out, shw, shb, prec);
    // fc_last: packed row P = 32*tau + 8g + 4h + c  <->  segment k = 8*tau + 2g + h, component c
    // (K % 8 != 0: the last tile is padded with zero rows for segments k >= K; the kernels give those segments weight 0)
    auto src_row = [&](int tau, int row) {
        const int g = row >> 3, h = (row >> 2) & 1, c = row & 3;
        const int k = 8 * tau + 2 * g + h;
        if (k >= K) return -1;
        return c == 0 ? k : K + 3 * k + (c - 1);
    };
    // the rows that produce the segment weights' pre-activations (reference rows 0 .. K-1) are packed times log2 e (layout.h S_PRESCALE)
    for (size_t i = 0; i < (size_t)K * 64; ++i) m.WL[i] *= S_PRESCALE;
    const bool b3 = prec == RNF_PREC_BF16X3;
    for (int tau = 0; tau < (K + 7) / 8; ++tau) {
        float *rec = out + (b3 ? Lay<2>::LAST + (size_t)tau * Lay<2>::LAST_TILE_FLOATS : MOB_LAST + (size_t)tau * MOB_LAST_TILE_FLOATS);
        auto row_of = [&](int, int i) {
            const int r = src_row(tau, i);
            return r < 0 ? (const float *)nullptr : m.WL.data() + (size_t)r * 64;
        };
        if (b3) pack_w64_b3(rec, 1, row_of); else if (prec) pack_w64_h(rec, 1, row_of); else pack_w64(rec, 1, row_of);
        pack_bias(rec + (b3 ? Lay<2>::LAST_TILE_BIAS : MOB_LAST_TILE_BIAS), 1, [&](int, int row) {
            const int r = src_row(tau, row);
            return r < 0 ? 0.f : (r < K ? fc_last_b[r] * S_PRESCALE : fc_last_b[r]);
        });
    }
    // bf16x3 flows project their features with the exact-fp32 projection kernels (the fp16 projection images are the ones that need the
    // equalisation's feature scale)
    if (F) { if (prec == RNF_PREC_F16X2) pack_featproj_h(out_feat, m.W0.data(), ni, 3, F, m.b0.data()); else pack_featproj(out_feat, m.W0.data(), ni, 3, F, m.b0.data()); }
    if (b3) return 0;                                   // nothing to audit: 24-bit operands, fp32's range
    if (prec && g_half_overflow) return fail("rnf_pack_mobius: a weight is outside the fp16 range; use RNF_PREC_FP32") + 1;
    if (prec) {
        for (size_t i = 0; i < (size_t)K * 64; ++i) m.WL[i] = std::ldexp(fc_last_w[i], -m.eq.e[0][i & 63]);      // the audit compares unscaled outputs
        g_last_audit = audit_mlp(m, fc_first_w, fc_first_b, hw, hb, fc_last_w, fc_last_b);
        if (g_audit && !(g_last_audit <= AUDIT_MAX_ERR))
            return fail("rnf_pack_mobius: the split-precision image of this layer is off by %.2e on the probe inputs (limit %.1e): its scales "
                        "are outside what fp16 pairs resolve; use RNF_PREC_FP32", g_last_audit, AUDIT_MAX_ERR) + 1;
    }
    return 0;
}

static int pack_cond(const float *fc_first_w, const float *fc_first_b, const float *l1_w, const float *l1_b, const float *l3_w,
                     const float *l3_b, const float *l5_w, const float *l5_b, const float *fc_last_w, const float *fc_last_b, int32_t F,
                     int32_t prec, int32_t n_out, float *out, float *out_feat);

extern "C" int rnf_pack_cond16(const float *fc_first_w, const float *fc_first_b, const float *l1_w, const float *l1_b,
                               const float *l3_w, const float *l3_b, const float *l5_w, const float *l5_b,
                               const float *fc_last_w, const float *fc_last_b, int32_t F, int32_t prec, float *out,
                               float *out_feat) {
    return pack_cond(fc_first_w, fc_first_b, l1_w, l1_b, l3_w, l3_b, l5_w, l5_b, fc_last_w, fc_last_b, F, prec, 16, out, out_feat);
}

// Condition9Trans / Condition9RotL / Condition9RotR / Condition9RotRSmith (flow/squeezetrans.py:234-247, flow/rottrans.py:108-181): the
// same record with a 9-row fc_last; output i sits where output i of Condition16Trans sits, so the kernels gather both the same way.
extern "C" int rnf_pack_cond9(const float *fc_first_w, const float *fc_first_b, const float *l1_w, const float *l1_b,
                              const float *l3_w, const float *l3_b, const float *l5_w, const float *l5_b,
                              const float *fc_last_w, const float *fc_last_b, int32_t F, int32_t prec, float *out,
                              float *out_feat) {
    return pack_cond(fc_first_w, fc_first_b, l1_w, l1_b, l3_w, l3_b, l5_w, l5_b, fc_last_w, fc_last_b, F, prec, 9, out, out_feat);
}

// Condition36Trans (flow/squeezetrans.py:334-347): two fc_last tiles, packed row P of tile 0 = output P, rows 0..3 of tile 1 = outputs 32..35
extern "C" int rnf_pack_cond36(const float *fc_first_w, const float *fc_first_b, const float *l1_w, const float *l1_b,
                               const float *l3_w, const float *l3_b, const float *l5_w, const float *l5_b,
                               const float *fc_last_w, const float *fc_last_b, int32_t F, int32_t prec, float *out,
                               float *out_feat) {
    return pack_cond(fc_first_w, fc_first_b, l1_w, l1_b, l3_w, l3_b, l5_w, l5_b, fc_last_w, fc_last_b, F, prec, 36, out, out_feat);
}
extern "C" int64_t rnf_cond36_packed_floats(void) { return COND36_FLOATS; }

static int pack_cond(const float *fc_first_w, const float *fc_first_b, const float *l1_w, const float *l1_b, const float *l3_w,
                     const float *l3_b, const float *l5_w, const float *l5_b, const float *fc_last_w, const float *fc_last_b, int32_t F,
                     int32_t prec, int32_t n_out, float *out, float *out_feat) {
    if (F <= 0 || F % 8) return fail("rnf_pack_cond16: feature_dim=%d must be a positive multiple of 8", F);
    if (prec != RNF_PREC_FP32 && prec != RNF_PREC_F16X2 && prec != RNF_PREC_BF16X3) return fail("rnf_pack_cond16: unknown precision %d", prec);
    g_half_overflow = false;
    g_last_audit = 0.0;
    std::memset(out, 0, sizeof(float) * cond_packed_floats_p(n_out == 36 ? 2 : 1, prec));   // zero fc_first image: x0 comes from the projection
    const float *hw[3] = {l1_w, l3_w, l5_w};
    const float *hb[3] = {l1_b, l3_b, l5_b};
    ScaledMlp m;                                       // equalize.h: canonical scaling before the fp16 split (split precision only)
    scale_mlp(fc_first_w, F, 0, fc_first_b, hw, hb, fc_last_w, n_out, prec == RNF_PREC_F16X2, m);
    const float *shw[3] = {m.hw[0].data(), m.hw[1].data(), m.hw[2].data()};
    const float *shb[3] = {m.hb[0].data(), m.hb[1].data(), m.hb[2].data()};
    // fc_first image: no rotation inputs (zero weights); the bias slot (element 1 of lane-half 1, multiplied by the constant 1 of the second
    // matrix step) carries b0 -- the projection G = W0 f has no bias of its own (round 5)
    for (int ot = 0; ot < 2; ++ot)
        for (int lane = 32; lane < 64; ++lane) out[MOB_FIRST + (ot * 64 + lane) * 2 + 1] = m.b0[32 * ot + (lane & 31)];
    pack_hidden(out, shw, shb, prec);
    // one fc_last tile: packed row 8g + 4h + c (g = 0,1) <-> output 4*(2g+h) + c (= M[2g + h][c] of the 4x4); rows >= 16 and outputs
    // >= n_out are zero
    auto src_row = [&](int row) {
        if (row >= 16) return -1;
        const int g = row >> 3, h = (row >> 2) & 1, c = row & 3;
        const int o = 4 * (2 * g + h) + c;
        return o < n_out ? o : -1;
    };
    for (int tau = 0; tau < (n_out == 36 ? 2 : 1); ++tau) {
        auto src = [&](int row) { return n_out == 36 ? (tau == 0 ? row : (row < 4 ? 32 + row : -1)) : src_row(row); };
        const bool b3 = prec == RNF_PREC_BF16X3;
        float *rec = out + (b3 ? Lay<2>::LAST + (size_t)tau * Lay<2>::LAST_TILE_FLOATS : MOB_LAST + (size_t)tau * MOB_LAST_TILE_FLOATS);
        auto row_of = [&](int, int i) { int s = src(i); return s < 0 ? (const float *)nullptr : m.WL.data() + (size_t)s * 64; };
        if (b3) pack_w64_b3(rec, 1, row_of); else if (prec) pack_w64_h(rec, 1, row_of); else pack_w64(rec, 1, row_of);
        pack_bias(rec + (b3 ? Lay<2>::LAST_TILE_BIAS : MOB_LAST_TILE_BIAS), 1, [&](int, int row) { int s = src(row); return s < 0 ? 0.f : fc_last_b[s]; });
    }
    if (prec == RNF_PREC_F16X2) pack_featproj_h(out_feat, m.W0.data(), F, 0, F, m.b0.data()); else pack_featproj(out_feat, m.W0.data(), F, 0, F, m.b0.data());
    if (prec == RNF_PREC_BF16X3) return 0;
    if (prec && g_half_overflow) return fail("rnf_pack_cond16: a weight is outside the fp16 range; use RNF_PREC_FP32") + 1;
    if (prec) {
        g_last_audit = audit_mlp(m, fc_first_w, fc_first_b, hw, hb, fc_last_w, fc_last_b);
        if (g_audit && !(g_last_audit <= AUDIT_MAX_ERR))
            return fail("rnf_pack_cond16: the split-precision image of this layer is off by %.2e on the probe inputs (limit %.1e): its scales "
                        "are outside what fp16 pairs resolve; use RNF_PREC_FP32", g_last_audit, AUDIT_MAX_ERR) + 1;
    }
    return 0;
}

// 4x4 inverse / determinant in double (Gauss-Jordan with partial pivoting)
static bool inv4_double(const double *m, double *inv, double *det) {
    double a[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) { a[i][j] = m[4 * i + j]; a[i][4 + j] = (i == j); }
    double d = 1.0;
    for (int c = 0; c < 4; ++c) {
        int p = c;
        for (int r = c + 1; r < 4; ++r) if (std::fabs(a[r][c]) > std::fabs(a[p][c])) p = r;
        if (a[p][c] == 0.0) return false;
        if (p != c) { for (int j = 0; j < 8; ++j) std::swap(a[p][j], a[c][j]); d = -d; }
        d *= a[c][c];
        double ip = 1.0 / a[c][c];
        for (int j = 0; j < 8; ++j) a[c][j] *= ip;
        for (int r = 0; r < 4; ++r) if (r != c) {
            double f = a[r][c];
            for (int j = 0; j < 8; ++j) a[r][j] -= f * a[c][j];
        }
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) inv[4 * i + j] = a[i][4 + j];
    *det = d;
    return true;
}

extern "C" int rnf_pack_affine16(const float *mat16, float *out) {
    double m[16], inv[16], det;
    for (int i = 0; i < 16; ++i) m[i] = mat16[i];
    if (!inv4_double(m, inv, &det)) return fail("rnf_pack_affine16: singular 4x4 matrix");
    for (int i = 0; i < 16; ++i) { out[i] = mat16[i]; out[17 + i] = (float)inv[i]; }
    out[16] = (float)std::log(std::fabs(det));
    out[33] = (float)(-std::log(std::fabs(det)));
    out[34] = out[35] = 0.f;
    affine16_table(m, out[16], false, out + AFF_TABLE_FWD);
    affine16_table(inv, out[33], false, out + AFF_TABLE_INV);
    return 0;
}

// n x n inverse in double (Gauss-Jordan with partial pivoting), n <= 6
static bool invn_double(const double *m, int n, double *inv) {
    double a[6][12];
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) { a[i][j] = m[n * i + j]; a[i][n + j] = (i == j); }
    for (int c = 0; c < n; ++c) {
        int p = c;
        for (int r = c + 1; r < n; ++r) if (std::fabs(a[r][c]) > std::fabs(a[p][c])) p = r;
        if (a[p][c] == 0.0) return false;
        if (p != c) for (int j = 0; j < 2 * n; ++j) std::swap(a[p][j], a[c][j]);
        const double ip = 1.0 / a[c][c];
        for (int j = 0; j < 2 * n; ++j) a[c][j] *= ip;
        for (int r = 0; r < n; ++r) if (r != c) {
            const double f = a[r][c];
            for (int j = 0; j < 2 * n; ++j) a[r][j] -= f * a[c][j];
        }
    }
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) inv[n * i + j] = a[i][n + j];
    return true;
}

// Uncondition9Trans / Uncondition36Trans (flow/squeezetrans.py:250-261, 350-361): [M | M^-1], n = 3 or 6
extern "C" int rnf_pack_gs(const float *mat, int32_t n, float *out) {
    if (n != 3 && n != 6) return fail("rnf_pack_gs: n=%d must be 3 or 6", n);
    if (!mat || !out) return fail("rnf_pack_gs: null pointer");
    double m[36], inv[36];
    for (int i = 0; i < n * n; ++i) m[i] = mat[i];
    if (!invn_double(m, n, inv)) return fail("rnf_pack_gs: singular %dx%d matrix", n, n);
    const int total = n == 3 ? GS9_FLOATS : GS36_FLOATS;
    for (int i = 0; i < total; ++i) out[i] = 0.f;
    for (int i = 0; i < n * n; ++i) { out[i] = mat[i]; out[n * n + i] = (float)inv[i]; }
    return 0;
}

extern "C" int64_t rnf_gs_packed_floats(int32_t n) { return n == 3 ? GS9_FLOATS : (n == 6 ? GS36_FLOATS : -1); }

extern "C" int rnf_pack_rot16(const float *mat16, float *out) {
    // orthogonality check in double: M M^T = I within fp32 rounding of a product of SVD factors
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double d = 0.0;
            for (int k = 0; k < 4; ++k) d += (double)mat16[4 * i + k] * mat16[4 * j + k];
            if (std::fabs(d - (i == j)) > 1e-4) return fail("rnf_pack_rot16: matrix is not orthogonal (|M M^T - I| = %g)", std::fabs(d - (i == j)));
        }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) { out[4 * i + j] = mat16[4 * i + j]; out[17 + 4 * i + j] = mat16[4 * j + i]; }
    out[16] = 0.f;
    out[33] = 0.f;
    out[34] = 1.f;
    out[35] = 0.f;
    double m[16], mt[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) { m[4 * i + j] = mat16[4 * i + j]; mt[4 * i + j] = mat16[4 * j + i]; }
    affine16_table(m, 0.f, true, out + AFF_TABLE_FWD);
    affine16_table(mt, 0.f, true, out + AFF_TABLE_INV);
    return 0;
}

// ------------------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------------------
constexpr int NW = 8;                                   // waves per workgroup (2 per SIMD): exact-fp32 and inverse kernels, featproj
constexpr int NW_FWD_H = 8;                             // forward split-precision kernel small launches
constexpr int NW_FWD_NARROW = 4;                        // same kernel, 1 wave per SIMD: launches that leave half of the CUs empty at 8 waves (small
                                                        // batches, training: forward latency 0.41 -> 0.34 ms at 1024 rotations)
constexpr int NW_FWD_WIDE = 16;                         // same kernel, 4 waves per SIMD (fits in 128 VGPRs): launches that fill every CU with
                                                        // 512-rotation workgroups; +12 % over 8 waves (profiles/r1/nw_sweep.txt)
#ifndef RNF_NW_FP
#define RNF_NW_FP 8
#endif
constexpr int NW_INV_BIG = 4;                           // inverse with 64 < K <= 128: one wave per SIMD
constexpr int NW_FP = RNF_NW_FP;                        // waves per workgroup of the feature projection (workgroups per CU: 8 / NW_FP)
#ifndef RNF_CHUNK_LOG2
#define RNF_CHUNK_LOG2 18           // (17 / 16 measured and not better: profiles/r6/ab_chunk_C4.jsonl)
#endif
constexpr long long CHUNK_SAMPLES = 1LL << RNF_CHUNK_LOG2;   // samples per launch when a feature projection scratch is needed
// head of the workspace: [0, 2048) block partials of the primary launch, [2048, 4095) partials of the exact-fp32 re-run, double 4095 =
// two int32: {guard of the current chunk, sticky "a re-run happened in this call"} (flow_kernels.h FlowArgs::guard)
constexpr size_t PARTIALS_BYTES = 4096 * sizeof(double);
constexpr int PARTIALS_FB_AT = 2048, GUARD_AT = 4095;

// Compute units of the CALLING THREAD'S CURRENT DEVICE (grids of the persistent kernels are sized by it).  Cached per device ordinal: one
// process may drive several GPUs (nn.DataParallel, agent.py:22; round 5 cached the first device's count for the whole process -- VERDICT r5 #7).
static int device_cus() {
    constexpr int MAX_DEV = 64;
    static std::atomic<int> cus[MAX_DEV];                  // zero-initialised
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return 256;
    int c = cus[dev].load(std::memory_order_relaxed);
    if (c == 0) {
        if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = 256;
        cus[dev].store(c, std::memory_order_relaxed);
    }
    return c;
}

extern "C" size_t rnf_workspace_bytes_shared(int64_t n, int32_t n_cond_layers, int64_t feature_div) {
    if (feature_div <= 0) return rnf_workspace_bytes(n, n_cond_layers);
    const size_t rows = (size_t)((n + feature_div - 1) / feature_div);
    return PARTIALS_BYTES + (size_t)n_cond_layers * rows * 64 * sizeof(float);
}

// K > 128: the inverse pass keeps the parameters of 64 segments per lane in registers and the rest in a per-wave stash behind the scratch
static size_t inv_stash_bytes(int32_t K) {
    const int KT = (K + 7) / 8;
    return KT > 16 ? (size_t)device_cus() * NW_INV_BIG * (size_t)(4 * (KT - 16)) * 64 * sizeof(float4) : 0;
}
extern "C" size_t rnf_workspace_bytes_segments(int64_t n, int32_t n_cond_layers, int32_t segments) {
    return rnf_workspace_bytes(n, n_cond_layers) + inv_stash_bytes(segments);
}

extern "C" size_t rnf_workspace_bytes(int64_t n, int32_t n_cond_layers) {
    size_t bytes = PARTIALS_BYTES;
    if (n_cond_layers > 0) {
        long long chunk = n < CHUNK_SAMPLES ? n : CHUNK_SAMPLES;
        long long groups = (chunk + 255) / 256 * 8;     // whole workgroup tiles, for either workgroup size
        const long long g16 = (chunk + 32 * NW_FWD_WIDE - 1) / (32 * NW_FWD_WIDE) * NW_FWD_WIDE;
        if (g16 > groups) groups = g16;
        bytes += (size_t)n_cond_layers * groups * G_FLOATS_PER_GROUP * sizeof(float);
    }
    return bytes;
}

template <typename K>
static hipError_t allow_lds(K kernel, size_t bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// RNF_WIDE=0 keeps the forward kernel at 8 waves per workgroup (measurement switch)
static bool wide_allowed() {
    static int mode = -1;
    if (mode < 0) {
        const char *e = std::getenv("RNF_WIDE");
        mode = (e && e[0] == '0') ? 0 : 1;
    }
    return mode == 1;
}

// staging mode: DMA (LDS-DMA double-phase prefetch, needs segments <= 64) unless RNF_STAGING=sync
static bool staging_dma() {
    static int mode = -1;
    if (mode < 0) {
        const char *e = std::getenv("RNF_STAGING");
        mode = (e && std::strcmp(e, "sync") == 0) ? 0 : 1;
    }
    return mode == 1;
}

// RNF_GUARD=0: no range guard / fp32 re-run behind split-precision calls (measurement switch)
static bool guard_allowed() {
#ifdef RNF_NO_GUARD
    return false;
#endif
    static int mode = -1;
    if (mode < 0) {
        const char *e = std::getenv("RNF_GUARD");
        mode = (e && e[0] == '0') ? 0 : 1;
    }
    return mode == 1;
}

// RNF_LEAN=0 keeps unconditional Moebius / affine stacks on the general instantiation (measurement switch)
static bool lean_allowed() {
#ifdef RNF_NO_LEAN
    return false;
#endif
    static int mode = -1;
    if (mode < 0) {
        const char *e = std::getenv("RNF_LEAN");
        mode = (e && e[0] == '0') ? 0 : 1;
    }
    return mode == 1;
}

// RNF_FUSED=1 / rnf_set_fused(1): conditional forward passes (every MLP layer conditional, F <= 256) run the FUSED instantiation of the
// stack kernel -- feature projection inside, no scratch round trip (HBM traffic = the algorithmic bytes) -- instead of the pre-pass + stack
// pair.  OFF by default: measured on C4 it is SLOWER (12.2 ms against 8.7 ms, profiles/r3/fused_c4.md): the features take 128 of the 256
// registers an 8-wave workgroup has per lane, the rest of the layer does not fit beside them, and its projection phases run in lockstep
// between workgroup barriers with nothing to overlap.
static int g_fused = -1;
static bool fused_allowed() {
    if (g_fused < 0) {
        const char *e = std::getenv("RNF_FUSED");
        g_fused = (e && e[0] == '1') ? 1 : 0;
    }
    return g_fused == 1;
}
extern "C" int rnf_set_fused(int on) {
    const int old = fused_allowed() ? 1 : 0;
    g_fused = on ? 1 : 0;
    return old;
}

// Block size of the training backward sweep: 0 = by batch size (default), 16 / 64 = force (RNF_TRAIN_BLOCK in the environment).
constexpr int64_t kTrainBlock64From = 6144;     // batches from this size on use the 64-rotation kernel (measured crossover: profiles/README.md)
static int g_train_block = -1;
static int train_block() {
    if (g_train_block < 0) {
        const char *e = std::getenv("RNF_TRAIN_BLOCK");
        const int v = e ? std::atoi(e) : 0;
        g_train_block = (v == 16 || v == 64) ? v : 0;
    }
    return g_train_block;
}
// Layers per launch of the device packer and of the backward sweep (their layer tables are kernel arguments): the table capacity, or fewer
// with RNF_LAYER_CHUNK=<n> in the environment -- a test switch: a chunked sweep must reproduce the single launch (tests/test_gpu_grad.py).
static int layer_chunk(int capacity) {
    const char *e = std::getenv("RNF_LAYER_CHUNK");
    const int v = e ? std::atoi(e) : 0;
    return (v >= 1 && v < capacity) ? v : capacity;
}
extern "C" int rnf_set_train_block(int rotations) {
    const int prev = train_block();
    g_train_block = (rotations == 16 || rotations == 64) ? rotations : 0;
    return prev;
}

// forward pass of a conditional flow with the feature projection inside the stack kernel (flow_kernels.h FUSED)
#ifndef RNF_NW_FUSED
#define RNF_NW_FUSED 8
#endif
constexpr int NW_FUSED = RNF_NW_FUSED;
static int launch_fused(const FlowArgs &a, int grid, size_t lds_bytes, hipStream_t stream) {
    auto kern = flow_stack_kernel<0, 0, NW_FUSED, true, 1, false, 2, true>;
    HIP_TRY(allow_lds(kern, lds_bytes));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW_FUSED * 64), lds_bytes, stream, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

// `lean` picks the kernel FAMILY -- 1: Moebius + constant-affine layers only (BASELINE C1 / C2 / C3), 2: the conditional counterpart (Moebius +
// constant-affine + Condition16Trans, every MLP conditional: C4), 0: the general kernel -- from the flow's structure and from whether the call
// runs guarded, NEVER from the batch size: the families differ in arithmetic (one-piece softplus of the lean kernels, so3_math.h), the
// workgroup widths of one family do not, so a rotation's result does not depend on the size of the launch (or chunk, or shard) it travels in.
template <int DIR, int KT_INV, bool PIPE, int PREC, bool EXT = false, bool ROWS = false>
static int launch_stack(const FlowArgs &a, int grid, size_t lds_bytes, hipStream_t stream, int nwk, int lean = 0) {
#define RNF_STACK_GO(NW_, LEAN_)                                                                                \
    do {                                                                                                        \
        auto kern = flow_stack_kernel<DIR, KT_INV, NW_, PIPE, PREC, EXT, LEAN_, false, ROWS>;                   \
        HIP_TRY(allow_lds(kern, lds_bytes));                                                                    \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NW_ * 64), lds_bytes, stream, a);                             \
        HIP_TRY(hipGetLastError());                                                                             \
        return 0;                                                                                               \
    } while (0)
    if constexpr (ROWS && DIR == 0) {                   // shared feature rows, forward: the conditional-lean family only (run_flow checks)
        if (lean != 2) return fail("internal: shared-row forward launches exist for the conditional-lean family only");
        if (nwk == NW_FWD_WIDE) RNF_STACK_GO(NW_FWD_WIDE, 2);
        if (nwk == NW_FWD_H) RNF_STACK_GO(NW_FWD_H, 2);
        if (nwk == NW_FWD_NARROW) RNF_STACK_GO(NW_FWD_NARROW, 2);
        return fail("internal: no %d-wave instantiation of the shared-row stack kernel", nwk);
    } else if constexpr (DIR == 0 && PREC == 1 && PIPE && !EXT) {
        if (lean == 2) {
            if (nwk == NW_FWD_WIDE) RNF_STACK_GO(NW_FWD_WIDE, 2);
            if (nwk == NW_FWD_H) RNF_STACK_GO(NW_FWD_H, 2);
            if (nwk == NW_FWD_NARROW) RNF_STACK_GO(NW_FWD_NARROW, 2);
        } else if (lean == 1) {
            if (nwk == NW_FWD_WIDE) RNF_STACK_GO(NW_FWD_WIDE, 1);
            if (nwk == NW_FWD_H) RNF_STACK_GO(NW_FWD_H, 1);
            if (nwk == NW_FWD_NARROW) RNF_STACK_GO(NW_FWD_NARROW, 1);
        } else {
            if (nwk == NW_FWD_WIDE) RNF_STACK_GO(NW_FWD_WIDE, 0);
            if (nwk == NW_FWD_NARROW) RNF_STACK_GO(NW_FWD_NARROW, 0);
        }
    }
    if constexpr (DIR == 1 && PREC == 1 && PIPE && !EXT && KT_INV <= 8) {
        // small inverse launches (round 5): 4-wave workgroups, one wave per SIMD, twice as many workgroups -- a launch that leaves half of
        // the CUs empty at 8 waves is a latency chain per wave (2^15 rotations: 0.73 -> 0.57 ms); same arithmetic, bit-equal rows
        if (nwk == NW_FWD_NARROW && !lean) RNF_STACK_GO(NW_FWD_NARROW, 0);
    }
    if constexpr (!(ROWS && DIR == 0)) {
        constexpr int NWK = (DIR == 0 && PREC == 1) ? NW_FWD_H : NW;
        if (nwk != NWK || lean) return fail("internal: no %d-wave instantiation of this stack kernel", nwk);
        RNF_STACK_GO(NWK, 0);
    }
#undef RNF_STACK_GO
}

// inverse with 64 < K <= 128 (flow_kernels.h mobius_inv_tiles: 16 tiles of segment parameters in registers, synchronous staging)
template <int PREC, bool EXT = false>
static int launch_big_inverse(const FlowArgs &a, int grid, size_t lds_bytes, hipStream_t stream) {
    auto kern = flow_stack_kernel<1, 16, NW_INV_BIG, false, PREC, EXT>;
    HIP_TRY(allow_lds(kern, lds_bytes));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW_INV_BIG * 64), lds_bytes, stream, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

struct RunOpts {
    int dir;                  // 0 forward, 1 inverse
    const float *fisher_A, *fisher_c;
    int64_t fisher_B;
    float *logp_out;
    double *sum_out;
    float *states = nullptr;  // training forward: per-layer input rotations [n_layers][n][9]
    int64_t feature_div = 0;  // > 0: feature row r serves rotations [r * feature_div, (r + 1) * feature_div)
    const float *side = nullptr;   // per-sample matrices of RNF_LAYER_SIDE* layers: [side slot][n][16]
};

static int run_flow(const float *rot, const float *feat, int64_t n, int32_t F, const float *blob, const int32_t *desc,
                    int32_t n_layers, int32_t K, float *rot_out, float *ldj_out, void *ws, size_t ws_bytes, void *stream_v,
                    const RunOpts &o) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    if (n < 0) return fail("n=%lld is negative", (long long)n);
    if (n_layers <= 0 || n_layers > MAX_LAYERS) return fail("n_layers=%d outside [1,%d]", n_layers, MAX_LAYERS);
    if (K <= 0) return fail("segments=%d must be positive", K);
    if (!blob || !desc || (n > 0 && !rot)) return fail("null rotation / blob / desc pointer");
    if (o.fisher_A && (o.fisher_B <= 0 || n % o.fisher_B)) return fail("n=%lld not divisible by fisher rows B=%lld (utils/fisher.py:226)", (long long)n, (long long)o.fisher_B);

    FlowArgs a;
    std::memset(&a, 0, sizeof(a));
    FeatProjArgs fp;
    std::memset(&fp, 0, sizeof(fp));
    // exact-fp32 images of the same layers in the same blob (desc columns D_PARAM_FB / D_FEAT_FB): a split-precision call is guarded and
    // re-run on them, on the device, when a sample comes out non-finite (an fp16 operand overflowed)
    bool have_fb = !o.states;
    std::vector<int> fbp(n_layers, -1), fbf(MAX_SLOTS, -1), fp_primary(MAX_SLOTS, -1);
    int n_slots = 0;
    int min_tiles = 1;                       // fc_last tiles the largest non-Moebius record needs resident in LDS
    bool any_mlp = false, ext = false;       // ext: the flow contains a layer kind only the extended kernel instantiation carries
    bool all_mlp_cond = true;                // every MLP layer consumes the feature vector (what the FUSED instantiation handles)
    int prec = -1, fb_prec = -1, rf_code = 0;
    bool lean = lean_allowed() && !o.states;   // Moebius + constant 4x4 affine layers only, nothing conditional, no saved states
    bool lean2 = lean_allowed() && !o.states;  // the conditional counterpart: + Condition16Trans, every MLP layer conditional (checked below)
    for (int l = 0; l < n_layers; ++l) {
        const int32_t *d = desc + (size_t)l * D_STRIDE;
        const int kind = d[D_KIND], perm = d[D_PERM], slot = d[D_SLOT];
        if (kind < RNF_KIND_MOBIUS || kind > RNF_KIND_LAST) return fail("layer %d: unknown kind %d", l, kind);
        if (perm < 0 || perm > 5) return fail("layer %d: perm_row %d outside [0,5]", l, perm);
        if (d[D_PARAM] < 0 || (d[D_PARAM] % 4 && !kind_is_side(kind))) return fail("layer %d: param offset %d must be a non-negative multiple of 4", l, d[D_PARAM]);
        if ((kind == RNF_KIND_COND16 || kind_is_cond9(kind) || kind == RNF_KIND_COND36) && slot < 0)
            return fail("layer %d: a conditional affine layer needs a cond_slot", l);
        if (kind_is_cond9(kind) || kind == RNF_KIND_COND36 || kind_is_side(kind)) ext = true;
        if (kind_is_side(kind) && !o.side) return fail("layer %d takes per-sample matrices: call the rnf_flow_*_side entry points with a side buffer", l);
        if ((kind != RNF_KIND_MOBIUS && kind != RNF_KIND_AFFINE16) || slot >= 0) lean = false;
        if (kind != RNF_KIND_MOBIUS && kind != RNF_KIND_AFFINE16 && kind != RNF_KIND_COND16) lean2 = false;
        if (kind == RNF_KIND_COND36) min_tiles = 2;
        if (slot >= 0) {
            if (slot >= MAX_SLOTS) return fail("layer %d: cond_slot %d >= %d", l, slot, MAX_SLOTS);
            if (d[D_FEAT] < 0 || d[D_FEAT] % 4) return fail("layer %d: feat offset %d invalid", l, d[D_FEAT]);
            fp.feat_off[slot] = d[D_FEAT];
            fp_primary[slot] = d[D_FEAT];
            if (slot + 1 > n_slots) n_slots = slot + 1;
        }
        if (kind_has_mlp(kind)) {
            any_mlp = true;
            if (slot < 0) all_mlp_cond = false;
            if (d[D_PARAM_FB] < 0 || d[D_PARAM_FB] % 4 || (slot >= 0 && (d[D_FEAT_FB] < 0 || d[D_FEAT_FB] % 4))) have_fb = false;
            fbp[l] = d[D_PARAM_FB];
            if (slot >= 0) fbf[slot] = d[D_FEAT_FB];
            const int p = d[D_PREC] & 255, pfb = (d[D_PREC] >> 8) & 255;      // bits 8..15: arithmetic of the fallback records (0: fp32, 2: bf16x3)
            if (kind == RNF_KIND_MOBIUS) {                                    // bits 16..17: first-pass order of the inverse root finder
                const int rc = (d[D_PREC] >> 16) & 3;
                if (rc == 3) return fail("layer %d: root-finder order code 3 is reserved", l);
                if (rc > rf_code) rf_code = rc;
            }
            if (p != RNF_PREC_FP32 && p != RNF_PREC_F16X2 && p != RNF_PREC_BF16X3) return fail("layer %d: unknown precision %d", l, p);
            if (pfb != RNF_PREC_FP32 && pfb != RNF_PREC_BF16X3) return fail("layer %d: fallback records must be RNF_PREC_FP32 or RNF_PREC_BF16X3, got %d", l, pfb);
            if (fb_prec >= 0 && pfb != fb_prec) return fail("layer %d: all fallback records of a flow must share one precision", l);
            fb_prec = pfb;
            if (prec >= 0 && p != prec) return fail("layer %d: all MLP layers of a flow must be packed with the same precision", l);
            prec = p;
        }
        a.layers[l] = make_int2(kind | (perm << 4) | ((slot + 1) << 8), d[D_PARAM]);
    }
    if (prec < 0) prec = 0;
    if (fb_prec < 0) fb_prec = 0;
    {   // bits 16..25 of x: iteration position + 1 of the next layer with an MLP image behind this one (0 = none), in the order the
        // pass walks the layers -- saves the kernel a dependent chain of scalar loads per layer
        int nxt = 0;
        for (int pos = n_layers - 1; pos >= 0; --pos) {
            const int l = o.dir ? n_layers - 1 - pos : pos;
            a.layers[l].x |= nxt << 16;
            if (kind_has_mlp(a.layers[l].x & 15)) nxt = pos + 1;
        }
    }
    if (n_slots > 0) {
        if (!feat) return fail("this flow consumes a feature vector but feature pointer is null (flow/mobiusflow.py:48-49)");
        if (F <= 0 || F % 8) return fail("feature_dim=%d must be a positive multiple of 8 (pad on the host)", F);
    }
    if (n == 0) {
        if (o.sum_out) { hipLaunchKernelGGL(nll_finalize_kernel, dim3(1), dim3(256), 0, stream, (const double *)nullptr, 0, 0.0, o.sum_out, 0); }
        return 0;
    }
    const bool shared = o.feature_div > 0 && n_slots > 0;
    const bool ext_layers = ext;                         // the flow holds a layer kind only the extended instantiation carries
    if (shared) ext = true;                              // shared feature rows: the extended instantiation, unless a ROWS one fits (below)
    if (shared && n % o.feature_div) return fail("n=%lld not divisible by feature_div=%lld", (long long)n, (long long)o.feature_div);
    const size_t ws_need = shared ? rnf_workspace_bytes_shared(n, n_slots, o.feature_div) : rnf_workspace_bytes(n, n_slots);
    if (ws_bytes < ws_need && (n_slots > 0 || o.sum_out)) return fail("workspace of %zu bytes is smaller than the %zu needed", ws_bytes, ws_need);
    if ((n_slots > 0 || o.sum_out) && !ws) return fail("workspace pointer is null");

    const int KT = (K + 7) / 8;
    // inverse: the segment parameters of a layer stay in registers through the root finder; instantiations hold 1, 2, 4, 8 tiles
    // (8-wave workgroups) or 16 (K <= 128: 4-wave workgroups with the whole register file, fc_last staged in two halves)
    const int kt_inv = KT <= 1 ? 1 : (KT <= 2 ? 2 : (KT <= 4 ? 4 : (KT <= 8 ? 8 : 16)));
    // (round 4: inverse passes with more than 64 segments also run for conditional 3x3 / 6x6 layers, side layers and shared feature rows --
    // flow/mobiusflow.py:7-14 takes any `segments` with any `rot` -- on the extended build of the 4-wave instantiation)

    if (o.dir == 1 && any_mlp && KT > 16) {            // overflow stash of the K > 128 inverse (flow_kernels.h mobius_inv_tiles), behind everything else
        if (shared) return fail("inverse pass with segments > 128 is not built for shared feature rows; got %d", K);
        if (!ws || ws_bytes < ws_need + inv_stash_bytes(K))
            return fail("workspace of %zu bytes is smaller than the %zu an inverse pass with %d segments needs (rnf_workspace_bytes_segments)",
                        ws_bytes, ws_need + inv_stash_bytes(K), K);
        a.inv_stash = reinterpret_cast<float4 *>(reinterpret_cast<char *>(ws) + ws_need);
    }
    double *partials = reinterpret_cast<double *>(ws);
    float *G = n_slots ? reinterpret_cast<float *>(reinterpret_cast<char *>(ws) + PARTIALS_BYTES) : nullptr;
    const int max_tiles = prec == 2 ? Lay<2>::MAX_TILES_IN_LDS : MOB_MAX_TILES_IN_LDS;
    int tiles_in_lds = any_mlp ? (KT < max_tiles ? KT : max_tiles) : 0;
    if (any_mlp && tiles_in_lds < min_tiles) tiles_in_lds = min_tiles;
    size_t lds_bytes = !any_mlp ? 0 : sizeof(float) * (prec == 2 ? (size_t)3 * Mlp<2>::REGION_FLOATS      // three ring regions (flow_kernels.h RING)
                                                                  : MOB_HEAD_FLOATS + (size_t)tiles_in_lds * MOB_LAST_TILE_FLOATS);
    if (lds_bytes < NW_FWD_WIDE * sizeof(double) * 2) lds_bytes = NW_FWD_WIDE * sizeof(double) * 2;
    {   // SIMD fairness governor (flow_kernels.h struct Fair): forward split-precision kernel; RNF_FAIR=0 switches it off
        static int fair = -1;
        if (fair < 0) { const char *e = std::getenv("RNF_FAIR"); fair = (e && e[0] == '0') ? 0 : 1; }
        a.fair_off = -1;
        if (fair && any_mlp && o.dir == 0 && prec == 1) {
            a.fair_off = (int)(lds_bytes / sizeof(float));
            lds_bytes += 64;
        }
    }
    const int cus = device_cus();
    const long long chunk_cap = (n_slots && !shared) ? CHUNK_SAMPLES : n;      // shared feature rows: the projection scratch is tiny
    const long long feat_rows = shared ? n / o.feature_div : 0;
    a.g_div = shared ? o.feature_div : 0;
    a.g_rows = feat_rows;

    a.blob = blob;
#ifdef RNF_STAMPS
    {   // diagnostic build: RNF_STAMPS_PTR=<device address of 8 zeroed uint64> (set by tools/phase_stamps.py)
        const char *sp = std::getenv("RNF_STAMPS_PTR");
        a.stamps = sp ? reinterpret_cast<unsigned long long *>(std::strtoull(sp, nullptr, 0)) : nullptr;
    }
#endif
    a.n_layers = n_layers;
    a.KT = KT;
    a.K = K;
    a.min_wsum = kMinWeightSum * (float)K;
    {   // Root finder of the inverse pass: order of its FIRST pass (flow_kernels.h mobius_inv_finish).  Third order (Halley) unless the flow
        // asks for the fourth-order first pass -- bits 16..17 of desc column 5 on its Moebius layers (1: third, 2: fourth; 0: this default).
        // The fourth order pays on sharply peaked conditioner outputs (a trained p(R | image): the third-order iteration then needs a third
        // pass for most waves; trained_c4 10.02 -> 9.66 ms) and costs mild weights its four extra instructions per segment pair (BASELINE's
        // synthetic C5q +4 %, C5 +0.9 %: profiles/r6/ab_centre.jsonl), so it is a property of the FLOW its owner sets (Flow.set_rootfinder_order,
        // the checkpoint sidecar) -- one value per flow, never per launch.  RNF_RF_FIRST=3|4 forces one.
        static int forced = -1;
        if (forced < 0) { const char *e = std::getenv("RNF_RF_FIRST"); forced = e ? (e[0] == '4' ? 4 : (e[0] == '3' ? 3 : 0)) : 0; }
        a.rf_first4 = forced ? (forced == 4) : (rf_code == 2);
    }
    a.side = o.side;
    a.side_n = n;
    a.states = o.states;
    a.states_n = n;
    a.fisher_A = o.fisher_A;
    a.fisher_c = o.fisher_c;
    a.fisher_div = o.fisher_A ? n / o.fisher_B : 1;

    const bool pipe = staging_dma() && KT <= MOB_MAX_TILES_IN_LDS && prec != 2;     // bf16x3: a K = 64 layer image is 171 KiB (layout.h Lay<2>): synchronous staging
    a.tab_off = -1;
    if ((pipe || prec == 2) && any_mlp) {                      // two LDS buffers for the blocks of constant-affine layers (flow_kernels.h stage_table)
        lds_bytes = (lds_bytes + 15) / 16 * 16;
        a.tab_off = (int)(lds_bytes / sizeof(float));
        lds_bytes += sizeof(float) * 2 * AFF_TABLE_LDS_STRIDE;
    }
    const int fair_off = a.fair_off;
    // guarded split-precision call: every chunk is followed by the exact-fp32 kernels, which return at once unless the chunk's guard fired
    // (an in-place call -- rotation_out == rotation -- cannot be re-run from its own overwritten input: it runs unguarded, on the kernel
    // instantiations whose softplus is overflow-safe on its own; include/rnf_hip.h "aliasing")
    const bool guarded = prec == 1 && any_mlp && have_fb && ws && ws_bytes >= PARTIALS_BYTES && guard_allowed() && !(rot_out && rot_out == rot);
    int *guard = guarded ? reinterpret_cast<int *>(reinterpret_cast<double *>(ws) + GUARD_AT) : nullptr;
    // FUSED: forward pass of a conditional flow whose every MLP layer is conditional, F <= 256, projection records equally spaced in the
    // blob (both packers lay them out that way).  Guarded launches only: the instantiation uses the one-piece softplus.
    bool fused = fused_allowed() && lean2 && o.dir == 0 && prec == 1 && pipe && any_mlp && n_slots > 0 && all_mlp_cond && !shared && !ext && !o.states &&
                 F <= FUSED_MAX_F && guarded && a.tab_off >= 0;
    int feat_stride = (int)((featproj_packed_floats(F) + 3) / 4 * 4);
    if (fused && n_slots > 1) feat_stride = fp_primary[1] - fp_primary[0];
    for (int sl = 0; fused && sl < n_slots; ++sl)
        if (fp_primary[sl] != fp_primary[0] + sl * feat_stride) fused = false;
    size_t lds_fused = 0;
    if (fused) {
        lds_fused = (lds_bytes + 15) / 16 * 16;
        a.pa_off = (int)(lds_fused / sizeof(float));
        lds_fused += sizeof(float) * FUSED_PA_FLOATS;
        a.feat_F = F;
        a.feat_base = fp_primary[0];
        a.feat_stride = feat_stride;
        if (lds_fused > 160 * 1024) fused = false;
    }
    FlowArgs afb;
    if (guarded) {
        HIP_TRY(hipMemsetAsync(guard, 0, 2 * sizeof(int), stream));
        afb = a;
        for (int l = 0; l < n_layers; ++l)
            if (kind_has_mlp(afb.layers[l].x & 15)) afb.layers[l].y = fbp[l];
    }
    // Shared feature rows on the fast kernels (round 5; pose estimation: agent.py:238-263 evaluates number_queries rotations per image
    // feature): rows of >= 32 rotations, guarded split-precision call with DMA staging, and either a forward pass of the conditional-lean
    // structure (SYMSOL-I: Condition16Trans + conditional Moebius + constant affine) or an inverse pass with K <= 64 segments of a flow
    // without extended layers.  Decided by the flow and the call's row length, never by the batch size.  Everything else with shared rows
    // stays on the extended instantiation.
    const bool rows_lean2 = lean2 && all_mlp_cond && guarded && a.tab_off >= 0;
    // GUARDED calls only, in both directions: the row records enter x0 through a matrix step (flow_kernels.h GFragRows), where a non-finite
    // record of one image would also poison the rotations of the NEXT image that share its wave (NaN x 0 = NaN); the guard sees that and the
    // exact-fp32 re-run, which reads the records per lane, restores per-image semantics.  Unguarded calls keep the extended instantiation.
    const bool rows_fast = shared && !ext_layers && prec == 1 && pipe && any_mlp && o.feature_div >= 32 && n < (1LL << 31) &&
                           feat_rows < (1LL << 24) && !o.states && guarded && (o.dir == 0 ? rows_lean2 : (KT <= 8));
    if (rows_fast) ext = false;
    bool first = true;
    for (long long base = 0; base < n; base += chunk_cap) {
        const long long cn = (n - base) < chunk_cap ? (n - base) : chunk_cap;
        // waves per workgroup of the stack kernel: the forward split-precision kernel goes 16 wide once 8-wave workgroups would
        // no longer fit the CUs in one round
        const bool wide = !fused && o.dir == 0 && prec == 1 && pipe && !ext && wide_allowed() && cn > (long long)cus * NW_FWD_H * 32;
        const bool narrow = !fused && o.dir == 0 && prec == 1 && pipe && !ext && wide_allowed() && cn <= (long long)cus * NW_FWD_NARROW * 32;
        const bool big_inv = o.dir == 1 && any_mlp && KT > 8;                  // 4-wave instantiation (512 registers per lane)
        const bool narrow_inv = o.dir == 1 && prec == 1 && pipe && !ext && any_mlp && KT <= 8 && wide_allowed() && cn <= (long long)cus * NW_FWD_NARROW * 32;
        const int nwk = fused ? NW_FUSED : (big_inv ? NW_INV_BIG : (wide ? NW_FWD_WIDE : ((narrow || narrow_inv) ? NW_FWD_NARROW : ((o.dir == 0 && prec == 1) ? NW_FWD_H : NW))));
        // kernel family of this call (launch_stack): fixed by the flow and by `guarded`, the same for every chunk and batch size
        const int family = (guarded && a.tab_off >= 0 && o.dir == 0 && prec == 1 && pipe && !ext) ? (lean ? 1 : (lean2 && all_mlp_cond && n_slots > 0 && (!shared || rows_fast) ? 2 : 0)) : 0;
        a.fair_off = (wide || narrow || family || (fused && NW_FUSED != 8)) ? -1 : fair_off;             // the governor pairs two waves per SIMD (general 8-wave kernel)
        const long long ntiles = (cn + nwk * 32 - 1) / (nwk * 32);
        const long long ntiles_fp = (cn + NW_FP * 32 - 1) / (NW_FP * 32);
        const int nw_fb = (o.dir == 1 && any_mlp && KT > 8) ? NW_INV_BIG : NW;
        const long long ntiles_fb = (cn + nw_fb * 32 - 1) / (nw_fb * 32);         // the exact-fp32 re-run uses NW-wave workgroups
        long long groups = (ntiles * nwk > ntiles_fp * NW_FP) ? ntiles * nwk : ntiles_fp * NW_FP;
        if (guarded && ntiles_fb * nw_fb > groups) groups = ntiles_fb * nw_fb;
        if ((cn + 127) / 128 * 4 > groups) groups = (cn + 127) / 128 * 4;
        if (guarded && base > 0) HIP_TRY(hipMemsetAsync(guard, 0, sizeof(int), stream));   // per-chunk guard; guard[1] stays
        int grid = (int)(ntiles < cus ? ntiles : cus);
        const int cus_fp = cus * (8 / NW_FP);
        const int grid_fp = (int)(ntiles_fp < cus_fp ? ntiles_fp : cus_fp);
        // feature projection of this chunk (shared rows: ONE projection of the feature rows, before the first chunk); `fb`: the
        // exact-fp32 projection of a guarded call, which runs only if the guard fired (every chunk, also with shared rows)
        auto project = [&](bool fb) -> int {
            const long long pn = shared ? feat_rows : cn;
            const long long pt = (pn + NW_FP * 32 - 1) / (NW_FP * 32);
            const int grid_p = shared ? (int)(pt < cus_fp ? pt : cus_fp) : grid_fp;
            fp.feat = shared ? feat : feat + base * F;
            fp.blob = blob;
            fp.G = G;
            fp.n = pn;
            fp.g_groups = groups;
            fp.F = F;
            fp.n_slots = n_slots;
            fp.row_mode = shared ? 1 : 0;
            fp.only_if = fb ? guard : nullptr;
            if (fb) for (int sl = 0; sl < n_slots; ++sl) fp.feat_off[sl] = fbf[sl];
            const int pprec = (fb || prec == 2) ? 0 : prec;       // bf16x3 flows: the exact-fp32 projection (their records hold the fp32 image)
            const int kchunk = F < FP_KCHUNK ? F : FP_KCHUNK;
            size_t fl = sizeof(float) * (pprec ? (size_t)2 * (FP_KCHUNK / 16) * 512 : (size_t)kchunk / 8 * 256);   // f16x2: two DMA buffers
            if (pprec && F > FP_KCHUNK && F <= 2 * FP_KCHUNK) {     // K split over wave pairs: no partial sums through the scratch (featproj_kernel.h)
                const long long pt2 = (pn + 127) / 128;
                const int grid2 = (int)(pt2 < cus ? pt2 : cus);
                auto kern = featproj_ksplit_kernel;
                HIP_TRY(allow_lds(kern, FP2_LDS_BYTES));
                hipLaunchKernelGGL(kern, dim3(grid2), dim3(8 * 64), FP2_LDS_BYTES, stream, fp);
            } else if (pprec && F > FP_KCHUNK) {                    // F > 512: K-chunks whose partial sums pass through the scratch
                auto kern = featproj_kernel<NW_FP, 1, true>;
                HIP_TRY(allow_lds(kern, fl));
                hipLaunchKernelGGL(kern, dim3(grid_p), dim3(NW_FP * 64), fl, stream, fp);
            } else if (pprec) {
                auto kern = featproj_kernel<NW_FP, 1>;
                HIP_TRY(allow_lds(kern, fl));
                hipLaunchKernelGGL(kern, dim3(grid_p), dim3(NW_FP * 64), fl, stream, fp);
            } else {
                auto kern = featproj_kernel<NW_FP, 0>;
                HIP_TRY(allow_lds(kern, fl));
                hipLaunchKernelGGL(kern, dim3(grid_p), dim3(NW_FP * 64), fl, stream, fp);
            }
            HIP_TRY(hipGetLastError());
            return 0;
        };
        if (n_slots && !fused && (!shared || base == 0)) {
            if (int rc = project(false)) return rc;
        }
        a.rot_in = rot + base * 9;
        a.G = G;
        a.rot_out = rot_out ? rot_out + base * 9 : nullptr;
        a.ldj_out = ldj_out ? ldj_out + base : nullptr;
        a.logp_out = o.logp_out ? o.logp_out + base : nullptr;
        a.partials = o.sum_out ? partials : nullptr;
        a.n = cn;
        a.sample_base = base;
        a.g_groups = groups;
        a.guard = guard;
        a.guard_mode = guarded ? 1 : 0;
        int rc;
        const bool rows_now = rows_fast;                   // (the exact-fp32 re-run below shadows it: shared rows there stay on the extended kernels)
#define RNF_LAUNCH(DIR_, KT_)                                                                                   \
    prec == 2 ? (ext ? launch_stack<DIR_, KT_, false, 2, true>(a, grid, lds_bytes, stream, nwk)                     \
                     : launch_stack<DIR_, KT_, false, 2>(a, grid, lds_bytes, stream, nwk)) :                        \
    rows_now ? launch_stack<DIR_, KT_, true, 1, false, true>(a, grid, lds_bytes, stream, nwk, family) :             \
    ext ? (pipe ? (prec ? launch_stack<DIR_, KT_, true, 1, true>(a, grid, lds_bytes, stream, nwk)                   \
                        : launch_stack<DIR_, KT_, true, 0, true>(a, grid, lds_bytes, stream, nwk))                  \
                : (prec ? launch_stack<DIR_, KT_, false, 1, true>(a, grid, lds_bytes, stream, nwk)                  \
                        : launch_stack<DIR_, KT_, false, 0, true>(a, grid, lds_bytes, stream, nwk))) :              \
    (pipe ? (prec ? launch_stack<DIR_, KT_, true, 1>(a, grid, lds_bytes, stream, nwk, family)                        \
                  : launch_stack<DIR_, KT_, true, 0>(a, grid, lds_bytes, stream, nwk))                              \
          : (prec ? launch_stack<DIR_, KT_, false, 1>(a, grid, lds_bytes, stream, nwk)                              \
                  : launch_stack<DIR_, KT_, false, 0>(a, grid, lds_bytes, stream, nwk)))
        if (fused) {
            a.feat = feat + base * F;
            a.stash = G;                                   // grid * 8 waves * 8 KB <= the projection scratch the workspace is sized for
            rc = launch_fused(a, grid, lds_fused, stream);
        }
        else if (o.dir == 0) rc = RNF_LAUNCH(0, 0);
        else if (kt_inv == 1) rc = RNF_LAUNCH(1, 1);
        else if (kt_inv == 2) rc = RNF_LAUNCH(1, 2);
        else if (kt_inv == 4) rc = RNF_LAUNCH(1, 4);
        else if (kt_inv == 8) rc = RNF_LAUNCH(1, 8);
        else if (prec == 2) rc = ext ? launch_big_inverse<2, true>(a, grid, lds_bytes, stream) : launch_big_inverse<2>(a, grid, lds_bytes, stream);
        else rc = ext ? (prec ? launch_big_inverse<1, true>(a, grid, lds_bytes, stream) : launch_big_inverse<0, true>(a, grid, lds_bytes, stream))
                      : (prec ? launch_big_inverse<1>(a, grid, lds_bytes, stream) : launch_big_inverse<0>(a, grid, lds_bytes, stream));
        if (rc) return rc;
        int grid_fb = 0;
        if (guarded) {                                   // the same chunk on the exact-fp32 kernels, skipped on the device unless the guard fired
            if (n_slots) {
                if (int rc2 = project(true)) return rc2;
                for (int sl = 0; sl < n_slots; ++sl) fp.feat_off[sl] = fp_primary[sl];
            }
            FlowArgs b = afb;
            b.rot_in = a.rot_in; b.G = a.G; b.rot_out = a.rot_out; b.ldj_out = a.ldj_out; b.logp_out = a.logp_out;
            b.partials = a.partials ? partials + PARTIALS_FB_AT : nullptr;
            b.n = cn; b.sample_base = base; b.g_groups = groups; b.guard = guard; b.guard_mode = 2; b.fair_off = -1;
            grid_fb = (int)(ntiles_fb < cus ? ntiles_fb : cus);
            const int grid_keep = grid;
            const bool ext_fb = ext || shared;            // shared rows: the exact-fp32 re-run reads them on the extended instantiation
            {
                FlowArgs &a = b;                          // RNF_LAUNCH names `a`, `grid`, `prec`, `nwk`, `ext`, `rows_now`, `lds_bytes`
                const int grid = grid_fb, prec = fb_prec, nwk = NW;
                const bool ext = ext_fb, rows_now = false;
                const int family = 0;                     // the strict kernels have no lean family
                (void)family;
                // bf16x3 fallback records (round 6: the guard's re-run target on host-packed flows -- 1.8x the guarded time instead of the
                // 3.3x of the exact-fp32 MFMA): the ring-staged kernels with their own LDS layout (three regions + the affine blocks)
                size_t lds_fb = lds_bytes;
                if (fb_prec == 2) {
                    lds_fb = sizeof(float) * (size_t)3 * Mlp<2>::REGION_FLOATS;
                    a.tab_off = (int)(lds_fb / sizeof(float));
                    lds_fb += sizeof(float) * 2 * AFF_TABLE_LDS_STRIDE;
                }
                const size_t lds_bytes = lds_fb;
                if (o.dir == 0) rc = RNF_LAUNCH(0, 0);
                else if (kt_inv == 1) rc = RNF_LAUNCH(1, 1);
                else if (kt_inv == 2) rc = RNF_LAUNCH(1, 2);
                else if (kt_inv == 4) rc = RNF_LAUNCH(1, 4);
                else if (kt_inv == 8) rc = RNF_LAUNCH(1, 8);
                else if (prec == 2) rc = ext ? launch_big_inverse<2, true>(a, grid, lds_bytes, stream) : launch_big_inverse<2>(a, grid, lds_bytes, stream);
                else rc = ext ? launch_big_inverse<0, true>(a, grid, lds_bytes, stream) : launch_big_inverse<0>(a, grid, lds_bytes, stream);
            }
            (void)grid_keep;
            if (rc) return rc;
        }
#undef RNF_LAUNCH
        if (o.sum_out) {
            hipLaunchKernelGGL(nll_finalize_kernel, dim3(1), dim3(256), 0, stream, (const double *)partials, grid, (double)cn, o.sum_out, first ? 0 : 1,
                               (const double *)(partials + PARTIALS_FB_AT), grid_fb, (const int *)guard);
            HIP_TRY(hipGetLastError());
        }
        first = false;
    }
    return 0;
}

extern "C" int rnf_flow_forward(const float *rot, const float *feat, int64_t n, int32_t F, const float *blob,
                                const int32_t *desc, int32_t n_layers, int32_t K, float *rot_out, float *ldj_out, void *ws,
                                size_t ws_bytes, void *stream) {
    RunOpts o{0, nullptr, nullptr, 0, nullptr, nullptr};
    return run_flow(rot, feat, n, F, blob, desc, n_layers, K, rot_out, ldj_out, ws, ws_bytes, stream, o);
}

// Shared feature rows (pose estimation, agent.py:238-263: every image feature is evaluated against number_queries rotations; the
// reference materialises feature.repeat): feature_dev has n / feature_div rows, row r serves rotations [r * feature_div, (r+1) * feature_div).
// The feature projection then runs once per ROW and its scratch is one 64-float record per (layer, row).
extern "C" int rnf_flow_forward_shared(const float *rot, const float *feat, int64_t n, int32_t F, int64_t feature_div, const float *blob,
                                       const int32_t *desc, int32_t n_layers, int32_t K, float *rot_out, float *ldj_out, void *ws,
                                       size_t ws_bytes, void *stream) {
    RunOpts o{0, nullptr, nullptr, 0, nullptr, nullptr};
    o.feature_div = feature_div;
    return run_flow(rot, feat, n, F, blob, desc, n_layers, K, rot_out, ldj_out, ws, ws_bytes, stream, o);
}

extern "C" int rnf_flow_inverse_shared(const float *rot, const float *feat, int64_t n, int32_t F, int64_t feature_div, const float *blob,
                                       const int32_t *desc, int32_t n_layers, int32_t K, float *rot_out, float *ldj_out, void *ws,
                                       size_t ws_bytes, void *stream) {
    RunOpts o{1, nullptr, nullptr, 0, nullptr, nullptr};
    o.feature_div = feature_div;
    return run_flow(rot, feat, n, F, blob, desc, n_layers, K, rot_out, ldj_out, ws, ws_bytes, stream, o);
}

// ------------------------------------------------------------------------------------------------------------
// training: forward that saves per-layer states, and the reverse sweep (train_kernels.h)
// ------------------------------------------------------------------------------------------------------------
// per-sample matrix layers (RNF_LAYER_SIDE16 / SIDE16_ROT / SIDE9): side_dev float[n_side_layers][n][16], desc param_offset = slot
extern "C" int rnf_flow_forward_side(const float *rot, const float *feat, int64_t n, int32_t F, const float *side, const float *blob,
                                     const int32_t *desc, int32_t n_layers, int32_t K, float *rot_out, float *ldj_out, void *ws,
                                     size_t ws_bytes, void *stream) {
    RunOpts o{0, nullptr, nullptr, 0, nullptr, nullptr};
    o.side = side;
    return run_flow(rot, feat, n, F, blob, desc, n_layers, K, rot_out, ldj_out, ws, ws_bytes, stream, o);
}
extern "C" int rnf_flow_inverse_side(const float *rot, const float *feat, int64_t n, int32_t F, const float *side, const float *blob,
                                     const int32_t *desc, int32_t n_layers, int32_t K, float *rot_out, float *ldj_out, void *ws,
                                     size_t ws_bytes, void *stream) {
    RunOpts o{1, nullptr, nullptr, 0, nullptr, nullptr};
    o.side = side;
    return run_flow(rot, feat, n, F, blob, desc, n_layers, K, rot_out, ldj_out, ws, ws_bytes, stream, o);
}
extern "C" int rnf_flow_log_prob_side(const float *rot, const float *feat, int64_t n, int32_t F, const float *side, const float *blob,
                                      const int32_t *desc, int32_t n_layers, int32_t K, const float *fisher_A, const float *fisher_c,
                                      int64_t fisher_B, float *rot_out, float *ldj_out, float *logp_out, double *sum_out, void *ws,
                                      size_t ws_bytes, void *stream) {
    if ((fisher_A == nullptr) != (fisher_c == nullptr)) return fail("fisher_A and fisher_c must both be given or both be null");
    RunOpts o{0, fisher_A, fisher_c, fisher_B, logp_out, sum_out};
    o.side = side;
    return run_flow(rot, feat, n, F, blob, desc, n_layers, K, rot_out, ldj_out, ws, ws_bytes, stream, o);
}

// ConditionalTransform(F, <= 16 outputs)(feature) alone (flow/condition.py:24-30): the nets of ConditionRot (flow/rottrans.py:40) and
// ConditionLU (flow/squeezetrans.py:117-119), whose outputs the reference post-processes with batched torch ops.  Records from
// rnf_pack_cond16 (rows beyond the net's outputs zero); out [n][16], output o of the net in column o.
template <int NWc, int PREC>
__global__ __launch_bounds__(NWc * 64) void cond_mlp_kernel(const float *G, long long g_groups, long long n, const float *layer, float *out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const long long ntiles = (n + NWc * 32 - 1) / (NWc * 32);
    stage_floats(lds, layer, MOB_HEAD_FLOATS + MOB_LAST_TILE_FLOATS, tid, NWc * 64);
    __syncthreads();
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long group = tile * NWc + wave;
        const long long sample = group * 32 + j;
        typename Mlp<PREC>::Act tt;
        Fair nofair{lds, wave, -1, 0};
        bool bad = false;
        GFrag<false> g{G + (size_t)group * G_FLOATS_PER_GROUP, false};
        Mlp<PREC>::head(lds, lane, h, 0.f, 0.f, 0.f, g, tt, nofair, bad);
        const f32x16 o = Mlp<PREC>::last(lds + MOB_LAST, lane, h, tt);
        if (sample < n) {
#pragma unroll
            for (int gq = 0; gq < 2; ++gq) {              // packed row 8g + 4h + c  <->  output 4 (2g + h) + c (rnf_pack_cond16)
                float4 v = make_float4(o[4 * gq], o[4 * gq + 1], o[4 * gq + 2], o[4 * gq + 3]);
                if (bad) v = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""));
                reinterpret_cast<float4 *>(out + sample * 16)[2 * gq + h] = v;
            }
        }
    }
}

extern "C" int rnf_cond_mlp_forward(const float *feat, int64_t n, int32_t F, const float *blob, int32_t layer_off, int32_t feat_off,
                                    int32_t prec, float *out, void *ws, size_t ws_bytes, void *stream_v) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    if (n < 0) return fail("n=%lld is negative", (long long)n);
    if (F <= 0 || F % 8) return fail("feature_dim=%d must be a positive multiple of 8 (pad on the host)", F);
    if (prec != RNF_PREC_FP32 && prec != RNF_PREC_F16X2) return fail("unknown precision %d", prec);
    if (layer_off < 0 || layer_off % 4 || feat_off < 0 || feat_off % 4) return fail("record offsets must be non-negative multiples of 4");
    if (n == 0) return 0;
    if (!feat || !blob || !out || !ws) return fail("rnf_cond_mlp_forward: null pointer");
    if (ws_bytes < rnf_workspace_bytes(n, 1)) return fail("workspace of %zu bytes is smaller than the %zu needed", ws_bytes, rnf_workspace_bytes(n, 1));
    float *G = reinterpret_cast<float *>(reinterpret_cast<char *>(ws) + PARTIALS_BYTES);
    const int cus = device_cus();
    const size_t lds_bytes = sizeof(float) * (MOB_HEAD_FLOATS + MOB_LAST_TILE_FLOATS);
    for (long long base = 0; base < n; base += CHUNK_SAMPLES) {
        const long long cn = (n - base) < CHUNK_SAMPLES ? (n - base) : CHUNK_SAMPLES;
        const long long ntiles = (cn + NW * 32 - 1) / (NW * 32), ntiles_fp = (cn + NW_FP * 32 - 1) / (NW_FP * 32);
        const long long groups = (ntiles * NW > ntiles_fp * NW_FP) ? ntiles * NW : ntiles_fp * NW_FP;
        FeatProjArgs fp;
        std::memset(&fp, 0, sizeof(fp));
        fp.feat = feat + base * F; fp.blob = blob; fp.G = G; fp.n = cn; fp.g_groups = groups; fp.F = F; fp.n_slots = 1; fp.row_mode = 0;
        fp.feat_off[0] = feat_off;
        const int cus_fp = cus * (8 / NW_FP);
        const int grid_fp = (int)(ntiles_fp < cus_fp ? ntiles_fp : cus_fp);
        const int kchunk = F < FP_KCHUNK ? F : FP_KCHUNK;
        const size_t fl = sizeof(float) * (prec ? (size_t)2 * (FP_KCHUNK / 16) * 512 : (size_t)kchunk / 8 * 256);
        const int grid = (int)(ntiles < cus ? ntiles : cus);
        if (prec) {
            if (F > FP_KCHUNK && F <= 2 * FP_KCHUNK) {
                const long long pt2 = (cn + 127) / 128;
                auto kp = featproj_ksplit_kernel;
                HIP_TRY(allow_lds(kp, FP2_LDS_BYTES));
                hipLaunchKernelGGL(kp, dim3((int)(pt2 < cus ? pt2 : cus)), dim3(8 * 64), FP2_LDS_BYTES, stream, fp);
            } else if (F > FP_KCHUNK) {
                auto kp = featproj_kernel<NW_FP, 1, true>;
                HIP_TRY(allow_lds(kp, fl));
                hipLaunchKernelGGL(kp, dim3(grid_fp), dim3(NW_FP * 64), fl, stream, fp);
            } else {
                auto kp = featproj_kernel<NW_FP, 1>;
                HIP_TRY(allow_lds(kp, fl));
                hipLaunchKernelGGL(kp, dim3(grid_fp), dim3(NW_FP * 64), fl, stream, fp);
            }
            auto kern = cond_mlp_kernel<NW, 1>;
            HIP_TRY(allow_lds(kern, lds_bytes));
            hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds_bytes, stream, (const float *)G, groups, cn, blob + layer_off, out + base * 16);
        } else {
            auto kp = featproj_kernel<NW_FP, 0>;
            HIP_TRY(allow_lds(kp, fl));
            hipLaunchKernelGGL(kp, dim3(grid_fp), dim3(NW_FP * 64), fl, stream, fp);
            auto kern = cond_mlp_kernel<NW, 0>;
            HIP_TRY(allow_lds(kern, lds_bytes));
            hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds_bytes, stream, (const float *)G, groups, cn, blob + layer_off, out + base * 16);
        }
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

// ConditionRot (flow/rottrans.py:37-66): rot = U^T V of svd(I + reshape(net(feature), 4, 4)) per sample, with the SIGN CONVENTIONS of the
// reference's torch.svd (LAPACK sgesdd; svd4_lapack.h restates that path for 4x4).  mlp_out [n][16] = the conditioner's outputs (row-major
// 4x4 per sample), rot_out [n][16] = the orthogonal matrix the RNF_LAYER_SIDE16_ROT layer applies to the quaternion.  One thread per sample.
// usv != nullptr: also U [n][16] (row-major), the singular values [n][4] and V^T [n][16], for the analytic backward of U^T V (rottrans.py).
__global__ void condrot_utv_kernel(const float *mlp_out, long long n, float *rot_out, float *u_out, float *s_out, float *vt_out, int *fail_flag) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float A[16], U[16], sv[4], VT[16];
    for (int k = 0; k < 16; ++k) A[k] = mlp_out[i * 16 + k] + ((k % 5 == 0) ? 1.0f : 0.0f);
    if (!svd4::svd(A, U, sv, VT) && fail_flag) atomicOr(fail_flag, 1);
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {                                   // rot = U^T V: rot[r][c] = u_r . v_c
            float a = 0.f;
            for (int k = 0; k < 4; ++k) a += U[4 * k + r] * VT[4 * c + k];
            rot_out[i * 16 + 4 * r + c] = a;
        }
    if (u_out) {
        for (int k = 0; k < 16; ++k) { u_out[i * 16 + k] = U[k]; vt_out[i * 16 + k] = VT[k]; }
        for (int k = 0; k < 4; ++k) s_out[i * 4 + k] = sv[k];
    }
}
static int condrot_launch(const float *mlp_out, int64_t n, float *rot_out, float *u, float *sv, float *vt, int32_t *fail_flag, void *stream) {
    if (n < 0) return fail("rnf_condrot: n=%lld", (long long)n);
    if (n == 0) return 0;
    if (!mlp_out || !rot_out) return fail("rnf_condrot: null pointer");
    hipLaunchKernelGGL(condrot_utv_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), mlp_out, (long long)n,
                       rot_out, u, sv, vt, reinterpret_cast<int *>(fail_flag));
    HIP_TRY(hipGetLastError());
    return 0;
}
extern "C" int rnf_condrot_matrices(const float *mlp_out, int64_t n, float *rot_out, int32_t *fail_flag, void *stream) {
    return condrot_launch(mlp_out, n, rot_out, nullptr, nullptr, nullptr, fail_flag, stream);
}
extern "C" int rnf_condrot_svd(const float *mlp_out, int64_t n, float *rot_out, float *u_out, float *s_out, float *vt_out, int32_t *fail_flag,
                               void *stream) {
    if (n > 0 && (!u_out || !s_out || !vt_out)) return fail("rnf_condrot_svd: null factor pointer");
    return condrot_launch(mlp_out, n, rot_out, u_out, s_out, vt_out, fail_flag, stream);
}

// ConditionLU (flow/squeezetrans.py:94-131) on the device (round 6; until round 5 the host ran the reference's einsum / torch.diag on the GPU):
//     weight[n] = w_p . (reshape(wl[n], C, C) * l_mask + l_eye) . (reshape(wu[n], C, C) * u_mask + dvec),   dvec[d] = s_sign[d] exp(ws[d][d])
// `torch.diag` of the 2-D [N, C] tensor s_sign * exp(ws) (squeezetrans.py:126-127) is its DIAGONAL ACROSS THE BATCH -- entry d of row d,
// d < C -- and the resulting C-vector is broadcast over the last axis of every sample's upper factor: it is added to EVERY row c of
// column d, and the weight of sample n depends on batch rows 0 .. C-1.  Reproduced as defined.  consts = w_p [C*C] | l_mask [C*C] |
// u_mask [C*C] | l_eye [C*C] | s_sign [C] (the module's buffers, as loaded from the checkpoint).  One thread per sample; out [n][16]:
// the C x C matrix row-major in the first C*C floats (+ identity when add_identity: Condition9TransLU, squeezetrans.py:269-271), rest 0.
template <int C>
__global__ void condlu_assemble_kernel(const float *wl, const float *wu, const float *ws, int sl, int su, int ss, long long n, const float *consts,
                                       int add_identity, float *out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    constexpr int CC = C * C;
    const float *wp = consts, *lm = consts + CC, *um = consts + 2 * CC, *le = consts + 3 * CC, *sg = consts + 4 * CC;
    float dvec[C], Lm[CC], Um[CC], PL[CC];
#pragma unroll
    for (int d = 0; d < C; ++d) dvec[d] = sg[d] * expf(ws[(long long)d * ss + d]);       // rows 0 .. C-1 of the BATCH
#pragma unroll
    for (int k = 0; k < CC; ++k) {
        Lm[k] = wl[i * sl + k] * lm[k] + le[k];
        Um[k] = wu[i * su + k] * um[k] + dvec[k % C];
    }
#pragma unroll
    for (int a = 0; a < C; ++a)
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float t = 0.f;
#pragma unroll
            for (int b = 0; b < C; ++b) t = fmaf(wp[a * C + b], Lm[b * C + c], t);
            PL[a * C + c] = t;
        }
    float o[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) o[k] = 0.f;
#pragma unroll
    for (int a = 0; a < C; ++a)
#pragma unroll
        for (int d = 0; d < C; ++d) {
            float t = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) t = fmaf(PL[a * C + c], Um[c * C + d], t);
            o[a * C + d] = t + ((add_identity && a == d) ? 1.0f : 0.0f);
        }
    float4 *dst = reinterpret_cast<float4 *>(out + i * 16);
    dst[0] = make_float4(o[0], o[1], o[2], o[3]); dst[1] = make_float4(o[4], o[5], o[6], o[7]);
    dst[2] = make_float4(o[8], o[9], o[10], o[11]); dst[3] = make_float4(o[12], o[13], o[14], o[15]);
}

// Backward of the assembly: g [n][16] = dL/d(weight).  dL/dL = P^T G U^T, dL/dU = (P L)^T G; g_wl = dL/dL * l_mask, g_wu = dL/dU * u_mask
// (row stride gstride), and the batch-coupled diagonal collects  g_dvec[d] = sum_n sum_c dL/dU[n][c][d]  (wave sums, then one float atomic
// per wave and column into dsum [C], zeroed by the launcher).  condlu_diag_kernel then writes g_ws: zero everywhere except
// g_ws[d][d] = g_dvec[d] * dvec[d]  (d dvec[d] / d ws[d][d] = dvec[d]).
template <int C>
__global__ void condlu_backward_kernel(const float *wl, const float *wu, const float *ws, int sl, int su, int ss, long long n, const float *consts,
                                       const float *g, float *g_wl, float *g_wu, int gsl, int gsu, float *dsum) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    constexpr int CC = C * C;
    const float *wp = consts, *lm = consts + CC, *um = consts + 2 * CC, *le = consts + 3 * CC, *sg = consts + 4 * CC;
    float dcol[C];
#pragma unroll
    for (int d = 0; d < C; ++d) dcol[d] = 0.f;
    if (i < n) {
        float dvec[C], Lm[CC], Um[CC], PL[CC], PtG[CC], G[CC];
#pragma unroll
        for (int d = 0; d < C; ++d) dvec[d] = sg[d] * expf(ws[(long long)d * ss + d]);
#pragma unroll
        for (int k = 0; k < CC; ++k) {
            Lm[k] = wl[i * sl + k] * lm[k] + le[k];
            Um[k] = wu[i * su + k] * um[k] + dvec[k % C];
            G[k] = g[i * 16 + k];
        }
#pragma unroll
        for (int a = 0; a < C; ++a)
#pragma unroll
            for (int c = 0; c < C; ++c) {
                float t = 0.f, u = 0.f;
#pragma unroll
                for (int b = 0; b < C; ++b) {
                    t = fmaf(wp[a * C + b], Lm[b * C + c], t);            // (P L)[a][c]
                    u = fmaf(wp[b * C + a], G[b * C + c], u);             // (P^T G)[a][c]
                }
                PL[a * C + c] = t;
                PtG[a * C + c] = u;
            }
#pragma unroll
        for (int b = 0; b < C; ++b)
#pragma unroll
            for (int c = 0; c < C; ++c) {
                float gl = 0.f, gu = 0.f;
#pragma unroll
                for (int d = 0; d < C; ++d) {
                    gl = fmaf(PtG[b * C + d], Um[c * C + d], gl);         // dL/dL[b][c] = sum_d (P^T G)[b][d] U[c][d]
                    gu = fmaf(PL[d * C + b], G[d * C + c], gu);           // dL/dU[b][c] = sum_a (P L)[a][b] G[a][c]
                }
                g_wl[i * gsl + b * C + c] = gl * lm[b * C + c];
                g_wu[i * gsu + b * C + c] = gu * um[b * C + c];
                dcol[c] += gu;
            }
    }
#pragma unroll
    for (int d = 0; d < C; ++d) {
        float v = dcol[d];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if ((threadIdx.x & 63) == 0 && v != 0.f) atomicAdd(dsum + d, v);
    }
}
template <int C>
__global__ void condlu_diag_kernel(const float *ws, int ss, long long n, const float *consts, const float *dsum, float *g_ws, int gss) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *sg = consts + 4 * C * C;
#pragma unroll
    for (int d = 0; d < C; ++d) g_ws[i * gss + d] = (i == d) ? dsum[d] * sg[d] * expf(ws[(long long)d * ss + d]) : 0.f;
}
extern "C" int rnf_condlu_matrices(const float *wl, const float *wu, const float *ws, int32_t stride_wl, int32_t stride_wu, int32_t stride_ws, int64_t n,
                                   int32_t C, const float *consts, int32_t add_identity, float *side_out, void *stream) {
    if (C != 3 && C != 4) return fail("rnf_condlu_matrices: in_channel %d (3 or 4)", C);
    if (n < 0 || stride_wl < C * C || stride_wu < C * C || stride_ws < C) return fail("rnf_condlu_matrices: n=%lld row strides %d %d %d", (long long)n, stride_wl, stride_wu, stride_ws);
    if (n == 0) return 0;
    // the reference broadcasts the C-vector torch.diag(...) of an [n, C] tensor (length min(n, C)) against [n, C, C]: fewer than C rows fail there
    if (n < C) return fail("rnf_condlu_matrices: a batch of %lld rows has no %d-entry batch diagonal (flow/squeezetrans.py:126-127 fails to broadcast)", (long long)n, C);
    if (!wl || !wu || !ws || !consts || !side_out) return fail("rnf_condlu_matrices: null pointer");
    const dim3 grid((unsigned)((n + 127) / 128)), block(128);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (C == 4) hipLaunchKernelGGL(condlu_assemble_kernel<4>, grid, block, 0, st, wl, wu, ws, stride_wl, stride_wu, stride_ws, (long long)n, consts, add_identity, side_out);
    else hipLaunchKernelGGL(condlu_assemble_kernel<3>, grid, block, 0, st, wl, wu, ws, stride_wl, stride_wu, stride_ws, (long long)n, consts, add_identity, side_out);
    HIP_TRY(hipGetLastError());
    return 0;
}
extern "C" int rnf_condlu_backward(const float *wl, const float *wu, const float *ws, int32_t stride_wl, int32_t stride_wu, int32_t stride_ws, int64_t n,
                                   int32_t C, const float *consts, const float *g_side, float *g_wl, float *g_wu, float *g_ws, float *scratch, void *stream) {
    if (C != 3 && C != 4) return fail("rnf_condlu_backward: in_channel %d (3 or 4)", C);
    if (n < C || stride_wl < C * C || stride_wu < C * C || stride_ws < C) return fail("rnf_condlu_backward: n=%lld row strides %d %d %d", (long long)n, stride_wl, stride_wu, stride_ws);
    if (!wl || !wu || !ws || !consts || !g_side || !g_wl || !g_wu || !g_ws || !scratch) return fail("rnf_condlu_backward: null pointer");
    const dim3 grid((unsigned)((n + 127) / 128)), block(128);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    HIP_TRY(hipMemsetAsync(scratch, 0, sizeof(float) * 4, st));
    if (C == 4) {
        hipLaunchKernelGGL(condlu_backward_kernel<4>, grid, block, 0, st, wl, wu, ws, stride_wl, stride_wu, stride_ws, (long long)n, consts, g_side, g_wl, g_wu, C * C, C * C, scratch);
        hipLaunchKernelGGL(condlu_diag_kernel<4>, grid, block, 0, st, ws, stride_ws, (long long)n, consts, scratch, g_ws, C);
    } else {
        hipLaunchKernelGGL(condlu_backward_kernel<3>, grid, block, 0, st, wl, wu, ws, stride_wl, stride_wu, stride_ws, (long long)n, consts, g_side, g_wl, g_wu, C * C, C * C, scratch);
        hipLaunchKernelGGL(condlu_diag_kernel<3>, grid, block, 0, st, ws, stride_ws, (long long)n, consts, scratch, g_ws, C);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int rnf_flow_forward_train(const float *rot, const float *feat, int64_t n, int32_t F, const float *blob, const int32_t *desc,
                                      int32_t n_layers, int32_t K, float *rot_out, float *ldj_out, float *states, void *ws,
                                      size_t ws_bytes, void *stream) {
    if (n > 0 && !states) return fail("states buffer is null");
    RunOpts o{0, nullptr, nullptr, 0, nullptr, nullptr};
    o.states = states;
    return run_flow(rot, feat, n, F, blob, desc, n_layers, K, rot_out, ldj_out, ws, ws_bytes, stream, o);
}

// Training forward from the plain blob on 16-rotation workgroups (train_block16.h: flow_train_forward16_kernel): no packed blob, exact fp32.
// Built for the layer kinds of the reference's training recipes (Moebius, Uncondition16Trans / UnconditionRot, Condition16Trans) in the
// forward direction, at most TR_MAX_LAYERS layers; the caller keeps every other flow on rnf_flow_forward_train.
// Floats of the activation buffer the training forward can leave for the backward sweep: one slot of (256 + 4K rounded up to 16) x 16 floats
// per conditioner layer and 16-rotation block (train_block16.h).
static int act_rows_for(int32_t K) { return b16::ACT_HEAD_ROWS + (4 * K + 15) / 16 * 16; }
extern "C" size_t rnf_train_acts_floats(int64_t n, int32_t n_conditioner_layers, int32_t segments) {
    if (n <= 0 || n_conditioner_layers <= 0 || segments <= 0) return 0;
    return (size_t)n_conditioner_layers * (size_t)((n + b16::SB - 1) / b16::SB) * (size_t)act_rows_for(segments) * 16;
}

extern "C" int rnf_flow_forward_train_plain(const float *rot, const float *feat, int64_t n, int32_t F, const float *plain, const int32_t *tdesc,
                                            int32_t n_layers, int32_t K, float *rot_out, float *ldj_out, float *states, float *acts,
                                            void *stream_v) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    if (n < 0) return fail("n=%lld is negative", (long long)n);
    if (n_layers < 1 || n_layers > TR_MAX_LAYERS) return fail("rnf_flow_forward_train_plain: n_layers=%d outside [1,%d]", n_layers, TR_MAX_LAYERS);
    if (K < 1 || K > 512) return fail("rnf_flow_forward_train_plain: 1..512 segments, got %d", K);
    if (F < 0) return fail("feature_dim %d is negative", F);
    if (n == 0) return 0;
    if (!rot || !plain || !tdesc || !rot_out || !ldj_out || !states) return fail("null pointer argument");
    b16::FwdArgs a;
    std::memset(&a, 0, sizeof(a));
    bool any_feature = false;
    for (int l = 0; l < n_layers; ++l) {
        const int32_t *d = tdesc + (size_t)l * 3;
        const int kind = d[0] & 15, orth = (d[0] >> 8) & 1;
        if ((d[0] & ~(15 | 256)) || (kind != RNF_KIND_MOBIUS && kind != RNF_KIND_AFFINE16 && kind != RNF_KIND_COND16))
            return fail("rnf_flow_forward_train_plain: layer %d (kind %d) is not built here; use rnf_flow_forward_train", l, d[0]);
        if (d[1] < 0 || d[1] > 5) return fail("layer %d: perm_row %d outside [0,5]", l, d[1]);
        if (d[2] < 0) return fail("layer %d: negative plain offset", l);
        if (kind == RNF_KIND_COND16 && F == 0) return fail("layer %d: a conditional affine layer needs a feature", l);
        any_feature = any_feature || (kind != RNF_KIND_AFFINE16 && F > 0);
        a.layers[l] = make_int2(kind | (d[1] << 4) | (orth << 8), d[2]);
    }
    if (any_feature && !feat) return fail("conditional layers but feature pointer is null");
    a.rot = rot; a.feature = F ? feat : nullptr; a.plain = plain; a.rot_out = rot_out; a.ldj_out = ldj_out; a.states = states;
    a.n = n; a.n_layers = n_layers; a.K = K; a.F = F;
    a.acts = acts; a.act_rows = act_rows_for(K);
    const size_t rows = (4 * (size_t)K + 63) / 64 * 64;
    const size_t lds_bytes = sizeof(float) * (b16::HEAD_FLOATS + rows * b16::LR);
    const long long nblocks = (n + b16::SB - 1) / b16::SB;
    const long long cap = (long long)device_cus() * 4;
    const dim3 grid((unsigned)(nblocks < cap ? nblocks : cap)), block(b16::WAVES * 64);
    if (F) {
        auto kern = b16::flow_train_forward16_kernel<true>;
        HIP_TRY(allow_lds(kern, lds_bytes));
        hipLaunchKernelGGL(kern, grid, block, lds_bytes, stream, a);
    } else {
        auto kern = b16::flow_train_forward16_kernel<false>;
        HIP_TRY(allow_lds(kern, lds_bytes));
        hipLaunchKernelGGL(kern, grid, block, lds_bytes, stream, a);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int rnf_pack_flow_device(const float *plain, const int32_t *pdesc, int32_t n_layers, int32_t K, int32_t F, int32_t prec,
                                    float *blob, int32_t *flags, void *stream_v) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    if (n_layers < 0 || n_layers > MAX_LAYERS) return fail("n_layers=%d outside [0,%d] (device packer)", n_layers, MAX_LAYERS);
    if (K <= 0) return fail("segments=%d must be positive", K);
    if (F < 0) return fail("feature_dim %d is negative", F);
    if (prec != RNF_PREC_FP32 && prec != RNF_PREC_F16X2) return fail("unknown precision %d", prec);
    if (n_layers == 0) return 0;
    if (!plain || !pdesc || !blob || !flags) return fail("null pointer argument");
    PackArgs a;
    std::memset(&a, 0, sizeof(a));
    a.plain = plain; a.blob = blob; a.flags = flags;
    a.n_layers = n_layers; a.K = K; a.F = F; a.Fp = (F + 7) / 8 * 8; a.prec = prec;
    a.equalise = (prec == RNF_PREC_F16X2 && equalise_allowed()) ? 1 : 0;
    a.feat_ms = g_feature_ms;
    for (int l = 0; l < n_layers; ++l) {
        const int32_t *d = pdesc + (size_t)l * 4;
        const int kind = d[0] & 15;
        if ((d[0] & ~(15 | 256)) || (kind != RNF_KIND_MOBIUS && kind != RNF_KIND_AFFINE16 && kind != RNF_KIND_COND16 && kind != RNF_KIND_GS9 &&
                                     kind != RNF_KIND_GS36 && kind != RNF_KIND_COND36 && !kind_is_cond9(kind) && !kind_is_side(kind)))
            return fail("layer %d: unknown kind %d", l, d[0]);
        if (d[1] < 0 || d[2] < 0 || d[2] % 4 || (d[3] >= 0 && d[3] % 4)) return fail("layer %d: bad offsets", l);
        if ((kind == RNF_KIND_COND16 || kind == RNF_KIND_COND36 || kind_is_cond9(kind)) && F == 0) return fail("layer %d: a conditional affine layer needs a feature", l);
        if (kind_has_mlp(kind) && (F > 0) != (d[3] >= 0)) return fail("layer %d: feature record offset does not match feature_dim", l);
    }
    const int pk_chunk = layer_chunk(PK_MAX_LAYERS);
    for (int base = 0; base < n_layers; base += pk_chunk) {            // the layer table travels as a kernel argument: PK_MAX_LAYERS per launch
        const int cnt = n_layers - base < pk_chunk ? n_layers - base : pk_chunk;
        for (int l = 0; l < cnt; ++l) {
            const int32_t *d = pdesc + (size_t)(base + l) * 4;
            a.layers[l] = PackLayer{d[0], d[1], d[2], d[3]};
        }
        a.n_layers = cnt;
        hipLaunchKernelGGL(pack_flow_kernel, dim3(8, cnt), dim3(256), 0, stream, a);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

extern "C" size_t rnf_plain_layer_floats(int32_t kind, int32_t segments, int32_t feature_dim) {
    if (kind == RNF_KIND_AFFINE16) return 16;
    if (kind == RNF_KIND_GS9) return 12;
    if (kind == RNF_KIND_GS36) return 36;
    if (kind_is_side(kind)) return 0;
    const size_t ni = (kind == RNF_KIND_MOBIUS ? 3 : 0) + (size_t)feature_dim,
                 no = kind == RNF_KIND_MOBIUS ? 4 * (size_t)segments : (kind_is_cond9(kind) ? 9 : (kind == RNF_KIND_COND36 ? 36 : 16));
    return 64 * ni + 64 + 3 * (4096 + 64) + no * 64 + no;
}

struct BackwardExtra {
    const float *side = nullptr;       // [n_side][n][16] per-sample matrices of the side layers
    float *side_grad = nullptr;        // [n_side][n][16] out: dL/d(matrix)
    const float *g_out_ext = nullptr;  // RNF_KIND_MLP_ONLY: dL/d(outputs) [n][NO]
    const float *acts = nullptr;       // activations saved by rnf_flow_forward_train_plain (16-rotation sweep only; else recomputed)
};
static int run_backward(const float *states, const float *rot_final, int dir, const float *feat, int64_t n, int32_t F, const float *plain,
                        const int32_t *tdesc, int32_t n_layers, int32_t K, const float *g_rot_out, const float *g_ldj, float *grads,
                        float *g_rot_in, float *g_feature, float *g_ldj_sum, void *stream_v, const BackwardExtra &x = BackwardExtra());

extern "C" int rnf_flow_backward(const float *states, const float *feat, int64_t n, int32_t F, const float *plain, const int32_t *tdesc,
                                 int32_t n_layers, int32_t K, const float *g_rot_out, const float *g_ldj, float *grads,
                                 float *g_rot_in, float *g_feature, float *g_ldj_sum, void *stream_v) {
    return run_backward(states, nullptr, 0, feat, n, F, plain, tdesc, n_layers, K, g_rot_out, g_ldj, grads, g_rot_in, g_feature, g_ldj_sum, stream_v);
}

// rnf_flow_backward with the conditioner activations that rnf_flow_forward_train_plain saved (same n, table and K): the 16-rotation sweep
// reads them back instead of recomputing every conditioner; the 64-rotation sweep of large batches ignores them.
extern "C" int rnf_flow_backward_saved(const float *states, const float *acts, const float *feat, int64_t n, int32_t F, const float *plain,
                                       const int32_t *tdesc, int32_t n_layers, int32_t K, const float *g_rot_out, const float *g_ldj, float *grads,
                                       float *g_rot_in, float *g_feature, float *g_ldj_sum, void *stream_v) {
    BackwardExtra x;
    x.acts = acts;
    for (int l = 0; acts && l < n_layers; ++l) {
        const int kind = tdesc[(size_t)l * 3] & 15;
        if (kind != RNF_KIND_MOBIUS && kind != RNF_KIND_AFFINE16 && kind != RNF_KIND_COND16)
            return fail("rnf_flow_backward_saved: layer %d (kind %d) has no saved activations (rnf_flow_forward_train_plain does not run it)", l, kind);
    }
    return run_backward(states, nullptr, 0, feat, n, F, plain, tdesc, n_layers, K, g_rot_out, g_ldj, grads, g_rot_in, g_feature, g_ldj_sum, stream_v, x);
}

extern "C" int rnf_flow_inverse_backward(const float *states, const float *rot_out, const float *feat, int64_t n, int32_t F, const float *plain,
                                         const int32_t *tdesc, int32_t n_layers, int32_t K, const float *g_rot_out, const float *g_ldj,
                                         float *grads, float *g_rot_in, float *g_feature, float *g_ldj_sum, void *stream_v) {
    if (n > 0 && !rot_out) return fail("rnf_flow_inverse_backward: the output rotations of the inverse pass are needed (they carry the roots)");
    return run_backward(states, rot_out, 1, feat, n, F, plain, tdesc, n_layers, K, g_rot_out, g_ldj, grads, g_rot_in, g_feature, g_ldj_sum, stream_v);
}

static int run_backward(const float *states, const float *rot_final, int dir, const float *feat, int64_t n, int32_t F, const float *plain,
                        const int32_t *tdesc, int32_t n_layers, int32_t K, const float *g_rot_out, const float *g_ldj, float *grads,
                        float *g_rot_in, float *g_feature, float *g_ldj_sum, void *stream_v, const BackwardExtra &x) {
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_v);
    if (n < 0) return fail("n=%lld is negative", (long long)n);
    if (n_layers < 0 || n_layers > MAX_LAYERS) return fail("n_layers=%d outside [0,%d] (training path)", n_layers, MAX_LAYERS);
    if (K < 1 || K > 512) return fail("training path supports 1..512 segments, got %d", K);
    if (F < 0) return fail("feature_dim %d is negative", F);
    if (n == 0) return 0;
    const bool mlp_only = n_layers == 1 && (tdesc[0] & 15) == RNF_KIND_MLP_ONLY;
    if (!plain || !tdesc || !g_ldj_sum) return fail("null pointer argument");
    if (!mlp_only && (!states || !g_ldj || !g_rot_in)) return fail("null pointer argument");
    TrainArgs a;
    std::memset(&a, 0, sizeof(a));
    bool rare = false;        // side / Gram-Schmidt / conditional 3x3 layers present: the instantiation that carries them (train_block16.h)
    std::vector<int2> table((size_t)n_layers);
    for (int l = 0; l < n_layers; ++l) {
        const int32_t *d = tdesc + (size_t)l * 3;
        const int kind = d[0] & 15, orth = (d[0] >> 8) & 1, aux = (d[0] >> 16) & 255;
        rare = rare || kind_is_side(kind) || kind == RNF_KIND_GS9 || kind == RNF_KIND_GS36 || kind == RNF_KIND_COND36 || kind_is_cond9(kind);
        if ((d[0] & ~(15 | 256 | (255 << 16))) ||
            (kind != RNF_KIND_MOBIUS && kind != RNF_KIND_AFFINE16 && kind != RNF_KIND_COND16 && kind != RNF_KIND_GS9 && kind != RNF_KIND_GS36 &&
             kind != RNF_KIND_COND36 && !kind_is_cond9(kind) && !kind_is_side(kind) && !(kind == RNF_KIND_MLP_ONLY && mlp_only)))
            return fail("layer %d: kind %d has no backward kernel", l, d[0]);
        if (kind_is_side(kind) && (!x.side || !x.side_grad)) return fail("layer %d is a side layer: side / side_grad buffers are needed (rnf_flow_backward_side)", l);
        if (kind == RNF_KIND_MLP_ONLY && (aux < 1 || aux > 64 || !x.g_out_ext || F <= 0)) return fail("rnf_cond_mlp_backward: 1..64 outputs, a feature and dL/d(outputs) are needed");
        if (d[1] < 0 || d[1] > 5) return fail("layer %d: perm_row %d outside [0,5]", l, d[1]);
        if (kind_has_mlp(kind) && F > 0 && !feat) return fail("conditional layer %d but feature pointer is null", l);
        if ((kind == RNF_KIND_COND16 || kind == RNF_KIND_COND36 || kind_is_cond9(kind)) && F == 0) return fail("layer %d: a conditional affine layer needs a feature", l);
        if (d[2] < 0) return fail("layer %d: negative plain offset", l);
        table[l] = make_int2(kind | (d[1] << 4) | (orth << 8) | (aux << 16), d[2]);
    }
    a.side = x.side; a.side_grad = x.side_grad; a.g_out_ext = x.g_out_ext;
    a.feature = F ? feat : nullptr; a.plain = plain; a.grads = grads; a.g_ldj = g_ldj;
    a.g_rot_in = g_rot_in; a.g_feature = F ? g_feature : nullptr;
    a.n = n; a.K = K; a.F = F;
    a.dir = dir;
#ifdef RNF_STAMPS
    {   // diagnostic build: RNF_TRAIN_STAMPS_PTR=<device address of 10 zeroed uint64> (tools/phase_stamps_train.py)
        const char *sp = std::getenv("RNF_TRAIN_STAMPS_PTR");
        a.stamps = sp ? reinterpret_cast<unsigned long long *>(std::strtoull(sp, nullptr, 0)) : nullptr;
    }
#endif
    const size_t rows = (4 * (size_t)K + 63) / 64 * 64;      // conditioner-output rows, padded to whole 64-row slabs (train_kernels.h)
    // Block size of the sweep: 16-rotation workgroups (train_block16.h: a quarter of the dependent matrix chain per workgroup, four times
    // as many workgroups, K up to 512) or 64-rotation ones (train_kernels.h: a quarter of the weight traffic and gradient atomics, K <= 64).
    const int blk_sel = train_block();
    const bool block16 = blk_sel == 16 || K > 64 || (blk_sel == 0 && n < kTrainBlock64From);
    if (blk_sel == 64 && K > 64) return fail("RNF_TRAIN_BLOCK=64: the 64-rotation backward kernel holds at most 64 segments, got %d", K);
    // The layer table travels as a kernel argument, TR_MAX_LAYERS entries per launch: a deeper stack is swept in chunks from the top, each
    // chunk starting from the rotation gradient the previous one left in g_rot_in (read at the start of a block, written at its end, by the
    // same workgroup) and reading its own slice of the saved states; dL/dldj is the same per-rotation value for every layer.
    const int tr_chunk = layer_chunk(TR_MAX_LAYERS);
    for (int hi = n_layers; hi > 0 || n_layers == 0; ) {
        const int lo = hi > tr_chunk ? hi - tr_chunk : 0;
        for (int l = lo; l < hi; ++l) a.layers[l - lo] = table[l];
        a.n_layers = hi - lo;
        a.states = states ? states + (size_t)lo * (size_t)n * 9 : nullptr;
        a.rot_final = hi == n_layers ? rot_final : states + (size_t)hi * (size_t)n * 9;      // (dir = 1: the output of the chunk's last position)
        a.g_rot_out = hi == n_layers ? g_rot_out : g_rot_in;
        a.g_ldj_sum = g_ldj_sum + lo;
        a.acts = block16 ? x.acts : nullptr;
        a.act_rows = act_rows_for(K);
        a.mlp_base = 0;
        for (int l = 0; l < lo; ++l) a.mlp_base += ((table[l].x & 15) == RNF_KIND_MOBIUS || (table[l].x & 15) == RNF_KIND_COND16) ? 1 : 0;
        if (block16) {
            const size_t lds_bytes = sizeof(float) * (b16::HEAD_FLOATS + rows * b16::LR);
            const long long nblocks = (n + b16::SB - 1) / b16::SB;
            const long long cap = (long long)device_cus() * 4;
            const dim3 grid((unsigned)(nblocks < cap ? nblocks : cap)), block(b16::WAVES * 64);
            auto launch16 = [&](auto kern) -> int {
                HIP_TRY(allow_lds(kern, lds_bytes));
                hipLaunchKernelGGL(kern, grid, block, lds_bytes, stream, a);
                return 0;
            };
            int rc;
            if (F) rc = rare ? launch16(b16::flow_train_backward16_kernel<true, true>) : launch16(b16::flow_train_backward16_kernel<true, false>);
            else rc = rare ? launch16(b16::flow_train_backward16_kernel<false, true>) : launch16(b16::flow_train_backward16_kernel<false, false>);
            if (rc) return rc;
        } else {
            const size_t lds_bytes = sizeof(float) * (TR_LDS_HEAD_FLOATS + rows * LROW);
            const long long nblocks = (n + 63) / 64;
            const int cap = device_cus() * 4;
            const dim3 grid((unsigned)(nblocks < cap ? nblocks : cap)), block(TR_WAVES * 64);
            if (F) {
                auto kern = flow_train_backward_kernel<true>;
                HIP_TRY(allow_lds(kern, lds_bytes));
                hipLaunchKernelGGL(kern, grid, block, lds_bytes, stream, a);
            } else {
                auto kern = flow_train_backward_kernel<false>;
                HIP_TRY(allow_lds(kern, lds_bytes));
                hipLaunchKernelGGL(kern, grid, block, lds_bytes, stream, a);
            }
        }
        HIP_TRY(hipGetLastError());
        if (a.n_layers) {
            hipLaunchKernelGGL(affine_logdet_grad_kernel, dim3((a.n_layers + 63) / 64), dim3(64), 0, stream, a);
            HIP_TRY(hipGetLastError());
        }
        hi = lo;
        if (n_layers == 0) break;
    }
    return 0;
}

// Training with side layers (Condition16TransLU / Condition9TransLU / ConditionRot): the pass that saves the layer inputs, and the
// backward sweep that also returns dL/d(per-sample matrix) -- the caller's autograd carries it through the reference's own tensor ops
// (einsum / torch.diag / torch.svd) into the conditioner networks, whose backward is rnf_cond_mlp_backward.
extern "C" int rnf_flow_train_side(int32_t dir, const float *rot, const float *feat, int64_t n, int32_t F, const float *side, const float *blob,
                                   const int32_t *desc, int32_t n_layers, int32_t K, float *rot_out, float *ldj_out, float *states, void *ws,
                                   size_t ws_bytes, void *stream) {
    if (dir != 0 && dir != 1) return fail("rnf_flow_train_side: dir=%d", (int)dir);
    if (n > 0 && (!states || !rot_out)) return fail("rnf_flow_train_side: states / rotation_out buffer is null");
    RunOpts o{dir, nullptr, nullptr, 0, nullptr, nullptr};
    o.states = states;
    o.side = side;
    return run_flow(rot, feat, n, F, blob, desc, n_layers, K, rot_out, ldj_out, ws, ws_bytes, stream, o);
}

extern "C" int rnf_flow_backward_side(int32_t dir, const float *states, const float *rot_out, const float *feat, int64_t n, int32_t F,
                                      const float *plain, const int32_t *tdesc, int32_t n_layers, int32_t K, const float *side, float *side_grad,
                                      const float *g_rot_out, const float *g_ldj, float *grads, float *g_rot_in, float *g_feature,
                                      float *g_ldj_sum, void *stream_v) {
    if (dir != 0 && dir != 1) return fail("rnf_flow_backward_side: dir=%d", (int)dir);
    if (dir == 1 && n > 0 && !rot_out) return fail("rnf_flow_backward_side: the output rotations of the inverse pass are needed (they carry the roots)");
    BackwardExtra x;
    x.side = side; x.side_grad = side_grad;
    static const float dummy = 0.f;
    return run_backward(states, dir ? rot_out : nullptr, dir, feat, n, F, plain ? plain : &dummy, tdesc, n_layers, K, g_rot_out, g_ldj, grads, g_rot_in,
                        g_feature, g_ldj_sum, stream_v, x);
}

// Backward of ONE ConditionalTransform(feature_dim -> n_out) evaluated on its own (rnf_cond_mlp_forward; flow/condition.py:24-30): plain =
// its parameters in reference order (fc_first.weight [64][F], .bias, layers.{1,3,5}.weight/.bias, fc_last.weight [n_out][64], .bias),
// g_out [n][n_out] -> grads (same layout as plain, ACCUMULATED into; nullptr: feature gradient only) and g_feature [n][F] (ACCUMULATED
// into, or nullptr).  One launch of the training backward kernel with the layer math replaced by the caller's dL/d(outputs).
extern "C" int rnf_cond_mlp_backward(const float *feat, int64_t n, int32_t F, const float *plain, int32_t n_out, const float *g_out, float *grads,
                                     float *g_feature, float *scratch1, void *stream_v) {
    if (!feat || !plain || !g_out || !scratch1) return fail("rnf_cond_mlp_backward: null pointer");
    if (n_out < 1 || n_out > 64) return fail("rnf_cond_mlp_backward: n_out=%d outside [1,64]", (int)n_out);
    const int32_t tdesc[3] = {RNF_KIND_MLP_ONLY | (n_out << 16), 0, 0};
    BackwardExtra x;
    x.g_out_ext = g_out;
    return run_backward(nullptr, nullptr, 0, feat, n, F, plain, tdesc, 1, 8, nullptr, nullptr, grads, nullptr, g_feature, scratch1, stream_v, x);
}

// Flow.inverse that also saves the rotation entering every iteration position of the inverse pass (position 0 = the last flow layer)
extern "C" int rnf_flow_inverse_train(const float *rot, const float *feat, int64_t n, int32_t F, const float *blob, const int32_t *desc,
                                      int32_t n_layers, int32_t K, float *rot_out, float *ldj_out, float *states, void *ws, size_t ws_bytes,
                                      void *stream) {
    if (!states && n > 0) return fail("rnf_flow_inverse_train: states pointer is null");
    if (!rot_out && n > 0) return fail("rnf_flow_inverse_train: rotation_out is needed by rnf_flow_inverse_backward");
    RunOpts o{1, nullptr, nullptr, 0, nullptr, nullptr};
    o.states = states;
    return run_flow(rot, feat, n, F, blob, desc, n_layers, K, rot_out, ldj_out, ws, ws_bytes, stream, o);
}

extern "C" int rnf_flow_inverse(const float *rot, const float *feat, int64_t n, int32_t F, const float *blob,
                                const int32_t *desc, int32_t n_layers, int32_t K, float *rot_out, float *ldj_out, void *ws,
                                size_t ws_bytes, void *stream) {
    RunOpts o{1, nullptr, nullptr, 0, nullptr, nullptr};
    return run_flow(rot, feat, n, F, blob, desc, n_layers, K, rot_out, ldj_out, ws, ws_bytes, stream, o);
}

extern "C" int rnf_flow_log_prob(const float *rot, const float *feat, int64_t n, int32_t F, const float *blob,
                                 const int32_t *desc, int32_t n_layers, int32_t K, const float *fisher_A,
                                 const float *fisher_c, int64_t fisher_B, float *rot_out, float *ldj_out, float *logp_out,
                                 double *sum_out, void *ws, size_t ws_bytes, void *stream) {
    if ((fisher_A == nullptr) != (fisher_c == nullptr)) return fail("fisher_A and fisher_c must both be given or both be null");
    RunOpts o{0, fisher_A, fisher_c, fisher_B, logp_out, sum_out};
    return run_flow(rot, feat, n, F, blob, desc, n_layers, K, rot_out, ldj_out, ws, ws_bytes, stream, o);
}

extern "C" int rnf_flow_log_prob_shared(const float *rot, const float *feat, int64_t n, int32_t F, int64_t feature_div, const float *blob,
                                        const int32_t *desc, int32_t n_layers, int32_t K, const float *fisher_A, const float *fisher_c,
                                        int64_t fisher_B, float *rot_out, float *ldj_out, float *logp_out, double *sum_out, void *ws,
                                        size_t ws_bytes, void *stream) {
    if ((fisher_A == nullptr) != (fisher_c == nullptr)) return fail("fisher_A and fisher_c must both be given or both be null");
    RunOpts o{0, fisher_A, fisher_c, fisher_B, logp_out, sum_out};
    o.feature_div = feature_div;
    return run_flow(rot, feat, n, F, blob, desc, n_layers, K, rot_out, ldj_out, ws, ws_bytes, stream, o);
}

// ------------------------------------------------------------------------------------------------------------
// small standalone kernels
// ------------------------------------------------------------------------------------------------------------
__global__ void fisher_log_prob_kernel(const float *rot, long long n, const float *A, const float *c, long long div, float *out) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const long long row = i / div;
        float tr = 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) tr = fmaf(rot[i * 9 + k], A[row * 9 + k], tr);
        out[i] = tr - c[row];
    }
}

extern "C" int rnf_fisher_log_prob(const float *rot, int64_t n, const float *A, const float *c, int64_t B, float *out, void *stream) {
    if (!rot || !A || !c || !out) return fail("rnf_fisher_log_prob: null pointer");
    if (B <= 0 || n % B) return fail("n=%lld not divisible by fisher rows B=%lld (utils/fisher.py:226)", (long long)n, (long long)B);
    if (n == 0) return 0;
    long long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(fisher_log_prob_kernel, dim3((int)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), rot, (long long)n, A, c, (long long)(n / B), out);
    HIP_TRY(hipGetLastError());
    return 0;
}

// eval_acc epilogue (agent.py:266-283, utils/utils.py:231-235): geodesic angle between an estimate and the closest of its K ground-truth
// rotations (K > 1: symmetric objects): acos(clip((max_k tr(E^T G_k) - 1) / 2, -1, 1)), one thread per estimate
__global__ void min_geodesic_kernel(const float *est, const float *gt, long long n, int k, float *out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float e[9];
    for (int c = 0; c < 9; ++c) e[c] = est[i * 9 + c];
    float best = -4.0f;
    for (int q = 0; q < k; ++q) {
        const float *g = gt + ((size_t)i * k + q) * 9;
        float tr = 0.f;
        for (int c = 0; c < 9; ++c) tr = fmaf(e[c], g[c], tr);
        best = fmaxf(best, tr);
    }
    out[i] = acosf(fminf(fmaxf((best - 1.0f) * 0.5f, -1.0f), 1.0f));
}

extern "C" int rnf_min_geodesic(const float *est, const float *gt, int64_t n, int32_t k, float *out, void *stream) {
    if (n < 0 || k <= 0) return fail("rnf_min_geodesic: n=%lld k=%d", (long long)n, k);
    if (n == 0) return 0;
    if (!est || !gt || !out) return fail("rnf_min_geodesic: null pointer");
    hipLaunchKernelGGL(min_geodesic_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), est, gt, (long long)n, k, out);
    HIP_TRY(hipGetLastError());
    return 0;
}

// log-constants of MatrixFisherN(A, norm_type) (utils/fisher.py:67-76,79-97): c = s0 + s1 + s2 + log norm with the PROPER singular values
// of A (fisher_math.h).  One thread per matrix, fp64, so that a per-sample A coming out of a network never goes through a host SVD and a
// device->host sync.  scratch (doubles): [0] = Q = sum_b |A_b|_F^2 (norm_type 0 is batch-coupled through it), [1] = W, [2 + 10 b ..] =
// per-row gradient accumulators of the backward pass.
__global__ void fisher_frobenius_kernel(const float *A, long long B, double *q) {
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < B * 9; i += (long long)gridDim.x * blockDim.x) acc += (double)A[i] * (double)A[i];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(q, acc);
}

__global__ void fisher_log_const_kernel(const float *A, long long B, int norm_type, const double *q, float *c_out) {
    const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double a[9];
    for (int k = 0; k < 9; ++k) a[k] = A[b * 9 + k];
    c_out[b] = (float)fisher_log_const(a, norm_type, norm_type == 0 ? q[0] : 0.0, nullptr);
}

static int fisher_const_launch(const float *A, int64_t B, int32_t norm_type, double *scratch, float *c_out, hipStream_t st) {
    if (norm_type == 0) {
        HIP_TRY(hipMemsetAsync(scratch, 0, 2 * sizeof(double), st));
        long long blocks = (B * 9 + 255) / 256;
        if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(fisher_frobenius_kernel, dim3((int)blocks), dim3(256), 0, st, A, (long long)B, scratch);
    }
    if (c_out) hipLaunchKernelGGL(fisher_log_const_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, st, A, (long long)B, norm_type, scratch, c_out);
    HIP_TRY(hipGetLastError());
    return 0;
}

// Proper SVD of every parameter matrix on the device (utils/fisher.py:53-76 proper_svd / proper_svd_N): U, V in SO(3) (vectors as columns,
// row-major), s with the smallest value signed by det A, and the Bingham parameters of the sampler lam = (0, 2(s1+s2), 2(s0+s2), 2(s0+s1))
// (utils/fisher.py:151-158).  fp64 Jacobi, one thread per matrix (fisher_math.h proper_svd3): MatrixFisherN._sample of a network-predicted
// A no longer goes through host LAPACK and a device->host copy per call.
__global__ void fisher_proper_svd_kernel(const float *A, long long B, float *U_out, float *V_out, float *s_out, float *lam_out) {
    const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double a[9], U[9], s[3], V[9];
    for (int k = 0; k < 9; ++k) a[k] = A[b * 9 + k];
    proper_svd3(a, U, s, V);
    for (int k = 0; k < 9; ++k) {
        if (U_out) U_out[b * 9 + k] = (float)U[k];
        if (V_out) V_out[b * 9 + k] = (float)V[k];
    }
    if (s_out) { s_out[b * 3] = (float)s[0]; s_out[b * 3 + 1] = (float)s[1]; s_out[b * 3 + 2] = (float)s[2]; }
    if (lam_out) {
        lam_out[b * 4] = 0.f;
        lam_out[b * 4 + 1] = (float)(2.0 * (s[1] + s[2]));
        lam_out[b * 4 + 2] = (float)(2.0 * (s[0] + s[2]));
        lam_out[b * 4 + 3] = (float)(2.0 * (s[0] + s[1]));
    }
}
extern "C" int rnf_fisher_proper_svd(const float *A, int64_t B, float *U_out, float *V_out, float *s_out, float *lam_out, void *stream) {
    if (B < 0) return fail("rnf_fisher_proper_svd: B=%lld", (long long)B);
    if (B == 0) return 0;
    if (!A) return fail("rnf_fisher_proper_svd: null pointer");
    hipLaunchKernelGGL(fisher_proper_svd_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), A, (long long)B,
                       U_out, V_out, s_out, lam_out);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int rnf_fisher_log_const(const float *A, int64_t B, float *c_out, void *stream) {
    if (!A || !c_out) return fail("rnf_fisher_log_const: null pointer");
    if (B <= 0) return fail("rnf_fisher_log_const: B=%lld", (long long)B);
    return fisher_const_launch(A, B, 1, nullptr, c_out, reinterpret_cast<hipStream_t>(stream));
}

// pytorch3d.transforms.matrix_to_quaternion (published 0.7.5 rule; call sites flow/squeezetrans.py:34, utils/fisher.py:243): real part first,
// the candidate with the largest |q_i|, denominators floored at 0.1 (so3_math.h rot_to_quat).  rot [n][9] row-major -> quat [n][4].
__global__ void matrix_to_quaternion_kernel(const float *rot, long long n, float *quat) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float *r = rot + i * 9;
        Rot R;
        R.c0 = v3f{r[0], r[3], r[6]}; R.c1 = v3f{r[1], r[4], r[7]}; R.c2 = v3f{r[2], r[5], r[8]};
        float q[4];
        rot_to_quat(R, q);
        quat[i * 4] = q[0]; quat[i * 4 + 1] = q[1]; quat[i * 4 + 2] = q[2]; quat[i * 4 + 3] = q[3];
    }
}
extern "C" int rnf_matrix_to_quaternion(const float *rot, int64_t n, float *quat, void *stream) {
    if (n < 0) return fail("rnf_matrix_to_quaternion: n=%lld", (long long)n);
    if (n == 0) return 0;
    if (!rot || !quat) return fail("rnf_matrix_to_quaternion: null pointer");
    long long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(matrix_to_quaternion_kernel, dim3((int)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), rot, (long long)n, quat);
    HIP_TRY(hipGetLastError());
    return 0;
}

// norm_type 2 (utils/fisher.py:98-101): Monte-Carlo estimate of the normaliser over `approx_num` uniform rotations -- norm = mean_k
// exp(tr(R_k^T A) - sum S), hence c = sum S + log norm.  The reference broadcasts [approx_num,3,3] * [N,3,3], i.e. it serves ONE matrix
// (N = 1).  Uniform rotations from normalised Gaussian quaternions on the counter-based Philox stream of the sampler (the reference uses
// pytorch3d.transforms.random_rotations on torch's generator: parity is statistical, error ~ 1/sqrt(approx_num)).
__global__ void fisher_mc_accum_kernel(const float *A, long long n_mc, unsigned long long seed, double *acc /* [0] = sum exp(tr - sumS) */) {
    double a[9], U[9], s[3], V[9];
    for (int k = 0; k < 9; ++k) a[k] = A[k];
    proper_svd3(a, U, s, V);
    const float sumS = (float)(s[0] + s[1] + s[2]);
    float af[9];
    for (int k = 0; k < 9; ++k) af[k] = A[k];
    const Philox rng{(unsigned)seed, (unsigned)(seed >> 32)};
    double part = 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_mc; i += (long long)gridDim.x * blockDim.x) {
        unsigned c0[4] = {(unsigned)i, (unsigned)(i >> 32), 0x4d43u, 0u};
        rng(c0);
        const float r0 = sqrtf(-2.0f * logf(u01(c0[0]))), r1 = sqrtf(-2.0f * logf(u01(c0[2])));
        float s0, k0, s1, k1;
        sincosf(kTwoPi * u01(c0[1]), &s0, &k0);
        sincosf(kTwoPi * u01(c0[3]), &s1, &k1);
        const float q[4] = {r0 * k0, r0 * s0, r1 * k1, r1 * s1};
        Rot R;
        quat_to_rot(q, q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3], R);
        const float tr = R.c0.x * af[0] + R.c1.x * af[1] + R.c2.x * af[2] + R.c0.y * af[3] + R.c1.y * af[4] + R.c2.y * af[5] + R.c0.z * af[6] +
                         R.c1.z * af[7] + R.c2.z * af[8];
        part += (double)expf(tr - sumS);
    }
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(acc, part);
}
__global__ void fisher_mc_final_kernel(const float *A, long long n_mc, const double *acc, float *c_out) {
    double a[9], U[9], s[3], V[9];
    for (int k = 0; k < 9; ++k) a[k] = A[k];
    proper_svd3(a, U, s, V);
    c_out[0] = (float)(s[0] + s[1] + s[2] + log(acc[0] / (double)n_mc));
}
extern "C" int rnf_fisher_log_const_mc(const float *A, int64_t B, int64_t approx_num, uint64_t seed, void *scratch, size_t scratch_bytes, float *c_out,
                                       void *stream) {
    if (!A || !c_out || !scratch) return fail("rnf_fisher_log_const_mc: null pointer");
    if (B != 1) return fail("rnf_fisher_log_const_mc: the reference's norm_type 2 broadcasts its random rotations against ONE matrix (got B=%lld)", (long long)B);
    if (approx_num <= 0) return fail("rnf_fisher_log_const_mc: approx_num=%lld must be positive", (long long)approx_num);
    if (scratch_bytes < sizeof(double)) return fail("rnf_fisher_log_const_mc: 8 bytes of scratch are needed");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    HIP_TRY(hipMemsetAsync(scratch, 0, sizeof(double), st));
    long long blocks = (approx_num + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(fisher_mc_accum_kernel, dim3((int)blocks), dim3(256), 0, st, A, (long long)approx_num, (unsigned long long)seed, static_cast<double *>(scratch));
    hipLaunchKernelGGL(fisher_mc_final_kernel, dim3(1), dim3(1), 0, st, A, (long long)approx_num, static_cast<const double *>(scratch), c_out);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" size_t rnf_fisher_scratch_bytes(int64_t B) { return (size_t)(2 + 10 * (B > 0 ? B : 0)) * sizeof(double); }

extern "C" int rnf_fisher_log_const_nt(const float *A, int64_t B, int32_t norm_type, void *scratch, size_t scratch_bytes, float *c_out, void *stream) {
    if (!A || !c_out) return fail("rnf_fisher_log_const_nt: null pointer");
    if (B <= 0) return fail("rnf_fisher_log_const_nt: B=%lld", (long long)B);
    if (norm_type != 0 && norm_type != 1)
        return fail("rnf_fisher_log_const_nt: norm_type=%d -- only the closed-form approximations 0 and 1 are built (utils/fisher.py:88-97)", (int)norm_type);
    if (norm_type == 0 && (!scratch || scratch_bytes < 2 * sizeof(double))) return fail("rnf_fisher_log_const_nt: norm_type 0 needs 16 bytes of scratch");
    return fisher_const_launch(A, B, norm_type, static_cast<double *>(scratch), c_out, reinterpret_cast<hipStream_t>(stream));
}

// d log p / dA (agent.py:57-65 keeps a predicted A in the graph):  g_A[b] = sum_{i in row b} g_i R_i  -  (sum_{i in row b} g_i) dc_b/dA_b
// (+ the batch coupling of norm_type 0 through Q).  Pass 1 accumulates T_b = sum g_i R_i and G_b = sum g_i in fp64: 256 consecutive
// samples per block step; a step inside one row reduces in the block and issues 10 atomics, a step that spans rows lets every sample
// add to its own row.
__global__ void fisher_grad_accum_kernel(const float *g, const float *rot, long long n, long long div, double *acc) {
    __shared__ double red[4][10];
    for (long long base = (long long)blockIdx.x * 256; base < n; base += (long long)gridDim.x * 256) {
        const long long i = base + threadIdx.x;
        const long long last = base + 255 < n ? base + 255 : n - 1;
        const bool one_row = base / div == last / div;                   // block uniform
        double v[10];
        const float gi = i < n ? g[i] : 0.f;
#pragma unroll
        for (int k = 0; k < 9; ++k) v[k] = i < n ? (double)gi * (double)rot[i * 9 + k] : 0.0;
        v[9] = gi;
        if (one_row) {
#pragma unroll
            for (int k = 0; k < 10; ++k)
                for (int o = 32; o > 0; o >>= 1) v[k] += __shfl_xor(v[k], o, 64);
            if ((threadIdx.x & 63) == 0)
                for (int k = 0; k < 10; ++k) red[threadIdx.x >> 6][k] = v[k];
            __syncthreads();
            if (threadIdx.x < 10) atomicAdd(acc + 10 * (base / div) + threadIdx.x, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
            __syncthreads();
        } else if (i < n) {
            double *dst = acc + 10 * (i / div);
#pragma unroll
            for (int k = 0; k < 10; ++k) atomicAdd(dst + k, v[k]);
        }
    }
}

// norm_type 0: W = sum_b G_b / D_b, D_b = 1 + Q/6 + det(A_b)/6  (the weight of dQ/dA_b' = 2 A_b' in every row's gradient)
__global__ void fisher_grad_w_kernel(const float *A, long long B, double *scratch) {
    double w = 0.0;
    for (long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x; b < B; b += (long long)gridDim.x * blockDim.x) {
        double a[9];
        for (int k = 0; k < 9; ++k) a[k] = A[b * 9 + k];
        w += scratch[2 + 10 * b + 9] / (1.0 + scratch[0] / 6.0 + det3d(a) / 6.0);
    }
    for (int o = 32; o > 0; o >>= 1) w += __shfl_xor(w, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(scratch + 1, w);
}

__global__ void fisher_grad_final_kernel(const float *A, long long B, int norm_type, const double *scratch, float *g_A) {
    const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double a[9], dc[9];
    for (int k = 0; k < 9; ++k) a[k] = A[b * 9 + k];
    fisher_log_const(a, norm_type, norm_type == 0 ? scratch[0] : 0.0, dc);
    const double *acc = scratch + 2 + 10 * b;
    for (int k = 0; k < 9; ++k) {
        double v = acc[k] - acc[9] * dc[k];
        if (norm_type == 0) v -= scratch[1] * a[k] / 3.0;
        g_A[b * 9 + k] = (float)v;
    }
}

extern "C" int rnf_fisher_log_prob_backward_param(const float *g_logp, const float *rot, int64_t n, const float *A, int64_t B, int32_t norm_type,
                                              void *scratch, size_t scratch_bytes, float *g_A, void *stream) {
    if (!g_logp || !rot || !A || !g_A || !scratch) return fail("rnf_fisher_log_prob_backward_param: null pointer");
    if (B <= 0 || n % B) return fail("n=%lld not divisible by fisher rows B=%lld (utils/fisher.py:226)", (long long)n, (long long)B);
    if (norm_type != 0 && norm_type != 1) return fail("rnf_fisher_log_prob_backward_param: norm_type=%d is not built", (int)norm_type);
    if (scratch_bytes < rnf_fisher_scratch_bytes(B)) return fail("rnf_fisher_log_prob_backward_param: scratch of %zu bytes, need %zu", scratch_bytes, rnf_fisher_scratch_bytes(B));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    double *sc = static_cast<double *>(scratch);
    HIP_TRY(hipMemsetAsync(sc, 0, rnf_fisher_scratch_bytes(B), st));
    if (norm_type == 0) {
        long long blocks = (B * 9 + 255) / 256;
        if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(fisher_frobenius_kernel, dim3((int)blocks), dim3(256), 0, st, A, (long long)B, sc);
    }
    if (n > 0) {
        long long blocks = (n + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(fisher_grad_accum_kernel, dim3((int)blocks), dim3(256), 0, st, g_logp, rot, (long long)n, (long long)(n / B), sc + 2);
    }
    if (norm_type == 0) {
        long long blocks = (B + 255) / 256;
        if (blocks > 1024) blocks = 1024;
        hipLaunchKernelGGL(fisher_grad_w_kernel, dim3((int)blocks), dim3(256), 0, st, A, (long long)B, sc);
    }
    hipLaunchKernelGGL(fisher_grad_final_kernel, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, st, A, (long long)B, norm_type, sc, g_A);
    HIP_TRY(hipGetLastError());
    return 0;
}

// d(tr(A^T R) - c)/dR = A: g_rot[i] = g_logp[i] * A[row(i)]
__global__ void fisher_log_prob_backward_kernel(const float *g_logp, long long n, const float *A, long long div, float *g_rot) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const long long row = i / div;
        const float g = g_logp[i];
#pragma unroll
        for (int k = 0; k < 9; ++k) g_rot[i * 9 + k] = g * A[row * 9 + k];
    }
}

extern "C" int rnf_fisher_log_prob_backward(const float *g_logp, int64_t n, const float *A, int64_t B, float *g_rot, void *stream) {
    if (!g_logp || !A || !g_rot) return fail("rnf_fisher_log_prob_backward: null pointer");
    if (B <= 0 || n % B) return fail("n=%lld not divisible by fisher rows B=%lld (utils/fisher.py:226)", (long long)n, (long long)B);
    if (n == 0) return 0;
    long long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(fisher_log_prob_backward_kernel, dim3((int)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), g_logp, (long long)n, A,
                       (long long)(n / B), g_rot);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int rnf_fisher_sample(const float *U, const float *V, const float *lam, int64_t B, int64_t n, uint64_t seed, float *out,
                                 int32_t *fail_flag_dev, void *stream) {
    if (!U || !V || !lam || !out) return fail("rnf_fisher_sample: null pointer");
    if (B <= 0 || n < 0) return fail("rnf_fisher_sample: B=%lld, n=%lld", (long long)B, (long long)n);
    if (n == 0) return 0;
    long long blocks = (B * n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(fisher_sample_kernel, dim3((int)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), U, V, lam,
                       (long long)B, (long long)n, (unsigned long long)seed, out, fail_flag_dev);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ConditionalTransform alone: y [n,3] -> [n,4K] in the reference's row order (bring-up / unit test of the MFMA chain)
template <int NWc, int PREC>
__global__ __launch_bounds__(NWc * 64) void conditioner_kernel(const float *y, long long n, const float *layer, int KT, int K, float *out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, h = lane >> 5;
    const long long ntiles = (n + NWc * 32 - 1) / (NWc * 32);
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long long sample = (tile * NWc + wave) * 32 + j;
        const bool valid = sample < n;
        float y0 = 0.f, y1 = 0.f, y2 = 0.f;
        if (valid) { y0 = y[sample * 3]; y1 = y[sample * 3 + 1]; y2 = y[sample * 3 + 2]; }
        __syncthreads();
        stage_floats(lds, layer, MOB_HEAD_FLOATS, tid, NWc * 64);
        __syncthreads();
        typename Mlp<PREC>::Act tt;
        Fair nofair{lds, wave, -1, 0};
        bool bad = false;
        Mlp<PREC>::head(lds, lane, h, y0, y1, y2, GFrag<false>{nullptr, false}, tt, nofair, bad);
        for (int tau = 0; tau < KT; ++tau) {
            __syncthreads();
            stage_floats(lds + MOB_LAST, layer + MOB_LAST + (size_t)tau * MOB_LAST_TILE_FLOATS, MOB_LAST_TILE_FLOATS, tid, NWc * 64);
            __syncthreads();
            f32x16 o = Mlp<PREC>::last(lds + MOB_LAST, lane, h, tt);
            if (valid) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int k = 8 * tau + 2 * g + h;
                        const int row = c == 0 ? k : K + 3 * k + (c - 1);
                        if (k < K) out[sample * 4 * K + row] = c == 0 ? S_UNSCALE * o[4 * g] : o[4 * g + c];   // (k >= K: pad rows; s: layout.h S_PRESCALE)
                    }
            }
        }
    }
}

extern "C" int rnf_conditioner_forward(const float *y, int64_t n, const float *layer, int32_t K, int32_t prec, float *out,
                                       void *stream) {
    if (K <= 0) return fail("segments=%d must be positive", K);
    if (prec != RNF_PREC_FP32 && prec != RNF_PREC_F16X2) return fail("unknown precision %d", prec);
    if (!y || !layer || !out) return fail("rnf_conditioner_forward: null pointer");
    if (n == 0) return 0;
    const size_t lds_bytes = sizeof(float) * (MOB_HEAD_FLOATS + MOB_LAST_TILE_FLOATS);
    long long ntiles = (n + NW * 32 - 1) / (NW * 32);
    int grid = (int)(ntiles < device_cus() ? ntiles : device_cus());
    if (prec) {
        auto kern = conditioner_kernel<NW, 1>;
        HIP_TRY(allow_lds(kern, lds_bytes));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds_bytes, reinterpret_cast<hipStream_t>(stream), y, (long long)n, layer, (K + 7) / 8, K, out);
    } else {
        auto kern = conditioner_kernel<NW, 0>;
        HIP_TRY(allow_lds(kern, lds_bytes));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds_bytes, reinterpret_cast<hipStream_t>(stream), y, (long long)n, layer, (K + 7) / 8, K, out);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}
