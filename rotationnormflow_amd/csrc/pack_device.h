// pack_device.h -- build the kernel blob ON the device from the plain parameter blob (training: the parameters change every
// iteration, so the host packers of rnf_api.hip -- a device->host copy, a CPU permutation and a host->device copy per step --
// would dominate the step).  One launch, one workgroup column per layer; every output float is gathered from its source
// weight with the same index maps as the host packers (pack_w64 / pack_w64_h / pack_bias / pack_featproj*), and
// tests/test_gpu_grad.py checks the two blobs bit for bit.
#pragma once
#include <hip/hip_runtime.h>

#include "layout.h"
#include "equalize.h"

namespace rnf {

constexpr int PK_MAX_LAYERS = 200;
constexpr int PK_FLAG_HALF_RANGE = 1, PK_FLAG_SINGULAR = 2;

struct PackLayer {
    int kind;        // RNF_KIND_* | orthogonal << 8
    int plain_off;   // floats, into the plain blob
    int rec_off;     // floats, into the kernel blob
    int feat_off;    // floats, feature-projection record (or -1)
};

struct PackArgs {
    const float *plain;
    float *blob;
    int *flags;      // OR of PK_FLAG_* (device int, zeroed by the caller)
    int n_layers, K, F, Fp, prec;
    int equalise;    // 1: move every MLP to its canonical scaling before the fp16 split (equalize.h)
    double feat_ms;  // mean square of a feature entry assumed by the equalisation
    PackLayer layers[PK_MAX_LAYERS];
};

__device__ __forceinline__ float half_part(float w, int lo, float lo_scale, int *flags) {
    if (!(fabsf(w) < 65504.0f)) atomicOr(flags, PK_FLAG_HALF_RANGE);      // also catches NaN / inf
    const _Float16 wh = (_Float16)w;
    return lo ? (w - (float)wh) * lo_scale : (float)wh;
}

__device__ __forceinline__ float pack2(float a, float b) {           // two fp16 in one 32-bit word, first element in the low half
    const _Float16 ha = (_Float16)a, hb = (_Float16)b;
    const unsigned short ua = __builtin_bit_cast(unsigned short, ha), ub = __builtin_bit_cast(unsigned short, hb);
    return __uint_as_float((unsigned)ua | ((unsigned)ub << 16));
}

// float `r` of a [n_ot][...] weight image of 64-column rows; val(ot, i, col) -> the (already scaled) source value of row i of out tile
// ot, column col (0 for a padding row)
template <class ValFn>
__device__ __forceinline__ float w64_image(int r, int prec, ValFn val, int *flags) {
    if (!prec) {                                                       // [ot][tg 8][lane 64] float4
        const int ot = r >> 11, tg = (r >> 8) & 7, lane = (r >> 2) & 63, c = r & 3;
        return val(ot, lane & 31, 8 * tg + 4 * (lane >> 5) + c);
    }
    float v[2];                                                        // [ot][s 4][hi, lo][lane 64] 8 x fp16
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int e = 2 * r + q;
        const int ot = e >> 12, s = (e >> 10) & 3, lo = (e >> 9) & 1, lane = (e >> 3) & 63, j = e & 7;
        const float w = val(ot, lane & 31, 16 * s + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3));
        v[q] = half_part(w, lo, W_LO_SCALE, flags);
    }
    return pack2(v[0], v[1]);
}

// float `r` of the feature-projection weight image: rows of W (leading dimension ldw) from column col0, F real columns
__device__ __forceinline__ float featproj_image(int r, int prec, const float *W, int ldw, int col0, int F, int Fp, int *flags, const int *e0) {
    if (!prec) {                                                       // [ot][tg Fp/8][lane] float4
        const int ng = Fp / 8;
        const int ot = r / (ng * 256), rem = r % (ng * 256);
        const int tg = rem >> 8, lane = (rem >> 2) & 63, c = rem & 3;
        const int k = 8 * tg + 4 * (lane >> 5) + c;
        return k < F ? ldexpf(W[(size_t)(32 * ot + (lane & 31)) * ldw + col0 + k], e0[32 * ot + (lane & 31)]) : 0.f;
    }
    const int ns = (Fp + 15) / 16;                                     // [ot][s][hi, lo][lane] 8 x fp16
    float v[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int e = 2 * r + q;
        const int ot = e / (ns * 1024), rem = e % (ns * 1024);
        const int s = rem >> 10, lo = (rem >> 9) & 1, lane = (rem >> 3) & 63, j = rem & 7;
        const int k = 16 * s + 8 * (lane >> 5) + j;
        const float w = k < F ? ldexpf(W[(size_t)(32 * ot + (lane & 31)) * ldw + col0 + k], e0[32 * ot + (lane & 31)]) : 0.f;
        v[q] = half_part(w, lo, FEAT_LO_SCALE, flags);
    }
    return pack2(v[0], v[1]);
}

// bias image [ot][h][16]: element q is b[32 ot + rho(r, h)]
__device__ __forceinline__ int bias_row(int q) { return 32 * (q >> 5) + rho(q & 15, (q >> 4) & 1); }

// 4x4 inverse / determinant in double, Gauss-Jordan with partial pivoting (same steps as the host's inv4_double)
__device__ inline bool inv4_double_dev(const double *m, double *inv, double *det) {
    double a[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) { a[i][j] = m[4 * i + j]; a[i][4 + j] = (i == j); }
    double d = 1.0;
    for (int c = 0; c < 4; ++c) {
        int p = c;
        for (int r = c + 1; r < 4; ++r) if (fabs(a[r][c]) > fabs(a[p][c])) p = r;
        if (a[p][c] == 0.0) return false;
        if (p != c) { for (int j = 0; j < 8; ++j) { const double t = a[p][j]; a[p][j] = a[c][j]; a[c][j] = t; } d = -d; }
        d *= a[c][c];
        const double ip = 1.0 / a[c][c];
        for (int j = 0; j < 8; ++j) a[c][j] *= ip;
        for (int r = 0; r < 4; ++r) if (r != c) {
            const double f = a[r][c];
            for (int j = 0; j < 8; ++j) a[r][j] -= f * a[c][j];
        }
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) inv[4 * i + j] = a[i][4 + j];
    *det = d;
    return true;
}

__global__ __launch_bounds__(256) void pack_flow_kernel(const PackArgs args) {
    const PackLayer L = args.layers[blockIdx.y];
    const int kind = L.kind & 15;
    const float *P = args.plain + L.plain_off;
    float *out = args.blob + L.rec_off;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
    const int prec = args.prec, K = args.K, F = args.F, Fp = args.Fp;
    if (kind_is_side(kind)) {                                          // per-sample matrices come from the caller: an empty record
        if (tid < 4) out[tid] = 0.f;
        return;
    }
    if (kind == RNF_KIND_AFFINE16) {                                   // rnf_pack_affine16 / rnf_pack_rot16
        if (tid != 0) return;
        if ((L.kind >> 8) & 1) {
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) { out[4 * i + j] = P[4 * i + j]; out[17 + 4 * i + j] = P[4 * j + i]; }
            out[16] = 0.f; out[33] = 0.f; out[34] = 1.f; out[35] = 0.f;
            double m[16], mt[16];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) { m[4 * i + j] = P[4 * i + j]; mt[4 * i + j] = P[4 * j + i]; }
            affine16_table(m, 0.f, true, out + AFF_TABLE_FWD);
            affine16_table(mt, 0.f, true, out + AFF_TABLE_INV);
            return;
        }
        double m[16], inv[16], det = 0.0;
        for (int i = 0; i < 16; ++i) m[i] = P[i];
        if (!inv4_double_dev(m, inv, &det)) {
            atomicOr(args.flags, PK_FLAG_SINGULAR);
            for (int i = 0; i < 16; ++i) inv[i] = __builtin_nan("");
        }
        for (int i = 0; i < 16; ++i) { out[i] = P[i]; out[17 + i] = (float)inv[i]; }
        out[16] = (float)log(fabs(det));
        out[33] = (float)(-log(fabs(det)));
        out[34] = out[35] = 0.f;
        affine16_table(m, out[16], false, out + AFF_TABLE_FWD);
        affine16_table(inv, out[33], false, out + AFF_TABLE_INV);
        return;
    }
    if (kind == RNF_KIND_GS9) {                                        // rnf_pack_gs(n = 3): [M | M^-1], cofactor inverse in double
        if (tid != 0) return;
        double m[9], c[9];
        for (int i = 0; i < 9; ++i) m[i] = P[i];
        c[0] = m[4] * m[8] - m[5] * m[7]; c[1] = m[2] * m[7] - m[1] * m[8]; c[2] = m[1] * m[5] - m[2] * m[4];
        c[3] = m[5] * m[6] - m[3] * m[8]; c[4] = m[0] * m[8] - m[2] * m[6]; c[5] = m[2] * m[3] - m[0] * m[5];
        c[6] = m[3] * m[7] - m[4] * m[6]; c[7] = m[1] * m[6] - m[0] * m[7]; c[8] = m[0] * m[4] - m[1] * m[3];
        const double det = m[0] * c[0] + m[1] * c[3] + m[2] * c[6];
        if (det == 0.0) atomicOr(args.flags, PK_FLAG_SINGULAR);
        for (int i = 0; i < 9; ++i) { out[i] = P[i]; out[9 + i] = (float)(c[i] / det); }
        out[18] = out[19] = 0.f;
        return;
    }
    if (kind == RNF_KIND_GS36) {                                       // rnf_pack_gs(n = 6): [M | M^-1], Gauss-Jordan in double
        if (tid != 0) return;
        double a[6][12];
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j < 6; ++j) { a[i][j] = P[6 * i + j]; a[i][6 + j] = (i == j); }
        bool ok = true;
        for (int c = 0; c < 6 && ok; ++c) {
            int pr = c;
            for (int r = c + 1; r < 6; ++r) if (fabs(a[r][c]) > fabs(a[pr][c])) pr = r;
            if (a[pr][c] == 0.0) { ok = false; break; }
            if (pr != c) for (int j = 0; j < 12; ++j) { const double t = a[pr][j]; a[pr][j] = a[c][j]; a[c][j] = t; }
            const double ip = 1.0 / a[c][c];
            for (int j = 0; j < 12; ++j) a[c][j] *= ip;
            for (int r = 0; r < 6; ++r) if (r != c) {
                const double f = a[r][c];
                for (int j = 0; j < 12; ++j) a[r][j] -= f * a[c][j];
            }
        }
        if (!ok) atomicOr(args.flags, PK_FLAG_SINGULAR);
        for (int i = 0; i < (int)GS36_FLOATS; ++i) out[i] = 0.f;
        for (int i = 0; i < 36; ++i) { out[i] = P[i]; out[36 + i] = ok ? (float)a[i / 6][6 + i % 6] : __builtin_nanf(""); }
        return;
    }
    const bool mob = kind == RNF_KIND_MOBIUS;
    const int yo = mob ? 3 : 0, ni = yo + F, NO = mob ? 4 * K : (kind_is_cond9(kind) ? 9 : (kind == RNF_KIND_COND36 ? 36 : 16));
    const float *W0 = P, *b0 = W0 + 64 * ni;
    const float *hw[3], *hb[3];
    hw[0] = b0 + 64; hb[0] = hw[0] + 4096;
    hw[1] = hb[0] + 64; hb[1] = hw[1] + 4096;
    hw[2] = hb[1] + 64; hb[2] = hw[2] + 4096;
    const float *WL = hb[2] + 64, *bL = WL + (size_t)NO * 64;
    const int n_tiles = mob ? (K + 7) / 8 : (kind == RNF_KIND_COND36 ? 2 : 1);
    const int rec_floats = MOB_HEAD_FLOATS + n_tiles * MOB_LAST_TILE_FLOATS;
    // split precision: the layer's canonical scaling (equalize.h), the same per-unit functions the host packer evaluates, one unit per
    // thread; every workgroup of the layer computes the exponents for itself
    __shared__ double q_prev[64], q_cur[64], q_first[64];
    __shared__ int ex[3][64];
    {
        const int i = threadIdx.x;
        if (i < 64) { ex[0][i] = ex[1][i] = ex[2][i] = 0; }
        if (args.equalise) {
            if (i < 64) { q_first[i] = eq_q_first(W0 + (size_t)i * ni, ni, yo, b0[i], args.feat_ms); q_prev[i] = q_first[i]; }
            __syncthreads();
            for (int l = 0; l < 3; ++l) {
                if (i < 64) q_cur[i] = eq_q_hidden(hw[l] + (size_t)i * 64, q_prev, hb[l][i]);
                __syncthreads();
                if (i < 64) {
                    if (l < 2) ex[l + 1][i] = eq_exponent(q_cur[i]);
                    else ex[0][i] = eq_exponent(q_first[i] + q_cur[i]);
                    q_prev[i] = q_cur[i];
                }
                __syncthreads();
            }
        } else {
            __syncthreads();
        }
    }
    // reference row of packed fc_last row `row` of tile tau (layout.h): Moebius: segment k = 8 tau + 2g + h, component c;
    // Condition16Trans: M[2g + h][c], rows >= 16 are zero padding
    auto src_row = [&](int tau, int row) {
        const int g = row >> 3, h = (row >> 2) & 1, c = row & 3;
        if (mob) { const int k = 8 * tau + 2 * g + h; return k >= K ? -1 : (c == 0 ? k : K + 3 * k + (c - 1)); }
        if (kind == RNF_KIND_COND36) return tau == 0 ? row : (row < 4 ? 32 + row : -1);   // packed row P of tile 0 = output P, tile 1: 32..35
        const int o = 4 * (2 * g + h) + c;                                // Condition9*: output i sits where output i of the 4x4 sits
        return (row >= 16 || o >= NO) ? -1 : o;
    };
    for (int idx = tid; idx < rec_floats; idx += nth) {
        float v;
        if (idx < MOB_HID) {                                           // fc_first image: float2 per lane
            const int ot = idx >> 7, lane = (idx >> 1) & 63, e = idx & 1;
            const int o = 32 * ot + (lane & 31), h = lane >> 5;
            // (round 5: b0 sits in the bias slot for conditional layers too; the projection record's bias image is zero)
            if (!mob) v = (e == 1 && h) ? ldexpf(b0[o], ex[0][o]) : 0.f;   // Condition16Trans: no rotation inputs, x0 = projection + b0
            else if (e == 0) v = ldexpf(W0[(size_t)o * ni + h], ex[0][o]);
            else v = h ? ldexpf(b0[o], ex[0][o]) : ldexpf(W0[(size_t)o * ni + 2], ex[0][o]);
        } else if (idx < MOB_HB) {
            const int q = idx - MOB_HID, Lh = q >> 12;
            const float *W = hw[Lh];
            const int *eo = ex[(Lh + 1) % 3], *ei = ex[Lh];          // x1 <- x0, x2 <- x1, x3 (scaled like x0) <- x2
            v = w64_image(q & 4095, prec, [&](int ot, int i, int col) { return ldexpf(W[(size_t)(32 * ot + i) * 64 + col], eo[32 * ot + i] - ei[col]); },
                          args.flags);
        } else if (idx < MOB_HEAD_FLOATS) {
            const int q = idx - MOB_HB, Lh = q >> 6, row = bias_row(q & 63);
            v = ldexpf(hb[Lh][row], ex[(Lh + 1) % 3][row]);
        } else {
            const int q = idx - MOB_LAST, tau = q / MOB_LAST_TILE_FLOATS, r = q % MOB_LAST_TILE_FLOATS;
            // Moebius: the rows of the segment weights' pre-activations (source rows 0 .. K-1) are packed times log2 e (layout.h)
            if (r < MOB_LAST_TILE_BIAS) {
                v = w64_image(r, prec, [&](int, int i, int col) {
                    const int s = src_row(tau, i);
                    if (s < 0) return 0.f;
                    const float w = ldexpf(WL[(size_t)s * 64 + col], -ex[0][col]);
                    return (mob && s < K) ? w * S_PRESCALE : w;
                }, args.flags);
            } else {
                const int s = src_row(tau, bias_row(r - MOB_LAST_TILE_BIAS));
                v = s < 0 ? 0.f : ((mob && s < K) ? bL[s] * S_PRESCALE : bL[s]);
            }
        }
        out[idx] = v;
    }
    if (L.feat_off >= 0) {                                             // feature projection record (pack_featproj / pack_featproj_h)
        float *fo = args.blob + L.feat_off;
        const int w_floats = prec ? 2 * ((Fp + 15) / 16) * 512 : 2 * (Fp / 8) * 256;
        const int total = (int)featproj_packed_floats(Fp);
        for (int idx = tid; idx < total; idx += nth) {
            float v = 0.f;
            if (idx < w_floats) v = featproj_image(idx, prec, W0, ni, yo, F, Fp, args.flags, ex[0]);
            // (bias image: zero -- see the fc_first image above)
            fo[idx] = v;
        }
    }
}

}  // namespace rnf
