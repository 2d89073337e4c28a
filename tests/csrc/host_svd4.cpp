// TEST INFRASTRUCTURE: csrc/svd4_lapack.h compiled for the HOST (tests/test_svd4.py compares it with torch.svd).  Not part of librnf_hip.so.
#include "../../rotationnormflow_amd/csrc/svd4_lapack.h"

extern "C" {
// A [n][16] row-major -> rot = U^T V [n][16], singular values [n][4]; returns the number of matrices whose QR sweeps did not converge
int hs_utv(const float *A, float *rot, float *sv, int n) {
    int bad = 0;
    for (int i = 0; i < n; ++i) {
        float U[16], VT[16];
        if (!rnf::svd4::svd(A + 16 * i, U, sv + 4 * i, VT)) ++bad;
        for (int r = 0; r < 4; ++r)
            for (int c = 0; c < 4; ++c) {
                float a = 0.f;
                for (int k = 0; k < 4; ++k) a += U[4 * k + r] * VT[4 * c + k];
                rot[16 * i + 4 * r + c] = a;
            }
    }
    return bad;
}
}
