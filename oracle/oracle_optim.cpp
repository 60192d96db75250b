// TEST INFRASTRUCTURE — restatement of the reference's cpu+optim acceleration loop (the parity oracle).
//
// Follows SimulationNBodyOptim.cpp:20-25 (zeroing the three temporaries) and :34-94 (for every pair
// j > i: one 1/sqrt, Newton's third law applied to both bodies).  The reference build (-O3
// -ffast-math, no -march) leaves this j loop SCALAR: one rsqrtss plus a Newton step per pair and
// strictly sequential sums.  This file is therefore compiled with -fno-tree-loop-vectorize (see
// oracle/Makefile) so that the summation order, and with it every bit of the result, is the
// reference's own: tests/test_oracle_vs_ref.py checks bit equality against oracle/_ref.
#include <cmath>
#include <cstring>
#include <vector>

#include "ftz.h"

extern "C" void oracle_accel_optim(unsigned long n, const float* qx_in, const float* qy_in, const float* qz_in,
                                   const float* m_in, float soft, float* ax_out, float* ay_out, float* az_out)
{
    const FlushDenormalsLikeReference ftz;
    std::vector<float> wx(n, 0.f), wy(n, 0.f), wz(n, 0.f);
    const float* __restrict__ qx = qx_in;
    const float* __restrict__ qy = qy_in;
    const float* __restrict__ qz = qz_in;
    const float* __restrict__ m = m_in;
    float* __restrict__ ax = wx.data();
    float* __restrict__ ay = wy.data();
    float* __restrict__ az = wz.data();
    const float soft2 = soft * soft;
    const float G = 6.67384e-11f;   // SimulationNBodyInterface.hpp:18

    for (unsigned long i = 0; i < n; ++i) {
        const float xi = qx[i], yi = qy[i], zi = qz[i], mi = m[i];
        float sx = ax[i], sy = ay[i], sz = az[i];
        for (unsigned long j = i + 1; j < n; ++j) {
            const float dx = qx[j] - xi;
            const float dy = qy[j] - yi;
            const float dz = qz[j] - zi;
            const float d2 = dx * dx + dy * dy + dz * dz + soft2;
            const float inv = 1.0f / std::sqrt(d2);
            const float inv3 = inv * inv * inv;
            const float scale = G * inv3;
            const float mj = m[j];
            const float on_i = scale * mj;   // pull of j on i
            const float on_j = scale * mi;   // pull of i on j (opposite sign)
            sx += on_i * dx; sy += on_i * dy; sz += on_i * dz;
            ax[j] -= on_j * dx; ay[j] -= on_j * dy; az[j] -= on_j * dz;
        }
        ax[i] = sx; ay[i] = sy; az[i] = sz;
    }
    std::memcpy(ax_out, ax, n * sizeof(float));
    std::memcpy(ay_out, ay, n * sizeof(float));
    std::memcpy(az_out, az, n * sizeof(float));
}
