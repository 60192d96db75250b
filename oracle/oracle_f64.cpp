// TEST INFRASTRUCTURE — fp64 direct-sum "truth" for the all-pairs acceleration.
//
// Same formula as the reference (SimulationNBodyOptim.cpp:60-76 / SimulationNBodyNaive.cpp:40-50):
//   a_i = sum_j G m_j (q_j - q_i) / (|q_j - q_i|^2 + soft^2)^(3/2)
// evaluated from the fp32 state in double precision, G taken as the reference's fp32 constant
// (SimulationNBodyInterface.hpp:18) widened to double.  Compiled WITHOUT -ffast-math.  It is the
// yardstick that tells the GPU kernel's rounding noise from cpu+optim's own (cpu+optim
// itself sits 1.2e-5 max / 3e-6 rms away from this at N = 30 000).
#include <cmath>

extern "C" {

// Accelerations (double) of bodies [i0, i1) due to all n bodies.  OpenMP over i.
void oracle_accel_f64(unsigned long n, unsigned long i0, unsigned long i1, const float* qx, const float* qy,
                      const float* qz, const float* m, float soft, double* ax, double* ay, double* az)
{
    const double G = (double)6.67384e-11f;
    const double soft2 = (double)soft * (double)soft;
#pragma omp parallel for schedule(static)
    for (unsigned long i = i0; i < i1; ++i) {
        const double xi = qx[i], yi = qy[i], zi = qz[i];
        double sx = 0.0, sy = 0.0, sz = 0.0;
        for (unsigned long j = 0; j < n; ++j) {
            const double dx = (double)qx[j] - xi, dy = (double)qy[j] - yi, dz = (double)qz[j] - zi;
            const double d2 = dx * dx + dy * dy + dz * dz + soft2;
            const double inv = 1.0 / std::sqrt(d2);
            const double f = G * (double)m[j] * inv * inv * inv;
            sx += f * dx; sy += f * dy; sz += f * dz;
        }
        ax[i - i0] = sx; ay[i - i0] = sy; az[i - i0] = sz;
    }
}

// Accelerations of an arbitrary subset of bodies (spot checks at N = 200k / 1M).
void oracle_accel_f64_subset(unsigned long n, unsigned long nsub, const unsigned long* idx, const float* qx,
                             const float* qy, const float* qz, const float* m, float soft, double* ax, double* ay,
                             double* az)
{
    const double G = (double)6.67384e-11f;
    const double soft2 = (double)soft * (double)soft;
#pragma omp parallel for schedule(static)
    for (unsigned long k = 0; k < nsub; ++k) {
        const unsigned long i = idx[k];
        const double xi = qx[i], yi = qy[i], zi = qz[i];
        double sx = 0.0, sy = 0.0, sz = 0.0;
        for (unsigned long j = 0; j < n; ++j) {
            const double dx = (double)qx[j] - xi, dy = (double)qy[j] - yi, dz = (double)qz[j] - zi;
            const double d2 = dx * dx + dy * dy + dz * dz + soft2;
            const double inv = 1.0 / std::sqrt(d2);
            const double f = G * (double)m[j] * inv * inv * inv;
            sx += f * dx; sy += f * dy; sz += f * dz;
        }
        ax[k] = sx; ay[k] = sy; az[k] = sz;
    }
}

// Mechanical energy with the reference's definitions (SimulationNBodyCUDAPropertyTracking.cu:217-304):
// kinetic = sum 1/2 m v^2, potential = -1/2 sum_i sum_{j != i} G m_i m_j / sqrt(r_ij^2 + soft^2).
void oracle_energy_f64(unsigned long n, const float* qx, const float* qy, const float* qz, const float* vx,
                       const float* vy, const float* vz, const float* m, float soft, double* kinetic, double* potential)
{
    const double G = (double)6.67384e-11f;
    const double soft2 = (double)soft * (double)soft;
    double ke = 0.0, pe = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : ke, pe)
    for (unsigned long i = 0; i < n; ++i) {
        ke += 0.5 * (double)m[i] * ((double)vx[i] * vx[i] + (double)vy[i] * vy[i] + (double)vz[i] * vz[i]);
        double phi = 0.0;
        for (unsigned long j = 0; j < n; ++j) {
            if (j == i) continue;
            const double dx = (double)qx[j] - qx[i], dy = (double)qy[j] - qy[i], dz = (double)qz[j] - qz[i];
            phi += G * (double)m[j] / std::sqrt(dx * dx + dy * dy + dz * dz + soft2);
        }
        pe -= 0.5 * (double)m[i] * phi;
    }
    *kinetic = ke;
    *potential = pe;
}

}  // extern "C"
