// TEST INFRASTRUCTURE — CPU oracle for the MUrB all-pairs force + integrate path.
//
// This is a from-scratch restatement (plain C++, C ABI for ctypes) of the reference's algorithm,
// used ONLY by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker.
// The product (libmurbhip.so, murb-hip) never links, loads or calls anything in oracle/.
//
// Parity status: PINNED.  Every function below is checked bit-for-bit (init, integrator) or to a
// stated tolerance (accelerations) against the real reference compiled from /root/reference
// (oracle/_ref/libmurbref.so, target `ref` of oracle/Makefile) by tests/test_oracle_vs_ref.py, and
// against the committed fixtures tests/golden/*.npz (made by tests/golden/make_golden.py from that
// same reference build) by tests/test_oracle_golden.py.
//
// Compile flags mirror the reference's host flags (CMakeLists.txt:6,128-131: C++20, -O3 -ffast-math,
// no -march) because the reference's results depend on them (rsqrtss+Newton for 1/sqrt, FTZ).
//
// Reference citations are relative to /root/reference.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ftz.h"

namespace {

constexpr float kG = 6.67384e-11f;   // src/common/core/SimulationNBodyInterface.hpp:18

// The three ways the reference maps rand() to a fraction.
inline float frac_up(int r) { return r / (float)RAND_MAX; }                        // Bodies.cpp:181,227
inline float frac_down(int r) { return (RAND_MAX - r) / (float)(RAND_MAX); }        // Bodies.cpp:184-186
inline float frac_centered(int r) { return (r - RAND_MAX / 2) / (float)(RAND_MAX / 2); }  // Bodies.cpp:204-210

struct State {
    float *qx, *qy, *qz, *vx, *vy, *vz, *m, *r;
    void put(unsigned long i, float mi, float ri, float x, float y, float z, float u, float v, float w) const
    {
        m[i] = mi; r[i] = ri; qx[i] = x; qy[i] = y; qz[i] = z; vx[i] = u; vy[i] = v; vz[i] = w;
    }
};

// A massless body drawn like the "random" scheme: what fills the SIMD padding zone
// (Bodies.cpp:201-213 and :244-256, identical in both schemes) — and, with a mass, the random scheme.
inline void draw_box_body(const State& s, unsigned long i, float mi, float ri)
{
    // six rand() calls in this order: qx qy qz vx vy vz
    float x = frac_centered(rand()) * (5.0e8 * 1.33);
    float y = frac_centered(rand()) * 5.0e8;
    float z = frac_centered(rand()) * 5.0e8 - 10.0e8;
    float u = frac_centered(rand()) * 1.0e2;
    float v = frac_centered(rand()) * 1.0e2;
    float w = frac_centered(rand()) * 1.0e2;
    s.put(i, mi, ri, x, y, z, u, v, w);
}

}  // namespace

extern "C" {

// Bodies.cpp:160-161 / :219-220 — padding to a multiple of the SIMD width, computed in float.
// The reference build has no -march, so mipp::N<float>() == 4 (SSE2); pass simd_width = 4 for it.
unsigned long oracle_padding(unsigned long n, int simd_width)
{
    const float w = (float)simd_width;
    const auto nvecs = std::ceil((float)n / w);
    return (unsigned long)((nvecs * w) - n);
}

// Bodies.cpp:14-25 (scheme dispatch), :158-214 (galaxy), :217-257 (random).
// Arrays hold n + oracle_padding(n, simd_width) entries.
void oracle_init(unsigned long n, const char* scheme, unsigned long seed, int simd_width, float* qx, float* qy,
                 float* qz, float* vx, float* vy, float* vz, float* m, float* r)
{
    const State s{qx, qy, qz, vx, vy, vz, m, r};
    const unsigned long pad = oracle_padding(n, simd_width);
    const bool galaxy = std::strcmp(scheme, "galaxy") == 0;
    srand((unsigned)seed);
    for (unsigned long i = 0; i < n; ++i) {
        if (!galaxy) {   // Bodies.cpp:226-240
            float mi = frac_up(rand()) * 5.0e21;
            float ri = mi * 0.5e-14;
            draw_box_body(s, i, mi, ri);
        } else if (i == 0) {   // the central heavy body, Bodies.cpp:170-179
            s.put(0, 2.0e24, 0.0f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f);
        } else {   // Bodies.cpp:181-194; four rand() calls: mass, horizontal, vertical, distance
            float mi = frac_up(rand()) * 5e20;
            float ri = mi * 2.5e-15;
            float hang = frac_down(rand()) * 2.0 * M_PI;
            float vang = frac_down(rand()) * 2.0 * M_PI;
            float dist = frac_down(rand()) * 1.0e8 + 1.0e8;
            float x = std::cos(vang) * std::sin(hang) * dist;
            float y = std::sin(vang) * dist;
            float z = std::cos(vang) * std::cos(hang) * dist;
            float u = y * 4.0e-6;
            float v = -x * 4.0e-6;
            s.put(i, mi, ri, x, y, z, u, v, 0.f);
        }
    }
    for (unsigned long i = n; i < n + pad; ++i) draw_box_body(s, i, 0.f, 0.f);
}

// cpu+optim lives in oracle_optim.cpp (own translation unit: it needs -fno-tree-loop-vectorize).
void oracle_accel_optim(unsigned long n, const float* qx, const float* qy, const float* qz, const float* m,
                        float soft, float* ax, float* ay, float* az);

// cpu+naive: SimulationNBodyNaive.cpp:34-53 (full N^2, pow(., 3/2)); the reference tests' golden model.
void oracle_accel_naive(unsigned long n, const float* qx, const float* qy, const float* qz, const float* m,
                        float soft, float* ax, float* ay, float* az)
{
    const FlushDenormalsLikeReference ftz;
    for (unsigned long i = 0; i < n; ++i) {
        float sx = 0.f, sy = 0.f, sz = 0.f;
        for (unsigned long j = 0; j < n; ++j) {
            const float dx = qx[j] - qx[i];
            const float dy = qy[j] - qy[i];
            const float dz = qz[j] - qz[i];
            const float d2 = std::pow(dx, 2) + std::pow(dy, 2) + std::pow(dz, 2);
            const float s2 = std::pow(soft, 2);
            const float k = kG * m[j] / std::pow(d2 + s2, 3.f / 2.f);
            sx += k * dx; sy += k * dy; sz += k * dz;
        }
        ax[i] = sx; ay[i] = sy; az[i] = sz;
    }
}

// Bodies.cpp:260-278 via :280-298 — the position/velocity update.  The `0.5` literal is a double,
// so (v + a*dt*0.5)*dt and the add to q run in fp64 and round to fp32 once, on the store.
void oracle_integrate(unsigned long n, float* qx, float* qy, float* qz, float* vx, float* vy, float* vz,
                      const float* ax, const float* ay, const float* az, float dt)
{
    const FlushDenormalsLikeReference ftz;
    for (unsigned long i = 0; i < n; ++i) {
        const float kx = ax[i] * dt, ky = ay[i] * dt, kz = az[i] * dt;
        const float nx = qx[i] + (vx[i] + kx * 0.5) * dt;
        const float ny = qy[i] + (vy[i] + ky * 0.5) * dt;
        const float nz = qz[i] + (vz[i] + kz * 0.5) * dt;
        const float nu = vx[i] + kx, nv = vy[i] + ky, nw = vz[i] + kz;
        qx[i] = nx; qy[i] = ny; qz[i] = nz;
        vx[i] = nu; vy[i] = nv; vz[i] = nw;
    }
}

// SimulationNBodyOptim.cpp:97-102 / SimulationNBodyNaive.cpp:56-61 — `iterations` whole steps.
// variant: 0 = cpu+optim, 1 = cpu+naive.  If acc_out != null it receives the last step's accelerations
// as three consecutive blocks of n floats.
void oracle_simulate(int variant, unsigned long n, int iterations, float soft, float dt, float* qx, float* qy,
                     float* qz, float* vx, float* vy, float* vz, const float* m, float* acc_out)
{
    std::vector<float> a(3 * n);
    float *ax = a.data(), *ay = ax + n, *az = ay + n;
    for (int it = 0; it < iterations; ++it) {
        if (variant == 0) oracle_accel_optim(n, qx, qy, qz, m, soft, ax, ay, az);
        else oracle_accel_naive(n, qx, qy, qz, m, soft, ax, ay, az);
        oracle_integrate(n, qx, qy, qz, vx, vy, vz, ax, ay, az, dt);
    }
    if (acc_out) std::memcpy(acc_out, a.data(), 3 * n * sizeof(float));
}

// Kick-drift-kick leapfrog as the reference STATES it (CUDABodies.cu:172-178):
//     v_{n+1/2} = v_n + a_n dt/2 ;  x_{n+1} = x_n + v_{n+1/2} dt ;  v_{n+1} = v_{n+1/2} + a_{n+1} dt/2
// in the one-force-per-step order of its own comment (:196-207), but with every a_n evaluated at x_n and
// the closing half kick applied.  (The reference's kernels do neither: devLeapfrogMiddle publishes x_n
// only after the force of iteration n has been taken at the previous positions, :262-268 with
// SimulationNBodyCUDALeapfrog.cu:127-133, and devLeapfrogLast drops the last kick, :316-319 — so there
// is no reference output to pin this against: PARITY UNPINNED, checked against this restatement and
// through energy conservation only.)  fp32 kicks, the drift with the reference integrator's fp64
// intermediate (Bodies.cpp:266-268); accelerations from the cpu+optim pair loop.
void oracle_leapfrog(unsigned long n, int iterations, float soft, float dt, float* qx, float* qy, float* qz, float* vx,
                     float* vy, float* vz, const float* m)
{
    if (iterations <= 0) return;
    std::vector<float> a(3 * n);
    float *ax = a.data(), *ay = ax + n, *az = ay + n;
    for (int it = 0; it < iterations; ++it) {
        oracle_accel_optim(n, qx, qy, qz, m, soft, ax, ay, az);
        const FlushDenormalsLikeReference ftz;
        const float h = it == 0 ? 0.5f * dt : 0.5f * (dt + dt);   // closing half of the last kick + opening half of this one
        for (unsigned long i = 0; i < n; ++i) {
            const float kx = ax[i] * h, ky = ay[i] * h, kz = az[i] * h;
            vx[i] = vx[i] + kx; vy[i] = vy[i] + ky; vz[i] = vz[i] + kz;
            qx[i] = (float)((double)qx[i] + (double)vx[i] * (double)dt);
            qy[i] = (float)((double)qy[i] + (double)vy[i] * (double)dt);
            qz[i] = (float)((double)qz[i] + (double)vz[i] * (double)dt);
        }
    }
    oracle_accel_optim(n, qx, qy, qz, m, soft, ax, ay, az);
    const FlushDenormalsLikeReference ftz;
    const float h = 0.5f * dt;
    for (unsigned long i = 0; i < n; ++i) {
        const float kx = ax[i] * h, ky = ay[i] * h, kz = az[i] * h;
        vx[i] = vx[i] + kx; vy[i] = vy[i] + ky; vz[i] = vz[i] + kz;
    }
}

// fp32 full-N^2 direct sum for i in [i0, i1) in the device twin's operation order
// (SimulationNBodyCUDATileFullDevice.cu:110-137: GM_j precomputed, FMA-shaped distance, f = GM_j*inv^3).
// Used by the world_size>1 tests: each rank evaluates only its slice of i.
void oracle_accel_slice_f32(unsigned long n, unsigned long i0, unsigned long i1, const float* qx, const float* qy,
                            const float* qz, const float* m, float soft, float* ax, float* ay, float* az)
{
    const float soft2 = soft * soft;
#pragma omp parallel for schedule(static)
    for (unsigned long i = i0; i < i1; ++i) {
        float sx = 0.f, sy = 0.f, sz = 0.f;
        for (unsigned long j = 0; j < n; ++j) {
            const float dx = qx[j] - qx[i], dy = qy[j] - qy[i], dz = qz[j] - qz[i];
            const float d2 = dx * dx + dy * dy + dz * dz + soft2;
            const float inv = 1.0f / std::sqrt(d2);
            const float f = (kG * m[j]) * (inv * inv * inv);
            sx += f * dx; sy += f * dy; sz += f * dz;
        }
        ax[i - i0] = sx; ay[i - i0] = sy; az[i - i0] = sz;
    }
}

}  // extern "C"
