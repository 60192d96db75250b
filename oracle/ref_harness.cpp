// TEST INFRASTRUCTURE — not part of the shipped product.
//
// C-ABI shim around the *real* reference classes, compiled together with the reference's own
// translation units where they lie under /root/reference (see oracle/Makefile, target `ref`).
// Nothing from /root/reference is copied: this file only calls the reference's public API
//   BodiesAllocator<float>(n, scheme)                 src/common/core/BodiesAllocator.hpp:18-30
//   SimulationNBody{Naive,Optim,SIMD,OpenMP}<float>   src/murb/implem/*.hpp
//   setDt / computeOneIteration / getBodies           src/common/core/SimulationNBodyInterface.hpp:45-87
//   getAccAoS                                          e.g. src/murb/implem/SimulationNBodyOptim.cpp:28-31
//   Bodies<float>::updatePositionsAndVelocities(accSoA) src/common/core/Bodies.cpp:280-288
// It is used (i) to pin oracle/murb_oracle.cpp against the reference and to generate
// tests/golden/*, in the build container only, and (ii) as bench.py's `cpu_baseline`
// (kind "reference": cpu+omp / cpu+simd / cpu+optim timed on the GPU box's host cores).
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "core/Bodies.hpp"
#include "core/BodiesAllocator.hpp"
#include "core/SimulationNBodyInterface.hpp"
#include "SimulationNBodyNaive.hpp"
#include "SimulationNBodyOptim.hpp"
#include "SimulationNBodySIMD.hpp"
#include "SimulationNBodyOpenMP.hpp"

#include "ftz.h"   // the murb executable runs flush-to-zero (linked with -ffast-math); so do these calls

namespace {
struct RefSim {
    std::string tag, scheme;
    SimulationNBodyInterface<float>* sim = nullptr;
    SimulationNBodyNaive<float>* naive = nullptr;
    SimulationNBodyOptim<float>* optim = nullptr;
    SimulationNBodySIMD<float>* simd = nullptr;
    SimulationNBodyOpenMP<float>* omp = nullptr;
};
}  // namespace

extern "C" {

void* murbref_create(const char* tag, unsigned long n, const char* scheme, float soft, float dt)
{
    auto* h = new RefSim;
    h->tag = tag;
    h->scheme = scheme;   // the allocator keeps a reference to this string
    BodiesAllocator<float> alloc(n, h->scheme);
    if (h->tag == "cpu+naive") h->sim = h->naive = new SimulationNBodyNaive<float>(alloc, soft);
    else if (h->tag == "cpu+optim") h->sim = h->optim = new SimulationNBodyOptim<float>(alloc, soft);
    else if (h->tag == "cpu+simd") h->sim = h->simd = new SimulationNBodySIMD<float>(alloc, soft);
    else if (h->tag == "cpu+omp") h->sim = h->omp = new SimulationNBodyOpenMP<float>(alloc, soft);
    else { delete h; return nullptr; }
    h->sim->setDt(dt);
    return h;
}

void murbref_destroy(void* p)
{
    auto* h = static_cast<RefSim*>(p);
    if (!h) return;
    delete h->sim;
    delete h;
}

unsigned long murbref_n(void* p) { return static_cast<RefSim*>(p)->sim->getBodies()->getN(); }
unsigned long murbref_padding(void* p) { return static_cast<RefSim*>(p)->sim->getBodies()->getPadding(); }
float murbref_flops_per_ite(void* p) { return static_cast<RefSim*>(p)->sim->getFlopsPerIte(); }
float murbref_allocated_bytes(void* p) { return static_cast<RefSim*>(p)->sim->getAllocatedBytes(); }

void murbref_step(void* p, int iterations)
{
    auto* h = static_cast<RefSim*>(p);
    const FlushDenormalsLikeReference ftz;
    for (int i = 0; i < iterations; ++i) h->sim->computeOneIteration();
}

// Copies n+padding entries of each SoA array (null pointers are skipped).
void murbref_get_state(void* p, float* qx, float* qy, float* qz, float* vx, float* vy, float* vz, float* m, float* r)
{
    auto* h = static_cast<RefSim*>(p);
    const auto& d = h->sim->getBodies()->getDataSoA();
    const size_t bytes = d.qx.size() * sizeof(float);
    if (qx) std::memcpy(qx, d.qx.data(), bytes);
    if (qy) std::memcpy(qy, d.qy.data(), bytes);
    if (qz) std::memcpy(qz, d.qz.data(), bytes);
    if (vx) std::memcpy(vx, d.vx.data(), bytes);
    if (vy) std::memcpy(vy, d.vy.data(), bytes);
    if (vz) std::memcpy(vz, d.vz.data(), bytes);
    if (m) std::memcpy(m, d.m.data(), bytes);
    if (r) std::memcpy(r, d.r.data(), bytes);
}

// Accelerations of the most recent iteration (n entries each).
int murbref_get_acc(void* p, float* ax, float* ay, float* az)
{
    auto* h = static_cast<RefSim*>(p);
    const std::vector<accAoS_t<float>>* a = nullptr;
    if (h->naive) a = &h->naive->getAccAoS();
    else if (h->optim) a = &h->optim->getAccAoS();
    else if (h->simd) a = &h->simd->getAccAoS();
    else if (h->omp) a = &h->omp->getAccAoS();
    if (!a) return -1;
    for (size_t i = 0; i < a->size(); ++i) { ax[i] = (*a)[i].ax; ay[i] = (*a)[i].ay; az[i] = (*a)[i].az; }
    return 0;
}

// The reference's integrator alone, driven with caller-supplied accelerations
// (the shape of src/test/implem/test_CUDABodies.cpp:42-75). State out: n entries each.
void murbref_integrate(unsigned long n, const char* scheme, const float* ax, const float* ay, const float* az,
                       float dt, int steps, float* qx, float* qy, float* qz, float* vx, float* vy, float* vz)
{
    const FlushDenormalsLikeReference ftz;
    Bodies<float> b(n, std::string(scheme));
    accSoA_t<float> acc;
    acc.ax.assign(ax, ax + n);
    acc.ay.assign(ay, ay + n);
    acc.az.assign(az, az + n);
    for (int s = 0; s < steps; ++s) b.updatePositionsAndVelocities(acc, dt);
    const auto& d = b.getDataSoA();
    std::memcpy(qx, d.qx.data(), n * sizeof(float));
    std::memcpy(qy, d.qy.data(), n * sizeof(float));
    std::memcpy(qz, d.qz.data(), n * sizeof(float));
    std::memcpy(vx, d.vx.data(), n * sizeof(float));
    std::memcpy(vy, d.vy.data(), n * sizeof(float));
    std::memcpy(vz, d.vz.data(), n * sizeof(float));
}

}  // extern "C"
