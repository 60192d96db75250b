// TEST INFRASTRUCTURE — the drop-in claim of INTEGRATION.md §A, executed.
//
// This file is compiled (oracle/Makefile, target `ref`) together with
//   * the REFERENCE's own core sources where they lie: src/common/core/{Bodies,BodiesAllocator,
//     SimulationNBodyInterface}.cpp and src/murb/implem/SimulationNBodyNaive.cpp, with the reference's
//     headers first on the include path, and
//   * the product's plugin files UNCHANGED: host/core/HIPBodies.cpp, host/implem/SimulationNBodyHIP.cpp
//     (their "core/Bodies.hpp", "core/SimulationNBodyInterface.hpp" then resolve to the reference's headers),
// and linked against libmurbhip.so.  It then runs what the reference's own hot-path test runs
// (src/test/implem/test_SimulationNBody.cpp:28-82): the reference's cpu+naive golden model against the target
// implementation behind the same SimulationNBodyInterface, positions compared after every iteration with the
// same relative tolerances — the target being SimulationNBodyHIP behind a HIPBodiesAllocator, i.e. exactly the
// substitution INTEGRATION.md describes for test_SimulationNBody.cpp:36-38.
//
// Output: one line per section and "dropin ok" / exit 1.  The binary lands in oracle/_ref/ (git-ignored, travels
// to the GPU box); tests/test_dropin_reference_tree.py builds it (CPU) and runs it (GPU).
#include <cmath>
#include <cstdio>
#include <memory>
#include <string>

#include "SimulationNBodyNaive.hpp"          // reference: src/murb/implem
#include "core/BodiesAllocator.hpp"          // reference: src/common/core
#include "core/HIPBodies.hpp"                // product:   nbody-eurohpc_amd/host/core
#include "implem/SimulationNBodyHIP.hpp"     // product:   nbody-eurohpc_amd/host/implem

// The device allocator a maintainer adds next to CUDABodiesAllocator (reference BodiesAllocator.hpp:33-46),
// as printed in INTEGRATION.md.
template <typename T> class HIPBodiesAllocator : public BodiesAllocatorInterface<T> {
  public:
    HIPBodiesAllocator(const unsigned long n, const std::string &scheme = "galaxy", const unsigned long randInit = 0)
        : n{n}, scheme{scheme}, randInit{randInit} {}
    std::unique_ptr<Bodies<T>> allocate_unique() const override { return std::make_unique<HIPBodies<T>>(n, scheme, randInit); }
    std::shared_ptr<Bodies<T>> allocate_shared() const override { return std::make_shared<HIPBodies<T>>(n, scheme, randInit); }

  private:
    const unsigned long n;
    const std::string &scheme;
    const unsigned long randInit;
};

// Catch::Matchers::WithinRel(target, eps): |a - b| <= eps * max(|a|, |b|)  (exact equality for eps == 0)
static bool within_rel(float a, float b, float eps) { return a == b || std::fabs(a - b) <= eps * std::fmax(std::fabs(a), std::fabs(b)); }

static bool section(const size_t n, const float soft, const float dt, const size_t nIte, const std::string &scheme,
                    const float eps)
{
    BodiesAllocator<float> naiveAllocator(n, scheme);
    SimulationNBodyNaive<float> simuRef(naiveAllocator, soft);
    simuRef.setDt(dt);
    HIPBodiesAllocator<float> targetAllocator(n, scheme);
    SimulationNBodyHIP<float> simuTest(targetAllocator, soft);
    simuTest.setDt(dt);
    unsigned long bad = 0;
    float worst = 0.f;
    for (size_t i = 0; i < nIte + 1; i++) {
        if (i > 0) {
            simuRef.computeOneIteration();
            simuTest.computeOneIteration();
        }
        const auto &r = simuRef.getBodies()->getDataSoA();
        const auto &t = simuTest.getBodies()->getDataSoA();   // lazy device -> host copy
        const float e = (i > 0) ? eps : 0.f;
        for (size_t b = 0; b < n; b++) {
            const float pr[3] = {r.qx[b], r.qy[b], r.qz[b]}, pt[3] = {t.qx[b], t.qy[b], t.qz[b]};
            for (int k = 0; k < 3; ++k) {
                if (!within_rel(pr[k], pt[k], e)) ++bad;
                const float d = std::fabs(pr[k] - pt[k]) / std::fmax(std::fmax(std::fabs(pr[k]), std::fabs(pt[k])), 1e-30f);
                if (d > worst) worst = d;
            }
        }
    }
    std::printf("fp32 - n=%zu - i=%zu - %s: flopsPerIte %.6g (reference naive %.6g), worst rel. position diff %.3e (eps %g), "
                "%lu outside\n", n, nIte, scheme.c_str(), (double)simuTest.getFlopsPerIte(), (double)simuRef.getFlopsPerIte(),
                (double)worst, (double)eps, bad);
    return bad == 0;
}

int main()
{
    bool ok = true;
    ok &= section(2048, 2e+08f, 3600.f, 1, "random", 1e-3f);   // test_SimulationNBody.cpp:76-77
    ok &= section(2049, 2e+08f, 3600.f, 3, "random", 1e-3f);
    ok &= section(2048, 2e+08f, 3600.f, 4, "galaxy", 1e-1f);   // :80-81
    ok &= section(2049, 2e+08f, 3600.f, 3, "galaxy", 1e-1f);
    std::puts(ok ? "dropin ok" : "dropin FAILED");
    return ok ? 0 : 1;
}
