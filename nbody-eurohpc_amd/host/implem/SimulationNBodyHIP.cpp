#include "implem/SimulationNBodyHIP.hpp"

#include <cstdio>
#include <cstdlib>

#include "murbhip.h"

template <typename T>
SimulationNBodyHIP<T>::SimulationNBodyHIP(const BodiesAllocatorInterface<T> &allocator, const T soft,
                                          const std::vector<int> &devices, int exchange)
    : SimulationNBodyInterface<T>(allocator, soft)
{
    const unsigned long n = this->getBodies()->getN();
    this->flopsPerIte = 20.f * (T)n * (T)n;   // the reference's count, every implementation (Optim.cpp:11)
    hipBodiesPtr = std::dynamic_pointer_cast<HIPBodies<T>>(this->bodies);
    if (!hipBodiesPtr) {
        std::fprintf(stderr, "SimulationNBodyHIP needs device bodies: construct it with a HIPBodiesAllocator\n");
        std::exit(EXIT_FAILURE);
    }
    hipBodiesPtr->bindDevice(this->soft, this->G, devices, exchange);
    accSoA.ax.resize(n); accSoA.ay.resize(n); accSoA.az.resize(n);
    // the device reaches its steady clock only after ~40 ms of work: get there before the caller's first
    // timed iteration.  Part of construction, like the reference's upload and GM precompute (FullDevice.cu:191-215).
    const char *ms = std::getenv("MURBHIP_WARMUP_MS");
    warmUp(ms && *ms ? std::atof(ms) : 50.0);
}

template <typename T> void SimulationNBodyHIP<T>::warmUp(double milliseconds)
{
    if (milliseconds > 0.0) murbhipCheck(murbhip_warmup(hipBodiesPtr->getContext(), milliseconds), "murbhip_warmup");
}

template <typename T> void SimulationNBodyHIP<T>::computeOneIteration()
{
    hipBodiesPtr->invalidateDataSoA();
    murbhipCheck(murbhip_step(hipBodiesPtr->getContext(), this->dt), "murbhip_step");
}

template <typename T> void SimulationNBodyHIP<T>::synchronize()
{
    murbhipCheck(murbhip_sync(hipBodiesPtr->getContext()), "murbhip_sync");
}

template <typename T> const accSoA_t<T> &SimulationNBodyHIP<T>::getAccSoA()
{
    murbhipCheck(murbhip_download_acc(hipBodiesPtr->getContext(), accSoA.ax.data(), accSoA.ay.data(),
                                      accSoA.az.data()), "murbhip_download_acc");
    return accSoA;
}

template class SimulationNBodyHIP<float>;
