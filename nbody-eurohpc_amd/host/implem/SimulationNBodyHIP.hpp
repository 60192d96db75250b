// `--im hip+tile` / `--im hip+tile+multi`: the MI355X implementation behind the reference's plugin
// interface.  It takes the place of SimulationNBodyCUDATileFullDevice
// (reference src/murb/implem/SimulationNBodyCUDATileFullDevice.hpp:11-35): device-resident bodies,
// one force launch + one integrate launch per iteration, nothing copied back unless asked.
#ifndef SIMULATION_N_BODY_HIP_HPP_
#define SIMULATION_N_BODY_HIP_HPP_

#include <memory>
#include <vector>

#include "core/HIPBodies.hpp"
#include "core/SimulationNBodyInterface.hpp"

template <typename T> class SimulationNBodyHIP : public SimulationNBodyInterface<T> {
  protected:
    std::shared_ptr<HIPBodies<T>> hipBodiesPtr;
    accSoA_t<T> accSoA;

  public:
    // `devices`: HIP ordinals to spread the bodies over ({0} = one GPU).  Needs a HIPBodiesAllocator.
    // Construction ends with an untimed device warm-up (warmUp below; MURBHIP_WARMUP_MS in the environment, default 50, 0 = none).
    SimulationNBodyHIP(const BodiesAllocatorInterface<T> &allocator, const T soft = 0.035f,
                       const std::vector<int> &devices = {0}, int exchange = 1);
    virtual ~SimulationNBodyHIP() = default;

    void computeOneIteration() override;   // enqueue only; the driver syncs (main.cpp:356-368)
    void synchronize();                    // that sync, for callers without HIP headers
    void warmUp(double milliseconds);      // force evaluations on the current state for about that long (state unchanged)
    const accSoA_t<T> &getAccSoA();        // accelerations of the last iteration (test hook)
};

#endif
