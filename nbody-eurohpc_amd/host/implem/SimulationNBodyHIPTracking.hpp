// `--im hip+tracking` and `--im hip+leapfrog`: the MI355X path with a per-iteration metrics history —
// the counterparts of the reference's gpu+tracking (SimulationNBodyCUDAPropertyTracking.hpp, energy of
// the state each iteration starts from, computeOneIteration() at .cu:121-133) and gpu+leapfrog
// (SimulationNBodyCUDALeapfrog.hpp: same history, leapfrog integrator).
//
// Filled per iteration: energy (kinetic + potential, reference definitions), |angular momentum| and the
// centre of mass — the reference reserves the last two columns but never computes them
// (SimulationNBodyCUDAPropertyTracking.cu:5-8).
#ifndef SIMULATION_N_BODY_HIP_TRACKING_HPP_
#define SIMULATION_N_BODY_HIP_TRACKING_HPP_

#include <memory>
#include <vector>

#include "core/SimulationHistory.hpp"
#include "implem/SimulationNBodyHIP.hpp"

template <typename T, typename Q = double> class SimulationNBodyHIPTracking : public SimulationNBodyHIP<T> {
  protected:
    std::shared_ptr<SimulationHistory<Q>> history;
    int currentIteration = 0;

  public:
    // leapfrog = true: kick-drift-kick instead of the reference's update (murbhip option "integrator")
    SimulationNBodyHIPTracking(const BodiesAllocatorInterface<T> &allocator, std::shared_ptr<SimulationHistory<Q>> history,
                               const T soft = 0.035f, const bool leapfrog = false, const std::vector<int> &devices = {0},
                               int exchange = 1);
    virtual ~SimulationNBodyHIPTracking() = default;

    void computeOneIteration() override;   // metrics of the current state -> history, then one step
    void computeMetrics();                 // fills row `currentIteration` (grows the history if needed)
    const std::shared_ptr<SimulationHistory<Q>> getHistory() const { return history; }
};

#endif
