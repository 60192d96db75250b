#include "implem/SimulationNBodyHIPTracking.hpp"

#include <cmath>

#include "murbhip.h"

template <typename T, typename Q>
SimulationNBodyHIPTracking<T, Q>::SimulationNBodyHIPTracking(const BodiesAllocatorInterface<T> &allocator,
                                                             std::shared_ptr<SimulationHistory<Q>> history, const T soft,
                                                             const bool leapfrog, const std::vector<int> &devices,
                                                             int exchange)
    : SimulationNBodyHIP<T>(allocator, soft, devices, exchange), history{history}
{
    if (!this->history) this->history = std::make_shared<SimulationHistory<Q>>();
    if (leapfrog)
        murbhipCheck(murbhip_set_option(this->hipBodiesPtr->getContext(), "integrator", 1), "murbhip_set_option(integrator)");
}

template <typename T, typename Q> void SimulationNBodyHIPTracking<T, Q>::computeMetrics()
{
    murbhip_ctx *ctx = this->hipBodiesPtr->getContext();
    double kinetic = 0, potential = 0, mom[10];
    murbhipCheck(murbhip_energy(ctx, &kinetic, &potential), "murbhip_energy");
    murbhipCheck(murbhip_moments(ctx, mom), "murbhip_moments");
    if (currentIteration >= history->getNumIterations()) history->setNumIterations(currentIteration + 1);
    history->setEnergyAt(currentIteration, (Q)(kinetic + potential));
    history->setAngMomentumAt(currentIteration, (Q)std::sqrt(mom[3] * mom[3] + mom[4] * mom[4] + mom[5] * mom[5]));
    const double mass = mom[9] != 0 ? mom[9] : 1;
    history->setDensityCenterAt(currentIteration, {(Q)(mom[6] / mass), (Q)(mom[7] / mass), (Q)(mom[8] / mass)});
}

template <typename T, typename Q> void SimulationNBodyHIPTracking<T, Q>::computeOneIteration()
{
    computeMetrics();
    SimulationNBodyHIP<T>::computeOneIteration();
    currentIteration++;
}

template class SimulationNBodyHIPTracking<float, double>;
template class SimulationNBodyHIPTracking<float, float>;
