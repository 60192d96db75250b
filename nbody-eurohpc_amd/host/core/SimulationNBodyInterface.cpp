#include "core/SimulationNBodyInterface.hpp"

#include <limits>

// Interface.cpp:11-17: the bodies come from the allocator; the byte count adds three acceleration
// arrays to what the bodies report.
template <typename T>
SimulationNBodyInterface<T>::SimulationNBodyInterface(const BodiesAllocatorInterface<T> &allocator, const T soft)
    : allocator{allocator}, bodies{allocator.allocate_shared()}, dt(std::numeric_limits<T>::infinity()), soft(soft),
      flopsPerIte(0)
{
    allocatedBytes = bodies->getAllocatedBytes() + (bodies->getN() + bodies->getPadding()) * sizeof(T) * 3;
}

template <typename T> const std::shared_ptr<Bodies<T>> &SimulationNBodyInterface<T>::getBodies() const { return bodies; }
template <typename T> void SimulationNBodyInterface<T>::setDt(T dtVal) { dt = dtVal; }
template <typename T> const T SimulationNBodyInterface<T>::getDt() const { return dt; }
template <typename T> const T SimulationNBodyInterface<T>::getFlopsPerIte() const { return flopsPerIte; }
template <typename T> const T SimulationNBodyInterface<T>::getAllocatedBytes() const { return allocatedBytes; }

template class SimulationNBodyInterface<float>;
template class SimulationNBodyInterface<double>;
