// The plugin boundary: every `--im` implementation derives from this and implements
// computeOneIteration() (reference src/common/core/SimulationNBodyInterface.hpp:16-88; same member
// names and meaning so an implementation written for the reference compiles against this header).
#ifndef SIMULATION_N_BODY_INTERFACE_HPP_
#define SIMULATION_N_BODY_INTERFACE_HPP_

#include <memory>
#include <string>

#include "core/Bodies.hpp"
#include "core/BodiesAllocator.hpp"

template <typename T> class SimulationNBodyInterface {
  protected:
    const T G = 6.67384e-11f;   // gravitational constant, m^3 kg^-1 s^-2 (Interface.hpp:18)
    const BodiesAllocatorInterface<T> &allocator;   // only valid during construction (main.cpp:210,238)
    std::shared_ptr<Bodies<T>> bodies;
    T dt;
    T soft;
    T flopsPerIte;
    T allocatedBytes;

    SimulationNBodyInterface(const BodiesAllocatorInterface<T> &allocator, const T soft = 0.035f);

  public:
    virtual void computeOneIteration() = 0;
    virtual ~SimulationNBodyInterface() = default;

    const std::shared_ptr<Bodies<T>> &getBodies() const;
    void setDt(T dtVal);
    const T getDt() const;
    const T getFlopsPerIte() const;
    const T getAllocatedBytes() const;
};

#endif
