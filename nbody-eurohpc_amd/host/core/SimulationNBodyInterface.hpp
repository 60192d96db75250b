// The plugin boundary: every `--im` implementation derives from this class template and implements
// computeOneIteration() (reference src/common/core/SimulationNBodyInterface.hpp:16-88 declares it,
// SimulationNBodyInterface.cpp:11-33 defines it).  Member names, types and meaning are the
// reference's, so an implementation written against the reference compiles against this header;
// unlike the reference the template is defined here in full (header only).
#ifndef SIMULATION_N_BODY_INTERFACE_HPP_
#define SIMULATION_N_BODY_INTERFACE_HPP_

#include <limits>
#include <memory>
#include <string>

#include "core/Bodies.hpp"
#include "core/BodiesAllocator.hpp"

template <typename T> class SimulationNBodyInterface {
  protected:
    const T G = 6.67384e-11f;   // gravitational constant in m^3 kg^-1 s^-2 (reference Interface.hpp:18)
    const BodiesAllocatorInterface<T> &allocator;   // the driver's allocators are stack locals (main.cpp:210,238):
                                                     // only touch this during construction
    std::shared_ptr<Bodies<T>> bodies;   // all the bodies of the simulation
    T dt;                                // time step, "not set" = +inf until setDt()
    T soft;                              // softening length
    T flopsPerIte;                       // set by the implementation (20 n^2 for every reference path)
    T allocatedBytes;                    // bodies + three acceleration arrays (reference Interface.cpp:15-16)

    SimulationNBodyInterface(const BodiesAllocatorInterface<T> &bodiesAllocator, const T softening = 0.035f)
        : allocator{bodiesAllocator}, bodies{bodiesAllocator.allocate_shared()}, dt{std::numeric_limits<T>::infinity()},
          soft{softening}, flopsPerIte{0}
    {
        const unsigned long slots = bodies->getN() + bodies->getPadding();
        allocatedBytes = bodies->getAllocatedBytes() + slots * sizeof(T) * 3;
    }

  public:
    virtual ~SimulationNBodyInterface() = default;

    // one iteration of the simulation: accelerations, then positions and velocities
    virtual void computeOneIteration() = 0;

    const std::shared_ptr<Bodies<T>> &getBodies() const { return bodies; }
    void setDt(T dtVal) { dt = dtVal; }
    const T getDt() const { return dt; }
    const T getFlopsPerIte() const { return flopsPerIte; }
    const T getAllocatedBytes() const { return allocatedBytes; }
};

#endif
