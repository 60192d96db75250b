// Factory indirection that lets a simulation be handed host Bodies or device-resident HIPBodies
// (reference src/common/core/BodiesAllocator.hpp:11-46, where the device flavour is CUDABodies).
#ifndef BODIES_ALLOCATOR_HPP_
#define BODIES_ALLOCATOR_HPP_

#include <memory>
#include <string>

#include "core/Bodies.hpp"
#include "core/HIPBodies.hpp"

template <typename T> class BodiesAllocatorInterface {
  public:
    virtual std::unique_ptr<Bodies<T>> allocate_unique() const = 0;
    virtual std::shared_ptr<Bodies<T>> allocate_shared() const = 0;
    virtual ~BodiesAllocatorInterface() = default;
};

template <typename T, typename B> class BodiesAllocatorOf : public BodiesAllocatorInterface<T> {
  public:
    // `scheme` is kept by reference like the reference does (BodiesAllocator.hpp:27): the allocator is
    // a short-lived stack object that only has to outlive the simulation's constructor.
    BodiesAllocatorOf(const unsigned long n, const std::string &scheme = "galaxy", const unsigned long randInit = 0)
        : n{n}, scheme{scheme}, randInit{randInit} {}
    std::unique_ptr<Bodies<T>> allocate_unique() const override { return std::make_unique<B>(n, scheme, randInit); }
    std::shared_ptr<Bodies<T>> allocate_shared() const override { return std::make_shared<B>(n, scheme, randInit); }

  private:
    const unsigned long n;
    const std::string &scheme;
    const unsigned long randInit;
};

template <typename T> using BodiesAllocator = BodiesAllocatorOf<T, Bodies<T>>;
// the drop-in hook: where the reference's driver builds a CUDABodiesAllocator (main.cpp:238)
template <typename T> using HIPBodiesAllocator = BodiesAllocatorOf<T, HIPBodies<T>>;

#endif
