// Device-resident bodies: the MI355X counterpart of the reference's CUDABodies
// (reference src/common/core/CUDABodies.hpp:24-65).  No HIP header is needed here: all device work
// goes through the C ABI of include/murbhip.h.
//
// Contract kept from the reference:
//   * construction = host initialisation (Bodies ctor) ; the device copy is made when a simulation
//     binds the bodies (the reference uploads in the CUDABodies ctor, CUDABodies.cu:4-10; here the
//     softening and G, which the context folds into the upload, are only known to the simulation);
//   * getDataSoA() returns CURRENT host data: a lazy device->host copy guarded by a dirty flag
//     (CUDABodies.cu:64-93); the vectors are never reallocated (the visualisation caches .data());
//   * updatePositionsAndVelocities(accSoA) = integrate on the device with host-supplied
//     accelerations (CUDABodies.cu:355-370); the AoS overload is not provided by the reference
//     either (CUDABodies.cu:373-376) and falls back to repacking here.
#ifndef HIP_BODIES_HPP_
#define HIP_BODIES_HPP_

#include <vector>

#include "core/Bodies.hpp"

struct murbhip_ctx;

template <typename T> class HIPBodies : public Bodies<T> {
  protected:
    murbhip_ctx *ctx = nullptr;
    mutable bool dataOnCPU = true;

  public:
    HIPBodies(const unsigned long n, const std::string &scheme = "galaxy", const unsigned long randInit = 0);
    ~HIPBodies() override;

    // Creates the device context on `devices` (one entry = single GPU, several = body-range
    // partition with a position exchange per step) and uploads the state.  exchange: 0 copies, 1 RCCL.
    void bindDevice(T soft, T G, const std::vector<int> &devices = {0}, int exchange = 1);
    murbhip_ctx *getContext() const { return ctx; }

    // The initial conditions again, this time generated ON THE DEVICE (murbhip_init_bodies: bit-identical to what the
    // constructor computed on the host for the same scheme and seed) — the bodies of a run then never cross PCIe.  Needs a
    // bound device; the host copy is refreshed lazily like after a step.  "galaxy" and "random" only.
    void initOnDevice(const std::string &scheme = "galaxy", const unsigned long randInit = 0);

    void invalidateDataSoA() { dataOnCPU = false; }
    const dataSoA_t<T> &getDataSoA() const override;
    const std::vector<dataAoS_t<T>> &getDataAoS() const override;

    void updatePositionsAndVelocities(const accSoA_t<T> &accelerations, T &dt) override;
    void updatePositionsAndVelocities(const std::vector<accAoS_t<T>> &accelerations, T &dt) override;
};

// print "<what>: <reason>" and exit(code): the reference's GPU error convention
// (SimulationNBodyCUDATileFullDevice.cu:10-17)
void murbhipCheck(int code, const char *what);

#endif
