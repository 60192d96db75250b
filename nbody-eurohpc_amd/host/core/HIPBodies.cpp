#include "core/HIPBodies.hpp"

#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "murbhip.h"

void murbhipCheck(int code, const char *what)
{
    if (code == 0) return;
    std::fprintf(stderr, "HIP path error %d in %s: %s\n", code, what, murbhip_error_string(code));
    std::exit(code < 0 ? -code : code);
}

template <typename T>
HIPBodies<T>::HIPBodies(const unsigned long n, const std::string &scheme, const unsigned long randInit)
    : Bodies<T>(n, scheme, randInit)
{
    static_assert(std::is_same<T, float>::value, "the device path is fp32, like the reference's driver (main.cpp:316)");
}

template <typename T> HIPBodies<T>::~HIPBodies() { murbhip_destroy(ctx); }

template <typename T> void HIPBodies<T>::bindDevice(T soft, T G, const std::vector<int> &devices, int exchange)
{
    if (ctx) { murbhip_destroy(ctx); ctx = nullptr; }
    if (devices.size() <= 1)
        murbhipCheck(murbhip_create(&ctx, this->n, soft, G, devices.empty() ? 0 : devices[0]), "murbhip_create");
    else {
        int rc = murbhip_create_sharded(&ctx, this->n, soft, G, (int)devices.size(), devices.data(), exchange);
        if (rc == MURBHIP_E_NO_RCCL && exchange == 1) {   // the in-process peer-copy exchange needs no library
            std::fprintf(stderr, "librccl could not be loaded: exchanging positions with peer copies instead\n");
            rc = murbhip_create_sharded(&ctx, this->n, soft, G, (int)devices.size(), devices.data(), 0);
        }
        murbhipCheck(rc, "murbhip_create_sharded");
    }
    const dataSoA_t<T> &d = this->dataSoA;
    murbhipCheck(murbhip_upload(ctx, d.qx.data(), d.qy.data(), d.qz.data(), d.vx.data(), d.vy.data(), d.vz.data(),
                                d.m.data()), "murbhip_upload");
    dataOnCPU = true;
}

template <typename T> void HIPBodies<T>::initOnDevice(const std::string &scheme, const unsigned long randInit)
{
    if (!ctx) {
        std::fprintf(stderr, "HIPBodies::initOnDevice needs a bound device (construct the simulation first)\n");
        std::exit(EXIT_FAILURE);
    }
    murbhipCheck(murbhip_init_bodies(ctx, scheme.c_str(), randInit), "murbhip_init_bodies");
    dataSoA_t<T> &d = this->dataSoA;   // masses and radii are not part of the lazy copy: fetch them once
    murbhipCheck(murbhip_download_mass(ctx, d.m.data(), d.r.data()), "murbhip_download_mass");
    dataOnCPU = false;
}

template <typename T> const dataSoA_t<T> &HIPBodies<T>::getDataSoA() const
{
    if (!dataOnCPU && ctx) {
        dataSoA_t<T> &d = this->dataSoA;   // masses and radii never change on the device
        murbhipCheck(murbhip_download_state(ctx, d.qx.data(), d.qy.data(), d.qz.data(), d.vx.data(), d.vy.data(),
                                            d.vz.data()), "murbhip_download_state");
        dataOnCPU = true;
    }
    return this->dataSoA;
}

template <typename T> const std::vector<dataAoS_t<T>> &HIPBodies<T>::getDataAoS() const
{
    const dataSoA_t<T> &d = getDataSoA();
    for (unsigned long i = 0; i < this->n; i++)
        this->dataAoS[i] = dataAoS_t<T>{d.qx[i], d.qy[i], d.qz[i], d.vx[i], d.vy[i], d.vz[i], d.m[i], d.r[i]};
    return this->dataAoS;
}

template <typename T> void HIPBodies<T>::updatePositionsAndVelocities(const accSoA_t<T> &a, T &dt)
{
    if (!ctx) { Bodies<T>::updatePositionsAndVelocities(a, dt); return; }
    invalidateDataSoA();
    murbhipCheck(murbhip_integrate_host_acc(ctx, a.ax.data(), a.ay.data(), a.az.data(), dt),
                 "murbhip_integrate_host_acc");
}

template <typename T> void HIPBodies<T>::updatePositionsAndVelocities(const std::vector<accAoS_t<T>> &a, T &dt)
{
    accSoA_t<T> s;
    s.ax.resize(this->n); s.ay.resize(this->n); s.az.resize(this->n);
    for (unsigned long i = 0; i < this->n; i++) { s.ax[i] = a[i].ax; s.ay[i] = a[i].ay; s.az[i] = a[i].az; }
    updatePositionsAndVelocities(s, dt);
}

template class HIPBodies<float>;
