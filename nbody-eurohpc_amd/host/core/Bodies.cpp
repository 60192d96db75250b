// See Bodies.hpp.  The arithmetic of the three initial-condition schemes and of the integrator has
// to come out bit-identical to the reference's (same rand() sequence, same float/double mix), which
// tests/test_abi_and_host.py (test_product_initial_conditions_*) checks against tests/golden/ (vectors made with the compiled reference).
// This file is compiled with the reference's host flags (-O3 -ffast-math, no -march).
#include "core/Bodies.hpp"

#include <cassert>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <stdexcept>

namespace {
// rand() -> fraction, the three shapes the reference uses
template <typename T> inline T fracUp(int r) { return r / (T)RAND_MAX; }                               // Bodies.cpp:181,227
template <typename T> inline T fracDown(int r) { return (RAND_MAX - r) / (T)(RAND_MAX); }              // Bodies.cpp:184-186
template <typename T> inline T fracCentered(int r) { return (r - RAND_MAX / 2) / (T)(RAND_MAX / 2); } // Bodies.cpp:204-210
}  // namespace

template <typename T>
Bodies<T>::Bodies(const unsigned long n, const std::string &scheme, const unsigned long randInit)
    : n(n), padding(0), allocatedBytes(0)
{
    assert(n > 0);
    if (scheme == "galaxy") initGalaxy(randInit);
    else if (scheme == "random") initRandomly(randInit);
    else initMilkyWayAndromeda();
}

template <typename T> void Bodies<T>::allocateBuffers()
{
    const unsigned long tot = n + padding;
    for (std::vector<T> *v : {&dataSoA.m, &dataSoA.r, &dataSoA.qx, &dataSoA.qy, &dataSoA.qz, &dataSoA.vx,
                              &dataSoA.vy, &dataSoA.vz})
        v->resize(tot);
    dataAoS.resize(tot);
    allocatedBytes = tot * sizeof(T) * 8 * 2;   // SoA + AoS copies (Bodies.cpp:40)
}

template <typename T> const unsigned long Bodies<T>::getN() const { return n; }
template <typename T> const unsigned short Bodies<T>::getPadding() const { return padding; }
template <typename T> const dataSoA_t<T> &Bodies<T>::getDataSoA() const { return dataSoA; }
template <typename T> const std::vector<dataAoS_t<T>> &Bodies<T>::getDataAoS() const { return dataAoS; }
template <typename T> const float Bodies<T>::getAllocatedBytes() const { return allocatedBytes; }

template <typename T>
void Bodies<T>::setBody(const unsigned long &iBody, const T &mi, const T &ri, const T &qix, const T &qiy, const T &qiz,
                        const T &vix, const T &viy, const T &viz)
{
    dataSoA.m[iBody] = mi; dataSoA.r[iBody] = ri;
    dataSoA.qx[iBody] = qix; dataSoA.qy[iBody] = qiy; dataSoA.qz[iBody] = qiz;
    dataSoA.vx[iBody] = vix; dataSoA.vy[iBody] = viy; dataSoA.vz[iBody] = viz;
    dataAoS[iBody] = dataAoS_t<T>{qix, qiy, qiz, vix, viy, viz, mi, ri};
}

template <typename T> void Bodies<T>::computePadding()
{
    // Bodies.cpp:160-161: the vector count is formed in T, with mipp::N<T>() lanes per vector
    const T lanes = (T)(MURB_SIMD_BYTES / sizeof(T));
    const auto nVecs = std::ceil((T)n / lanes);
    padding = (nVecs * lanes) - n;
}

// One body drawn in the "random" box: six rand() calls in the order qx qy qz vx vy vz
// (Bodies.cpp:231-237; the padding zone of BOTH schemes uses the same draw, :201-213 / :244-256).
template <typename T> void Bodies<T>::drawBoxBody(unsigned long iBody, T mi, T ri)
{
    T qix = fracCentered<T>(rand()) * (5.0e8 * 1.33);
    T qiy = fracCentered<T>(rand()) * 5.0e8;
    T qiz = fracCentered<T>(rand()) * 5.0e8 - 10.0e8;
    T vix = fracCentered<T>(rand()) * 1.0e2;
    T viy = fracCentered<T>(rand()) * 1.0e2;
    T viz = fracCentered<T>(rand()) * 1.0e2;
    setBody(iBody, mi, ri, qix, qiy, qiz, vix, viy, viz);
}

template <typename T> void Bodies<T>::fillPaddingZone()
{
    for (unsigned long iBody = n; iBody < n + padding; iBody++) drawBoxBody(iBody, 0, 0);
}

// A heavy body at rest in the centre and n-1 light ones on tangential orbits (Bodies.cpp:158-214).
template <typename T> void Bodies<T>::initGalaxy(const unsigned long randInit)
{
    computePadding();
    allocateBuffers();
    srand(randInit);
    setBody(0, 2.0e24, 0, 0, 0, 0, 0, 0, 0);
    for (unsigned long iBody = 1; iBody < n; iBody++) {
        // four rand() calls per body: mass, horizontal angle, vertical angle, distance
        T mi = fracUp<T>(rand()) * 5e20;
        T ri = mi * 2.5e-15;
        T horizontalAngle = fracDown<T>(rand()) * 2.0 * M_PI;
        T verticalAngle = fracDown<T>(rand()) * 2.0 * M_PI;
        T distToCenter = fracDown<T>(rand()) * 1.0e8 + 1.0e8;
        T qix = std::cos(verticalAngle) * std::sin(horizontalAngle) * distToCenter;
        T qiy = std::sin(verticalAngle) * distToCenter;
        T qiz = std::cos(verticalAngle) * std::cos(horizontalAngle) * distToCenter;
        T vix = qiy * 4.0e-6;
        T viy = -qix * 4.0e-6;
        setBody(iBody, mi, ri, qix, qiy, qiz, vix, viy, 0);
    }
    fillPaddingZone();
}

// Uniform box (Bodies.cpp:217-257).
template <typename T> void Bodies<T>::initRandomly(const unsigned long randInit)
{
    computePadding();
    allocateBuffers();
    srand(randInit);
    for (unsigned long iBody = 0; iBody < n; iBody++) {
        T mi = fracUp<T>(rand()) * 5.0e21;
        T ri = mi * 0.5e-14;
        drawBoxBody(iBody, mi, ri);
    }
    fillPaddingZone();
}

// Two-galaxy collision read from "milkyway_andromeda.tab" in the working directory: one body per
// non-empty line, "m qx qy qz vx vy vz", rescaled per component galaxy (Bodies.cpp:83-153).  The
// data file is not part of the reference repository either; a missing file is a runtime_error.
template <typename T> void Bodies<T>::initMilkyWayAndromeda()
{
    const std::string path = "milkyway_andromeda.tab";
    std::ifstream in(path);
    if (!in.is_open()) throw std::runtime_error("cannot open " + path);
    std::vector<std::string> rows;
    for (std::string line; std::getline(in, line);)
        if (!line.empty()) rows.push_back(line);
    n = rows.size();
    allocateBuffers();

    // file order: disk MW, disk M31, bulge MW, bulge M31, halo MW, halo M31
    const unsigned long disk = 16384, bulge = 8192, halo = 16384;
    for (unsigned long iBody = 0; iBody < n; iBody++) {
        std::istringstream iss(rows[iBody]);
        T mi, qix, qiy, qiz, vix, viy, viz;
        iss >> mi >> qix >> qiy >> qiz >> vix >> viy >> viz;
        if (iss.fail()) throw std::runtime_error("parse error at line " + std::to_string(iBody + 1) + " of " + path);
        const bool milkyWay = iBody < disk || (iBody >= 2 * disk && iBody < 2 * disk + bulge) ||
                              (iBody >= 2 * (disk + bulge) && iBody < 2 * (disk + bulge) + halo);
        const double massUnit = milkyWay ? 4.5e10 : 9.4e10;   // solar masses
        const double lenUnit = milkyWay ? 4.0 : 6.0;          // kpc
        const double velUnit = milkyWay ? 220 : 260;          // km/s
        mi *= massUnit;
        qix *= lenUnit; qiy *= lenUnit; qiz *= lenUnit;
        vix *= velUnit; viy *= velUnit; viz *= velUnit;
        setBody(iBody, mi, 1e5, qix, qiy, qiz, vix, viy, viz);
    }
}

// Bodies.cpp:260-278.  `0.5` is a double literal on purpose: (v + a*dt*0.5)*dt and the sum with q run
// in double and are rounded to T once (the device integrator reproduces exactly this).
template <typename T>
void Bodies<T>::updatePositionAndVelocity(const unsigned long iBody, const T mi, const T ri, const T qix, const T qiy,
                                          const T qiz, const T vix, const T viy, const T viz, const T aix, const T aiy,
                                          const T aiz, T &dt)
{
    const T kx = aix * dt, ky = aiy * dt, kz = aiz * dt;
    const T qixNew = qix + (vix + kx * 0.5) * dt;
    const T qiyNew = qiy + (viy + ky * 0.5) * dt;
    const T qizNew = qiz + (viz + kz * 0.5) * dt;
    const T vixNew = vix + kx, viyNew = viy + ky, vizNew = viz + kz;
    setBody(iBody, mi, ri, qixNew, qiyNew, qizNew, vixNew, viyNew, vizNew);
}

template <typename T> void Bodies<T>::updatePositionsAndVelocities(const accSoA_t<T> &a, T &dt)
{
    for (unsigned long i = 0; i < n; i++)
        updatePositionAndVelocity(i, dataSoA.m[i], dataSoA.r[i], dataSoA.qx[i], dataSoA.qy[i], dataSoA.qz[i],
                                  dataSoA.vx[i], dataSoA.vy[i], dataSoA.vz[i], a.ax[i], a.ay[i], a.az[i], dt);
}

template <typename T> void Bodies<T>::updatePositionsAndVelocities(const std::vector<accAoS_t<T>> &a, T &dt)
{
    for (unsigned long i = 0; i < n; i++)
        updatePositionAndVelocity(i, dataSoA.m[i], dataSoA.r[i], dataSoA.qx[i], dataSoA.qy[i], dataSoA.qz[i],
                                  dataSoA.vx[i], dataSoA.vy[i], dataSoA.vz[i], a[i].ax, a[i].ay, a[i].az, dt);
}

template class Bodies<double>;
template class Bodies<float>;
