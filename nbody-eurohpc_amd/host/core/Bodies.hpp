// Host-side body storage, API-compatible with the reference's Bodies<T>
// (reference src/common/core/Bodies.hpp:79-225): same type names, same public members, same
// meaning, so code written against the reference's `--im` plugin boundary compiles unchanged.
// Written from scratch; behaviour notes cite the reference implementation they mirror.
#ifndef BODIES_HPP_
#define BODIES_HPP_

#include <string>
#include <vector>

// structure of arrays / array of structures of body characteristics (Bodies.hpp:15-44)
template <typename T> struct dataSoA_t { std::vector<T> qx, qy, qz, vx, vy, vz, m, r; };
template <typename T> struct dataAoS_t { T qx, qy, qz, vx, vy, vz, m, r; };
// accelerations (Bodies.hpp:54-71)
template <typename T> struct accSoA_t { std::vector<T> ax, ay, az; };
template <typename T> struct accAoS_t { T ax, ay, az; };

#ifndef MURB_SIMD_BYTES
#define MURB_SIMD_BYTES 16   // the reference builds without -march: MIPP = SSE2, mipp::N<float>() == 4
#endif

template <typename T> class Bodies {
  protected:
    unsigned long n;
    mutable dataSoA_t<T> dataSoA;
    mutable std::vector<dataAoS_t<T>> dataAoS;
    unsigned short padding;   // fictional bodies filling the last SIMD vector (Bodies.cpp:160-161)
    float allocatedBytes;

  public:
    // scheme: "galaxy", "random", anything else = read milkyway_andromeda.tab (Bodies.cpp:14-25)
    Bodies(const unsigned long n, const std::string &scheme = "galaxy", const unsigned long randInit = 0);
    virtual ~Bodies() = default;

    const unsigned long getN() const;
    const unsigned short getPadding() const;
    virtual const dataSoA_t<T> &getDataSoA() const;
    virtual const std::vector<dataAoS_t<T>> &getDataAoS() const;
    const float getAllocatedBytes() const;

    // time integration, applied after each iteration (Bodies.cpp:260-298)
    virtual void updatePositionsAndVelocities(const accSoA_t<T> &accelerations, T &dt);
    virtual void updatePositionsAndVelocities(const std::vector<accAoS_t<T>> &accelerations, T &dt);

    void initGalaxy(const unsigned long randInit = 0);
    void initRandomly(const unsigned long randInit = 0);
    void initMilkyWayAndromeda();

  protected:
    void updatePositionAndVelocity(const unsigned long iBody, const T mi, const T ri, const T qix, const T qiy,
                                   const T qiz, const T vix, const T viy, const T viz, const T aix, const T aiy,
                                   const T aiz, T &dt);
    inline void setBody(const unsigned long &iBody, const T &mi, const T &ri, const T &qix, const T &qiy, const T &qiz,
                        const T &vix, const T &viy, const T &viz);
    void allocateBuffers();

  private:
    void computePadding();
    void fillPaddingZone();
    void drawBoxBody(unsigned long iBody, T mi, T ri);
};

#endif /* BODIES_HPP_ */
