// Wall-clock stopwatch with the reference's metric definitions (reference src/common/utils/Perf.hpp:6-31,
// Perf.cpp:26-35): times in ms from gettimeofday, Gflop/s divides by 1024^3 (not 1e9).
#ifndef PERF_HPP_
#define PERF_HPP_

#include <cstddef>

class Perf {
    unsigned long tStart = 0, tStop = 0;   // microseconds

  public:
    Perf() = default;
    explicit Perf(float ms) : tStart(0), tStop((unsigned long)(ms * 1000)) {}
    void start();
    void stop();
    void reset() { tStart = tStop = 0; }
    float getElapsedTime() const { return (tStop - tStart) / 1000.f; }                       // ms
    float getGflops(float flops) const;                                                       // flops / s / 1024^3
    float getFPS(const size_t nFrames = 1) const { return (nFrames * 1000.f) / getElapsedTime(); }
    float getMemoryBandwidth(unsigned long memops, unsigned short nBytes) const;              // GiB/s
    Perf operator+(const Perf &p) const;
    Perf &operator+=(const Perf &p);

  protected:
    static unsigned long getTime();
};

#endif
