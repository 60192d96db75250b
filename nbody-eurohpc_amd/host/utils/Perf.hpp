// Stopwatch with the metric definitions the reference's driver prints
// (reference src/common/utils/Perf.hpp:6-31 / Perf.cpp:26-35): elapsed milliseconds, frames per
// second, and "Gflop/s" as flops / seconds / 1024^3 — a binary giga, which is why murb's --gf figures
// are 7 % below decimal GFLOP/s.  Header only; time source is std::chrono::steady_clock.
#ifndef PERF_HPP_
#define PERF_HPP_

#include <chrono>
#include <cstddef>

class Perf {
    using clock = std::chrono::steady_clock;
    long long beginUs = 0, endUs = 0;   // the measured interval is [beginUs, endUs], in microseconds

    static long long nowUs()
    {
        return std::chrono::duration_cast<std::chrono::microseconds>(clock::now().time_since_epoch()).count();
    }
    static constexpr double kBinaryGiga = 1024.0 * 1024.0 * 1024.0;

  public:
    Perf() = default;
    explicit Perf(float ms) : beginUs(0), endUs((long long)(ms * 1000.f)) {}

    void start() { beginUs = nowUs(); }
    void stop() { endUs = nowUs(); }
    void reset() { beginUs = endUs = 0; }

    float getElapsedTime() const { return (float)(endUs - beginUs) / 1000.f; }   // milliseconds
    float getFPS(const size_t nFrames = 1) const { return (float)nFrames * 1000.f / getElapsedTime(); }
    float getGflops(float flops) const { return (float)((double)flops * (1000.0 / getElapsedTime()) / kBinaryGiga); }
    float getMemoryBandwidth(unsigned long memops, unsigned short nBytes) const
    {
        return (float)((double)memops * nBytes * (1000.0 / getElapsedTime()) / kBinaryGiga);
    }

    // accumulate intervals: (a + b) and (a += b) measure the sum of both durations
    Perf operator+(const Perf &other) const
    {
        Perf sum;
        sum.endUs = (endUs - beginUs) + (other.endUs - other.beginUs);
        return sum;
    }
    Perf &operator+=(const Perf &other)
    {
        endUs += other.endUs - other.beginUs;
        return *this;
    }
};

#endif
