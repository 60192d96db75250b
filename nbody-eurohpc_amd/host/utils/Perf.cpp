#include "utils/Perf.hpp"

#include <sys/time.h>

unsigned long Perf::getTime()
{
    struct timeval t;
    return gettimeofday(&t, nullptr) == 0 ? t.tv_sec * 1000000ul + t.tv_usec : 0ul;
}

void Perf::start() { tStart = getTime(); }
void Perf::stop() { tStop = getTime(); }

// Perf.cpp:28 — note the binary divisor: the "Gflop/s" murb prints are 2^30 flop/s
float Perf::getGflops(float flops) const { return (flops * (1000 / getElapsedTime())) / 1024.0 / 1024.0 / 1024.0; }

float Perf::getMemoryBandwidth(unsigned long memops, unsigned short nBytes) const
{
    return (memops * nBytes * (1000 / getElapsedTime())) / 1024.0 / 1024.0 / 1024.0;
}

Perf Perf::operator+(const Perf &p) const
{
    Perf sum;
    sum.tStop = (p.tStop - p.tStart) + (tStop - tStart);
    return sum;
}

Perf &Perf::operator+=(const Perf &p)
{
    tStop += p.tStop - p.tStart;
    return *this;
}
