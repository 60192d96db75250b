// TEST INFRASTRUCTURE.  The reference's murb executable is linked with -ffast-math
// (CMakeLists.txt:128-131), so GCC's crtfastmath.o switches it to flush-to-zero / denormals-are-zero
// at start-up.  That is visible in its results: cpu+optim forms G * inv^3 first
// (SimulationNBodyOptim.cpp:69), which is subnormal for pairs farther apart than ~1.8e9 m (they occur
// in the `random` scheme) and is flushed to 0 — those pairs then exert no force.  A shared object
// loaded into Python does not reliably inherit that mode, so the oracle sets it for the duration of
// each call instead.
#pragma once
#include <xmmintrin.h>
struct FlushDenormalsLikeReference {
    unsigned saved;
    FlushDenormalsLikeReference() : saved(_mm_getcsr()) { _mm_setcsr(saved | 0x8040u); }   // FTZ | DAZ
    ~FlushDenormalsLikeReference() { _mm_setcsr(saved); }
};
