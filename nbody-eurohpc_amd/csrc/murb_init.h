// Initial conditions ON THE DEVICE: the reference's Bodies::initGalaxy / initRandomly
// (src/common/core/Bodies.cpp:158-257) restated for the GPU so that the body state never exists on the host —
// bit-identical to what the reference's host code produces on the same machine, which pins three things:
//
//   1. glibc's rand(): the TYPE_3 additive feedback generator of random_r.c, w[k] = w[k-3] + w[k-31] (mod 2^32),
//      output w[k] >> 1, seeded by srandom_r (a 16807 Lehmer sequence into 31 words, 310 outputs discarded).  The
//      recurrence is linear, so a thread can jump straight to draw k: x^k mod (x^31 - x^28 - 1) over Z/2^32 by square and
//      multiply (22 squarings of a 31-coefficient polynomial for 4 M draws), applied to 61 base words the host derives
//      from the seed.  Every thread then produces 248 consecutive draws.  (murb_rand_fill_kernel)
//   2. the float/double mix of the reference's expressions AS COMPILED with its flags (-O3 -ffast-math, no -march): read
//      off the object code of the mirrored host file, e.g. `f * 1.0e8 + 1.0e8` is evaluated as (f + 1.0) * 1.0e8 in double,
//      `r / (float)RAND_MAX` as r * 2^-31f, and the box scheme's x = fc * (5e8 * 1.33) as ONE float multiply by
//      0x3f1e8c61.  FP contraction is off in this file; every operation below is one IEEE operation of the host code.
//   3. glibc's sincosf (sysdeps/ieee754/flt-32/s_sincosf.c, the ARM optimized-routines algorithm): double-precision
//      range reduction by pi/2 and two degree-8/9 polynomials.  glibc 2.35 ships two builds of it and picks one at load
//      time: compiled with -mfma -mavx2 (every a + b * c contracted into an FMA) on CPUs with FMA and AVX2, plain SSE2
//      otherwise.  Both are here (template parameter); the host tells the kernel which one its libm uses.
//
// Checked on the GPU box against the product's host initialisation (which is bit-identical to the compiled reference):
// tests/test_gpu_parity.py::test_device_initial_conditions_*.
#ifndef MURB_INIT_H_
#define MURB_INIT_H_

#include <hip/hip_runtime.h>

#include "murb_layout.h"

#define MURB_RAND_DEG 31
#define MURB_RAND_BLOCKS 8
#define MURB_RAND_CHUNK (MURB_RAND_DEG * MURB_RAND_BLOCKS)   /* draws per thread */

struct MurbRandBase {
    unsigned int u[2 * MURB_RAND_DEG - 1];   // the 31 words in front of the first draw and the 30 that follow
};

// p <- p * p mod (x^31 - x^28 - 1), coefficients mod 2^32
__device__ __forceinline__ void murb_rand_poly_square(unsigned int (&p)[MURB_RAND_DEG])
{
    unsigned int sq[2 * MURB_RAND_DEG - 1];
#pragma unroll
    for (int d = 0; d < 2 * MURB_RAND_DEG - 1; ++d) sq[d] = 0u;
#pragma unroll
    for (int a = 0; a < MURB_RAND_DEG; ++a)
#pragma unroll
        for (int b = 0; b < MURB_RAND_DEG; ++b) sq[a + b] += p[a] * p[b];
#pragma unroll
    for (int d = 2 * MURB_RAND_DEG - 2; d >= MURB_RAND_DEG; --d) {   // x^d = x^(d-3) + x^(d-31)
        sq[d - 3] += sq[d];
        sq[d - MURB_RAND_DEG] += sq[d];
    }
#pragma unroll
    for (int m = 0; m < MURB_RAND_DEG; ++m) p[m] = sq[m];
}

// p <- p * x mod (x^31 - x^28 - 1)
__device__ __forceinline__ void murb_rand_poly_mulx(unsigned int (&p)[MURB_RAND_DEG])
{
    const unsigned int top = p[MURB_RAND_DEG - 1];
#pragma unroll
    for (int m = MURB_RAND_DEG - 1; m > 0; --m) p[m] = p[m - 1];
    p[0] = top;
    p[MURB_RAND_DEG - 3] += top;
}

// out[k] = the k-th value rand() returns after srand(seed), k in [0, total): thread t produces draws
// [t * MURB_RAND_CHUNK, (t + 1) * MURB_RAND_CHUNK).
__global__ __launch_bounds__(64) void murb_rand_fill_kernel(const MurbRandBase base, const unsigned long total, unsigned int* out)
{
    const unsigned long k0 = ((unsigned long)blockIdx.x * blockDim.x + threadIdx.x) * MURB_RAND_CHUNK;
    if (k0 >= total) return;
    unsigned int p[MURB_RAND_DEG];
#pragma unroll
    for (int m = 0; m < MURB_RAND_DEG; ++m) p[m] = m == 0 ? 1u : 0u;
    for (int bit = 63 - __builtin_clzl(k0 | 1ul); bit >= 0; --bit) {
        murb_rand_poly_square(p);
        if ((k0 >> bit) & 1ul) murb_rand_poly_mulx(p);
    }
    // w[j] = word k0 + j of the generator: sum over m of p[m] * u[m + j]
    unsigned int w[MURB_RAND_DEG];
#pragma unroll
    for (int j = 0; j < MURB_RAND_DEG; ++j) {
        unsigned int acc = 0u;
#pragma unroll
        for (int m = 0; m < MURB_RAND_DEG; ++m) acc += p[m] * base.u[m + j];
        w[j] = acc;
    }
    for (int blk = 0; blk < MURB_RAND_BLOCKS; ++blk) {
        // the next 31 words in place: word k + 31 = word k + word k + 28 (for j >= 3 the second one is already new)
#pragma unroll
        for (int j = 0; j < MURB_RAND_DEG; ++j) {
            w[j] += w[(j + MURB_RAND_DEG - 3) % MURB_RAND_DEG];
            const unsigned long k = k0 + (unsigned long)(blk * MURB_RAND_DEG + j);
            if (k < total) out[k] = w[j] >> 1;
        }
    }
}

// ---- glibc 2.35 sincosf ---------------------------------------------------------------------------------------------
// __sincosf_table[2] of s_sincosf_data.c as laid out in the shipped libm (sign[4], hpi_inv * 2^24, hpi, then c0 c1 s1 c2 s2
// c3 s3 c4); entry 1 has the cosine coefficients negated (quadrants 2 and 3).
template <bool FMA> __device__ __forceinline__ double murb_mad(double a, double b, double c)
{
#pragma clang fp contract(off)
    if (FMA) return __builtin_fma(a, b, c);
    const double t = a * b;
    return t + c;
}

template <bool FMA>
__device__ __forceinline__ void murb_sincosf_poly(double x, double x2, bool negcos, int n, float* sinp, float* cosp)
{
#pragma clang fp contract(off)
    const double sg = negcos ? -1.0 : 1.0;
    const double c0 = sg * 0x1.0000000000000p+0, c1 = sg * -0x1.ffffffd0c621cp-2, c2 = sg * 0x1.55553e1068f19p-5;
    const double c3 = sg * -0x1.6c087e89a359dp-10, c4 = sg * 0x1.99343027bf8c3p-16;
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    const double x3 = x2 * x, x4 = x2 * x2;
    const double c2p = murb_mad<FMA>(x2, c4, c3), s1p = murb_mad<FMA>(x2, s3, s2), c1p = murb_mad<FMA>(x2, c1, c0);
    const double x5 = x2 * x3, x6 = x2 * x4;
    const double s = murb_mad<FMA>(x3, s1, x), c = murb_mad<FMA>(x4, c2, c1p);
    const float sv = (float)murb_mad<FMA>(s1p, x5, s), cv = (float)murb_mad<FMA>(c2p, x6, c);
    *sinp = (n & 1) ? cv : sv;
    *cosp = (n & 1) ? sv : cv;
}

template <bool FMA> __device__ __forceinline__ void murb_sincosf(float y, float* sinp, float* cosp)
{
#pragma clang fp contract(off)
    const double x = (double)y;
    const unsigned int top = (__float_as_uint(y) >> 20) & 0x7ffu;   // abstop12
    if (top < 0x3f4u) {                       // |y| < pi/4
        if (top < 0x398u) { *sinp = y; *cosp = 1.0f; return; }   // |y| < 2^-12
        murb_sincosf_poly<FMA>(x, x * x, false, 0, sinp, cosp);
    } else if (top < 0x42fu) {                // |y| < 120: reduce_fast
        const double r = x * 0x1.45f306dc9c883p+23;
        const int n = ((int)r + 0x800000) >> 24;
        const double xr = murb_mad<FMA>(-(double)n, 0x1.921fb54442d18p+0, x);
        const double s = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0;
        murb_sincosf_poly<FMA>(xr * s, xr * xr, (n & 2) != 0, n, sinp, cosp);
    } else {                                  // not reachable from the initial conditions (angles are in (0, 2 pi])
        *sinp = __uint_as_float(0x7fc00000u);
        *cosp = __uint_as_float(0x7fc00000u);
    }
}

// ---- bodies ----------------------------------------------------------------------------------------------------------
struct MurbInitArgs {
    const unsigned int* draws;   // rand() values in call order
    float4* rec0;                // both position buffers, all slots
    float4* rec1;
    float4* vel;                 // local slice
    float* mass;                 // local slice, as the host would upload it
    float* radius;               // local slice
    unsigned long n;
    unsigned int world, rank;
    unsigned int slice;          // slots per rank
    float g;
};

__device__ __forceinline__ void murb_init_store(const MurbInitArgs& a, unsigned long i, float m, float r, float qx, float qy, float qz,
                                                float vx, float vy, float vz)
{
#pragma clang fp contract(off)
    // reference partition rule (SimulationNBodyMultiNode.cpp:76-91): ranks < rem own base + 1 bodies
    const unsigned long base = a.n / a.world, rem = a.n % a.world;
    unsigned long owner, first;
    if (i < rem * (base + 1)) { owner = i / (base + 1); first = owner * (base + 1); }
    else { owner = rem + (base ? (i - rem * (base + 1)) / base : 0); first = rem * (base + 1) + (owner - rem) * base; }
    const unsigned long slot = owner * a.slice + (i - first);
    const unsigned long ra = murb_rec_a(slot >> 1);
    const int h = (int)(slot & 1ul);
    const float gm = a.g * m;
    float* A0 = reinterpret_cast<float*>(a.rec0 + ra);
    float* B0 = reinterpret_cast<float*>(a.rec0 + ra + MURB_TILE_PAIRS);
    float* A1 = reinterpret_cast<float*>(a.rec1 + ra);
    float* B1 = reinterpret_cast<float*>(a.rec1 + ra + MURB_TILE_PAIRS);
    A0[h] = qx; A0[2 + h] = qy; B0[h] = qz; B0[2 + h] = gm;
    A1[h] = qx; A1[2 + h] = qy; B1[h] = qz; B1[2 + h] = gm;
    if (owner != a.rank) return;
    const unsigned long ls = i - first;
    const unsigned long va = murb_rec_a(ls >> 1);
    float* VA = reinterpret_cast<float*>(a.vel + va);
    float* VB = reinterpret_cast<float*>(a.vel + va + MURB_TILE_PAIRS);
    VA[h] = vx; VA[2 + h] = vy; VB[h] = vz;
    a.mass[ls] = m;
    a.radius[ls] = r;
}

// Bodies::initGalaxy (Bodies.cpp:158-214) as compiled: body 0 heavy and at rest, body i >= 1 from draws 4 (i - 1) .. + 3
template <bool FMA> __global__ __launch_bounds__(256) void murb_init_galaxy_kernel(const MurbInitArgs a)
{
#pragma clang fp contract(off)
    const unsigned long i = (unsigned long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    if (i == 0) { murb_init_store(a, 0, 2.0e24f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f); return; }
    const unsigned int* y = a.draws + 4ul * (i - 1);
    const float k31 = 0x1p-31f;                                   // 1 / (float)RAND_MAX
    const float fm = (float)(int)y[0] * k31;
    const float mi = (float)((double)fm * 5e20);
    const float ri = (float)((double)mi * 2.5e-15);
    const float ah = (float)((double)((float)(int)(2147483647u - y[1]) * k31) * 0x1.921fb54442d18p+2);   // * 2 pi
    const float av = (float)((double)((float)(int)(2147483647u - y[2]) * k31) * 0x1.921fb54442d18p+2);
    float sh, ch, sv, cv;
    murb_sincosf<FMA>(ah, &sh, &ch);
    murb_sincosf<FMA>(av, &sv, &cv);
    const float dist = (float)(((double)((float)(int)(2147483647u - y[3]) * k31) + 1.0) * 1.0e8);
    const float t = sh * cv;
    const float qx = t * dist;
    const float qy = sv * dist;
    const float u = cv * ch;
    const float qz = u * dist;
    const float vx = (float)((double)qy * 4.0e-6);
    const float vy = (float)((double)(-qx) * 4.0e-6);
    murb_init_store(a, i, mi, ri, qx, qy, qz, vx, vy, 0.f);
}

// Bodies::initRandomly (Bodies.cpp:217-257) as compiled: body i from draws 7 i .. + 6 (mass, then the six of the box)
__global__ __launch_bounds__(256) void murb_init_random_kernel(const MurbInitArgs a)
{
#pragma clang fp contract(off)
    const unsigned long i = (unsigned long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const unsigned int* y = a.draws + 7ul * i;
    const float fm = (float)(int)y[0] * 0x1p-31f;
    const float mi = (float)((double)fm * 5.0e21);
    const float ri = (float)((double)mi * 0.5e-14);
    const int half = 0x3fffffff;                                  // RAND_MAX / 2
    const float qx = (float)((int)y[1] - half) * __uint_as_float(0x3f1e8c61u);   // 2^-30 * 5e8 * 1.33, one float
    const float qy = (float)((int)y[2] - half) * __uint_as_float(0x3eee6b28u);   // 2^-30 * 5e8
    const float qz = (float)((double)((float)((int)y[3] - half) * 0x1p-30f) * 5.0e8 - 1.0e9);
    const float kv = __uint_as_float(0x33c80000u);                               // 2^-30 * 1e2
    const float vx = (float)((int)y[4] - half) * kv;
    const float vy = (float)((int)y[5] - half) * kv;
    const float vz = (float)((int)y[6] - half) * kv;
    murb_init_store(a, i, mi, ri, qx, qy, qz, vx, vy, vz);
}

#endif
