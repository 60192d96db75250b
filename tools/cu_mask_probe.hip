// Lab tool: which CUs does a stream created with hipExtStreamCreateWithCUMask use?  Launches many small workgroups on a
// stream whose mask leaves out the `reserve` highest bits and counts, per XCD, the distinct CUs they ran on.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/cu_mask_probe.hip -o tools/cu_mask_probe ; tools/cu_mask_probe [reserve]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
__global__ void where(unsigned* out)
{
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_ID
        out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 20); // XCC_ID
    }
    for (volatile int k = 0; k < 2000; ++k) {}   // long enough that the workgroups spread over every CU available
}
int main(int argc, char** argv)
{
    const int reserve = argc > 1 ? atoi(argv[1]) : 8;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    std::vector<uint32_t> mask((cus + 31) / 32, 0u);
    for (int b = 0; b < cus - reserve; ++b) mask[b / 32] |= 1u << (b % 32);
    hipStream_t s; CK(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
    const int n = 16384;
    unsigned* d; CK(hipMalloc(&d, 2 * n * sizeof(unsigned)));
    hipLaunchKernelGGL(where, dim3(n), dim3(256), 0, s, d);
    CK(hipStreamSynchronize(s));
    std::vector<unsigned> h(2 * n); CK(hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost));
    std::map<unsigned, std::set<unsigned>> per_xcc;
    for (int k = 0; k < n; ++k) per_xcc[h[2 * k + 1] & 0xf].insert((h[2 * k] >> 8) & 0xff);   // HW_ID bits 8-15: cu, sh, se
    printf("%d CUs, mask leaves out the %d highest bits:", cus, reserve);
    int total = 0;
    for (auto& kv : per_xcc) { printf("  XCD %u: %zu CUs", kv.first, kv.second.size()); total += (int)kv.second.size(); }
    printf("  (total %d)\n", total);
    return 0;
}
