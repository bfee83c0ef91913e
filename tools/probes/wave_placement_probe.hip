// Development probe (standalone: hipcc --offload-arch=gfx950 -O3 -o wave_placement_probe wave_placement_probe.hip).
// Where does the dispatcher put wave w of a workgroup?  The looped kernels give the waves of a block different ROLES
// (sweeping / ownship / copy): if wave 0 of every block landed on SIMD 0, all the critical sweeping waves of a CU would
// share one SIMD's issue port.  Every wave records HW_ID (gfx9: [3:0] wave slot, [5:4] SIMD, [11:8] CU, [13:12] SH,
// [15:14] SE... as the ISA manual lays it out) and XCC_ID, then spins so that the whole grid is co-resident.
//   usage: wave_placement_probe [blocks=1024] [threads=192] [vgpr_hog=1]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int HOG>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) probe(uint32_t* out, int spin)
{
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin) { }
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
        out[2 * w] = hw;
        out[2 * w + 1] = xcc;
    }
}

int main(int argc, char** argv)
{
    const int blocks = argc > 1 ? atoi(argv[1]) : 1024, threads = argc > 2 ? atoi(argv[2]) : 192;
    const int wpb = threads / 64, nw = blocks * wpb;
    uint32_t* d = nullptr;
    CHECK(hipMalloc((void**)&d, (size_t)nw * 8));
    CHECK(hipMemset(d, 0xff, (size_t)nw * 8));
    hipLaunchKernelGGL((probe<1>), dim3(blocks), dim3(threads), 0, 0, d, 3000 /* 30 us of the 100 MHz clock */);
    CHECK(hipDeviceSynchronize());
    std::vector<uint32_t> h((size_t)nw * 2);
    CHECK(hipMemcpy(h.data(), d, (size_t)nw * 8, hipMemcpyDeviceToHost));
    // histogram: wave index in block -> SIMD
    std::vector<std::vector<int>> hist(wpb, std::vector<int>(4, 0));
    std::map<uint32_t, std::vector<int>> per_cu;      // (xcc, se, sh, cu) -> waves per SIMD by role 0
    for (int w = 0; w < nw; w++) {
        const uint32_t hw = h[2 * (size_t)w], xcc = h[2 * (size_t)w + 1] & 0xf;
        const int simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        hist[w % wpb][simd]++;
        auto& v = per_cu[(xcc << 16) | (se << 8) | (sh << 4) | cu];
        if (v.empty()) v.assign(8, 0);
        v[simd]++;
        if (w % wpb == 0) v[4 + simd]++;
    }
    printf("%d blocks x %d waves; wave index in block -> count per SIMD 0..3\n", blocks, wpb);
    for (int i = 0; i < wpb; i++) printf("  wave %d: %5d %5d %5d %5d\n", i, hist[i][0], hist[i][1], hist[i][2], hist[i][3]);
    printf("%zu distinct (xcc, se, sh, cu); first 12: waves per SIMD | role-0 waves per SIMD\n", per_cu.size());
    int n = 0, worst = 0;
    for (auto& kv : per_cu) {
        const auto& v = kv.second;
        for (int s = 0; s < 4; s++) if (v[4 + s] > worst) worst = v[4 + s];
        if (n++ < 12) printf("  %06x: %d %d %d %d | %d %d %d %d\n", kv.first, v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
    }
    printf("most role-0 waves on one SIMD of one CU: %d\n", worst);
    printf("first 8 blocks, (simd of wave 0, 1, ...):");
    for (int b = 0; b < 8 && b < blocks; b++) {
        printf(" [");
        for (int i = 0; i < wpb; i++) printf("%d", (h[2 * (size_t)(b * wpb + i)] >> 4) & 3);
        printf("]");
    }
    printf("\n");
    return 0;
}
