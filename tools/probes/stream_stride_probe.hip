// Development probe (standalone: hipcc --offload-arch=gfx950 -O3 -o stream_stride_probe stream_stride_probe.hip).
// Does the bandwidth of a kernel that walks S read streams and W write streams in lockstep depend on the SPACING
// between the streams (DRAM bank / Infinity-Cache set conflicts between streams that sit a power of two apart)?
// The step kernel at 1 M envs walks ~6 fat streams (px py vx vy in, px py out, obs out) whose arrays are each 128 MiB.
//   usage: stream_stride_probe [bytes_per_stream_MiB=128] [reads=4] [writes=2]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int R, int W>
__global__ void __launch_bounds__(256) walk(const char* base, size_t spacing, size_t n16)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n16) return;
    uint4 acc = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int s = 0; s < R; s++) {
        const uint4 v = ((const uint4*)(base + (size_t)s * spacing))[i];
        acc.x += v.x; acc.y ^= v.y; acc.z += v.z; acc.w ^= v.w;
    }
#pragma unroll
    for (int s = 0; s < W; s++) {
        acc.x += s;
        // the write streams overwrite the first W read streams in place (as the step does with px / py)
        ((uint4*)(base + (size_t)s * spacing))[i] = acc;
    }
}

int main(int argc, char** argv)
{
    const size_t mib = argc > 1 ? (size_t)atol(argv[1]) : 128;
    const size_t bytes = mib << 20, n16 = bytes / 16;
    const int R = 4, W = 2;
    const size_t max_spacing = bytes + (64u << 20);
    char* buf = nullptr;
    CHECK(hipMalloc((void**)&buf, max_spacing * R + bytes));
    CHECK(hipMemset(buf, 1, max_spacing * R + bytes));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const size_t extras[] = {0, 256, 4096, 4096 + 256, 65536, 65536 + 4096, 1u << 20, (1u << 20) + 65536, 2u << 20, (2u << 20) + 4096,
                             3u << 20, 5u << 20, (5u << 20) + 4096 * 3, 8u << 20, 11u << 20, 16u << 20, 17u << 20, 32u << 20, 33u << 20, 37u << 20,
                             64u << 20};
    printf("buffer %p, %zu MiB per stream, %d read + %d in-place write streams, %.0f MB per launch\n", (void*)buf, mib, R, W,
           (double)(R + W) * bytes / 1e6);
    for (size_t x : extras) {
        const size_t spacing = bytes + x;
        for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL((walk<R, W>), dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, buf, spacing, n16);
        CHECK(hipEventRecord(e0, 0));
        const int reps = 6;
        for (int rep = 0; rep < reps; rep++) hipLaunchKernelGGL((walk<R, W>), dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, 0, buf, spacing, n16);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("spacing = size + %9zu B: %.1f us per launch, %.2f TB/s\n", x, ms / reps * 1e3, (double)(R + W) * bytes / (ms / reps * 1e-3) / 1e12);
    }
    return 0;
}
