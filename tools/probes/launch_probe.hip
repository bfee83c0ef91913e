// Development probe: how long does an (almost) empty kernel of the step kernel's launch shape take?
// hipcc --offload-arch=gfx950 -O3 -o launch_probe launch_probe.hip && ./launch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int WORK>
__global__ void __launch_bounds__(256) probe(float* out, int n)
{
    extern __shared__ float lds[];
    float v = 0.f;
    if (WORK) {
        // touch LDS and do a little arithmetic so that the wave is not retired at once
        lds[threadIdx.x] = (float)threadIdx.x;
        __syncthreads();
        for (int i = 0; i < WORK; i++) v = v * 1.0001f + lds[(threadIdx.x + i) & 255];
    }
    if (n < 0) out[blockIdx.x * 256 + threadIdx.x] = v;      // never taken: keeps v alive
}

template <int WORK>
static void run(const char* name, int grid, int block, size_t lds)
{
    float* out = nullptr;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 50; i++) hipLaunchKernelGGL(probe<WORK>, dim3(grid), dim3(block), lds, 0, out, 0);
    hipDeviceSynchronize();
    std::vector<float> t;
    for (int rep = 0; rep < 200; rep++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(probe<WORK>, dim3(grid), dim3(block), lds, 0, out, 0);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        t.push_back(ms * 1e3f);
    }
    std::sort(t.begin(), t.end());
    // back-to-back: 200 launches between one pair of events
    hipEventRecord(e0, 0);
    for (int i = 0; i < 200; i++) hipLaunchKernelGGL(probe<WORK>, dim3(grid), dim3(block), lds, 0, out, 0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s grid %5d x %3d lds %6zu: single launch median %.2f us (min %.2f), back-to-back %.2f us per launch\n",
           name, grid, block, lds, t[t.size() / 2], t[0], ms * 1e3f / 200);
    hipFree(out);
}

int main()
{
    run<0>("empty", 2048, 256, 19456);
    run<0>("empty, no LDS", 2048, 256, 0);
    run<0>("empty, 512-thread blocks", 1024, 512, 38912);
    run<0>("empty, 1024 blocks", 1024, 256, 19456);
    run<0>("empty, 256 blocks", 256, 256, 19456);
    run<64>("64 fma + barrier", 2048, 256, 19456);
    run<1024>("1024 fma + barrier", 2048, 256, 19456);
    return 0;
}
