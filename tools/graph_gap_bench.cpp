// tools/graph_gap_bench.cpp - what a chain of N dependent kernel nodes costs in a HIP graph on this chip when the kernels do nothing:
// the floor under the step's "graph time minus sum of kernel times" (134 nodes at B = 1), i.e. what removing a node is worth.
// Build: hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/graph_gap_bench.cpp -o tools/graph_gap_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
__global__ void tiny(float* p, int blocks_that_write) { if (blockIdx.x < blocks_that_write && threadIdx.x == 0) p[blockIdx.x] += 1.f; }
__global__ void spin(float* p, long long ticks) { const long long t0 = __builtin_amdgcn_s_memtime(); while (__builtin_amdgcn_s_memtime() - t0 < ticks) {} if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipStream_t st; CK(hipStreamCreate(&st));
    float* d; CK(hipMalloc((void**)&d, 1 << 20)); CK(hipMemset(d, 0, 1 << 20));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int nodes[] = {1, 16, 134, 268};
    const int grids[] = {1, 256, 2048};
    for (int grid : grids)
        for (int n : nodes) {
            hipGraph_t g; hipGraphExec_t ge;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
            for (int i = 0; i < n; ++i) hipLaunchKernelGGL(tiny, dim3(grid), dim3(256), 0, st, d, 1);
            CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, st));
            CK(hipStreamSynchronize(st));
            const int reps = 50;
            CK(hipEventRecord(e0, st)); for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("empty kernels, grid %4d x 256: %3d nodes: %8.1f us per graph launch = %5.2f us per node\n", grid, n, ms * 1000.f / reps, ms * 1000.f / reps / n);
            hipGraphExecDestroy(ge); hipGraphDestroy(g);
        }
    // the same chain eagerly (stream order, no graph)
    for (int n : {134}) {
        for (int i = 0; i < 3 * n; ++i) hipLaunchKernelGGL(tiny, dim3(256), dim3(256), 0, st, d, 1);
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st)); for (int r = 0; r < 20; ++r) for (int i = 0; i < n; ++i) hipLaunchKernelGGL(tiny, dim3(256), dim3(256), 0, st, d, 1);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("eager stream, grid 256 x 256: %d launches: %8.1f us = %5.2f us per launch\n", n, ms * 1000.f / 20, ms * 1000.f / 20 / n);
    }
    // 20 us kernels: does the gap change when the kernels are long enough for the next packet to be pre-processed?
    for (int n : {16, 134}) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, st, d, 2000LL);      // 2000 ticks of the 100 MHz counter = 20 us
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st)); for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("20 us spin kernels (2000 s_memtime ticks), grid 256 x 256: %3d nodes: %8.1f us per graph launch = %6.2f us per node\n", n, ms * 1000.f / 10, ms * 1000.f / 10 / n);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    return 0;
}
