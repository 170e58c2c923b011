// Per-launch cost of dependent kernels in one stream (what 77 launches of a forward pay at least): chains of N launches of
//   (a) an empty kernel with one workgroup, (b) an empty kernel with 768 workgroups of 256 threads,
//   (c) a kernel that writes 47 MB with sc1 stores and one that reads it (the traffic of a 76x76 layer boundary),
// timed with hipEvents around the chain; and the same chain replayed as a hipGraph.
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench_launch tools/ubench_launch.hip && tools/ubench_launch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void k_empty(int* p) { if (p && threadIdx.x == 12345) *p = 1; }
__global__ void k_write(float4* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const f32x4 v = {1.f, 2.f, 3.f, 4.f};
        asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p + i), "v"(v) : "memory");
    }
}
__global__ void k_read(const float4* p, size_t n, float* out) {
    float s = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = p[i]; s += v.x + v.w; }
    if (s == 12345.678f) *out = s;
}
int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t n = 47u * 1024 * 1024 / 16;
    float4* buf; float* out; CK(hipMalloc(&buf, n * 16)); CK(hipMalloc(&out, 4));
    const int N = 200;
    auto timeit = [&](const char* name, auto launch) -> int {
        for (int i = 0; i < 20; ++i) launch();
        CK(hipStreamSynchronize(st));
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < N; ++i) launch();
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("%-44s %.2f us per launch\n", name, best * 1000.f / N);
        return 0;
    };
    timeit("empty kernel, 1 workgroup", [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, (int*)nullptr); });
    timeit("empty kernel, 768 x 256 threads", [&] { hipLaunchKernelGGL(k_empty, dim3(768), dim3(256), 0, st, (int*)nullptr); });
    timeit("empty kernel, 768 x 256 threads, 40 KB LDS", [&] { hipLaunchKernelGGL(k_empty, dim3(768), dim3(256), 40960, st, (int*)nullptr); });
    timeit("write 47 MB (sc1)", [&] { hipLaunchKernelGGL(k_write, dim3(2048), dim3(256), 0, st, buf, n); });
    timeit("read 47 MB", [&] { hipLaunchKernelGGL(k_read, dim3(2048), dim3(256), 0, st, buf, n, out); });
    timeit("write 47 MB then read it (pair)", [&] { hipLaunchKernelGGL(k_write, dim3(2048), dim3(256), 0, st, buf, n); hipLaunchKernelGGL(k_read, dim3(2048), dim3(256), 0, st, buf, n, out); });
    // graph of N empty launches
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_empty, dim3(768), dim3(256), 0, st, (int*)nullptr);
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) { CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
    printf("%-44s %.2f us per launch\n", "graph of 200 empty 768 x 256 kernels", best * 1000.f / N);
    return 0;
}
