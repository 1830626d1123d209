// Developer probe: what does a kernel boundary cost on this box?  Chains of dependent launches on one stream, timed with
// HIP events: (a) an empty one-workgroup kernel, (b) an empty full-chip kernel (256 workgroups x 512 threads, 128 KiB of
// dynamic LDS like the persistent GEMM), (c) the same with a short spin so that launches cannot overlap their tails.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/launch_gap_probe tools/launch_gap_probe.hip && /tmp/launch_gap_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void empty_kernel(int* p) {
    if (p && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) *p = 1;
}

__global__ void lds_kernel(int* p, int spin) {
    extern __shared__ int lds[];
    if (spin) {
        const long long t0 = __builtin_amdgcn_s_memtime();
        while (__builtin_amdgcn_s_memtime() - t0 < spin) {}
    }
    if (p && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) *p = lds[0];
}

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <typename F>
static float chain(F launch, int n, hipStream_t s) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 50; ++i) launch();
    hipStreamSynchronize(s);
    hipEventRecord(a, s);
    for (int i = 0; i < n; ++i) launch();
    hipEventRecord(b, s);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / n;
}

int main() {
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    int* d;
    CHECK(hipMalloc(&d, 4));
    CHECK(hipFuncSetAttribute((const void*)lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    const int n = 2000;
    printf("empty, 1 workgroup x 64:              %.2f us per launch\n", chain([&] { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, d); }, n, s));
    printf("empty, 256 workgroups x 512:          %.2f us per launch\n", chain([&] { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(512), 0, s, d); }, n, s));
    printf("128 KiB LDS, 256 x 512:               %.2f us per launch\n", chain([&] { hipLaunchKernelGGL(lds_kernel, dim3(256), dim3(512), 131072, s, d, 0); }, n, s));
    printf("128 KiB LDS, 256 x 512, spin 10 us:   %.2f us per launch (10 us of it is the spin at 100 MHz memtime)\n",
           chain([&] { hipLaunchKernelGGL(lds_kernel, dim3(256), dim3(512), 131072, s, d, 1000); }, n, s));
    printf("128 KiB LDS, 1024 x 512:              %.2f us per launch\n", chain([&] { hipLaunchKernelGGL(lds_kernel, dim3(1024), dim3(512), 131072, s, d, 0); }, n, s));
    printf("empty, 16384 workgroups x 256:        %.2f us per launch\n", chain([&] { hipLaunchKernelGGL(empty_kernel, dim3(16384), dim3(256), 0, s, d); }, n, s));
    // the same chains captured in a graph (one launch of 200 kernel nodes)
    hipGraph_t g;
    hipGraphExec_t ge;
    CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(lds_kernel, dim3(256), dim3(512), 131072, s, d, 0);
    CHECK(hipStreamEndCapture(s, &g));
    CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    printf("graph of 200 x (128 KiB LDS, 256 x 512): %.2f us per kernel node\n", chain([&] { hipGraphLaunch(ge, s); }, 20, s) / 200);
    return 0;
}
