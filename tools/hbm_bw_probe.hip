// Developer probe: what HBM delivers on this box for pure writes, pure reads and copies (16-byte accesses, 2 GiB buffers):
// the ceilings the write-heavy kernels (conv0, the GEMM epilogue bursts, rownorm) are priced against.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/hbm_bw_probe tools/hbm_bw_probe.hip && /tmp/hbm_bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void write_kernel(float4* dst, size_t n, float v) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = make_float4(v, v, v, v);
}
__global__ void read_kernel(const float4* src, size_t n, float* out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float4 v = src[i];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 12345.678f) *out = s;
}
__global__ void copy_kernel(const float4* src, float4* dst, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

template <typename F>
static float timed(F f) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int i = 0; i < 5; ++i) f();
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms / 5;
}

int main() {
    const size_t bytes = 2ull << 30, n = bytes / 16;
    float4 *a, *b;
    float* out;
    hipMalloc(&a, bytes);
    hipMalloc(&b, bytes);
    hipMalloc(&out, 4);
    hipMemset(a, 0, bytes);
    for (int wg = 1024; wg <= 16384; wg *= 4) {
        const float w = timed([&] { hipLaunchKernelGGL(write_kernel, dim3(wg), dim3(256), 0, 0, a, n, 1.f); });
        const float r = timed([&] { hipLaunchKernelGGL(read_kernel, dim3(wg), dim3(256), 0, 0, a, n, out); });
        const float c = timed([&] { hipLaunchKernelGGL(copy_kernel, dim3(wg), dim3(256), 0, 0, a, b, n); });
        printf("%5d workgroups x 256: write %.2f TB/s   read %.2f TB/s   copy %.2f TB/s (read + write bytes)\n", wg, bytes / w / 1e9,
               bytes / r / 1e9, 2.0 * bytes / c / 1e9);
    }
    return 0;
}
