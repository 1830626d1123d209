// Developer probe: what a pure store stream reaches on an MI355X, by access shape (round-2 review, item 6a: the 3.9-5.1 TB/s of
// tools/hbm_bw_probe.hip is a property of its grid-stride loop, the microarch guide records 6.0-6.2 TB/s for plain stores).
//   hipcc --offload-arch=gfx950 -O3 -o build/store_bw_probe tools/store_bw_probe.hip && build/store_bw_probe
// Every wave writes whole KiB (64 lanes x 16 B per instruction).  Varied: contiguous bytes per wave before it jumps (CHUNK), waves
// per workgroup, workgroups per CU, persistent stripes against a grid-stride walk, plain / non-temporal stores, buffer size
// (2 GiB: beyond the 256-MiB Infinity Cache; 192 MiB: inside it).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float v4 __attribute__((ext_vector_type(4)));

// wave `gw` of `nw` writes chunks gw, gw + nw, ... of `chunk_kib` KiB each
template <bool NT>
__global__ void store_chunks(v4* dst, size_t bytes, int chunk_kib, float v) {
    const int lane = threadIdx.x & 63;
    const size_t gw = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
    const size_t chunk = (size_t)chunk_kib << 10, nchunks = bytes / chunk;
    const v4 val = {v, v, v, v};
    for (size_t c = gw; c < nchunks; c += nw) {
        v4* p = dst + (c * chunk) / 16 + lane;
        for (int k = 0; k < chunk_kib; ++k) {
            if (NT) __builtin_nontemporal_store(val, p + k * 64);
            else p[k * 64] = val;
        }
    }
}
// every workgroup owns one contiguous stripe of the buffer, its waves walk it KiB by KiB side by side
template <bool NT>
__global__ void store_stripes(v4* dst, size_t bytes, float v) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, waves = blockDim.x >> 6;
    const size_t stripe = bytes / gridDim.x;
    v4* base = dst + ((size_t)blockIdx.x * stripe) / 16;
    const v4 val = {v, v, v, v};
    for (size_t kib = wave; kib < (stripe >> 10); kib += waves) {
        if (NT) __builtin_nontemporal_store(val, base + kib * 64 + lane);
        else base[kib * 64 + lane] = val;
    }
}

// the store pattern of conv0_kernel without its arithmetic: workgroup (block of 128 frames, utterance), 4 waves, a wave writes
// the 1-KiB rows of frames f, f + 1 (f = 2 wave, step 8) to the hi plane and to the lo plane `plane` elements behind it;
// FPW consecutive frames per wave instead of 2 to see what longer contiguous runs per wave buy
template <int FPW>
__global__ void store_conv0_like(v4* dst, int T1, size_t plane_v4, float v) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.y, f0 = blockIdx.x * 128;
    const v4 val = {v, v, v, v};
    for (int f = wave * FPW; f < 128; f += 4 * FPW)
        for (int u = 0; u < FPW; ++u) {
            const int t = f0 + f + u;
            if (t < T1) {
                v4* p = dst + ((size_t)n * T1 + t) * 64 + lane;
                *p = val;
                p[plane_v4] = val;
            }
        }
}

template <typename F>
static double timed_ms(F f) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    f();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < 5; ++i) f();
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms / 5;
}

int main() {
    const size_t big = 2ull << 30, small = 192ull << 20;
    v4* a;
    (void)hipMalloc(&a, big);
    (void)hipMemset(a, 0, big);
    int cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) == hipSuccess) cus = prop.multiProcessorCount;
    {
        const int N = 32, T1 = 31999;
        const size_t plane_v4 = (size_t)N * T1 * 64;  // 1 KiB rows
        const double bytes = 2.0 * plane_v4 * 16;
        const dim3 grid((T1 + 127) / 128, N);
        const double t2 = timed_ms([&] { hipLaunchKernelGGL(store_conv0_like<2>, grid, dim3(256), 0, 0, a, T1, plane_v4, 1.f); });
        const double t8 = timed_ms([&] { hipLaunchKernelGGL(store_conv0_like<8>, grid, dim3(256), 0, 0, a, T1, plane_v4, 1.f); });
        const double t32 = timed_ms([&] { hipLaunchKernelGGL(store_conv0_like<32>, grid, dim3(256), 0, 0, a, T1, plane_v4, 1.f); });
        printf("conv0-like stores (32 x 31999 frames, two planes of 1-KiB rows, %.2f GB): 2 frames per wave and hop %.3f ms = %.2f TB/s,"
               " 8 frames %.3f ms = %.2f TB/s, 32 frames %.3f ms = %.2f TB/s\n", bytes / 1e9, t2, bytes / t2 / 1e9, t8, bytes / t8 / 1e9,
               t32, bytes / t32 / 1e9);
    }
    for (size_t bytes : {big, small}) {
        printf("---- %zu MiB buffer ----\n", bytes >> 20);
        for (int threads : {256, 512, 1024})
            for (int per_cu : {1, 2, 4})
                for (int chunk : {1, 4, 16, 64}) {
                    if ((threads >> 6) * per_cu > 32) continue;
                    const dim3 grid(cus * per_cu), block(threads);
                    const double p = timed_ms([&] { hipLaunchKernelGGL(store_chunks<false>, grid, block, 0, 0, a, bytes, chunk, 1.f); });
                    const double n = timed_ms([&] { hipLaunchKernelGGL(store_chunks<true>, grid, block, 0, 0, a, bytes, chunk, 1.f); });
                    printf("chunks  %4d thr x %d WG/CU, %2d KiB per wave and hop: plain %.2f TB/s   nt %.2f TB/s\n", threads, per_cu, chunk,
                           bytes / p / 1e9, bytes / n / 1e9);
                }
        for (int threads : {256, 512, 1024})
            for (int per_cu : {1, 2, 4, 8}) {
                if ((threads >> 6) * per_cu > 32) continue;
                const dim3 grid(cus * per_cu), block(threads);
                const double p = timed_ms([&] { hipLaunchKernelGGL(store_stripes<false>, grid, block, 0, 0, a, bytes, 1.f); });
                const double n = timed_ms([&] { hipLaunchKernelGGL(store_stripes<true>, grid, block, 0, 0, a, bytes, 1.f); });
                printf("stripes %4d thr x %d WG/CU (%5.1f MiB per workgroup): plain %.2f TB/s   nt %.2f TB/s\n", threads, per_cu,
                       (double)bytes / grid.x / 1048576.0, bytes / p / 1e9, bytes / n / 1e9);
            }
    }
    return 0;
}
