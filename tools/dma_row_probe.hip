// Developer probe: does the row-segment size of the operand DMA matter?  The ping-pong GEMM fetches every operand row in
// 64-byte segments (32 K-elements of one 16-bit plane: half a 128-byte cache line per request); an interleaved plane layout
// would make them 128 bytes.  This kernel only moves bytes L2 -> LDS with buffer_load_dwordx4 ... lds, the GEMM's way:
// every workgroup (8 waves) walks `rows` x `k_bytes` panels (row stride `ld` bytes) slice by slice, each wave-instruction
// fetching 1 KiB as 16 rows x 64 B, 8 rows x 128 B or 4 rows x 256 B.  Same bytes, same footprint, different request shape.
//   hipcc --offload-arch=gfx950 -O3 -o build/dma_row_probe tools/dma_row_probe.hip && build/dma_row_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// SEG = bytes per row segment (64, 128, 256).  A slice = 512 rows x 64 B worth of bytes (32 KiB) = 32 pieces of 1 KiB; the 8
// waves fetch 4 pieces each per slice into a 4-slot ring (no consumer: the probe measures the fetch path alone).
template <int SEG>
__global__ __launch_bounds__(512) void dma_rows(const unsigned char* __restrict__ src, int64_t ld, int rows_per_panel, int slices,
                                                int panels, unsigned long long* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int LPR = SEG / 16;        // lanes per row segment
    constexpr int RPP = 64 / LPR;        // rows per 1-KiB piece
    constexpr int PIECES = 32 * 1024 / 1024;
    for (int panel = blockIdx.x; panel < panels; panel += gridDim.x) {
        // two distinct panels (1 MiB touched each): an L2-resident set, so that the probe isolates the L2 -> LDS path
        const unsigned char* base = src + (int64_t)(panel & 1) * rows_per_panel * ld;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, -1, 0x00020000);
        for (int s = 0; s < slices; ++s) {
            unsigned char* slot = smem + (s & 3) * 32768;
#pragma unroll
            for (int j = 0; j < PIECES / 8; ++j) {
                const int piece = wave * (PIECES / 8) + j;
                const int row = piece * RPP + lane / LPR;  // 0 .. rows_per_panel - 1: one segment of every row per slice
                const uint32_t off = (uint32_t)((int64_t)row * ld + (int64_t)s * SEG + (lane % LPR) * 16);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)(slot + piece * 1024), 16, off, 0, 0, 0);
            }
            if ((s & 3) == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (sink && threadIdx.x == 0 && blockIdx.x == 0) sink[0] = smem[lane];
}

template <typename F>
static double timed_ms(F f, int reps) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) f();
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

template <int SEG>
static void run(const unsigned char* buf, unsigned long long* sink, int cus) {
    // per slice a workgroup fetches 32 KiB; rows_per_panel x SEG bytes per slice => rows_per_panel = 32 KiB / SEG
    const int rows_per_panel = 32768 / SEG, slices = 32, panels = 256 * 32;
    const int64_t ld = 8192;  // 8-KiB rows: consecutive slices walk along the row (K direction)
    (void)hipFuncSetAttribute((const void*)dma_rows<SEG>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    const double ms = timed_ms([&] { hipLaunchKernelGGL(dma_rows<SEG>, dim3(cus), dim3(512), 131072, 0, buf, ld, rows_per_panel, slices, panels, sink); }, 20);
    const double bytes = (double)panels * slices * 32768.0;
    printf("row segments of %3d B: %.3f ms per launch, %.2f TB/s L2 -> LDS (%.1f GB per launch, 2 MiB touched)\n", SEG, ms, bytes / ms / 1e9,
           bytes / 1e9);
}

int main() {
    unsigned char* buf;
    unsigned long long* sink;
    const size_t bytes = (size_t)64 * 512 * 8192 + (1 << 20);
    (void)hipMalloc(&buf, bytes);
    (void)hipMemset(buf, 1, bytes);
    (void)hipMalloc(&sink, 64);
    int cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, 0) == hipSuccess) cus = prop.multiProcessorCount;
    for (int rep = 0; rep < 2; ++rep) {
        run<64>(buf, sink, cus);
        run<128>(buf, sink, cus);
        run<256>(buf, sink, cus);
    }
    return 0;
}
