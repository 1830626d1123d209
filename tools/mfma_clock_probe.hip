// Developer probe: what the matrix pipe sustains with nothing else in the way (no LDS, no memory): MFMA-only loops on every
// CU, v_mfma_f32_16x16x32_f16 against v_mfma_f32_32x32x16_f16, 1 or 2 waves per SIMD; reports the shader clock held during
// the loop (s_memtime cycles per s_memrealtime tick of 10 ns) and the achieved rate against the 2.5 PFLOP/s dense peak.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_clock_probe tools/mfma_clock_probe.hip && /tmp/mfma_clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE>
__global__ __launch_bounds__(512) void mfma_loop(float* out, unsigned long long* stamps, int iters) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
    unsigned long long c0, r0, c1, r1;
    float sink = 0.f;
    if (SHAPE == 16) {
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
        }
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
        for (int i = 0; i < 16; ++i) sink += acc[i][0];
    } else {
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i)
            for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(r0)::"memory");
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
        }
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c1), "=s"(r1)::"memory");
        for (int i = 0; i < 8; ++i) sink += acc[i][0];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = sink;
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = c1 - c0;
        stamps[2 * blockIdx.x + 1] = r1 - r0;
    }
}

// dependent chains: every MFMA accumulates into one of NACC accumulators in turn (NACC = 1: each MFMA waits for its predecessor;
// the attention kernel chains 3-12 MFMAs on one accumulator)
template <int NACC>
__global__ __launch_bounds__(1024) void mfma_chain32(float* out, int iters) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x - i)); }
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[i % NACC]) : "v"(a), "v"(b));
    }
    float sink = 0.f;
    for (int i = 0; i < NACC; ++i) sink += acc[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sink;
}

template <int NACC>
static void run_chain(float* out, int threads) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(mfma_chain32<NACC>, dim3(256), dim3(threads), 0, 0, out, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double flop = 8.0 * iters * 32.0 * 32 * 16 * 2 * (threads / 64) * 256;
    printf("32x32x16 chains on %d accumulator(s), %d wave(s) per SIMD: %.0f TFLOP/s (%.1f %% of 2500)\n", NACC, threads / 256, flop / ms / 1e9,
           flop / ms / 1e9 / 25.0);
}

int main() {
    float* out;
    unsigned long long* st;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&st, 256 * 16);
    unsigned long long h[512];
    const int iters = 20000;
    for (int shape : {16, 32}) {
        for (int threads : {256, 512}) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEvent_t e0, e1;
                hipEventCreate(&e0);
                hipEventCreate(&e1);
                hipEventRecord(e0, 0);
                if (shape == 16) hipLaunchKernelGGL(mfma_loop<16>, dim3(256), dim3(threads), 0, 0, out, st, iters);
                else hipLaunchKernelGGL(mfma_loop<32>, dim3(256), dim3(threads), 0, 0, out, st, iters);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                hipMemcpy(h, st, 256 * 16, hipMemcpyDeviceToHost);
                double cyc = 0, rt = 0;
                for (int i = 0; i < 256; ++i) { cyc += h[2 * i]; rt += h[2 * i + 1]; }
                const double mfmas = (shape == 16 ? 16.0 : 8.0) * iters;          // per wave
                const double flop = mfmas * (shape == 16 ? 16.0 * 16 * 32 * 2 : 32.0 * 32 * 16 * 2) * (threads / 64) * 256;
                if (rep == 1)
                    printf("%dx%d, %d wave(s) per SIMD: %.3f ms  %.0f TFLOP/s (%.1f %% of 2500)  clock %.2f GHz  %.1f cycles per MFMA and wave\n",
                           shape, shape, threads / 256, ms, flop / ms / 1e9, flop / ms / 1e9 / 25.0, cyc / rt / 10.0,
                           cyc / 256 / mfmas);
            }
        }
    }
    float* out2;
    hipMalloc(&out2, 256 * 1024 * 4);
    for (int threads : {256, 512, 1024}) {
        run_chain<1>(out2, threads);
        run_chain<2>(out2, threads);
        run_chain<8>(out2, threads);
    }
    return 0;
}
