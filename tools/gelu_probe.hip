// Developer probe: accuracy (against float64 on the host) and cost (cycles per value and wave) of the GELU used by the
// epilogues, beside the Abramowitz-Stegun 7.1.26 form it replaced.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=fast -I allophant_amd/csrc -o /tmp/gelu_probe tools/gelu_probe.hip && /tmp/gelu_probe
#include "amx_common.h"
#include <cmath>
#include <cstdio>
#include <vector>

using namespace amx;

__device__ __forceinline__ float gelu_as(float x) {  // round-1 form: v_rcp_f32 + v_exp_f32
    const float ax = fabsf(x);
    const float z = ax * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    const float e = __builtin_amdgcn_exp2f(-(z * z) * 1.4426950408889634f);
    float q = fmaf(t, 1.061405429f, -1.453152027f);
    q = fmaf(t, q, 1.421413741f);
    q = fmaf(t, q, -0.284496736f);
    q = fmaf(t, q, 0.254829592f);
    q *= t;
    const float erf_abs = fmaf(-q, e, 1.0f);
    return fmaf(0.5f * ax, erf_abs, 0.5f * x);
}

template <int WHICH>
__global__ void eval_kernel(const float* x, float* y, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (WHICH == 0) y[i] = gelu_as(x[i]);
    else if (WHICH == 1) y[i] = gelu_fast(x[i]);
    else {
        const f32x2 r = gelu_fast2(f32x2{x[i], x[i ^ 1]});
        y[i] = r[0];
    }
}

// 16 independent values per lane, `iters` rounds: cycles per value and wave
template <int WHICH>
__global__ void cost_kernel(float* out, unsigned long long* cycles, int iters, float seed) {
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = seed + 0.01f * (threadIdx.x + 64 * i);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (WHICH == 2) {
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                const f32x2 r = gelu_fast2(f32x2{v[i], v[i + 1]});
                v[i] = r[0] - 0.3f;
                v[i + 1] = r[1] - 0.3f;
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = (WHICH == 0 ? gelu_as(v[i]) : gelu_fast(v[i])) - 0.3f;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

int main() {
    const int n = 1 << 22;
    std::vector<float> hx(n), hy(n);
    for (int i = 0; i < n; ++i) hx[i] = -9.f + 18.f * (float)i / (float)(n - 1);
    float *dx, *dy;
    unsigned long long* dc;
    hipMalloc(&dx, n * 4);
    hipMalloc(&dy, n * 4);
    hipMalloc(&dc, 8);
    hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
    const char* names[3] = {"Abramowitz-Stegun (rcp + exp)", "exp-only, scalar", "exp-only, packed pairs"};
    for (int which = 0; which < 3; ++which) {
        if (which == 0) hipLaunchKernelGGL(eval_kernel<0>, dim3(n / 256), dim3(256), 0, 0, dx, dy, n);
        if (which == 1) hipLaunchKernelGGL(eval_kernel<1>, dim3(n / 256), dim3(256), 0, 0, dx, dy, n);
        if (which == 2) hipLaunchKernelGGL(eval_kernel<2>, dim3(n / 256), dim3(256), 0, 0, dx, dy, n);
        hipMemcpy(hy.data(), dy, n * 4, hipMemcpyDeviceToHost);
        double worst = 0, at = 0;
        for (int i = 0; i < n; ++i) {
            const double x = hx[i], ref = 0.5 * x * (1.0 + erf(x / sqrt(2.0)));
            const double e = fabs((double)hy[i] - ref);
            if (e > worst) { worst = e; at = x; }
        }
        unsigned long long cyc = 0;
        const int iters = 2000;
        // one wave per SIMD (256 threads on one CU), then four waves per SIMD
        for (int waves = 1; waves <= 4; waves *= 4) {
            const dim3 block(256 * waves > 1024 ? 1024 : 256 * waves);
            if (which == 0) hipLaunchKernelGGL(cost_kernel<0>, dim3(1), block, 0, 0, dy, dc, iters, 0.5f);
            if (which == 1) hipLaunchKernelGGL(cost_kernel<1>, dim3(1), block, 0, 0, dy, dc, iters, 0.5f);
            if (which == 2) hipLaunchKernelGGL(cost_kernel<2>, dim3(1), block, 0, 0, dy, dc, iters, 0.5f);
            hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost);
            printf("%-32s max abs error %.3e at x = %+.4f   %d wave(s) per SIMD: %.1f memtime ticks per value and wave\n", names[which], worst,
                   at, waves, (double)cyc / (iters * 16.0));
        }
    }
    return 0;
}
