// Probe: rounding behaviour of the fp32 accumulator inside v_mfma_f32_16x16x32_f16 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void k(float* out, float cin, float av, float bv, int nk) {
    int l = threadIdx.x;
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (f16)0.f; b[j] = (f16)0.f; }
    // nk products of av*bv along k (spread over lanes groups / elements)
    int cnt = 0;
    for (int g = 0; g < 4; ++g) for (int j = 0; j < 8; ++j) { if (cnt < nk && (l >> 4) == g) { a[j] = (f16)av; b[j] = (f16)bv; } ++cnt; }
    f32x4 c = {cin, cin, cin, cin};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    if (l == 0) out[0] = c[0];
}
int main() {
    float* d; hipMalloc(&d, 64);
    struct { float c, a, b; int nk; const char* what; } cases[] = {
        {1.0f, 0.75f, ldexpf(1.f, -23), 1, "1 + 0.75ulp      (RNE -> 1+1ulp, RTZ -> 1)"},
        {1.0f, 0.5f, ldexpf(1.f, -23), 1, "1 + 0.5ulp (tie) (RNE-even -> 1)"},
        {1.0f, 0.25f, ldexpf(1.f, -23), 1, "1 + 0.25ulp      (-> 1)"},
        {1.0f, 0.25f, ldexpf(1.f, -23), 4, "1 + 4*0.25ulp    (exact sum -> 1+1ulp; sequential fp32 rounding -> 1)"},
        {1.0f, 0.25f, ldexpf(1.f, -23), 3, "1 + 3*0.25ulp    (exact-then-RNE -> 1+1ulp; RTZ -> 1)"},
        {-1.0f, -0.75f, ldexpf(1.f, -23), 1, "-1 - 0.75ulp    (RNE -> -(1+1ulp), RTZ -> -1)"},
        {1.0f, -0.25f, ldexpf(1.f, -23), 1, "1 - 0.25ulp(=0.5 ulp below) (RNE -> 1; RTZ -> 1-0.5ulp)"},
    };
    for (auto& cs : cases) {
        k<<<1, 64>>>(d, cs.c, cs.a, cs.b, cs.nk);
        float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("ROUND %-70s got %.9g  (delta/ulp = %g)\n", cs.what, h, (h - cs.c) / ldexp(1.0, -23));
    }
    return 0;
}
