// Hardware probe for gfx950: verifies the MFMA operand / accumulator lane maps the kernels in
// allophant_amd/csrc rely on, fp16 subnormal handling inside MFMA, and global_load_lds placement.
// Build: hipcc --offload-arch=gfx950 -O2 -o probe_gfx950 tools/probe_gfx950.hip ; run on an MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef _Float16 f16;
typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

// A is [M][K] row-major, B is given as Bt [N][K] row-major (K-contiguous), C [M][N]
template <typename T, typename V8>
__global__ void k_mfma16(const T* A, const T* Bt, float* C) {
    int l = threadIdx.x;
    V8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = A[(l & 15) * 32 + 8 * (l >> 4) + j];
        b[j] = Bt[(l & 15) * 32 + 8 * (l >> 4) + j];
    }
    f32x4 c = {0, 0, 0, 0};
    if constexpr (sizeof(T) == 2 && __is_same(T, f16))
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    else
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) C[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}

template <typename T, typename V8>
__global__ void k_mfma32(const T* A, const T* Bt, float* C) {
    int l = threadIdx.x;
    V8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = A[(l & 31) * 16 + 8 * (l >> 5) + j];
        b[j] = Bt[(l & 31) * 16 + 8 * (l >> 5) + j];
    }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0;
    if constexpr (__is_same(T, f16))
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    else
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}

// Chained: X = A(32x16) * B(16x32) ; then Z = Vt(32 x 32keys) * X(32keys x 32) with X's accumulator regs as B operand
// using the documented permuted k order: element j of lane half h of k-step s is X row 16s + 8(j>>2) + 4h + (j&3).
__global__ void k_chain(const f16* A, const f16* Bt, const f16* Vt /*[32 d][32 key]*/, float* Z) {
    int l = threadIdx.x;
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = A[(l & 31) * 16 + 8 * (l >> 5) + j];
        b[j] = Bt[(l & 31) * 16 + 8 * (l >> 5) + j];
    }
    f32x16 x;
    for (int r = 0; r < 16; ++r) x[r] = 0;
    x = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, x, 0, 0, 0);
    f32x16 z;
    for (int r = 0; r < 16; ++r) z[r] = 0;
    int h = l >> 5, d = l & 31;
    for (int s = 0; s < 2; ++s) {
        f16x8 p, v;
        for (int j = 0; j < 8; ++j) {
            p[j] = (f16)x[8 * s + j];
            int key = 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
            v[j] = Vt[d * 32 + key];
        }
        z = __builtin_amdgcn_mfma_f32_32x32x16_f16(v, p, z, 0, 0, 0);
    }
    // Z[d][query]: col = query = l&31, row = d = (r&3)+8(r>>2)+4h
    for (int r = 0; r < 16; ++r) Z[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + (l & 31)] = z[r];
}

__global__ void k_glds(const unsigned* src, unsigned* dst) {
    __shared__ __attribute__((aligned(16))) unsigned lds[64 * 4 * 2];
    int l = threadIdx.x;
    // lane l reads 16 B from src + (63-l)*16 bytes (reversed) -> expect LDS[l*4..] = src[(63-l)*4..]
    const unsigned* g = src + (63 - l) * 4;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + 256),
                                     (__attribute__((address_space(3))) void*)(lds + 256), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = 0; i < 8; ++i) dst[l * 8 + i] = lds[l * 8 + i];
}

__global__ void k_denorm(float* out) {
    int l = threadIdx.x;
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (f16)0.0f; b[j] = (f16)0.0f; }
    // A[row][k=0] = 2^-20 (fp16 subnormal), B[k=0][col] = 2^10 ; expect C = 2^-10 if subnormals are kept
    if ((l >> 4) == 0) { a[0] = (f16)9.5367431640625e-07f; b[0] = (f16)1024.0f; }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    if (l == 0) { out[0] = c[0]; out[1] = (float)a[0]; }
    // VALU conversion behaviour: f32 -> f16 of a value in the fp16 subnormal range
    float tiny = 3.0e-6f * (1 + l * 0.0f);
    f16 t = (f16)tiny;
    if (l == 0) { out[2] = (float)t; }
}

template <typename T>
static void fill(std::vector<T>& v, std::vector<float>& f, int n) {
    v.resize(n); f.resize(n);
    for (int i = 0; i < n; ++i) { int r = (rand() % 9) - 4; f[i] = (float)r; v[i] = (T)(float)r; }
}

template <typename T, typename V8>
static void run16(const char* name) {
    std::vector<T> A, B; std::vector<float> Af, Bf;
    fill(A, Af, 16 * 32); fill(B, Bf, 16 * 32);
    T *dA, *dB; float* dC;
    CK(hipMalloc(&dA, A.size() * sizeof(T))); CK(hipMalloc(&dB, B.size() * sizeof(T))); CK(hipMalloc(&dC, 256 * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * sizeof(T), hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * sizeof(T), hipMemcpyHostToDevice));
    k_mfma16<T, V8><<<1, 64>>>(dA, dB, dC);
    std::vector<float> C(256);
    CK(hipMemcpy(C.data(), dC, 256 * 4, hipMemcpyDeviceToHost));
    double err = 0;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        float s = 0; for (int k = 0; k < 32; ++k) s += Af[i * 32 + k] * Bf[j * 32 + k];
        err = fmax(err, fabs(s - C[i * 16 + j]));
    }
    printf("PROBE %s maxerr=%g %s\n", name, err, err == 0 ? "OK" : "MISMATCH");
}

template <typename T, typename V8>
static void run32(const char* name) {
    std::vector<T> A, B; std::vector<float> Af, Bf;
    fill(A, Af, 32 * 16); fill(B, Bf, 32 * 16);
    T *dA, *dB; float* dC;
    CK(hipMalloc(&dA, A.size() * sizeof(T))); CK(hipMalloc(&dB, B.size() * sizeof(T))); CK(hipMalloc(&dC, 1024 * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * sizeof(T), hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, B.data(), B.size() * sizeof(T), hipMemcpyHostToDevice));
    k_mfma32<T, V8><<<1, 64>>>(dA, dB, dC);
    std::vector<float> C(1024);
    CK(hipMemcpy(C.data(), dC, 1024 * 4, hipMemcpyDeviceToHost));
    double err = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        float s = 0; for (int k = 0; k < 16; ++k) s += Af[i * 16 + k] * Bf[j * 16 + k];
        err = fmax(err, fabs(s - C[i * 32 + j]));
    }
    printf("PROBE %s maxerr=%g %s\n", name, err, err == 0 ? "OK" : "MISMATCH");
}

int main() {
    srand(7);
    run16<f16, f16x8>("mfma_f32_16x16x32_f16");
    run16<bf16, bf16x8>("mfma_f32_16x16x32_bf16");
    run32<f16, f16x8>("mfma_f32_32x32x16_f16");
    run32<bf16, bf16x8>("mfma_f32_32x32x16_bf16");
    {   // chain
        std::vector<f16> A, B, V; std::vector<float> Af, Bf, Vf;
        fill(A, Af, 32 * 16); fill(B, Bf, 32 * 16); fill(V, Vf, 32 * 32);
        for (auto& x : Af) x = (float)((int)x % 2); for (size_t i = 0; i < A.size(); ++i) A[i] = (f16)Af[i];
        f16 *dA, *dB, *dV; float* dZ;
        CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dV, 2048)); CK(hipMalloc(&dZ, 4096));
        CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice));
        CK(hipMemcpy(dV, V.data(), 2048, hipMemcpyHostToDevice));
        k_chain<<<1, 64>>>(dA, dB, dV, dZ);
        std::vector<float> Z(1024);
        CK(hipMemcpy(Z.data(), dZ, 4096, hipMemcpyDeviceToHost));
        double err = 0;
        for (int d = 0; d < 32; ++d) for (int q = 0; q < 32; ++q) {
            float s = 0;
            for (int key = 0; key < 32; ++key) {
                float x = 0; for (int k = 0; k < 16; ++k) x += Af[key * 16 + k] * Bf[q * 16 + k];   // X[key][q]
                s += Vf[d * 32 + key] * x;
            }
            err = fmax(err, fabs(s - Z[d * 32 + q]));
        }
        printf("PROBE chain_acc_as_B_operand maxerr=%g %s\n", err, err == 0 ? "OK" : "MISMATCH");
    }
    {   // glds
        std::vector<unsigned> src(512), dst(512);
        for (int i = 0; i < 512; ++i) src[i] = i;
        unsigned *ds, *dd;
        CK(hipMalloc(&ds, 2048)); CK(hipMalloc(&dd, 2048));
        CK(hipMemcpy(ds, src.data(), 2048, hipMemcpyHostToDevice));
        k_glds<<<1, 64>>>(ds, dd);
        CK(hipMemcpy(dst.data(), dd, 2048, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) {
            if (dst[l * 4 + i] != (unsigned)((63 - l) * 4 + i)) ++bad;
            if (dst[256 + l * 4 + i] != (unsigned)(256 + (63 - l) * 4 + i)) ++bad;
        }
        printf("PROBE global_load_lds_dwordx4 lane-linear-dest bad=%d %s (dst[0..7]=%u %u %u %u %u %u %u %u)\n", bad,
               bad == 0 ? "OK" : "MISMATCH", dst[0], dst[1], dst[2], dst[3], dst[4], dst[5], dst[6], dst[7]);
    }
    {
        float* d; CK(hipMalloc(&d, 64));
        k_denorm<<<1, 64>>>(d);
        float h[4]; CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
        printf("PROBE f16 subnormal through MFMA: c=%g (expect %g if kept, 0 if flushed); a_as_f32=%g cvt(3e-6)=%g\n",
               h[0], ldexp(1.0, -10), h[1], h[2]);
    }
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("PROBE device %s CUs=%d clock=%d MHz memclk=%d MHz L2=%d\n", p.gcnArchName, p.multiProcessorCount,
           p.clockRate / 1000, p.memoryClockRate / 1000, p.l2CacheSize);
    return 0;
}
