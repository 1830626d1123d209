// Developer probe: which CUs / XCDs does a CU-masked stream (hipExtStreamCreateWithCUMask) dispatch to on MI355X?
// Launches a census kernel (one wave per workgroup, 2048 workgroups spinning briefly so that they spread over every
// enabled CU) on streams with different mask patterns and prints the (XCC_ID, SE, CU) population it saw.
//   hipcc --offload-arch=gfx950 -O2 -o build/cumask_probe tools/cumask_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <map>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void census(uint32_t* out, int spin) {
    uint32_t xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) {}
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
}

static int run(const char* name, const std::vector<uint32_t>& mask) {
    hipStream_t s;
    if (mask.empty()) CK(hipStreamCreate(&s));
    else CK(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
    const int n = 4096;
    uint32_t* d;
    CK(hipMalloc(&d, n * 8));
    hipLaunchKernelGGL(census, dim3(n), dim3(64), 0, s, d, 20000);
    CK(hipStreamSynchronize(s));
    std::vector<uint32_t> h(2 * n);
    CK(hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost));
    std::map<uint32_t, std::set<uint32_t>> per_xcc;
    for (int i = 0; i < n; ++i) {
        const uint32_t xcc = h[2 * i] & 0xF, hw = h[2 * i + 1];
        const uint32_t cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;
        per_xcc[xcc].insert((se << 8) | (sh << 4) | cu);
    }
    int total = 0;
    printf("%-34s:", name);
    for (auto& kv : per_xcc) { printf(" xcc%u:%zu", kv.first, kv.second.size()); total += (int)kv.second.size(); }
    printf("  -> %d CUs\n", total);
    // first 16 workgroups: which XCC did block i land on?
    printf("    blocks 0..15 on xcc:");
    for (int i = 0; i < 16; ++i) printf(" %u", h[2 * i] & 0xF);
    printf("\n");
    CK(hipFree(d));
    CK(hipStreamDestroy(s));
    return 0;
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    printf("device: %s, %d CUs\n", p.name, p.multiProcessorCount);
    const int words = 8;  // 256 bits
    if (run("no mask", {})) return 1;
    std::vector<uint32_t> m(words, 0);
    for (int i = 0; i < 128; ++i) m[i / 32] |= 1u << (i % 32);
    if (run("bits 0..127", m)) return 1;
    m.assign(words, 0);
    for (int i = 0; i < 256; i += 2) m[i / 32] |= 1u << (i % 32);
    if (run("even bits", m)) return 1;
    m.assign(words, 0);
    for (int i = 0; i < 256; ++i) if ((i / 8) % 2 == 0) m[i / 32] |= 1u << (i % 32);
    if (run("bits with (i/8) even", m)) return 1;
    m.assign(words, 0);
    for (int i = 0; i < 256; ++i) if ((i / 16) % 2 == 0) m[i / 32] |= 1u << (i % 32);
    if (run("bits with (i/16) even", m)) return 1;
    m.assign(words, 0);
    for (int i = 0; i < 256; ++i) if ((i / 8) % 2 == 1) m[i / 32] |= 1u << (i % 32);
    if (run("bits with (i/8) odd", m)) return 1;
    return 0;
}
