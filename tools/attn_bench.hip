// Developer micro-benchmark of the attention kernel (includes the translation unit directly so that diagnostic macros apply):
// time per launch at the config-2 geometry and, with -DAMX_ATTN_STAMP, cycles per phase of a key tile.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iallophant_amd/csrc -Iinclude [-DAMX_ATTN_STAMP] -o build/attn_bench tools/attn_bench.hip
//   build/attn_bench [N] [T]
#include "../allophant_amd/csrc/amx_attention.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace amx;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 32, T = argc > 2 ? atoi(argv[2]) : 499, H = 16;
    const int Tp = (T + 63) / 64 * 64;
    const size_t plane = (size_t)N * H * Tp * 64;
    std::vector<unsigned short> h(2 * plane);
    srand(1);
    for (size_t i = 0; i < 2 * plane; ++i) {  // f16: hi plane ~ +-[0.25, 2) * 2^-2, lo plane ~ 2^-11 of that
        const int lo = i >= plane;
        h[i] = (unsigned short)(((rand() & 1) << 15) | (((lo ? 1 : 11) + rand() % 3) << 10) | (rand() & 1023));
    }
    void *q, *k, *v, *out;
    int* fl;
    CK(hipMalloc(&q, 2 * plane * 2)); CK(hipMalloc(&k, 2 * plane * 2)); CK(hipMalloc(&v, 2 * plane * 2));
    CK(hipMalloc(&out, (size_t)2 * N * T * H * 64 * 2));
    CK(hipMalloc(&fl, N * 4));
    CK(hipMemcpy(q, h.data(), 2 * plane * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(k, h.data(), 2 * plane * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(v, h.data(), 2 * plane * 2, hipMemcpyHostToDevice));
    std::vector<int> lens(N, T);
    CK(hipMemcpy(fl, lens.data(), N * 4, hipMemcpyHostToDevice));
    AttnParams p{};
    p.q = q; p.k = k; p.v = v; p.qk_plane = (int64_t)plane; p.out = out; p.out_plane = (int64_t)N * T * H * 64;
    p.frame_len = fl; p.N = N; p.H = H; p.T = T; p.Tp = Tp; p.dh = 64;
    const int qblocks = (T + 255) / 256, wgs = 8 * ((N * H + 7) / 8) * qblocks;
#ifdef AMX_ATTN_STAMP
    unsigned long long* st;
    CK(hipMalloc(&st, (size_t)wgs * 8 * 12 * 8));
    CK(hipMemset(st, 0, (size_t)wgs * 8 * 12 * 8));
    p.stamps = st;
#endif
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) launch_attention(PREC_F16X3, p, 0);
    CK(hipDeviceSynchronize());
    hipEventRecord(a, 0);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) launch_attention(PREC_F16X3, p, 0);
    hipEventRecord(b, 0);
    CK(hipEventSynchronize(b));
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double flop = 4.0 * N * H * (double)T * T * 64;
    printf("attention f16x3 N=%d T=%d: %.1f us per launch, %.0f TFLOP/s algorithmic (x3 issued: %.0f)\n", N, T, ms * 1e3 / reps,
           flop / (ms / reps) / 1e9, 3 * flop / (ms / reps) / 1e9);
#ifdef AMX_ATTN_STAMP
    std::vector<unsigned long long> hs((size_t)wgs * 8 * 12);
    CK(hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost));
    double s[9] = {0};
    double cnt = 0, tiles = 0;
    unsigned long long t_first = ~0ull, t_last = 0;
    for (size_t w = 0; w < (size_t)wgs * 8; ++w) {
        if (!hs[w * 12 + 9]) continue;
        for (int i = 0; i < 8; ++i) s[i] += (double)hs[w * 12 + i];
        tiles += (double)hs[w * 12 + 8];
        cnt += 1;
        if (hs[w * 12 + 10] < t_first) t_first = hs[w * 12 + 10];
        if (hs[w * 12 + 11] > t_last) t_last = hs[w * 12 + 11];
    }
    // timeline of the last launch: wave start times and lifetimes in us (100 MHz clock), in 10 buckets of the launch
    {
        const double span = (double)(t_last - t_first) / 100.0;
        int started[10] = {0}, ended[10] = {0};
        double life = 0, life_early = 0, life_late = 0;
        int n_early = 0, n_late = 0;
        for (size_t w = 0; w < (size_t)wgs * 8; ++w) {
            if (!hs[w * 12 + 9]) continue;
            const double t0 = (double)(hs[w * 12 + 10] - t_first) / 100.0, t1 = (double)(hs[w * 12 + 11] - t_first) / 100.0;
            int b0 = (int)(t0 / span * 10), b1 = (int)(t1 / span * 10);
            started[b0 > 9 ? 9 : b0]++;
            ended[b1 > 9 ? 9 : b1]++;
            life += t1 - t0;
            if (t0 < span * 0.25) { life_early += t1 - t0; n_early++; } else { life_late += t1 - t0; n_late++; }
        }
        printf("timeline: first wave start to last wave end %.1f us; mean wave lifetime %.1f us (started in the first quarter: %.1f us x %d, "
               "later: %.1f us x %d)\n", span, life / cnt, n_early ? life_early / n_early : 0.0, n_early, n_late ? life_late / n_late : 0.0, n_late);
        {
            double cyc = 0, rt = 0;
            for (size_t w = 0; w < (size_t)wgs * 8; ++w) {
                if (!hs[w * 12 + 9]) continue;
                cyc += (double)(hs[w * 12 + 6] + hs[w * 12 + 7]);
                rt += (double)(hs[w * 12 + 11] - hs[w * 12 + 10]);
            }
            printf("shader clock held over the wave lifetimes: %.2f GHz (s_memtime cycles per 10 ns tick of s_memrealtime)\n", cyc / rt / 10.0);
        }
        printf("waves started per tenth of the span:");
        for (int i = 0; i < 10; ++i) printf(" %d", started[i]);
        printf("\nwaves ended   per tenth of the span:");
        for (int i = 0; i < 10; ++i) printf(" %d", ended[i]);
        printf("\n");
    }
    printf("per key tile and wave (cycles, %0.f waves, stamps cost ~40 each): S (LDS reads + MFMA issue) %.0f | mask / max / rescale "
           "(waits for the S MFMAs) %.0f | exp + sum %.0f | P split + V reads + PV issue %.0f | vmcnt/lgkmcnt wait %.0f | barrier %.0f\n",
           cnt, s[0] / tiles, s[1] / tiles, s[2] / tiles, s[3] / tiles, s[4] / tiles, s[5] / tiles);
    printf("per wave: prologue + main loop %.0f cycles, epilogue %.0f cycles, %.1f tiles\n", s[6] / cnt, s[7] / cnt, tiles / cnt);
#endif
    return 0;
}
