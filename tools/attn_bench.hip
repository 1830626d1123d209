// Developer micro-benchmark of the attention kernel (includes the translation unit directly so that diagnostic macros apply):
// time per launch at the config-2 geometry and, with -DAMX_ATTN_STAMP, cycles per phase of a key tile.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iallophant_amd/csrc -Iinclude [-DAMX_ATTN_STAMP] -o build/attn_bench tools/attn_bench.hip
//   build/attn_bench [N] [T]
#include "../allophant_amd/csrc/amx_attention.hip"
#include "experiments/attn3_pingpong.inc"
#include "experiments/attn4_subblock_pipeline.inc"
#include "experiments/attn5_one_wave_per_simd.inc"
#include "experiments/attn6_straddled_softmax.inc"
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <cstring>
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
    const double flop = 4.0 * N * H * (double)T * T * 64;
    const int reps = 20;
    // attn_kernel (256-query workgroups, 32 queries per wave) against attn2_kernel (persistent 512-query workgroups, 64 queries
    // per wave): outputs compared element by element (different softmax block sizes: close, not bitwise), then timed
    const size_t out_elems = (size_t)2 * N * T * H * 64;
    std::vector<unsigned short> o1(out_elems), o2(out_elems);
    const int v2_waves = getenv("ATTN2_WAVES") ? atoi(getenv("ATTN2_WAVES")) : 8;
    const int base_waves = getenv("ATTN_BASE_WAVES") ? atoi(getenv("ATTN_BASE_WAVES")) : 8;  // 4: the short-batch form of attn_kernel
    auto run1 = [&]() {
        if (base_waves == 2) launch_attn2<f16, 2, 4, 2>(p, 256, 0);  // the library's long-utterance kernel as the baseline
        else if (base_waves == 4) launch_attn<f16, 2, 4, 64>(p, 0);
        else launch_attn<f16, 2, 8, 64>(p, 0);
    };
    auto run2 = [&]() {
        if (v2_waves == 2) launch_attn<f16, 2, 4, 64, 2>(p, 0);  // attn_kernel with the key tiles split over two wave groups
        else if (v2_waves == 4) launch_attn2<f16, 2, 4, 2>(p, 256, 0);
        else if (v2_waves == 3) launch_attn4<f16, 2>(p, 256, 0);
        else if (v2_waves == 5) launch_attn3<f16, 2>(p, 256, 0);
        else if (v2_waves == 6) launch_attn5<f16, 2>(p, 256, 0);
        else if (v2_waves == 7) launch_attn6<f16, 2>(p, 256, 0);
        else launch_attn2<f16, 2, 8, 4>(p, 256, 0);
    };
    CK(hipMemset(out, 0, out_elems * 2));
    run1();
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(o1.data(), out, out_elems * 2, hipMemcpyDeviceToHost));
    CK(hipMemset(out, 0, out_elems * 2));
    run2();
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(o2.data(), out, out_elems * 2, hipMemcpyDeviceToHost));
    {
        auto h2f = [](unsigned short h) {
            _Float16 x;
            memcpy(&x, &h, 2);
            return (float)x;
        };
        const size_t plane_o = (size_t)N * T * H * 64;
        double worst = 0, scale = 0;
        for (size_t i = 0; i < plane_o; ++i) {
            // interleaved planes are not used here (separate planes: out_plane != 32): value = hi + lo
            const double v1 = (double)h2f(o1[i]) + h2f(o1[i + plane_o]), v2 = (double)h2f(o2[i]) + h2f(o2[i + plane_o]);
            worst = fmax(worst, fabs(v1 - v2));
            scale = fmax(scale, fabs(v1));
        }
        printf("attn2 vs attn: max |difference| %.3e (largest output %.3f)%s\n", worst, scale,
               memcmp(o1.data(), o2.data(), out_elems * 2) == 0 ? " -- BITWISE equal" : "");
    }
    for (int variant = 0; variant < 2; ++variant) {
        for (int i = 0; i < 3; ++i) { if (variant) run2(); else run1(); }
        CK(hipDeviceSynchronize());
        hipEventRecord(a, 0);
        for (int i = 0; i < reps; ++i) { if (variant) run2(); else run1(); }
        hipEventRecord(b, 0);
        CK(hipEventSynchronize(b));
        float ms;
        hipEventElapsedTime(&ms, a, b);
        printf("%s f16x3 N=%d T=%d: %.1f us per launch, %.0f TFLOP/s algorithmic (x3 issued: %.0f)\n", variant ? (v2_waves == 7 ? "attn6 (one wave per SIMD, softmax straddling two MFMA groups)" : v2_waves == 6 ? "attn5 (one wave per SIMD, sub-blocks pipelined)" : v2_waves == 2 ? "attn<4 waves, key split 2>" : v2_waves == 4 ? "attn2<4 waves, 2 slots>" : v2_waves == 5 ? "attn3 (ping-pong groups)" : v2_waves == 3 ? "attn4 (sub-blocks pipelined)" : "attn2<8 waves, 4 slots>") : (base_waves == 2 ? "attn2<4 waves, 2 slots>" : base_waves == 4 ? "attn<4 waves>" : "attn<8 waves>"), N, T,
               ms * 1e3 / reps, flop / (ms / reps) / 1e9, 3 * flop / (ms / reps) / 1e9);
    }
#ifdef AMX_ATTN2_REPORT
    if (v2_waves == 6 || v2_waves == 7) {
        const int items5 = 8 * ((N * H + 7) / 8) * ((T + 255) / 256);
        CK(hipMemset(st, 0, (size_t)wgs * 8 * 12 * 8));
        run2();
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> h5((size_t)items5 * 4 * 8);
        CK(hipMemcpy(h5.data(), st, h5.size() * 8, hipMemcpyDeviceToHost));
        double ph[6] = {0}, tiles = 0;
        for (size_t w = 0; w < (size_t)items5 * 4; ++w) {
            const unsigned long long* o = &h5[w * 8];
            if (!o[7]) continue;
            for (int i = 0; i < 6; ++i) ph[i] += (double)o[i];
            tiles += (double)o[6];
        }
        if (v2_waves == 7)
            printf("attn6 cycles per 64-key tile and wave (stamped build): part I head %.0f | part I body %.0f | K reads / hand-off %.0f | part II head %.0f | part II body %.0f | sum %.0f\n",
                   ph[0] / tiles, ph[1] / tiles, ph[2] / tiles, ph[3] / tiles, ph[4] / tiles, (ph[0] + ph[1] + ph[2] + ph[3] + ph[4]) / tiles);
        else
        printf("attn5 cycles per 64-key tile and wave (stamped build): wait + barrier %.0f | K reads + DMA issue %.0f | A %.0f | B %.0f | C %.0f | D %.0f | sum %.0f\n",
               ph[0] / tiles, ph[1] / tiles, ph[2] / tiles, ph[3] / tiles, ph[4] / tiles, ph[5] / tiles,
               (ph[0] + ph[1] + ph[2] + ph[3] + ph[4] + ph[5]) / tiles);
    } else if (v2_waves == 5) {
        CK(hipMemset(st, 0, (size_t)wgs * 8 * 12 * 8));
        run2();
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> h3((size_t)256 * 8 * 8);
        CK(hipMemcpy(h3.data(), st, h3.size() * 8, hipMemcpyDeviceToHost));
        for (int g = 0; g < 2; ++g) {
            double ph[4] = {0}, tiles = 0;
            for (int wg = 0; wg < 256; ++wg)
                for (int w = 4 * g; w < 4 * g + 4; ++w) {
                    const unsigned long long* o = &h3[((size_t)wg * 8 + w) * 8];
                    for (int i = 0; i < 4; ++i) ph[i] += (double)o[i];
                    tiles += (double)o[4];
                }
            if (tiles > 0)
                printf("attn3 group %d, cycles per tile and wave: scores %.0f | barrier behind them %.0f | DMA + softmax + PV + tile wait %.0f (softmax part %.0f) | barrier %.0f\n",
                       g, ph[0] / tiles, ph[1] / tiles - 0, ph[2] / tiles, 0.0, ph[3] / tiles);
        }
    } else
    {   // attn2 item anatomy: the stamps of the last attn2 launch (items x waves x 8 words)
        const int w2 = v2_waves == 4 ? 4 : 8, qb2 = w2 * 64;
        const int items2 = 8 * ((N * H + 7) / 8) * ((T + qb2 - 1) / qb2);
        CK(hipMemset(st, 0, (size_t)wgs * 8 * 12 * 8));
        run2();
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> h2((size_t)items2 * w2 * 14);
        CK(hipMemcpy(h2.data(), st, h2.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long first = ~0ull, last = 0;
        double pro = 0, loop = 0, epi = 0, cyc = 0, tiles = 0, cnt = 0;
        for (size_t w = 0; w < (size_t)items2 * w2; ++w) {
            const unsigned long long* o = &h2[w * 8];
            if (!o[6]) continue;
            first = o[0] < first ? o[0] : first;
            last = o[3] > last ? o[3] : last;
            pro += (double)(o[1] - o[0]); loop += (double)(o[2] - o[1]); epi += (double)(o[3] - o[2]);
            cyc += (double)o[4]; tiles += (double)o[5]; cnt += 1;
        }
        printf("attn2 anatomy (%d waves): span %.1f us; per item and wave: prologue %.2f us, key loop %.2f us (%.2f us and %.0f cycles per tile, "
               "clock %.2f GHz), epilogue %.2f us\n", w2, (double)(last - first) / 100.0, pro / cnt / 100.0, loop / cnt / 100.0,
               loop / tiles / 100.0, cyc / tiles, cyc / loop / 10.0, epi / cnt / 100.0);
        {
            double ph[6] = {0}, ph_lo[6] = {0}, ph_hi[6] = {0}, t_lo = 0, t_hi = 0;
            for (size_t w = 0; w < (size_t)items2 * w2; ++w) {
                if (!h2[w * 8 + 6]) continue;
                const unsigned long long* q = &h2[(size_t)items2 * w2 * 8 + w * 6];
                const bool hi = (w % w2) >= (size_t)w2 / 2;
                for (int i = 0; i < 6; ++i) { ph[i] += (double)q[i]; (hi ? ph_hi : ph_lo)[i] += (double)q[i]; }
                (hi ? t_hi : t_lo) += (double)h2[w * 8 + 5];
            }
            printf("attn2 cycles per tile and wave (stamps ~40 each): DMA issue %.0f | scores %.0f | mask/max/exp %.0f | P split + PV %.0f | tile wait %.0f | "
                   "barrier %.0f\n", ph[0] / tiles, ph[1] / tiles, ph[2] / tiles, ph[3] / tiles, ph[4] / tiles, ph[5] / tiles);
            printf("   waves 0-3: %.0f %.0f %.0f %.0f %.0f %.0f   waves 4-7: %.0f %.0f %.0f %.0f %.0f %.0f\n", ph_lo[0] / t_lo, ph_lo[1] / t_lo,
                   ph_lo[2] / t_lo, ph_lo[3] / t_lo, ph_lo[4] / t_lo, ph_lo[5] / t_lo, ph_hi[0] / t_hi, ph_hi[1] / t_hi, ph_hi[2] / t_hi,
                   ph_hi[3] / t_hi, ph_hi[4] / t_hi, ph_hi[5] / t_hi);
        }
    }
#endif
#ifdef AMX_ATTN_STAMP_V1
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
