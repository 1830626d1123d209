// Developer micro-benchmark + self-check of the GEMM kernels of liballophant_amx (includes the translation unit directly so
// that ablation macros apply).  Build (-DAMX_DEVELOPER: the AMX_* A/B switches read the environment, e.g. AMX_PP_FORCE_NI=3):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DAMX_DEVELOPER -Iallophant_amd/csrc -Iinclude -o build/gemm_bench tools/gemm_bench.hip
//   hipcc ... -DAMX_ABLATE_NO_EPI -o build/gemm_bench_noepi tools/gemm_bench.hip
// Run:  build/gemm_bench check   (ping-pong kernel vs generic tile kernel on edge-case shapes)
//       build/gemm_bench time    (model shapes of BASELINE config 2, both kernels, f16x3 and bf16)
#include "../allophant_amd/csrc/amx_gemm.hip"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace amx;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

static float f16_to_f32(unsigned short h) {
    _Float16 v; memcpy(&v, &h, 2); return (float)v;
}
static float bf16_to_f32(unsigned short h) {
    unsigned int u = (unsigned int)h << 16; float f; memcpy(&f, &u, 4); return f;
}
static float to_f32(int prec, unsigned short h) { return (prec == PREC_F16 || prec == PREC_F16X3) ? f16_to_f32(h) : bf16_to_f32(h); }

// random 16-bit values: hi plane ~ +-[0.25, 2), lo plane ~ 2^-11 of that (f16) -- plausible split operands
static void fill16(void* d, size_t n, int seed, int prec, bool lo_plane) {
    std::vector<unsigned short> h(n);
    srand(seed);
    const bool f16 = prec == PREC_F16 || prec == PREC_F16X3;
    for (size_t i = 0; i < n; ++i) {
        int sign = rand() & 1;
        if (f16) {
            int e = (lo_plane ? 2 : 13) + rand() % 3;
            h[i] = (unsigned short)((sign << 15) | (e << 10) | (rand() & 1023));
        } else {
            int e = (lo_plane ? 116 : 125) + rand() % 3;
            h[i] = (unsigned short)((sign << 15) | (e << 7) | (rand() & 127));
        }
    }
    CK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice));
}
// the operand of a product: one plane, or (two-plane modes) the INTERLEAVED layout the kernels read (amx_common.h pidx():
// [hi x 32 | lo x 32] per 32 K elements); n = logical elements, a multiple of 32 in the two-plane modes
static void fill_operand(void* d, size_t n, int seed_hi, int seed_lo, int prec) {
    if (prec_planes(prec) == 1) { fill16(d, n, seed_hi, prec, false); return; }
    std::vector<unsigned short> hi(n), lo(n), out(2 * n);
    {
        void* tmp; CK(hipMalloc(&tmp, n * 2));
        fill16(tmp, n, seed_hi, prec, false); CK(hipMemcpy(hi.data(), tmp, n * 2, hipMemcpyDeviceToHost));
        fill16(tmp, n, seed_lo, prec, true); CK(hipMemcpy(lo.data(), tmp, n * 2, hipMemcpyDeviceToHost));
        CK(hipFree(tmp));
    }
    for (size_t i = 0; i < n; ++i) {
        const size_t ph = (size_t)pidx((int64_t)i, true);
        out[ph] = hi[i];
        out[ph + PLANE_IL] = lo[i];
    }
    CK(hipMemcpy(d, out.data(), 2 * n * 2, hipMemcpyHostToDevice));
}
static int64_t plane_of(int prec, size_t separate) { return prec_planes(prec) > 1 ? PLANE_IL : (int64_t)separate; }

static void fill32(float* d, size_t n, int seed, float scale) {
    std::vector<float> h(n);
    srand(seed);
    for (size_t i = 0; i < n; ++i) h[i] = scale * ((rand() % 2001) - 1000) / 1000.f;
    CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
}

// fp64 reference of the dense product (operands = sum of their planes): C[m][n] = scale * sum_k A[m][k] W[n][k] + bias[n] + res[m][n]
template <typename T>
__global__ void ref_kernel(const T* A, int64_t a_plane, const T* W, int64_t w_plane, int NT, int M, int N, int K, float scale,
                           const float* bias, const float* res, double* out) {
    int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N || m >= M) return;
    double acc = 0;
    const bool il = NT > 1;  // two planes: interleaved
    for (int k = 0; k < K; ++k) {
        const int64_t ia = pidx((int64_t)m * K + k, il), iw = pidx((int64_t)n * K + k, il);
        double a = (double)(float)A[ia], w = (double)(float)W[iw];
        if (NT > 1) { a += (double)(float)A[a_plane + ia]; w += (double)(float)W[w_plane + iw]; }
        acc += a * w;
    }
    out[(int64_t)m * N + n] = acc * scale + bias[n] + (res ? (double)res[(int64_t)m * N + n] : 0.0);
}

struct Case {
    const char* name;
    int M, N, K;
    int act, residual, mask, planes_out, f32_out, qkv;
    int conv_rows_per_batch, conv_lda;  // 0: dense
    float scale;
};

// variant 0 always gets a split-K workspace (launch_gemm decides whether to use it); variant 1 is the unsplit generic
// kernel, or -- with g_compare_nosplit -- the same automatic choice without a workspace
static bool g_compare_nosplit = false;
static float* g_splitk_ws = nullptr;
static const int64_t SPLITK_ELEMS = (int64_t)18 << 20;

static double run_case(int prec, const Case& c, bool timing_only, double* us_pp, double* us_gen) {
    if (!g_splitk_ws) CK(hipMalloc(&g_splitk_ws, SPLITK_ELEMS * 4));
    const int NT = prec_planes(prec);
    const int64_t rows_per_batch = c.conv_rows_per_batch ? c.conv_rows_per_batch : c.M;
    const int64_t lda = c.conv_lda ? c.conv_lda : c.K;
    const int64_t nbatch = (c.M + rows_per_batch - 1) / rows_per_batch;
    const int64_t a_batch_stride = c.conv_rows_per_batch ? ((rows_per_batch - 1) * lda + c.K + 32 * 3) : 0;
    const size_t a_el = c.conv_rows_per_batch ? (size_t)(nbatch * a_batch_stride) : (size_t)c.M * c.K;
    const size_t w_el = (size_t)c.N * c.K;
    const size_t o_el = (size_t)c.M * c.N;
    void *A, *W;
    CK(hipMalloc(&A, a_el * 2 * NT)); CK(hipMalloc(&W, w_el * 2 * NT));
    fill_operand(A, a_el, 1, 3, prec); fill_operand(W, w_el, 2, 4, prec);
    float *bias, *res; int* row_len;
    CK(hipMalloc(&bias, c.N * 4)); CK(hipMalloc(&res, o_el * 4));
    fill32(bias, c.N, 5, 0.5f); fill32(res, o_el, 6, 1.0f);
    const int T = 499, Tp = 512, dh = 64, H = c.N / 3 / dh;
    const int nb = (c.M + T - 1) / T;
    std::vector<int> rl(nb + 1);
    for (int i = 0; i <= nb; ++i) rl[i] = 300 + (i * 37) % 199;
    CK(hipMalloc(&row_len, (nb + 1) * 4)); CK(hipMemcpy(row_len, rl.data(), (nb + 1) * 4, hipMemcpyHostToDevice));
    const size_t qk_el = (size_t)nb * H * Tp * dh;

    double worst = 0;
    std::vector<std::vector<unsigned char>> results[2];
    for (int variant = 0; variant < 2; ++variant) {  // 0: ping-pong kernel, 1: generic kernel
        float* outf = nullptr; void* outp = nullptr; void *q = nullptr, *k = nullptr, *vt = nullptr;
        if (c.f32_out) { CK(hipMalloc(&outf, o_el * 4)); CK(hipMemset(outf, 0xEE, o_el * 4)); }
        if (c.planes_out) { CK(hipMalloc(&outp, o_el * 2 * NT)); CK(hipMemset(outp, 0xEE, o_el * 2 * NT)); }
        if (c.qkv) {
            CK(hipMalloc(&q, qk_el * 2 * NT)); CK(hipMalloc(&k, qk_el * 2 * NT)); CK(hipMalloc(&vt, qk_el * 2 * NT));
            CK(hipMemset(q, 0, qk_el * 2 * NT)); CK(hipMemset(k, 0, qk_el * 2 * NT)); CK(hipMemset(vt, 0, qk_el * 2 * NT));
        }
        GemmParams g{};
        g.A = A; g.a_plane = plane_of(prec, a_el); g.lda = lda; g.rows_per_batch = rows_per_batch; g.a_batch_stride = a_batch_stride;
        g.W = W; g.w_plane = plane_of(prec, w_el); g.ldw = c.K; g.M = c.M; g.N = c.N; g.K = c.K;
        g.scale = c.scale; g.bias = bias; g.act = c.act;
        if (c.residual) { g.residual = res; g.ldr = c.N; }
        if (c.mask) { g.row_len = row_len; g.rows_T = T; }
        if (c.f32_out) { g.out_f32 = outf; g.ldo = c.N; }
        if (c.planes_out) { g.out_p = outp; g.out_plane = plane_of(prec, o_el); g.ldp = c.N; }
        if (c.qkv) {
            g.mode = 1; g.q = q; g.k = k; g.v = vt; g.qk_plane = qk_el; g.T = T; g.Tp = Tp; g.H = H; g.dh = dh;
            if (c.mask) g.row_len = row_len;  // (the model's QKV product carries no row mask: the branch-free scatter epilogue)
        }
        g_force_generic_gemm = variant == 1 && !g_compare_nosplit;
        if (variant == 0) { g.splitk_ws = g_splitk_ws; g.splitk_ws_elems = SPLITK_ELEMS; }
        launch_gemm(prec, g, 0);
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int iters = timing_only ? 20 : 2;
        for (int i = 0; i < 2; ++i) launch_gemm(prec, g, 0);
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < iters; ++i) launch_gemm(prec, g, 0);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        (variant == 0 ? *us_pp : *us_gen) = ms * 1e3 / iters;
        if (!timing_only) {
            auto grab = [&](void* d, size_t bytes) {
                std::vector<unsigned char> h(bytes);
                if (d) CK(hipMemcpy(h.data(), d, bytes, hipMemcpyDeviceToHost));
                results[variant].push_back(std::move(h));
            };
            grab(outf, c.f32_out ? o_el * 4 : 0);
            grab(outp, c.planes_out ? o_el * 2 * NT : 0);
            grab(q, c.qkv ? qk_el * 2 * NT : 0);
            grab(k, c.qkv ? qk_el * 2 * NT : 0);
            grab(vt, c.qkv ? qk_el * 2 * NT : 0);
        }
        if (outf) CK(hipFree(outf)); if (outp) CK(hipFree(outp));
        if (q) { CK(hipFree(q)); CK(hipFree(k)); CK(hipFree(vt)); }
    }
    g_force_generic_gemm = false;
    if (!timing_only) {
        // f32 output: relative-to-magnitude error; planes: compare hi + lo as floats
        for (size_t r = 0; r < results[0].size(); ++r) {
            auto& x = results[0][r]; auto& y = results[1][r];
            if (x.empty()) continue;
            if (r == 0) {
                const float* a = (const float*)x.data(); const float* b = (const float*)y.data();
                for (size_t i = 0; i < x.size() / 4; ++i) {
                    double e = fabs((double)a[i] - b[i]) / (1.0 + fabs((double)b[i]));
                    if (!(e <= worst)) worst = (e == e) ? e : 1e30;
                }
            } else {
                const unsigned short* a = (const unsigned short*)x.data(); const unsigned short* b = (const unsigned short*)y.data();
                size_t n = x.size() / 2 / NT;
                // the output planes of a product are interleaved like its operands; Q / K / V (r >= 2) are separate planes
                const bool il = NT > 1 && r == 1;
                int shown = 0;
                for (size_t i = 0; i < n; ++i) {
                    const size_t ph = (size_t)pidx((int64_t)i, il), pl = il ? ph + PLANE_IL : n + i;
                    double va = to_f32(prec, a[ph]), vb = to_f32(prec, b[ph]);
                    if (NT > 1) { va += to_f32(prec, a[pl]); vb += to_f32(prec, b[pl]); }
                    double e = fabs(va - vb) / (1.0 + fabs(vb));
                    if (!(e <= worst)) worst = (e == e) ? e : 1e30;
                    if (getenv("GEMM_BENCH_VERBOSE") && !(e <= 0.05) && shown < 24) {  // developer diagnostic: where the outputs differ
                        printf("      buffer %zu element %zu (row %zu, col %zu of %d): ping-pong %.5g generic %.5g\n", r, i,
                               r == 1 ? i / c.N : i / 64, r == 1 ? i % c.N : i % 64, r == 1 ? c.N : 64, va, vb);
                        ++shown;
                    }
                }
            }
        }
    }
    if (!timing_only && c.f32_out && !c.mask && !c.act && !c.conv_rows_per_batch && !c.qkv) {
        double* truth; CK(hipMalloc(&truth, o_el * 8));
        dim3 grid((c.N + 127) / 128, c.M);
        const bool is_f16 = prec == PREC_F16 || prec == PREC_F16X3;
        if (is_f16) hipLaunchKernelGGL(ref_kernel<amx::f16>, grid, dim3(128), 0, 0, (const amx::f16*)A, plane_of(prec, a_el), (const amx::f16*)W, plane_of(prec, w_el), NT, c.M, c.N, c.K, c.scale, bias, c.residual ? res : nullptr, truth);
        else hipLaunchKernelGGL(ref_kernel<amx::bf16>, grid, dim3(128), 0, 0, (const amx::bf16*)A, plane_of(prec, a_el), (const amx::bf16*)W, plane_of(prec, w_el), NT, c.M, c.N, c.K, c.scale, bias, c.residual ? res : nullptr, truth);
        std::vector<double> t(o_el); CK(hipMemcpy(t.data(), truth, o_el * 8, hipMemcpyDeviceToHost));
        double e[2] = {0, 0};
        for (int v = 0; v < 2; ++v) {
            const float* a = (const float*)results[v][0].data();
            for (size_t i = 0; i < o_el; ++i) { double d = fabs((double)a[i] - t[i]) / (1.0 + fabs(t[i])); if (!(d <= e[v])) e[v] = d; }
        }
        printf("      vs fp64 truth: ping-pong %.3e, generic %.3e (max rel err)\n", e[0], e[1]);
        CK(hipFree(truth));
    }
    CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(bias)); CK(hipFree(res)); CK(hipFree(row_len));
    return worst;
}

#ifdef AMX_PP_STAMP
// phase anatomy of the ping-pong main loop: per-segment cycles of the LOAD / MFMA phases and the in-kernel clock
static void run_stamp(int prec, int M, int N, int K, const char* name) {
    const int NT = prec_planes(prec);
    size_t a_el = (size_t)M * K, w_el = (size_t)N * K, o_el = (size_t)M * N;
    void *A, *W; float *bias, *outf;
    CK(hipMalloc(&A, a_el * 2 * NT)); CK(hipMalloc(&W, w_el * 2 * NT)); CK(hipMalloc(&bias, N * 4)); CK(hipMalloc(&outf, o_el * 4));
    fill_operand(A, a_el, 1, 3, prec); fill_operand(W, w_el, 2, 4, prec);
    fill32(bias, N, 5, 0.5f);
    const int nblk = 256;  // persistent grid: at most one workgroup per CU
    unsigned long long* st; CK(hipMalloc(&st, (size_t)nblk * 20 * 8)); CK(hipMemset(st, 0, (size_t)nblk * 20 * 8));
    GemmParams g{};
    g.A = A; g.a_plane = plane_of(prec, a_el); g.lda = K; g.rows_per_batch = M; g.W = W; g.w_plane = plane_of(prec, w_el); g.ldw = K;
    g.M = M; g.N = N; g.K = K;
    g.scale = 1.f; g.bias = bias; g.out_f32 = outf; g.ldo = N; g.stamps = st;
    for (int i = 0; i < 30; ++i) launch_gemm(prec, g, 0);  // warm: let the clock settle under load
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h((size_t)nblk * 20);
    CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
    const int nseg = K / 32;  // LOAD + MFMA segment pairs per tile: one per 32-deep K slice in either mode
    for (int grp = 0; grp < 2; ++grp) {
        double s[9] = {0}; int cnt = 0; double tiles = 0;
        for (int b = 0; b < nblk; ++b) {
            const unsigned long long* o = &h[((size_t)b * 2 + grp) * 10];
            if (!o[7]) continue;
            for (int i = 0; i < 9; ++i) s[i] += (double)o[i];
            tiles += (double)o[9];
            ++cnt;
        }
        if (!cnt) continue;
        const double segs = tiles * nseg;  // segments summed over all sampled workgroups
        printf("STAMP prec=%d %-10s grp%d  per segment: load %.0f  bar1 %.0f  mfma %.0f  vmwait %.0f  bar2 %.0f | per tile: main loop %.0f cyc of %.0f total (%.1f tiles/WG)  clock %.0f MHz\n",
               prec, name, grp, s[0] / segs, s[1] / segs, s[2] / segs, s[3] / segs, s[4] / segs, s[5] / tiles, s[8] / tiles, tiles / cnt,
               s[8] / (s[6] / 100.0));
    }
    CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(bias)); CK(hipFree(outf)); CK(hipFree(st));
}
#endif

int main(int argc, char** argv) {
#ifdef AMX_PP_STAMP
    if (argc > 1 && !strcmp(argv[1], "stamp")) {
        int precs[] = {PREC_F16X3, PREC_BF16, PREC_F16};
        if (argc > 4) {  // stamp <M> <N> <K>: one shape, f16x3
            run_stamp(PREC_F16X3, atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), "custom");
            return 0;
        }
        for (int prec : precs) {
            run_stamp(prec, 15968, 4096, 1024, "ffn1-like");
            run_stamp(prec, 15968, 1024, 4096, "ffn2-like");
        }
        return 0;
    }
#endif
    if (argc > 6 && !strcmp(argv[1], "soak")) {
        // sustained load for tools/power_probe.sh: gemm_bench soak <prec> <M> <N> <K> <seconds>  (FFN1-like: GELU -> planes)
        const int prec = atoi(argv[2]), M = atoi(argv[3]), N = atoi(argv[4]), K = atoi(argv[5]);
        const double seconds = atof(argv[6]);
        const int NT = prec_planes(prec);
        size_t a_el = (size_t)M * K, w_el = (size_t)N * K, o_el = (size_t)M * N;
        void *A, *W, *outp; float* bias;
        CK(hipMalloc(&A, a_el * 2 * NT)); CK(hipMalloc(&W, w_el * 2 * NT)); CK(hipMalloc(&bias, N * 4)); CK(hipMalloc(&outp, o_el * 2 * NT));
        fill_operand(A, a_el, 1, 3, prec); fill_operand(W, w_el, 2, 4, prec);
        fill32(bias, N, 5, 0.5f);
        GemmParams g{};
        g.A = A; g.a_plane = plane_of(prec, a_el); g.lda = K; g.rows_per_batch = M; g.W = W; g.w_plane = plane_of(prec, w_el); g.ldw = K;
        g.M = M; g.N = N; g.K = K;
        g.scale = 1.f; g.bias = bias; g.act = 1; g.out_p = outp; g.out_plane = plane_of(prec, o_el); g.ldp = N;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        double total_ms = 0;
        while (total_ms < seconds * 1e3) {
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < 200; ++i) launch_gemm(prec, g, 0);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            total_ms += ms;
            printf("SOAK %.1f us per launch (200 launches)\n", ms * 1e3 / 200); fflush(stdout);
        }
        return 0;
    }
    const bool check = argc < 2 || !strcmp(argv[1], "check");
    const bool timing = argc < 2 || !strcmp(argv[1], "time");
    if (argc > 5 && !strcmp(argv[1], "one")) {
        // one shape, for rocprofv3 --kernel-trace --stats: gemm_bench one <prec> <M> <N> <K> [kind: 0 f32+res, 1 gelu->planes]
        const int prec = atoi(argv[2]), kind = argc > 6 ? atoi(argv[6]) : 0;
        Case c = {"one", atoi(argv[3]), atoi(argv[4]), atoi(argv[5]), kind, kind == 0, 0, kind == 1, kind == 0, 0, 0, 0, 1.f};
        g_compare_nosplit = true;
        double a = 0, b = 0;
        for (int rep = 0; rep < 5; ++rep) run_case(prec, c, true, &a, &b);
        printf("ONE prec=%d M=%d N=%d K=%d: with workspace %.1f us | without %.1f us\n", prec, c.M, c.N, c.K, a, b);
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "small")) {
        // products of short batches (rows = frames of 1 x 3 s, 1 x 10 s, 4 x 10 s, 1 x 60 s): split-K against no split
        g_compare_nosplit = true;
        const int rows[] = {149, 499, 1996, 2999, 3992, 7984};
        int precs[] = {PREC_F16X3, PREC_BF16};
        for (int prec : precs)
            for (int M : rows) {
                Case shapes[] = {
                    {"ffn1 (gelu -> planes)", M, 4096, 1024, 1, 0, 0, 1, 0, 0, 0, 0, 1.f},
                    {"ffn2 (+res -> f32)", M, 1024, 4096, 0, 1, 0, 0, 1, 0, 0, 0, 1.f},
                    {"qkv (scatter)", M, 3072, 1024, 0, 0, 0, 0, 0, 1, 0, 0, 1.f},
                    {"out-proj (+res -> f32)", M, 1024, 1024, 0, 1, 0, 0, 1, 0, 0, 0, 1.f},
                };
                for (auto& c : shapes) {
                    double a = 0, b = 0;
                    run_case(prec, c, true, &a, &b);
                    double fl = 2.0 * c.M * c.N * c.K;
                    printf("SMALL prec=%d %-24s M=%5d N=%d K=%d : split-K %.1f us %.1f TF/s | no split %.1f us %.1f TF/s\n", prec,
                           c.name, c.M, c.N, c.K, a, fl / a * 1e-6, b, fl / b * 1e-6);
                }
            }
        return 0;
    }
    if (check) {
        Case cases[] = {
            {"dense gelu->planes, M tail", 1500, 512, 256, 1, 0, 0, 1, 0, 0, 0, 0, 1.0f},
            {"dense f32 +res +mask, N tail", 2100, 640, 384, 0, 1, 1, 0, 1, 0, 0, 0, 0.5f},
            {"dense f32 + planes (no bias act)", 1024, 256, 128, 0, 0, 0, 1, 1, 0, 0, 0, 1.0f},
            {"conv-like overlapping rows", 2800, 512, 384, 0, 0, 0, 0, 1, 0, 700, 256, 1.0f},
            {"qkv scatter", 1497, 384, 256, 0, 0, 0, 0, 0, 1, 0, 0, 1.0f},
            {"qkv scatter + row mask", 1497, 384, 256, 0, 0, 1, 0, 0, 1, 0, 0, 1.0f},
            {"long K", 1100, 256, 4096, 0, 1, 0, 0, 1, 0, 0, 0, 1.0f},
            // enough tiles for the 256 x 256 kernel (the cases above run on 128 x 256 tiles)
            {"256-row tiles f32 +res +mask", 16000, 1024, 256, 0, 1, 1, 0, 1, 0, 0, 0, 1.0f},
            {"256-row tiles gelu->planes, M tail", 15968, 2048, 128, 1, 0, 0, 1, 0, 0, 0, 0, 1.0f},
            {"256-row tiles qkv scatter", 15968, 3072, 256, 0, 0, 0, 0, 0, 1, 0, 0, 1.0f},
            {"256-row tiles conv-like", 31996, 512, 1536, 0, 0, 0, 0, 1, 0, 7999, 1024, 1.0f},
            // split-K (few tiles): ping-pong kernel with K chunks, generic kernel with grid.z chunks, + fix-up epilogue
            {"split pp gelu->planes", 2000, 1024, 1024, 1, 0, 0, 1, 0, 0, 0, 0, 1.0f},
            {"split pp qkv scatter", 1996, 3072, 1024, 0, 0, 0, 0, 0, 1, 0, 0, 1.0f},
            {"split pp qkv scatter + row mask", 1996, 3072, 1024, 0, 0, 1, 0, 0, 1, 0, 0, 1.0f},
            {"split pp f32 +res +mask +planes", 1300, 512, 2048, 0, 1, 1, 1, 1, 0, 0, 0, 0.25f},
            {"split generic f32 +res +mask, N%4", 300, 1022, 1024, 0, 1, 1, 0, 1, 0, 0, 0, 0.5f},
            {"split generic gelu->planes", 149, 4096, 1024, 1, 0, 0, 1, 0, 0, 0, 0, 1.0f},
            {"split generic conv-like", 598, 512, 1536, 0, 0, 0, 0, 1, 0, 299, 1024, 1.0f},
            {"split generic qkv scatter", 499, 3072, 1024, 0, 0, 0, 0, 0, 1, 0, 0, 1.0f},
            // shapes the plan gives 192-column tiles (NI = 3), and -- under AMX_PP_FORCE_NI=3 -- their N tails
            {"192-wide qkv scatter 8 x 10 s", 3992, 3072, 1024, 0, 0, 0, 0, 0, 1, 0, 0, 1.0f},
            {"192-wide qkv scatter 16 x 10 s", 7984, 3072, 512, 0, 0, 0, 0, 0, 1, 0, 0, 1.0f},
            {"N tail f32 +res +mask", 2100, 640, 384, 0, 1, 1, 0, 1, 0, 0, 0, 0.5f},
            {"N tail gelu->planes", 4000, 1088, 256, 1, 0, 0, 1, 0, 0, 0, 0, 1.0f},
            {"N = 4 mod 192 f32 + planes", 2048, 580, 128, 0, 0, 0, 1, 1, 0, 0, 0, 1.0f},
        };
        int precs[] = {PREC_F16X3, PREC_BF16X3, PREC_F16, PREC_BF16};
        int bad = 0;
        for (int prec : precs)
            for (auto& c : cases) {
                double a, b;
                double e = run_case(prec, c, false, &a, &b);
                // x3: both kernels are ~fp32-exact; 1 plane: identical products, different summation order + fast GELU,
                // results quantised to 16 bit on plane outputs
                double tol = prec_planes(prec) > 1 ? 2e-4 : ((c.planes_out || c.qkv) ? (prec == PREC_BF16 ? 8e-3 : 1e-3) : 2e-4);
                printf("CHECK prec=%d %-36s max rel err %.3e  %s\n", prec, c.name, e, e <= tol ? "ok" : "FAIL");
                if (!(e <= tol)) ++bad;
            }
        printf("gemm check: %s\n", bad ? "FAILED" : "all ok");
        if (bad) return 1;
    }
    if (timing) {
        Case shapes[] = {
            {"ffn1 (gelu -> planes)", 15968, 4096, 1024, 1, 0, 0, 1, 0, 0, 0, 0, 1.f},
            {"ffn2 (+res -> f32)", 15968, 1024, 4096, 0, 1, 0, 0, 1, 0, 0, 0, 1.f},
            {"qkv (scatter)", 15968, 3072, 1024, 0, 0, 0, 0, 0, 1, 0, 0, 1.f},
            {"out-proj (+res -> f32)", 15968, 1024, 1024, 0, 1, 0, 0, 1, 0, 0, 0, 1.f},
            {"conv1 (f32 out)", 511968, 512, 1536, 0, 0, 0, 0, 1, 0, 15999, 1024, 1.f},
            {"conv2 (f32 out)", 255968, 512, 1536, 0, 0, 0, 0, 1, 0, 7999, 1024, 1.f},
        };
        int precs[] = {PREC_F16X3, PREC_BF16, PREC_F16};
        for (int prec : precs)
            for (auto& c : shapes) {
                double a = 0, b = 0;
                run_case(prec, c, true, &a, &b);
                double fl = 2.0 * c.M * c.N * c.K;
                printf("TIME prec=%d %-24s M=%d N=%d K=%d : pp %.1f us %.1f TF/s | generic %.1f us %.1f TF/s  (algorithmic; x3 issues 3x)\n",
                       prec, c.name, c.M, c.N, c.K, a, fl / a * 1e-6, b, fl / b * 1e-6);
            }
    }
    return 0;
}
