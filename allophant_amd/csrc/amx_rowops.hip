// Row-wise / bandwidth-bound kernels of the Allophant forward path on gfx950 (64-wide wavefronts):
// input normalisation statistics, fused conv layer 0 + LayerNorm + GELU, row LayerNorm(+GELU)(+LayerNorm) with 16-bit
// plane output, positional-conv input image, classifier-input concatenation (hidden | softmax(dependency logits)),
// per-head log-softmax with [T,N,C] time-major output, greedy CTC decode, and the one-off weight packers.
#include "amx_common.h"
#include <cstdlib>

namespace amx {

namespace {

// ----------------------------------------------------------------------------------------------------------------
// input normalisation (reference acoustic_model.py:762-767): statistics only; applied on the fly by conv layer 0
// ----------------------------------------------------------------------------------------------------------------
constexpr int STAT_CHUNKS = 64;

__global__ __launch_bounds__(256) void audio_stats_kernel(const float* __restrict__ audio, const int64_t* __restrict__ lengths,
                                                          int64_t L, double* __restrict__ partial) {
    const int n = blockIdx.y, c = blockIdx.x;
    int64_t per = (L + STAT_CHUNKS - 1) / STAT_CHUNKS;
    per = (per + 3) & ~(int64_t)3;
    int64_t lo = c * per, hi = lo + per < L ? lo + per : L;
    const int64_t len = lengths[n];
    const float* row = audio + (int64_t)n * L;
    double s_all = 0, s_val = 0, q_val = 0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
        double x = row[i];
        s_all += x;
        if (i < len) { s_val += x; q_val += x * x; }
    }
    __shared__ double red[3][4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s_all += __shfl_xor(s_all, o);
        s_val += __shfl_xor(s_val, o);
        q_val += __shfl_xor(q_val, o);
    }
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = s_all;
        red[1][threadIdx.x >> 6] = s_val;
        red[2][threadIdx.x >> 6] = q_val;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        double v = red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3];
        partial[((int64_t)n * STAT_CHUNKS + c) * 3 + threadIdx.x] = v;
    }
}

__global__ void audio_stats_final_kernel(const double* __restrict__ partial, const int64_t* __restrict__ lengths, int N,
                                         float* __restrict__ mean_rstd, int do_normalize) {
    int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    if (!do_normalize) { mean_rstd[2 * n] = 0.f; mean_rstd[2 * n + 1] = 1.f; return; }
    double s_all = 0, s_val = 0, q_val = 0;
    for (int c = 0; c < STAT_CHUNKS; ++c) {
        s_all += partial[((int64_t)n * STAT_CHUNKS + c) * 3 + 0];
        s_val += partial[((int64_t)n * STAT_CHUNKS + c) * 3 + 1];
        q_val += partial[((int64_t)n * STAT_CHUNKS + c) * 3 + 2];
    }
    double len = (double)lengths[n];
    double mean = s_all / len;  // the reference sums the whole padded row (zero padding contract)
    double var = (q_val - 2.0 * mean * s_val + len * mean * mean) / len;
    if (var < 0) var = 0;
    mean_rstd[2 * n] = (float)mean;
    mean_rstd[2 * n + 1] = (float)(1.0 / sqrt(var + 1e-7));
}

// ----------------------------------------------------------------------------------------------------------------
// conv layer 0 (C_in = 1, kernel KW, stride s) + LayerNorm over channels + exact GELU  ->  16-bit planes [N*T1, C]
// One wave per output frame, lane owns CPL consecutive channels (weights in registers), window staged in LDS.
// HBM-bound: reads 4 B/sample once, writes 2*NT bytes per output element.
// ----------------------------------------------------------------------------------------------------------------
constexpr int C0_FRAMES = 128;  // frames per workgroup: amortises the per-lane weight loads

// GN (the group-norm feature extractor, `feat_extract_norm="group"`: GroupNorm(C groups of one channel) over TIME behind conv
// layer 0 instead of a LayerNorm over channels): the statistics come from conv0_gn_stats_kernel, and `gamma` / `beta` are the
// per-(utterance, channel) scale and shift [N, C] that conv0_gn_final_kernel folded them into -- y = acc * scale + shift, no
// cross-lane reduction.
template <typename T, int NT, int CPL, int KW, bool GN>
__global__ __launch_bounds__(256) void conv0_kernel(const float* __restrict__ audio, const int64_t* __restrict__ lengths,
                                                    const float* __restrict__ mean_rstd, int64_t L, int T1, int C, int k,
                                                    int stride, const float* __restrict__ w, const float* __restrict__ b,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    float eps, int do_normalize, T* __restrict__ out, int64_t out_plane,
                                                    int skip_padding) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* win = (float*)smem;
    const int n = blockIdx.y;
    const int f0 = blockIdx.x * C0_FRAMES;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nwin = C0_FRAMES * stride + k;
    const int64_t len = lengths[n];
    // ragged batch: frames past the utterance's own (len - k) / stride + 1 feed no valid frame of any later layer
    if (skip_padding && (int64_t)f0 * stride + k > len && f0 > 0) return;
    const float mean = mean_rstd[2 * n], rstd = mean_rstd[2 * n + 1];
    const int64_t s0 = (int64_t)f0 * stride;
    for (int i = threadIdx.x; i < nwin; i += 256) {
        int64_t pos = s0 + i;
        float x = 0.f;
        if (pos < L) {
            x = audio[(int64_t)n * L + pos];
            if (do_normalize) x = pos < len ? (x - mean) * rstd : 0.f;
        }
        win[i] = x;
    }
    const int c0 = lane * CPL;
    const bool active = c0 < C;
    if (GN) { gamma += (int64_t)n * C; beta += (int64_t)n * C; }
    float wr[CPL][KW], br[CPL], gr[CPL], be[CPL];
    if ((CPL * KW) % 4 == 0 && k == KW && c0 + CPL <= C) {
        // this lane's CPL x KW weights are contiguous and 16-byte aligned: branch-free vector loads (the per-element
        // predicated loads of the general path cost as much as the block's arithmetic)
        float flat[CPL * KW];
        const float4* src = (const float4*)(w + (int64_t)c0 * KW);
#pragma unroll
        for (int q = 0; q < CPL * KW / 4; ++q) {
            const float4 v = src[q];
            flat[4 * q] = v.x; flat[4 * q + 1] = v.y; flat[4 * q + 2] = v.z; flat[4 * q + 3] = v.w;
        }
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
#pragma unroll
            for (int j = 0; j < KW; ++j) wr[i][j] = flat[i * KW + j];
            br[i] = b[c0 + i];
            gr[i] = gamma[c0 + i];
            be[i] = beta[c0 + i];
        }
    } else {
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            int c = c0 + i;
            bool ok = c < C;
#pragma unroll
            for (int j = 0; j < KW; ++j) wr[i][j] = (ok && j < k) ? w[c * k + j] : 0.f;
            br[i] = ok ? b[c] : 0.f;
            gr[i] = ok ? gamma[c] : 0.f;
            be[i] = ok ? beta[c] : 0.f;
        }
    }
    __syncthreads();
    const float invC = 1.0f / (float)C;
    // two frames per wave iteration: the two dependent chains (conv -> mean -> variance -> GELU) interleave
    constexpr int FPI = 2;
    if constexpr (CPL % 2 == 0) {
        // channel pairs on the packed fp32 pipe (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32, packed 16-bit converts): the
        // kernel is VALU-bound, and this halves the instructions of the convolution, the normalisation and the split
        typedef typename Vec2<T>::type V2;
        constexpr int CP = CPL / 2;
        f32x2 w2[CP][KW], b2[CP], g2[CP], be2[CP];
#pragma unroll
        for (int i = 0; i < CP; ++i) {
#pragma unroll
            for (int j = 0; j < KW; ++j) w2[i][j] = f32x2{wr[2 * i][j], wr[2 * i + 1][j]};
            b2[i] = f32x2{br[2 * i], br[2 * i + 1]};
            g2[i] = f32x2{gr[2 * i], gr[2 * i + 1]};
            be2[i] = f32x2{be[2 * i], be[2 * i + 1]};
        }
        for (int f = wave * FPI; f < C0_FRAMES; f += 4 * FPI) {
            f32x2 acc[FPI][CP];
#pragma unroll
            for (int u = 0; u < FPI; ++u)
#pragma unroll
                for (int i = 0; i < CP; ++i) acc[u][i] = b2[i];
#ifdef AMX_C0_ABLATE_TAPS
            // developer ablation (wrong results): the k-tap contraction reduced to ONE tap -- what the kernel would cost if the
            // taps ran somewhere else (e.g. on the matrix pipe): an upper bound of what an MFMA form of them could save
#pragma unroll
            for (int u = 0; u < FPI; ++u) {
                const float x = win[(f + u) * stride];
                const f32x2 xx = {x, x};
#pragma unroll
                for (int i = 0; i < CP; ++i) acc[u][i] = w2[i][0] * xx + acc[u][i];
            }
#else
#pragma unroll
            for (int j = 0; j < KW; ++j) {
#pragma unroll
                for (int u = 0; u < FPI; ++u) {
                    const float x = j < k ? win[(f + u) * stride + j] : 0.f;  // f + u < C0_FRAMES: inside the staged window
                    const f32x2 xx = {x, x};
#pragma unroll
                    for (int i = 0; i < CP; ++i) acc[u][i] = w2[i][j] * xx + acc[u][i];
                }
            }
#endif
            // channels beyond C carry zero weights and bias, so they add nothing to the sums (c0 + CPL <= C or lane idle)
            float mu[FPI], rs[FPI];
            if constexpr (!GN) {
#pragma unroll
                for (int u = 0; u < FPI; ++u) {
                    f32x2 s2 = acc[u][0];
#pragma unroll
                    for (int i = 1; i < CP; ++i) s2 += acc[u][i];
                    mu[u] = wave_sum(active ? s2[0] + s2[1] : 0.f) * invC;
                }
#pragma unroll
                for (int u = 0; u < FPI; ++u) {
                    const f32x2 m2 = {mu[u], mu[u]};
                    f32x2 q2 = {0.f, 0.f};
#pragma unroll
                    for (int i = 0; i < CP; ++i) {
                        const f32x2 d = acc[u][i] - m2;
                        q2 = d * d + q2;
                    }
                    rs[u] = 1.0f / sqrtf(wave_sum(active ? q2[0] + q2[1] : 0.f) * invC + eps);
                }
            } else {
#pragma unroll
                for (int u = 0; u < FPI; ++u) { mu[u] = 0.f; rs[u] = 1.f; }
            }
#pragma unroll
            for (int u = 0; u < FPI; ++u) {
                const int t = f0 + f + u;
                V2 hi[CP], lo[CP];
                const f32x2 r2 = {rs[u], rs[u]}, m2 = {mu[u], mu[u]};
#pragma unroll
                for (int i = 0; i < CP; ++i) {
                    const f32x2 v = GN ? acc[u][i] * g2[i] + be2[i] : (acc[u][i] - m2) * r2 * g2[i] + be2[i];
                    f32x2 y = gelu_fast2(v);
                    // one register pair for y: its 16-bit image and the residual are taken from the same value (see split16)
                    asm volatile("" : "+v"(y));
                    hi[i] = __builtin_convertvector(y, V2);
                    if (NT > 1) {
                        asm volatile("" : "+v"(hi[i]));
                        const f32x2 back = {(float)hi[i][0], (float)hi[i][1]};
                        lo[i] = __builtin_convertvector(y - back, V2);
                    }
                }
                if (active && t < T1) {
                    T* dst = out + pidx(((int64_t)n * T1 + t) * C + c0, plane_is_il<NT>(out_plane));
                    if constexpr (CPL == 8) {
                        union { V2 h[4]; typename Vec8<T>::type v; } hv, lv;
#pragma unroll
                        for (int i = 0; i < 4; ++i) { hv.h[i] = hi[i]; if (NT > 1) lv.h[i] = lo[i]; }
                        *(typename Vec8<T>::type*)dst = hv.v;
                        if (NT > 1) *(typename Vec8<T>::type*)(dst + out_plane) = lv.v;
                    } else {
#pragma unroll
                        for (int i = 0; i < CP; ++i) {
                            *(V2*)(dst + 2 * i) = hi[i];
                            if (NT > 1) *(V2*)(dst + out_plane + 2 * i) = lo[i];
                        }
                    }
                }
            }
        }
        return;
    }
    for (int f = wave * FPI; f < C0_FRAMES; f += 4 * FPI) {
        float acc[FPI][CPL];
#pragma unroll
        for (int u = 0; u < FPI; ++u)
#pragma unroll
            for (int i = 0; i < CPL; ++i) acc[u][i] = br[i];
#pragma unroll
        for (int j = 0; j < KW; ++j) {
#pragma unroll
            for (int u = 0; u < FPI; ++u) {
                const float x = j < k ? win[(f + u) * stride + j] : 0.f;  // f + u < C0_FRAMES: inside the staged window
#pragma unroll
                for (int i = 0; i < CPL; ++i) acc[u][i] = fmaf(wr[i][j], x, acc[u][i]);
            }
        }
        float mu[FPI], rs[FPI];
        if constexpr (!GN) {
#pragma unroll
            for (int u = 0; u < FPI; ++u) {
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < CPL; ++i) s += (c0 + i < C) ? acc[u][i] : 0.f;
                mu[u] = wave_sum(s) * invC;
            }
#pragma unroll
            for (int u = 0; u < FPI; ++u) {
                float q = 0.f;
#pragma unroll
                for (int i = 0; i < CPL; ++i) {
                    float d = acc[u][i] - mu[u];
                    q += (c0 + i < C) ? d * d : 0.f;
                }
                rs[u] = 1.0f / sqrtf(wave_sum(q) * invC + eps);
            }
        } else {
#pragma unroll
            for (int u = 0; u < FPI; ++u) { mu[u] = 0.f; rs[u] = 1.f; }
        }
#pragma unroll
        for (int u = 0; u < FPI; ++u) {
            const int t = f0 + f + u;
            T hi[CPL], lo[CPL];
#pragma unroll
            for (int i = 0; i < CPL; ++i) {
                float y = gelu_fast((acc[u][i] - mu[u]) * rs[u] * gr[i] + be[i]);
                split16<T, NT>(y, hi[i], lo[i]);
            }
            if (active && t < T1) {
                T* dst = out + pidx(((int64_t)n * T1 + t) * C + c0, plane_is_il<NT>(out_plane));
#pragma unroll
                for (int i = 0; i < CPL; ++i)
                    if (c0 + i < C) {
                        dst[i] = hi[i];
                        if (NT > 1) dst[out_plane + i] = lo[i];
                    }
            }
        }
    }
}

// ----------------------------------------------------------------------------------------------------------------
// conv layer 0 on the matrix pipe (round 5; k = 10, C = 512, LayerNorm variant -- every released wav2vec 2.0 / XLS-R shape).
//
// conv0_kernel above is VALU-bound (0.45 ms of its 0.57 ms per config-2 step without its stores): ~390 VALU instructions per
// frame pair, of which the k-tap contraction is 28 % (measured: with ONE tap the kernel takes 0.41 ms, profiles/r05_conv0_*)
// and the two 64-lane LayerNorm reductions another 20 %.  Here
//   * the taps run as ONE v_mfma_f32_16x16x32_f16 per (16 channels x 16 frames): K = 32 holds the three split-precision terms of
//     the 10-tap product side by side -- A row (channel) [w_hi(10) | w_lo(10) | w_hi(10) | 0 0], B column (frame)
//     [x_hi(10) | x_hi(10) | x_lo(10) | 0 0] -- so hi.hi + lo.hi + hi.lo (2^-22 relative, fp32 accumulate) is a single MFMA on
//     a pipe that idles in this kernel.  Both operands are fp16 planes whatever the handle's mode: the samples of a frame are
//     scaled by an exact power of two (largest |x| of the frame into [512, 1024)) and the weights by one per tensor, so the lo
//     planes stay normal numbers and silence is as accurate as speech (LayerNorm makes the result scale-free per frame);
//   * the LayerNorm statistics need no pass over the channels: mean_c(w_c . x + b_c) = wbar . x + bbar and
//     var_c(...) = x~^T G x~ with x~ = [x; 1] and G the 11 x 11 covariance of the rows [w_c, b_c] over the channels -- exact
//     identities, evaluated per frame in fp64 from tables amx_create computes in fp64 (`stats`: 11 + 121 doubles), i.e. more
//     accurate than the fp32 two-pass sums they replace, and 132 FMAs per frame instead of 2 x 512 + two butterflies;
//   * a lane owns ONE frame and 4 consecutive channels per tile (MFMA output layout), so the normalisation is two packed FMAs
//     with lane-constant factors; outputs go through a per-wave LDS patch and leave as whole 128-byte lines.
// Workgroup = 128 frames (8 frame tiles) x 512 channels; wave w owns channels [128 w, 128 w + 128) with its weight fragments in
// registers; the frame operands (8 KiB), scales and statistics of the 128 frames are built once per workgroup.
// ----------------------------------------------------------------------------------------------------------------
constexpr int C0M_FRAMES = 128, C0M_K = 10, C0M_C = 512;
static_assert(CONV0_MFMA_STATS == 11 + 121, "stats table: mean of [w_c, b_c] over c (11 doubles), covariance G (11 x 11, row-major)");

template <int NT>
constexpr int c0m_rowb() { return (NT == 2 ? 512 : 256) + 16; }  // bytes of a staged frame row of one wave (+ 16: bank skew)
inline size_t conv0_mfma_lds_bytes(int NT, int stride) {
    const size_t win = ((size_t)(C0M_FRAMES * stride + C0M_K) * 4 + 15) & ~(size_t)15;
    const size_t stage = (size_t)4 * 16 * (NT == 2 ? c0m_rowb<2>() : c0m_rowb<1>());
    const size_t image = (size_t)C0M_C * 64;  // fp16 weight image [512][32], aliased with the staging patches
    return win + 4 * C0M_FRAMES * 4 + (size_t)C0M_FRAMES * 64 + (stage > image ? stage : image);
}

// GN (the group-norm feature extractor): `gamma` / `beta` are the per-(utterance, channel) scale and shift [N, C] of the GroupNorm
// over time (conv0_gn_cov_final_kernel), y = (conv + b) * scale + shift: no per-frame statistics, `stats` unused.
template <typename T, int NT, bool GN>
__global__ __launch_bounds__(256, 2) void conv0_mfma_kernel(const float* __restrict__ audio, const int64_t* __restrict__ lengths,
                                                            const float* __restrict__ mean_rstd, int64_t L, int T1, int stride,
                                                            const float* __restrict__ w /*[512][10]*/, const float* __restrict__ b,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const double* __restrict__ stats, float w_scale, float eps,
                                                            int do_normalize, T* __restrict__ out, int64_t out_plane,
                                                            int skip_padding) {
    typedef _Float16 h16;
    typedef __attribute__((ext_vector_type(8))) _Float16 h16x8;
    typedef typename Vec2<T>::type V2;
    constexpr int ROWB = c0m_rowb<NT>();
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int n = blockIdx.y;
    const int f0 = blockIdx.x * C0M_FRAMES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwin = C0M_FRAMES * stride + C0M_K;
    const int64_t len = lengths[n];
    if (skip_padding && (int64_t)f0 * stride + C0M_K > len && f0 > 0) return;  // (see conv0_kernel)
    if (GN) { gamma += (int64_t)n * C0M_C; beta += (int64_t)n * C0M_C; }
    float* win = (float*)smem;
    const int win_bytes = (nwin * 4 + 15) & ~15;
    float* fr_a = (float*)(smem + win_bytes);          // per frame: acc -> conv value / sigma:  rstd / (2^s * w_scale)
    float* fr_c = fr_a + C0M_FRAMES;                    //            - mean * rstd
    float* fr_r = fr_c + C0M_FRAMES;                    //            rstd
    unsigned char* bplanes = (unsigned char*)(fr_r + 2 * C0M_FRAMES);  // [8 frame tiles][64 MFMA lanes][16 B]
    unsigned char* stage = bplanes + C0M_FRAMES * 64;   // per-wave output patches; first: the fp16 weight image [512][32]

    // ---- the window of this workgroup's frames, input normalisation applied on load ----
    const float mean = mean_rstd[2 * n], rstd_in = mean_rstd[2 * n + 1];
    const int64_t s0 = (int64_t)f0 * stride;
    for (int i = tid; i < nwin; i += 256) {
        const int64_t pos = s0 + i;
        float x = 0.f;
        if (pos < L) {
            x = audio[(int64_t)n * L + pos];
            if (do_normalize) x = pos < len ? (x - mean) * rstd_in : 0.f;
        }
        win[i] = x;
    }
    // ---- weight image: row c = [w_hi(10) | w_lo(10) | w_hi(10) | 0 0] of w_c * w_scale as fp16 (two channels per thread) ----
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
        const int c = tid + 256 * rep;
        h16 hi[C0M_K], lo[C0M_K];
#pragma unroll
        for (int j = 0; j < C0M_K; ++j) {
            float v = w[c * C0M_K + j] * w_scale;
            asm volatile("" : "+v"(v));
            hi[j] = (h16)v;
            asm volatile("" : "+v"(hi[j]));
            lo[j] = (h16)(v - (float)hi[j]);
        }
        h16 row[32];
#pragma unroll
        for (int j = 0; j < C0M_K; ++j) { row[j] = hi[j]; row[10 + j] = lo[j]; row[20 + j] = hi[j]; }
        row[30] = (h16)0.f; row[31] = (h16)0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            h16x8 v8;
#pragma unroll
            for (int e = 0; e < 8; ++e) v8[e] = row[8 * q + e];
            *(h16x8*)(stage + c * 64 + q * 16) = v8;
        }
    }
    __syncthreads();
    // ---- per frame (threads 0..127): power-of-two scale, fp64 LayerNorm statistics, the frame's B column ----
    if (tid < C0M_FRAMES) {
        const int f = tid;
        float x[C0M_K];
        float amax = 0.f;
#pragma unroll
        for (int j = 0; j < C0M_K; ++j) { x[j] = win[f * stride + j]; amax = fmaxf(amax, fabsf(x[j])); }
        // exact power of two that puts the largest sample of the frame into [512, 1024) (1 for an all-zero / non-finite frame)
        int e = 0;
        if (amax > 0.f && amax < INFINITY) e = 9 - (((__builtin_bit_cast(int, amax) >> 23) & 255) - 127);
        e = e < -100 ? -100 : (e > 120 ? 120 : e);
        const float sc = __builtin_bit_cast(float, (127 + e) << 23), inv_sc = __builtin_bit_cast(float, (127 - e) << 23);
        if constexpr (GN) {
            fr_r[f] = 1.f;
            fr_c[f] = 0.f;
            fr_a[f] = inv_sc / w_scale;  // (both exact powers of two)
        } else {
            double xt[11];
#pragma unroll
            for (int j = 0; j < C0M_K; ++j) xt[j] = (double)x[j];
            xt[10] = 1.0;
            double mu = 0.0, var = 0.0;
#pragma unroll
            for (int a = 0; a < 11; ++a) {
                mu = fma(stats[a], xt[a], mu);
                double rowsum = 0.0;
#pragma unroll
                for (int c2 = 0; c2 < 11; ++c2) rowsum = fma(stats[11 + a * 11 + c2], xt[c2], rowsum);
                var = fma(rowsum, xt[a], var);
            }
            if (!(var > 0.0)) var = 0.0;
            const double rs = 1.0 / sqrt(var + (double)eps);
            fr_r[f] = (float)rs;
            fr_c[f] = (float)(-mu * rs);
            fr_a[f] = (float)(rs * (double)inv_sc / (double)w_scale);
        }
        h16 hi[C0M_K], lo[C0M_K];
#pragma unroll
        for (int j = 0; j < C0M_K; ++j) {
            float v = x[j] * sc;
            asm volatile("" : "+v"(v));
            hi[j] = (h16)v;
            asm volatile("" : "+v"(hi[j]));
            lo[j] = (h16)(v - (float)hi[j]);
        }
        h16 row[32];
#pragma unroll
        for (int j = 0; j < C0M_K; ++j) { row[j] = hi[j]; row[10 + j] = hi[j]; row[20 + j] = lo[j]; }
        row[30] = (h16)0.f; row[31] = (h16)0.f;
        const int ft = f >> 4, jj = f & 15;
#pragma unroll
        for (int q = 0; q < 4; ++q) {  // chunk q = the 8 K values MFMA lane (jj, group q) feeds
            h16x8 v8;
#pragma unroll
            for (int e2 = 0; e2 < 8; ++e2) v8[e2] = row[8 * q + e2];
            *(h16x8*)(bplanes + ((ft * 4 + q) * 16 + jj) * 16) = v8;
        }
    }
    // ---- this wave's weight fragments (A operand: lane (i, g) = channel i of the tile, K 8g .. 8g + 7) and channel constants ----
    const int li = lane & 15, g = lane >> 4;
    h16x8 wf[8];
    f32x2 b2[8][2], g2[8][2], be2[8][2];
#pragma unroll
    for (int tt = 0; tt < 8; ++tt) {
        wf[tt] = *(const h16x8*)(stage + ((wave * 8 + tt) * 16 + li) * 64 + g * 16);
        const int c = wave * 128 + tt * 16 + 4 * g;
        const float4 bb = *(const float4*)(b + c), gg = *(const float4*)(gamma + c), ee = *(const float4*)(beta + c);
        b2[tt][0] = f32x2{bb.x, bb.y}; b2[tt][1] = f32x2{bb.z, bb.w};
        g2[tt][0] = f32x2{gg.x, gg.y}; g2[tt][1] = f32x2{gg.z, gg.w};
        be2[tt][0] = f32x2{ee.x, ee.y}; be2[tt][1] = f32x2{ee.z, ee.w};
        if constexpr (GN) {  // (conv + b) * scale + shift = conv * scale + (b * scale + shift)
            be2[tt][0] = b2[tt][0] * g2[tt][0] + be2[tt][0];
            be2[tt][1] = b2[tt][1] * g2[tt][1] + be2[tt][1];
        }
    }
    __syncthreads();  // frame columns and statistics are complete; the weight image is dead (its LDS becomes the patches)
    unsigned char* patch = stage + wave * (16 * ROWB);
    const bool o_il = plane_is_il<NT>(out_plane);
    for (int ft = 0; ft < C0M_FRAMES / 16; ++ft) {
        if (f0 + ft * 16 >= T1) break;  // (wave-uniform) nothing of this tile exists
        const h16x8 xf = *(const h16x8*)(bplanes + (ft * 64 + lane) * 16);
        const int f = ft * 16 + li;
        const float a = fr_a[f], cc = fr_c[f], rr = fr_r[f];
        const f32x2 a2 = {a, a}, c2 = {cc, cc}, r2 = {rr, rr};
#pragma unroll
        for (int tt = 0; tt < 8; ++tt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[tt], xf, acc, 0, 0, 0);
            V2 hi[2], lo[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const f32x2 v = {acc[2 * q], acc[2 * q + 1]};
                // LayerNorm((v * inv_scale + b) ; mu, rstd) * gamma + beta with the per-frame factors folded: a = rstd * inv_scale
                const f32x2 u = GN ? v * a2 : v * a2 + (b2[tt][q] * r2 + c2);
                const f32x2 y = gelu_fast2(u * g2[tt][q] + be2[tt][q]);
                lo[q] = V2{(T)0.f, (T)0.f};
                split16x2<T, NT>(y, hi[q], lo[q]);
            }
            // channels co .. co + 3 of the wave's 128: the frame row as it lies in HBM (interleaved planes: [hi x 32 | lo x 32])
            const int co = tt * 16 + 4 * g;
            unsigned char* dst = patch + li * ROWB + (NT == 2 ? (co >> 5) * 128 + (co & 31) * 2 : co * 2);
            union { V2 h[2]; uint64_t u; } ph, pl;
            ph.h[0] = hi[0]; ph.h[1] = hi[1];
            *(uint64_t*)dst = ph.u;
            if (NT == 2) {
                pl.h[0] = lo[0]; pl.h[1] = lo[1];
                *(uint64_t*)(dst + 64) = pl.u;
            }
        }
        // the wave's 16 x (128 channels) patch leaves as 16-byte pieces of whole lines (LDS operations of one wave complete in
        // order: these reads see the writes above, and the next tile's writes follow them)
        constexpr int CHUNKS_PER_ROW = (ROWB - 16) / 16, ROUNDS = 16 * CHUNKS_PER_ROW / 64;
#pragma unroll
        for (int q = 0; q < ROUNDS; ++q) {
            const int id = q * 64 + lane, row = id / CHUNKS_PER_ROW, chunk = id - row * CHUNKS_PER_ROW;
            const uint4 v = *(const uint4*)(patch + row * ROWB + chunk * 16);
            const int t = f0 + ft * 16 + row;
            if (t < T1) {
                T* base = out + pidx(((int64_t)n * T1 + t) * C0M_C + wave * 128, o_il);
                *(uint4*)((unsigned char*)base + chunk * 16) = v;
            }
        }
    }
}

// ----------------------------------------------------------------------------------------------------------------
// Group-norm feature extractor (transformers Wav2Vec2GroupNormConvLayer: conv -> GroupNorm(num_groups = C) -> GELU): the
// statistics of conv layer 0's raw output per (utterance, channel) over ALL T1 frames of the padded length -- upstream
// normalises the padded batch tensor, so frames of an utterance's padding count (their conv output is the bias).  The conv
// (k taps per output) is recomputed rather than stored: pass 1 here, pass 2 = conv0_kernel<GN>.  fp64 sums of the fp32 conv
// values, fixed order (block partials, then a serial combine): deterministic.
// ----------------------------------------------------------------------------------------------------------------
constexpr int GN_FRAMES = 1024;  // frames per workgroup of the statistics pass

template <int KW>
__global__ __launch_bounds__(256) void conv0_gn_stats_kernel(const float* __restrict__ audio, const int64_t* __restrict__ lengths,
                                                             const float* __restrict__ mean_rstd, int64_t L, int T1, int C, int k,
                                                             int stride, const float* __restrict__ w, const float* __restrict__ b,
                                                             int do_normalize, double* __restrict__ partial /*[N][blocks][C][2]*/) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* win = (float*)smem;
    const int n = blockIdx.y, blk = blockIdx.x;
    const int f0 = blk * GN_FRAMES;
    const int frames = min(GN_FRAMES, T1 - f0);
    const int nwin = (frames - 1) * stride + k;
    const int64_t len = lengths[n];
    const float mean = mean_rstd[2 * n], rstd = mean_rstd[2 * n + 1];
    const int64_t s0 = (int64_t)f0 * stride;
    for (int i = threadIdx.x; i < nwin; i += 256) {
        const int64_t pos = s0 + i;
        float x = 0.f;
        if (pos < L) {
            x = audio[(int64_t)n * L + pos];
            if (do_normalize) x = pos < len ? (x - mean) * rstd : 0.f;
        }
        win[i] = x;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float wr[KW];
#pragma unroll
        for (int j = 0; j < KW; ++j) wr[j] = j < k ? w[c * k + j] : 0.f;
        const float bias = b[c];
        double s = 0.0, q = 0.0;
        for (int f = 0; f < frames; f += 8) {
            // eight frames in fp32, then one fp64 step: the fp32 partial sums stay short (8 terms)
            float s8 = 0.f, q8 = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (f + u < frames) {
                    float acc = bias;
#pragma unroll
                    for (int j = 0; j < KW; ++j) acc = fmaf(wr[j], j < k ? win[(f + u) * stride + j] : 0.f, acc);
                    s8 += acc;
                    q8 = fmaf(acc, acc, q8);
                }
            }
            s += (double)s8;
            q += (double)q8;
        }
        double* dst = partial + (((int64_t)n * gridDim.x + blk) * C + c) * 2;
        dst[0] = s;
        dst[1] = q;
    }
}

// The same statistics with the register blocking of conv0_kernel (C == 512, k == KW): a lane owns 8 consecutive channels with
// their taps in registers, a wave walks the frames of its block two at a time on the packed fp32 pipe (one LDS broadcast read
// per tap feeds 8 channels; the general kernel above reads LDS once per multiply), eight frames in fp32 between fp64 steps, and
// the four waves of the workgroup are summed through LDS in wave order.  0.60 -> 0.2 ms at 32 x 10 s.
template <int KW>
__global__ __launch_bounds__(256) void conv0_gn_stats8_kernel(const float* __restrict__ audio, const int64_t* __restrict__ lengths,
                                                              const float* __restrict__ mean_rstd, int64_t L, int T1, int C, int stride,
                                                              const float* __restrict__ w, const float* __restrict__ b, int do_normalize,
                                                              double* __restrict__ partial /*[N][blocks][C][2]*/) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* win = (float*)smem;
    constexpr int CP = 4;  // channel pairs per lane
    const int n = blockIdx.y, blk = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int f0 = blk * GN_FRAMES;
    const int frames = min(GN_FRAMES, T1 - f0);
    const int nwin = (frames - 1) * stride + KW;
    const int64_t len = lengths[n];
    const float mean = mean_rstd[2 * n], rstd = mean_rstd[2 * n + 1];
    const int64_t s0 = (int64_t)f0 * stride;
    for (int i = threadIdx.x; i < nwin; i += 256) {
        const int64_t pos = s0 + i;
        float x = 0.f;
        if (pos < L) {
            x = audio[(int64_t)n * L + pos];
            if (do_normalize) x = pos < len ? (x - mean) * rstd : 0.f;
        }
        win[i] = x;
    }
    const int c0 = lane * 8;
    f32x2 w2[CP][KW], b2[CP];
#pragma unroll
    for (int i = 0; i < CP; ++i) {
#pragma unroll
        for (int j = 0; j < KW; ++j) w2[i][j] = f32x2{w[(c0 + 2 * i) * KW + j], w[(c0 + 2 * i + 1) * KW + j]};
        b2[i] = f32x2{b[c0 + 2 * i], b[c0 + 2 * i + 1]};
    }
    __syncthreads();
    double sd[2 * CP], qd[2 * CP];
#pragma unroll
    for (int i = 0; i < 2 * CP; ++i) { sd[i] = 0.0; qd[i] = 0.0; }
    // the wave's frames: wave, wave + 4, ... in groups of 8 (two at a time), fp32 inside a group
    for (int g = wave * 8; g < frames; g += 32) {
        f32x2 s8[CP], q8[CP];
#pragma unroll
        for (int i = 0; i < CP; ++i) { s8[i] = f32x2{0.f, 0.f}; q8[i] = f32x2{0.f, 0.f}; }
#pragma unroll
        for (int u = 0; u < 8; u += 2) {
            f32x2 acc[2][CP];
#pragma unroll
            for (int v = 0; v < 2; ++v)
#pragma unroll
                for (int i = 0; i < CP; ++i) acc[v][i] = b2[i];
#pragma unroll
            for (int j = 0; j < KW; ++j)
#pragma unroll
                for (int v = 0; v < 2; ++v) {
                    const int f = g + u + v;
                    const float x = f < frames ? win[f * stride + j] : 0.f;
                    const f32x2 xx = {x, x};
#pragma unroll
                    for (int i = 0; i < CP; ++i) acc[v][i] = w2[i][j] * xx + acc[v][i];
                }
#pragma unroll
            for (int v = 0; v < 2; ++v) {
                if (g + u + v < frames) {
#pragma unroll
                    for (int i = 0; i < CP; ++i) {
                        s8[i] += acc[v][i];
                        q8[i] = acc[v][i] * acc[v][i] + q8[i];
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < CP; ++i) {
            sd[2 * i] += (double)s8[i][0]; sd[2 * i + 1] += (double)s8[i][1];
            qd[2 * i] += (double)q8[i][0]; qd[2 * i + 1] += (double)q8[i][1];
        }
    }
    // the four waves of the block, summed in wave order (the window is dead: its LDS holds the partials)
    __syncthreads();
    double* red = (double*)smem;  // [4 waves][512 channels][2]
#pragma unroll
    for (int i = 0; i < 2 * CP; ++i) {
        red[((size_t)wave * 512 + c0 + i) * 2] = sd[i];
        red[((size_t)wave * 512 + c0 + i) * 2 + 1] = qd[i];
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        double s = 0.0, q = 0.0;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) { s += red[((size_t)wv * 512 + c) * 2]; q += red[((size_t)wv * 512 + c) * 2 + 1]; }
        double* dst = partial + (((int64_t)n * gridDim.x + blk) * C + c) * 2;
        dst[0] = s;
        dst[1] = q;
    }
}

// mean / variance over the T1 frames -> scale = gamma * rstd, shift = beta - mean * scale per (utterance, channel)
__global__ void conv0_gn_final_kernel(const double* __restrict__ partial, int blocks, int N, int C, int T1,
                                      const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                      float* __restrict__ scale, float* __restrict__ shift) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)N * C) return;
    const int n = (int)(i / C), c = (int)(i - (int64_t)n * C);
    double s = 0.0, q = 0.0;
    for (int blk = 0; blk < blocks; ++blk) {
        const double* src = partial + (((int64_t)n * blocks + blk) * C + c) * 2;
        s += src[0];
        q += src[1];
    }
    const double mean = s / (double)T1;
    double var = q / (double)T1 - mean * mean;  // biased variance, like torch.nn.GroupNorm
    if (var < 0) var = 0;
    const double rs = 1.0 / sqrt(var + (double)eps);
    const double sc = (double)gamma[c] * rs;
    scale[i] = (float)sc;
    shift[i] = (float)((double)beta[c] - mean * sc);
}

// ----------------------------------------------------------------------------------------------------------------
// GroupNorm statistics without recomputing the convolution (round 5; k = 10, the conv0_mfma_kernel shapes).  The conv output of
// channel c at frame t is w_c . x_t + b_c with x_t the frame's 10 (normalised) samples, so over the T1 frames of the padded length
//   mean_t = w_c . xbar + b_c,    var_t = w_c^T Cov w_c,    xbar = mean_t x_t,  Cov = mean_t x_t x_t^T - xbar xbar^T  (10 x 10)
// -- one pass over the audio for 10 + 55 sums in fp64 (conv0_gn_cov_kernel), then 110 fp64 FMAs per (utterance, channel)
// (conv0_gn_cov_final_kernel) instead of a second evaluation of every conv output.  Deterministic: block partials in fixed
// order, tree reductions of fixed shape.
// ----------------------------------------------------------------------------------------------------------------
constexpr int GNC_FRAMES = 2048;   // frames per workgroup
constexpr int GNC_SUMS = 10 + 55;  // S1[j], S2[a <= b]

__global__ __launch_bounds__(256) void conv0_gn_cov_kernel(const float* __restrict__ audio, const int64_t* __restrict__ lengths,
                                                           const float* __restrict__ mean_rstd, int64_t L, int T1, int stride,
                                                           int do_normalize, double* __restrict__ partial /*[N][blocks][65]*/) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* win = (float*)smem;
    const int n = blockIdx.y, blk = blockIdx.x;
    const int f0 = blk * GNC_FRAMES;
    const int frames = min(GNC_FRAMES, T1 - f0);
    const int nwin = (frames - 1) * stride + C0M_K;
    const int64_t len = lengths[n];
    const float mean = mean_rstd[2 * n], rstd = mean_rstd[2 * n + 1];
    const int64_t s0 = (int64_t)f0 * stride;
    for (int i = threadIdx.x; i < nwin; i += 256) {
        const int64_t pos = s0 + i;
        float x = 0.f;
        if (pos < L) {
            x = audio[(int64_t)n * L + pos];
            if (do_normalize) x = pos < len ? (x - mean) * rstd : 0.f;
        }
        win[i] = x;
    }
    __syncthreads();
    double acc[GNC_SUMS];
#pragma unroll
    for (int i = 0; i < GNC_SUMS; ++i) acc[i] = 0.0;
    for (int f = threadIdx.x; f < frames; f += 256) {
        double x[C0M_K];
#pragma unroll
        for (int j = 0; j < C0M_K; ++j) x[j] = (double)win[f * stride + j];
        int p = C0M_K;
#pragma unroll
        for (int a = 0; a < C0M_K; ++a) {
            acc[a] += x[a];
#pragma unroll
            for (int b2 = a; b2 < C0M_K; ++b2) { acc[p] = fma(x[a], x[b2], acc[p]); ++p; }
        }
    }
    // wave totals (xor tree), then the four waves in order
#pragma unroll
    for (int i = 0; i < GNC_SUMS; ++i) {
        double v = acc[i];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
        acc[i] = v;
    }
    __syncthreads();  // the window is dead: its LDS takes the wave totals
    double* red = (double*)smem;  // [4][65]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < GNC_SUMS; ++i) red[wave * GNC_SUMS + i] = acc[i];
    }
    __syncthreads();
    if (threadIdx.x < GNC_SUMS) {
        const double v = ((red[threadIdx.x] + red[GNC_SUMS + threadIdx.x]) + red[2 * GNC_SUMS + threadIdx.x]) + red[3 * GNC_SUMS + threadIdx.x];
        partial[((int64_t)n * gridDim.x + blk) * GNC_SUMS + threadIdx.x] = v;
    }
}

// one workgroup of 512 threads per utterance: totals of the block partials -> xbar, Cov; thread c -> scale / shift of channel c
__global__ __launch_bounds__(512) void conv0_gn_cov_final_kernel(const double* __restrict__ partial, int blocks, int T1,
                                                                 const float* __restrict__ w /*[512][10]*/, const float* __restrict__ b,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                                 float* __restrict__ scale, float* __restrict__ shift) {
    __shared__ double tot[GNC_SUMS], xbar[C0M_K], cov[C0M_K][C0M_K];
    const int n = blockIdx.x, c = threadIdx.x;
    if (c < GNC_SUMS) {
        double v = 0.0;
        for (int blk = 0; blk < blocks; ++blk) v += partial[((int64_t)n * blocks + blk) * GNC_SUMS + c];
        tot[c] = v;
    }
    __syncthreads();
    if (c < C0M_K) xbar[c] = tot[c] / (double)T1;
    __syncthreads();
    if (c < C0M_K * C0M_K) {
        const int a = c / C0M_K, b2 = c % C0M_K, lo = a < b2 ? a : b2, hi = a < b2 ? b2 : a;
        // index of (lo, hi) in the upper-triangular order of conv0_gn_cov_kernel
        const int p = C0M_K + lo * C0M_K - lo * (lo - 1) / 2 + (hi - lo);
        cov[a][b2] = tot[p] / (double)T1 - xbar[a] * xbar[b2];
    }
    __syncthreads();
    double wr[C0M_K];
#pragma unroll
    for (int j = 0; j < C0M_K; ++j) wr[j] = (double)w[c * C0M_K + j];
    double mean = (double)b[c], var = 0.0;
#pragma unroll
    for (int a = 0; a < C0M_K; ++a) {
        mean = fma(wr[a], xbar[a], mean);
        double rowsum = 0.0;
#pragma unroll
        for (int b2 = 0; b2 < C0M_K; ++b2) rowsum = fma(cov[a][b2], wr[b2], rowsum);
        var = fma(rowsum, wr[a], var);
    }
    if (!(var > 0.0)) var = 0.0;  // biased variance, like torch.nn.GroupNorm
    const double rs = 1.0 / sqrt(var + (double)eps);
    const double sc = (double)gamma[c] * rs;
    scale[(int64_t)n * C0M_C + c] = (float)sc;
    shift[(int64_t)n * C0M_C + c] = (float)((double)beta[c] - mean * sc);
}

// ----------------------------------------------------------------------------------------------------------------
// Row kernel: x[M, D] fp32 -> [LayerNorm(g1,b1) -> optional GELU] -> [LayerNorm(g2,b2)] -> planes / fp32.
// One wave per row, D <= 256 V, D % 4 == 0, row kept in registers (float4 x V per lane; V = 4 up to hidden 1024 -- XLS-R 300M and
// every released checkpoint --, V = 8 up to 2048: the XLS-R 1B / 2B widths 1280 / 1920, round 6).
// ----------------------------------------------------------------------------------------------------------------
template <typename T, int NT, int V = 4>
// (x and out_f32 carry no __restrict__: the post-LN encoder normalises the residual stream IN PLACE, out_f32 == x; a wave loads
// its whole row into registers before it stores any of it, and rows belong to one wave each)
__global__ __launch_bounds__(256) void rownorm_kernel(const float* x, int64_t ldx, int64_t M, int D,
                                                      const float* __restrict__ g1, const float* __restrict__ b1, int gelu,
                                                      const float* __restrict__ g2, const float* __restrict__ b2,
                                                      float eps1, float eps2, T* __restrict__ out_p, int64_t out_plane,
                                                      int64_t ldp, float* out_f32, int64_t ldo,
                                                      const int* __restrict__ row_off, const int* __restrict__ frame_len, int T_rows) {
    const int lane = threadIdx.x & 63;
    int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* src = x + row * ldx;
    if (row_off) {
        // ragged batch: input row n * T_rows + t goes to packed row row_off[n] + t; frames beyond the utterance are dropped
        const int n = (int)(row / T_rows), t = (int)(row - (int64_t)n * T_rows);
        if (t >= frame_len[n]) return;
        row = (int64_t)row_off[n] + t;
    }
    float4 v[V];
    const float invD = 1.0f / (float)D;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        int c = (i * 64 + lane) * 4;
        v[i] = c < D ? *(const float4*)(src + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    auto norm = [&](const float* g, const float* b, float eps, bool act) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < V; ++i) s += v[i].x + v[i].y + v[i].z + v[i].w;  // out-of-range entries are zero
        const float mu = wave_sum(s) * invD;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            int c = (i * 64 + lane) * 4;
            if (c < D) {
                float dx = v[i].x - mu, dy = v[i].y - mu, dz = v[i].z - mu, dw = v[i].w - mu;
                q += dx * dx + dy * dy + dz * dz + dw * dw;
            }
        }
        const float rs = 1.0f / sqrtf(wave_sum(q) * invD + eps);
#pragma unroll
        for (int i = 0; i < V; ++i) {
            int c = (i * 64 + lane) * 4;
            if (c < D) {
                float4 gg = *(const float4*)(g + c), bb = *(const float4*)(b + c);
                float4 y;
                y.x = (v[i].x - mu) * rs * gg.x + bb.x;
                y.y = (v[i].y - mu) * rs * gg.y + bb.y;
                y.z = (v[i].z - mu) * rs * gg.z + bb.z;
                y.w = (v[i].w - mu) * rs * gg.w + bb.w;
                if (act) {
                    const f32x2 g01 = gelu_fast2(f32x2{y.x, y.y}), g23 = gelu_fast2(f32x2{y.z, y.w});
                    y.x = g01[0]; y.y = g01[1]; y.z = g23[0]; y.w = g23[1];
                }
                v[i] = y;
            }
        }
    };
    if (g1) norm(g1, b1, eps1, gelu != 0);
    if (g2) norm(g2, b2, eps2, false);
#pragma unroll
    for (int i = 0; i < V; ++i) {
        int c = (i * 64 + lane) * 4;
        if (c < D) {
            if (out_f32) *(float4*)(out_f32 + row * ldo + c) = v[i];
            if (out_p) {
                T hi[4], lo[4];
                split16<T, NT>(v[i].x, hi[0], lo[0]);
                split16<T, NT>(v[i].y, hi[1], lo[1]);
                split16<T, NT>(v[i].z, hi[2], lo[2]);
                split16<T, NT>(v[i].w, hi[3], lo[3]);
                T* dst = out_p + pidx(row * ldp + c, plane_is_il<NT>(out_plane));
                typedef typename Vec4<T>::type V4;
                V4 hv = {hi[0], hi[1], hi[2], hi[3]};
                *(V4*)dst = hv;
                if (NT > 1) {
                    V4 lv = {lo[0], lo[1], lo[2], lo[3]};
                    *(V4*)(dst + out_plane) = lv;
                }
            }
        }
    }
}

// ----------------------------------------------------------------------------------------------------------------
// LayerNorm fold (GemmParams.ln_partial / row_coef, amx_common.h).  The power of two a row's planes are written under: 16 * rstd
// rounded down to a power of two, i.e. sigma * s in (8, 16] -- |x - mu| <= sqrt(D) sigma keeps a row itself below 512, and the NEXT
// state of the row (written under this scale before its own statistics are known) may grow a hundredfold before an fp16 plane
// overflows (which the range report of the pass would then show); values below 2^-6 sigma lose bits of their lo plane.
// ----------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float ln_plane_scale(float rstd) {
    const float t = fminf(fmaxf(16.0f * rstd, 1.0e-12f), 1.0e12f);
    return __uint_as_float(__float_as_uint(t) & 0x7F800000u);
}

// the first norm of the stack: exact statistics from the fp32 row (the arithmetic of rownorm_kernel), planes of (x - mu) * s
template <typename T, int NT, int V = 4>
__global__ __launch_bounds__(256) void ln_rowprep_kernel(const float* __restrict__ x, int64_t ldx, int64_t M, int D, float eps,
                                                         T* __restrict__ out_p, int64_t out_plane, int64_t ldp,
                                                         float4* __restrict__ rowps, float2* __restrict__ coef) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* src = x + row * ldx;
    float4 v[V];
    const float invD = 1.0f / (float)D;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int c = (i * 64 + lane) * 4;
        v[i] = c < D ? *(const float4*)(src + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) s += v[i].x + v[i].y + v[i].z + v[i].w;
    const float mu = wave_sum(s) * invD;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < D) {
            const float dx = v[i].x - mu, dy = v[i].y - mu, dz = v[i].z - mu, dw = v[i].w - mu;
            q += dx * dx + dy * dy + dz * dz + dw * dw;
        }
    }
    const float rs = 1.0f / sqrtf(wave_sum(q) * invD + eps);
    const float sc = ln_plane_scale(rs);
    if (lane == 0) {
        rowps[row] = make_float4(mu, sc, mu, sc);
        coef[row] = make_float2(rs / sc, 0.f);
    }
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < D) {
            T hi[4], lo[4];
            split16<T, NT>((v[i].x - mu) * sc, hi[0], lo[0]);
            split16<T, NT>((v[i].y - mu) * sc, hi[1], lo[1]);
            split16<T, NT>((v[i].z - mu) * sc, hi[2], lo[2]);
            split16<T, NT>((v[i].w - mu) * sc, hi[3], lo[3]);
            T* dst = out_p + pidx(row * ldp + c, plane_is_il<NT>(out_plane));
            typedef typename Vec4<T>::type V4;
            V4 hv = {hi[0], hi[1], hi[2], hi[3]};
            *(V4*)dst = hv;
            if (NT > 1) {
                V4 lv = {lo[0], lo[1], lo[2], lo[3]};
                *(V4*)(dst + out_plane) = lv;
            }
        }
    }
}

// one thread per row: Chan's pairwise update over the row's 64-column blocks, in ascending order
__global__ __launch_bounds__(256) void ln_finalize_kernel(const float2* __restrict__ partial, int blocks, int64_t M, float eps,
                                                          float4* __restrict__ rowps, float2* __restrict__ coef) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= M) return;
    const float2* pr = partial + row * blocks;
    float2 b[32];  // (hidden <= 2048)
    float s1 = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j)
        if (j < blocks) {
            b[j] = pr[j];
            s1 += b[j].x;
        }
    // the producer wrote the planes under (pivot z, scale w) and formed the statistics of u = (x - z) * w: w is a power of two, so
    // the sums are those of x - z times w (times w^2) exactly
    const float4 ps = rowps[row];
    const float inv_w = __uint_as_float(0x7F000000u - __float_as_uint(ps.w));
    const float invD = 1.0f / (float)(blocks * 64);
    const float mu = s1 * invD;  // mean of u
    float m2 = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j)
        if (j < blocks) {
            const float d = b[j].x * (1.0f / 64.0f) - mu;
            m2 += fmaf(64.0f * d, d, b[j].y);
        }
    const float m1 = mu * inv_w;  // mean of x - pivot
    const float rs = 1.0f / sqrtf(m2 * invD * inv_w * inv_w + eps);
    coef[row] = make_float2(rs * inv_w, -rs * m1);
    rowps[row] = make_float4(ps.z, ps.w, ps.z + m1, ln_plane_scale(rs));
}

// ----------------------------------------------------------------------------------------------------------------
// positional-conv input image: h[N*T, D] fp32 -> planes [G][N][Tpad][cg], zero rows in front/behind every utterance
// ----------------------------------------------------------------------------------------------------------------
template <typename T, int NT>
__global__ __launch_bounds__(256) void posconv_pack_kernel(const float* __restrict__ h, int N, int Tn, int D, int G,
                                                           int pad_front, int Tpad, T* __restrict__ out, int64_t out_plane,
                                                           const int* __restrict__ row_off, const int* __restrict__ frame_len) {
    const int cg = D / G;
    const int64_t total4 = (int64_t)G * N * Tpad * cg / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        int64_t e = i * 4;
        int ci = (int)(e % cg);
        int64_t r = e / cg;
        int tp = (int)(r % Tpad);
        r /= Tpad;
        int n = (int)(r % N);
        int g = (int)(r / N);
        int t = tp - pad_front;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        // packed rows (row_off): utterance n owns rows row_off[n] .. + frame_len[n]; its other frames are the zeros the
        // feature projection's row mask leaves in the padded layout
        if (row_off) {
            if (t >= 0 && t < frame_len[n]) v = *(const float4*)(h + ((int64_t)row_off[n] + t) * D + g * cg + ci);
        } else if (t >= 0 && t < Tn) v = *(const float4*)(h + ((int64_t)n * Tn + t) * D + g * cg + ci);
        T hi[4], lo[4];
        split16<T, NT>(v.x, hi[0], lo[0]);
        split16<T, NT>(v.y, hi[1], lo[1]);
        split16<T, NT>(v.z, hi[2], lo[2]);
        split16<T, NT>(v.w, hi[3], lo[3]);
        typedef typename Vec4<T>::type V4;
        V4 hv = {hi[0], hi[1], hi[2], hi[3]};
        *(V4*)(out + e) = hv;
        if (NT > 1) {
            V4 lv = {lo[0], lo[1], lo[2], lo[3]};
            *(V4*)(out + out_plane + e) = lv;
        }
    }
}

// ----------------------------------------------------------------------------------------------------------------
// classifier input: cat([hidden | softmax(dependency logits)]) as 16-bit planes, one wave per row
// (reference acoustic_model.py:497-514)
// ----------------------------------------------------------------------------------------------------------------
template <typename T, int NT>
__global__ __launch_bounds__(256) void concat_kernel(const ConcatPart* __restrict__ parts, int n_parts,
                                                     const float* __restrict__ logits, int64_t ld_logits, int64_t M,
                                                     T* __restrict__ out, int64_t out_plane, int64_t ldp, int kpad) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const bool il = plane_is_il<NT>(out_plane);
    const int64_t row0 = row * ldp;  // logical offset of the row; element c lives at pidx(row0 + c)
    int kend = 0;
    for (int pi = 0; pi < n_parts; ++pi) {
        const ConcatPart pt = parts[pi];
        if (pt.type == 0) {
            const float* src = pt.src + row * pt.width;
            for (int c = lane; c < pt.width; c += 64) {
                T hi, lo;
                split16<T, NT>(src[c], hi, lo);
                T* dst = out + pidx(row0 + pt.dst_col + c, il);
                dst[0] = hi;
                if (NT > 1) dst[out_plane] = lo;
            }
        } else {
            const float* src = logits + row * ld_logits + pt.src_col;
            float m = -INFINITY;
            for (int c = lane; c < pt.width; c += 64) m = fmaxf(m, src[c]);
            m = wave_max(m);
            float s = 0.f;
            for (int c = lane; c < pt.width; c += 64) s += expf(src[c] - m);
            s = wave_sum(s);
            for (int c = lane; c < pt.width; c += 64) {
                T hi, lo;
                split16<T, NT>(expf(src[c] - m) / s, hi, lo);
                T* dst = out + pidx(row0 + pt.dst_col + c, il);
                dst[0] = hi;
                if (NT > 1) dst[out_plane] = lo;
            }
        }
        int e = pt.dst_col + pt.width;
        kend = e > kend ? e : kend;
    }
    for (int c = kend + lane; c < kpad; c += 64) {
        T* dst = out + pidx(row0 + c, il);
        dst[0] = (T)0.f;
        if (NT > 1) dst[out_plane] = (T)0.f;
    }
}

// ----------------------------------------------------------------------------------------------------------------
// time-layer classifier heads (ProjectingMultiheadAttention, reference acoustic_model.py:237-268)
// ----------------------------------------------------------------------------------------------------------------
// LayerNorm(C) of the projected rows + sinusoidal positions (acoustic_model.py:34-69: column 2k = sin(t * base_k),
// column 2k+1 = cos(t * base_k); `pe_base` holds base per column) -> 16-bit planes [M, kpad] for the in_proj product.
// One wave per row (row = n*T + t); C is arbitrary (class sizes are small, the composed head has C = embedding_size).
template <typename T, int NT>
__global__ __launch_bounds__(256) void time_ln_pe_kernel(const float* __restrict__ x, int64_t M, int C, int T_frames,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float eps, const float* __restrict__ pe_base,
                                                         T* __restrict__ out, int64_t out_plane, int kpad) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* src = x + row * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += src[c];
    const float mu = wave_sum(s) / (float)C;
    float q = 0.f;
    for (int c = lane; c < C; c += 64) {
        float d = src[c] - mu;
        q += d * d;
    }
    const float rs = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
    const float t = (float)(row % T_frames);
    const bool il = plane_is_il<NT>(out_plane);
    for (int c = lane; c < kpad; c += 64) {
        float y = 0.f;
        if (c < C) {
            y = (src[c] - mu) * rs * gamma[c] + beta[c];
            if (pe_base) {
                const float arg = t * pe_base[c];
                y += (c & 1) ? cosf(arg) : sinf(arg);
            }
        }
        T hi, lo;
        split16<T, NT>(y, hi, lo);
        T* dst = out + pidx(row * kpad + c, il);
        dst[0] = hi;
        if (NT > 1) dst[out_plane] = lo;
    }
}

// softmax(q k^T / sqrt(dh) masked to the valid keys of the utterance) v over the frames of one utterance, fp32 on the
// vector ALU: one wave per (utterance, head, query frame).  Lanes are (key group, d) pairs: DP = pow2 >= min(dh, 64)
// lanes hold the dh axis, 64 / DP groups stride over the keys.  Scores go through LDS, `kc` keys at a time (per wave: kc scores,
// the scaled query and the output accumulator): an utterance with more valid keys than that is walked in chunks with the running
// maximum / sum of the online softmax, so there is no length limit (nn.MultiheadAttention has none, acoustic_model.py:255-268);
// up to kc keys -- every utterance below ~ 3 minutes -- the arithmetic is that of the plain two-pass softmax (0 * 0 + x steps).
// Queries of padded frames are computed like upstream (only keys are masked, nn.MultiheadAttention key_padding_mask).
template <typename T, int NT>
__global__ __launch_bounds__(256) void time_attention_kernel(const float* __restrict__ qkv, const int* __restrict__ frame_len,
                                                             int T_frames, int C, int dh, int dp, int kc, T* __restrict__ out,
                                                             int64_t out_plane, int kpad) {
    extern __shared__ float time_lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int t = blockIdx.x * 4 + wave, hd = blockIdx.y, n = blockIdx.z;
    if (t >= T_frames) return;
    float* sc = time_lds + (size_t)wave * (kc + 2 * dh);
    float* qs = sc + kc;
    float* oa = qs + dh;
    const int64_t ld = 3 * (int64_t)C;
    const float* base = qkv + (int64_t)n * T_frames * ld + hd * dh;
    const float scale = 1.0f / sqrtf((float)dh);
    for (int d = lane; d < dh; d += 64) {
        qs[d] = base[(int64_t)t * ld + d] * scale;
        oa[d] = 0.f;
    }
    const int len = frame_len[n];
    const int groups = 64 / dp, kg = lane / dp, dl = lane % dp;
    float m = -INFINITY, l = 0.f;
    for (int k0 = 0; k0 < len; k0 += kc) {
        const int nk = len - k0 < kc ? len - k0 : kc;
        float mc = -INFINITY;
        for (int key = lane; key < nk; key += 64) {
            const float* kr = base + (int64_t)(k0 + key) * ld + C;
            float a = 0.f;
            for (int d = 0; d < dh; ++d) a = fmaf(qs[d], kr[d], a);
            sc[key] = a;
            mc = fmaxf(mc, a);
        }
        const float mn = fmaxf(m, wave_max(mc));
        const float f = expf(m - mn);  // first chunk: exp(-inf) = 0
        float sum = 0.f;
        for (int key = lane; key < nk; key += 64) {
            float e = expf(sc[key] - mn);
            sc[key] = e;
            sum += e;
        }
        l = l * f + wave_sum(sum);
        m = mn;
        for (int d0 = 0; d0 < dh; d0 += dp) {
            const int d = d0 + dl;
            float o = 0.f;
            if (d < dh) {
                const float* vr = base + (int64_t)k0 * ld + 2 * C + d;
                for (int key = kg; key < nk; key += groups) o = fmaf(sc[key], vr[(int64_t)key * ld], o);
            }
            for (int off = dp; off < 64; off <<= 1) o += __shfl_xor(o, off);
            if (kg == 0 && d < dh) oa[d] = oa[d] * f + o;
        }
    }
    const float inv = 1.0f / l;
    const int64_t orow = ((int64_t)n * T_frames + t) * kpad;
    for (int d0 = 0; d0 < dh; d0 += dp) {
        const int d = d0 + dl;
        if (kg == 0 && d < dh) {
            T hi, lo;
            split16<T, NT>(oa[d] * inv, hi, lo);
            T* dst = out + pidx(orow + hd * dh + d, plane_is_il<NT>(out_plane));
            dst[0] = hi;
            if (NT > 1) dst[out_plane] = lo;
        }
    }
    if (hd == 0)
        for (int c = C + lane; c < kpad; c += 64) {
            T* dst = out + pidx(orow + c, plane_is_il<NT>(out_plane));
            dst[0] = (T)0.f;
            if (NT > 1) dst[out_plane] = (T)0.f;
        }
}

// ----------------------------------------------------------------------------------------------------------------
// per-head log-softmax (reference estimator.py:1041-1045), batch-major logits -> time-major [T,N,C] outputs.
// Frames t >= frame_len[n] are published as zeros: upstream leaves layout-dependent garbage there (padded queries run
// through every layer), here the padded and the packed row layouts must hand consumers of whole [T, N, C] tensors the
// same bytes (the contract: frames beyond `lengths` carry no information).
// ----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void logsoftmax_out_kernel(const OutDesc* __restrict__ descs, int n_out,
                                                             const float* __restrict__ logits, int64_t ld, int N, int T,
                                                             const int* __restrict__ frame_len, int log_probs,
                                                             float* __restrict__ out, int* __restrict__ nonfinite,
                                                             const int* __restrict__ row_off) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // row = n*T + t
    if (row >= (int64_t)N * T) return;
    const int n = (int)(row / T), t = (int)(row % T);
    if (t >= frame_len[n]) {
        for (int o = 0; o < n_out; ++o) {
            const OutDesc d = descs[o];
            float* dst = out + (int64_t)T * N * d.prefix + ((int64_t)t * N + n) * d.C;
            for (int c = lane; c < d.C; c += 64) dst[c] = 0.f;
        }
        return;
    }
    const float* src_row = logits + (row_off ? (int64_t)row_off[n] + t : row) * ld;  // packed rows: utterances back to back
    if (nonfinite) {
        // range check of the 16-bit planes (fp16 overflows at 65504): whatever overflowed upstream has become an infinity or
        // a NaN in this frame's logits by now (amx_check_finite reads the counter)
        bool bad = false;
        for (int o = 0; o < n_out; ++o) {
            const OutDesc d = descs[o];
            for (int c = lane; c < d.C; c += 64) bad |= !__builtin_isfinite(src_row[d.col + c]);
        }
        if (__builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0) atomicAdd(nonfinite, 1);
    }
    // narrow outputs (the attribute classifiers: 4 classes each): one LANE per output, everything lane-local -- a wave
    // reduction per 4-class output (36 of them per frame) made this kernel 20x slower than its 11 MB of traffic
    constexpr int NARROW = 8;
    for (int o0 = 0; o0 < n_out; o0 += 64) {
        const int o = o0 + lane;
        if (o < n_out) {
            const OutDesc d = descs[o];
            if (d.C <= NARROW) {
                const float* src = src_row + d.col;
                float* dst = out + (int64_t)T * N * d.prefix + ((int64_t)t * N + n) * d.C;
                float v[NARROW];
#pragma unroll
                for (int c = 0; c < NARROW; ++c) v[c] = c < d.C ? src[c] : -INFINITY;
                float lse = 0.f;
                if (log_probs) {
                    float m = v[0];
#pragma unroll
                    for (int c = 1; c < NARROW; ++c) m = fmaxf(m, v[c]);
                    float sum = 0.f;
#pragma unroll
                    for (int c = 0; c < NARROW; ++c) sum += c < d.C ? expf(v[c] - m) : 0.f;
                    lse = m + logf(sum);
                }
#pragma unroll
                for (int c = 0; c < NARROW; ++c)
                    if (c < d.C) dst[c] = v[c] - lse;
            }
        }
    }
    // wide outputs (phoneme / phone inventories): the whole wave per output
    for (int o = 0; o < n_out; ++o) {
        const OutDesc d = descs[o];
        if (d.C <= NARROW) continue;
        const float* src = src_row + d.col;
        float* dst = out + (int64_t)T * N * d.prefix + ((int64_t)t * N + n) * d.C;
        float lse = 0.f;
        if (log_probs) {
            float m = -INFINITY;
            for (int c = lane; c < d.C; c += 64) m = fmaxf(m, src[c]);
            m = wave_max(m);
            float s = 0.f;
            for (int c = lane; c < d.C; c += 64) s += expf(src[c] - m);
            s = wave_sum(s);
            lse = m + logf(s);
        }
        for (int c = lane; c < d.C; c += 64) dst[c] = src[c] - lse;
    }
}

// ----------------------------------------------------------------------------------------------------------------
// greedy CTC (reference predictions.py:194-207): argmax, collapse repeats, drop blank 0, 1-based start timesteps,
// score = sum of the per-frame maxima.  One block per (output, utterance).
// ----------------------------------------------------------------------------------------------------------------
// one workgroup decodes one utterance of one output: frame t of the utterance is the C floats at base + t * stride_t.
// The argmax indices of CTC_CHUNK frames at a time live in LDS (`chunk` = min(T, CTC_CHUNK) ints); an utterance longer than that is
// walked chunk by chunk with the last index and the output position carried over, so there is no length limit (the reference's
// decoder has none, predictions.py:194-207).  A thread adds the maxima of frames tid, tid + 256, ... in ascending order whatever
// the chunking (CTC_CHUNK % 256 == 0) and the block total is one fixed tree: scores do not depend on T's relation to the chunk.
constexpr int CTC_CHUNK = 8192;
__device__ __forceinline__ void greedy_ctc_block(const float* __restrict__ base, int64_t stride_t, int C, int len, int chunk,
                                                 int blank, int64_t* __restrict__ tok, int64_t* __restrict__ ts, int* __restrict__ count,
                                                 float* __restrict__ score_out, unsigned char* smem) {
    int* idx = (int*)smem;          // [chunk]
    int* scan = idx + chunk;        // [256]
    float* fred = (float*)(scan + 256);  // [256]
    float score = 0.f;
    int pos_base = 0, prev = -1;
    for (int b = 0; b < len; b += chunk) {
        const int n = len - b < chunk ? len - b : chunk;
        for (int t = threadIdx.x; t < n; t += 256) {
            const float* p = base + (int64_t)(b + t) * stride_t;
            float best = p[0];
            int bi = 0;
            for (int c = 1; c < C; ++c) {
                float v = p[c];
                if (v > best) { best = v; bi = c; }
            }
            idx[t] = bi;
            score += best;
        }
        __syncthreads();
        // contiguous segment per thread
        const int seg = (n + 255) / 256;
        const int lo = threadIdx.x * seg, hi = lo + seg < n ? lo + seg : n;
        int cnt = 0;
        for (int t = lo; t < hi; ++t) {
            const bool start = b + t == 0 || idx[t] != (t > 0 ? idx[t - 1] : prev);
            cnt += (start && idx[t] != blank) ? 1 : 0;
        }
        scan[threadIdx.x] = cnt;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            int v = threadIdx.x >= off ? scan[threadIdx.x - off] : 0;
            __syncthreads();
            scan[threadIdx.x] += v;
            __syncthreads();
        }
        int pos = pos_base + scan[threadIdx.x] - cnt;  // exclusive prefix
        for (int t = lo; t < hi; ++t) {
            const bool start = b + t == 0 || idx[t] != (t > 0 ? idx[t - 1] : prev);
            if (start && idx[t] != blank) {
                tok[pos] = idx[t];
                ts[pos] = b + t + 1;
                ++pos;
            }
        }
        const int total = scan[255], last = idx[n - 1];
        __syncthreads();  // every thread has read idx / scan of this chunk before the next one overwrites them
        pos_base += total;
        prev = last;
    }
    fred[threadIdx.x] = score;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        float f = threadIdx.x + off < 256 ? fred[threadIdx.x + off] : 0.f;
        __syncthreads();
        if ((threadIdx.x & (2 * off - 1)) == 0) fred[threadIdx.x] += f;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *count = pos_base;
        *score_out = fred[0];
    }
}

__global__ __launch_bounds__(256) void greedy_ctc_kernel(const OutDesc* __restrict__ descs, const float* __restrict__ out,
                                                         const int* __restrict__ frame_len, int N, int T,
                                                         int64_t* __restrict__ tokens, int64_t* __restrict__ timesteps,
                                                         int* __restrict__ counts, float* __restrict__ scores) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int o = blockIdx.y, n = blockIdx.x;
    const OutDesc d = descs[o];
    const int len = frame_len[n] < T ? frame_len[n] : T;
    const float* base = out + (int64_t)T * N * d.prefix + (int64_t)n * d.C;
    greedy_ctc_block(base, (int64_t)N * d.C, d.C, len, T < CTC_CHUNK ? T : CTC_CHUNK, 0, tokens + ((int64_t)o * N + n) * T, timesteps + ((int64_t)o * N + n) * T,
                     counts + o * N + n, scores + o * N + n, smem);
}

// the reference decoder's own signature (predictions.py:194): one [N, T, C] emission tensor with arbitrary strides
__global__ __launch_bounds__(256) void greedy_ctc_emissions_kernel(const float* __restrict__ emissions, int64_t stride_n,
                                                                   int64_t stride_t, const int* __restrict__ frame_len, int T,
                                                                   int C, int blank, int64_t* __restrict__ tokens,
                                                                   int64_t* __restrict__ timesteps, int* __restrict__ counts,
                                                                   float* __restrict__ scores) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int n = blockIdx.x;
    const int len = frame_len[n] < T ? frame_len[n] : T;
    greedy_ctc_block(emissions + (int64_t)n * stride_n, stride_t, C, len < 0 ? 0 : len, T < CTC_CHUNK ? T : CTC_CHUNK, blank, tokens + (int64_t)n * T,
                     timesteps + (int64_t)n * T, counts + n, scores + n, smem);
}

// ----------------------------------------------------------------------------------------------------------------
// weight packers
// ----------------------------------------------------------------------------------------------------------------
template <typename T, int NT>
__global__ void pack_matrix_kernel(const float* __restrict__ src, int rows, int cols, int64_t srs, int64_t scs, float scale,
                                   T* __restrict__ dst, int64_t dst_plane, int64_t ldd, int cols_pad) {
    int64_t total = (int64_t)rows * cols_pad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int r = (int)(i / cols_pad), c = (int)(i % cols_pad);
        float v = c < cols ? src[r * srs + c * scs] * scale : 0.f;
        T hi, lo;
        split16<T, NT>(v, hi, lo);
        T* d = dst + pidx((int64_t)r * ldd + c, plane_is_il<NT>(dst_plane));
        d[0] = hi;
        if (NT > 1) d[dst_plane] = lo;
    }
}

template <typename T, int NT>
__global__ void pack_conv_w_kernel(const float* __restrict__ src, int Co, int Ci, int k, float scale, T* __restrict__ dst,
                                   int64_t dst_plane, int tap_minor_slice) {
    // dst[co][j*Ci + ci] = src[co][ci][j]; tap-minor (slice S): dst[co][((ci / S) * k + pos) * S + ci % S] = src[co][ci][(k - pos) % k]
    int64_t total = (int64_t)Co * Ci * k;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int ci, j, co;
        if (tap_minor_slice > 0) {
            const int S = tap_minor_slice;
            const int within = (int)(i % S);
            int64_t r = i / S;
            j = (int)(r % k);
            j = (k - j) % k;  // position 0, 1, 2 of a channel slice holds tap 0, 2, 1 (GemmParams.a_taps: shared rows back to back)
            r /= k;
            ci = (int)(r % (Ci / S)) * S + within;
            co = (int)(r / (Ci / S));
        } else {
            ci = (int)(i % Ci);
            int64_t r = i / Ci;
            j = (int)(r % k);
            co = (int)(r / k);
        }
        T hi, lo;
        split16<T, NT>(src[((int64_t)co * Ci + ci) * k + j] * scale, hi, lo);
        T* d = dst + pidx(i, plane_is_il<NT>(dst_plane));  // rows of Ci * k elements
        d[0] = hi;
        if (NT > 1) d[dst_plane] = lo;
    }
}

__global__ __launch_bounds__(256) void posconv_norm_kernel(const float* __restrict__ v, int64_t rows /*D*cg*/, int k,
                                                           float* __restrict__ norm) {
    // weight_norm(dim=2): one norm per kernel tap over dims (0,1)
    const int tap = blockIdx.x;
    double s = 0;
    for (int64_t r = threadIdx.x; r < rows; r += 256) {
        double x = v[r * k + tap];
        s += x * x;
    }
    __shared__ double red[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) norm[tap] = (float)sqrt(red[0] + red[1] + red[2] + red[3]);
}

template <typename T, int NT>
__global__ void pack_posconv_w_kernel(const float* __restrict__ g, const float* __restrict__ v, const float* __restrict__ norm,
                                      int D, int cg, int k, float scale, T* __restrict__ dst, int64_t dst_plane) {
    // dst[grp][co][tap*cg + ci] = v[grp*cg + co][ci][tap] * (g[tap] / norm[tap])
    int64_t total = (int64_t)D * cg * k;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int ci = (int)(i % cg);
        int64_t r = i / cg;
        int tap = (int)(r % k);
        int co_all = (int)(r / k);  // grp*cg + co
        float w = v[((int64_t)co_all * cg + ci) * k + tap] * (g[tap] / norm[tap]) * scale;
        T hi, lo;
        split16<T, NT>(w, hi, lo);
        dst[i] = hi;
        if (NT > 1) dst[dst_plane + i] = lo;
    }
}

template <typename T, int NT>
__global__ void compose_kernel(const float* __restrict__ emb, int E, const int64_t* __restrict__ idx, int P1, int F, float scale,
                               float* __restrict__ composed, T* __restrict__ dst, int64_t dst_plane, int64_t ldd) {
    // EmbeddingBag(mode="sum") per phone (reference acoustic_model.py:219-232); negative index = unused slot
    int64_t total = (int64_t)P1 * E;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int e = (int)(i % E), p = (int)(i / E);
        float s = 0.f;
        for (int f = 0; f < F; ++f) {
            int64_t r = idx[(int64_t)p * F + f];
            if (r >= 0) s += emb[r * E + e];
        }
        composed[i] = s;
        T hi, lo;
        split16<T, NT>(s * scale, hi, lo);
        T* d = dst + pidx((int64_t)p * ldd + e, plane_is_il<NT>(dst_plane));
        d[0] = hi;
        if (NT > 1) d[dst_plane] = lo;
    }
}

__global__ void scale_copy_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t n, float scale) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = src[i] * scale;
}

inline int grid_for(int64_t n, int block = 256, int cap = 8192) {
    int64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    return (int)(g < cap ? g : cap);
}

}  // namespace

#define AMX_DISPATCH(prec, CALL)                                   \
    switch (prec) {                                                \
        case PREC_BF16: { typedef bf16 T16; constexpr int NT = 1; CALL; } break;   \
        case PREC_F16: { typedef f16 T16; constexpr int NT = 1; CALL; } break;     \
        case PREC_BF16X3: { typedef bf16 T16; constexpr int NT = 2; CALL; } break; \
        default: { typedef f16 T16; constexpr int NT = 2; CALL; } break;           \
    }

void launch_audio_stats(const float* audio, const int64_t* lengths, int N, int64_t L, double* partial, float* mean_rstd,
                        int do_normalize, hipStream_t s) {
    if (do_normalize) hipLaunchKernelGGL(audio_stats_kernel, dim3(STAT_CHUNKS, N), dim3(256), 0, s, audio, lengths, L, partial);
    hipLaunchKernelGGL(audio_stats_final_kernel, dim3((N + 63) / 64), dim3(64), 0, s, partial, lengths, N, mean_rstd,
                       do_normalize);
}

template <typename T, int NT, int KW, bool GN>
static void conv0_launch_cpl(const float* audio, const int64_t* lengths, const float* mean_rstd, int N, int64_t L, int T1,
                             int C, int k, int stride, const float* w, const float* b, const float* gamma,
                             const float* beta, float eps, int do_normalize, void* out, int64_t out_plane, int skip_padding,
                             hipStream_t s) {
    dim3 grid((T1 + C0_FRAMES - 1) / C0_FRAMES, N);
    size_t lds = (size_t)(C0_FRAMES * stride + k) * sizeof(float);
    int cpl = C >= 64 ? C / 64 : 1;
#define C0_GO(CPL)                                                                                                  \
    hipLaunchKernelGGL((conv0_kernel<T, NT, CPL, KW, GN>), grid, dim3(256), lds, s, audio, lengths, mean_rstd, L, T1, C, k, \
                       stride, w, b, gamma, beta, eps, do_normalize, (T*)out, out_plane, skip_padding)
    if (cpl == 8) C0_GO(8);
    else if (cpl == 4) C0_GO(4);
    else if (cpl == 2) C0_GO(2);
    else C0_GO(1);
#undef C0_GO
}

bool conv0_mfma_eligible(int C, int k, int stride) {
    static const bool off = dev_switch("AMX_NO_CONV0_MFMA");  // developer A/B switch: the VALU kernel
    return !off && C == C0M_C && k == C0M_K && stride >= 1 && conv0_mfma_lds_bytes(2, stride) <= 64 * 1024;
}

template <typename T, int NT, bool GN = false>
static void conv0_mfma_launch(const float* audio, const int64_t* lengths, const float* mean_rstd, int N, int64_t L, int T1, int stride,
                              const float* w, const float* b, const float* gamma, const float* beta, const double* stats, float w_scale,
                              float eps, int do_normalize, void* out, int64_t out_plane, int skip_padding, hipStream_t s) {
    dim3 grid((T1 + C0M_FRAMES - 1) / C0M_FRAMES, N);
    hipLaunchKernelGGL((conv0_mfma_kernel<T, NT, GN>), grid, dim3(256), conv0_mfma_lds_bytes(NT, stride), s, audio, lengths, mean_rstd, L, T1,
                       stride, w, b, gamma, beta, stats, w_scale, eps, do_normalize, (T*)out, out_plane, skip_padding);
}

void launch_conv0(int prec, const float* audio, const int64_t* lengths, const float* mean_rstd, int N, int64_t L, int T1,
                  int C, int k, int stride, const float* w, const float* b, const float* gamma, const float* beta,
                  float eps, int do_normalize, void* out, int64_t out_plane, int skip_padding, hipStream_t s,
                  const double* mfma_stats, float w_scale) {
    // (two planes: the patch image of the kernel is the interleaved layout)
    if (mfma_stats && conv0_mfma_eligible(C, k, stride) && (prec_planes(prec) == 1 || out_plane == PLANE_IL)) {
        AMX_DISPATCH(prec, (conv0_mfma_launch<T16, NT>(audio, lengths, mean_rstd, N, L, T1, stride, w, b, gamma, beta, mfma_stats, w_scale,
                                                     eps, do_normalize, out, out_plane, skip_padding, s)));
        return;
    }
    if (k == 10) {
        AMX_DISPATCH(prec, (conv0_launch_cpl<T16, NT, 10, false>(audio, lengths, mean_rstd, N, L, T1, C, k, stride, w, b, gamma, beta,
                                                                eps, do_normalize, out, out_plane, skip_padding, s)));
    } else {
        AMX_DISPATCH(prec, (conv0_launch_cpl<T16, NT, 16, false>(audio, lengths, mean_rstd, N, L, T1, C, k, stride, w, b, gamma, beta,
                                                                eps, do_normalize, out, out_plane, skip_padding, s)));
    }
}

// dynamic LDS of the conv-0 kernels (the staged audio window of a workgroup): amx_create refuses strides whose window would
// pass the 64 KiB every kernel may use without raising its limit -- a launch over it would fail and leave stale outputs
size_t conv0_window_lds_bytes(int k, int stride, int group_norm) {
    const size_t apply = (size_t)(C0_FRAMES * stride + k) * sizeof(float);
    const size_t stats = group_norm ? (size_t)((GN_FRAMES - 1) * stride + k) * sizeof(float) : 0;
    return apply > stats ? apply : stats;
}

size_t conv0_groupnorm_partial_bytes(int N, int T1, int C) {
    const size_t recompute = (size_t)N * ((T1 + GN_FRAMES - 1) / GN_FRAMES) * C * 2 * sizeof(double);
    const size_t covariance = (size_t)N * ((T1 + GNC_FRAMES - 1) / GNC_FRAMES) * GNC_SUMS * sizeof(double);
    return recompute > covariance ? recompute : covariance;
}

void launch_conv0_groupnorm(int prec, const float* audio, const int64_t* lengths, const float* mean_rstd, int N, int64_t L, int T1,
                            int C, int k, int stride, const float* w, const float* b, const float* gamma, const float* beta,
                            float eps, int do_normalize, double* partial, float* scale, float* shift, void* out, int64_t out_plane,
                            int skip_padding, hipStream_t s, float mfma_w_scale) {
    // wav2vec 2.0 shape: statistics from the utterance's sample covariance (no second evaluation of the convolution), the
    // apply pass on the matrix pipe (conv0_mfma_kernel<GN>)
    const size_t cov_lds = (size_t)((GNC_FRAMES - 1) * stride + C0M_K) * sizeof(float);
    if (mfma_w_scale > 0.f && conv0_mfma_eligible(C, k, stride) && (prec_planes(prec) == 1 || out_plane == PLANE_IL) && cov_lds <= 64 * 1024) {
        const int cblocks = (T1 + GNC_FRAMES - 1) / GNC_FRAMES;
        hipLaunchKernelGGL(conv0_gn_cov_kernel, dim3(cblocks, N), dim3(256), cov_lds, s, audio, lengths, mean_rstd, L, T1, stride, do_normalize,
                           partial);
        hipLaunchKernelGGL(conv0_gn_cov_final_kernel, dim3(N), dim3(512), 0, s, partial, cblocks, T1, w, b, gamma, beta, eps, scale, shift);
        AMX_DISPATCH(prec, (conv0_mfma_launch<T16, NT, true>(audio, lengths, mean_rstd, N, L, T1, stride, w, b, scale, shift, nullptr,
                                                           mfma_w_scale, eps, do_normalize, out, out_plane, skip_padding, s)));
        return;
    }
    const int blocks = (T1 + GN_FRAMES - 1) / GN_FRAMES;
    const size_t lds = (size_t)((GN_FRAMES - 1) * stride + k) * sizeof(float);
    static const bool plain_stats = dev_switch("AMX_GN_PLAIN_STATS");  // developer A/B switch
    if (k == 10 && C == 512 && !plain_stats) {
        // register-blocked form (wav2vec 2.0 shape); its LDS also holds the 4 x 512 x 2 fp64 partials of the block
        const size_t lds8 = lds > (size_t)4 * 512 * 2 * sizeof(double) ? lds : (size_t)4 * 512 * 2 * sizeof(double);
        hipLaunchKernelGGL(conv0_gn_stats8_kernel<10>, dim3(blocks, N), dim3(256), lds8, s, audio, lengths, mean_rstd, L, T1, C, stride, w, b,
                           do_normalize, partial);
    } else if (k == 10)
        hipLaunchKernelGGL(conv0_gn_stats_kernel<10>, dim3(blocks, N), dim3(256), lds, s, audio, lengths, mean_rstd, L, T1, C, k, stride,
                           w, b, do_normalize, partial);
    else
        hipLaunchKernelGGL(conv0_gn_stats_kernel<16>, dim3(blocks, N), dim3(256), lds, s, audio, lengths, mean_rstd, L, T1, C, k, stride,
                           w, b, do_normalize, partial);
    const int64_t nc = (int64_t)N * C;
    hipLaunchKernelGGL(conv0_gn_final_kernel, dim3((unsigned)((nc + 255) / 256)), dim3(256), 0, s, partial, blocks, N, C, T1, gamma, beta,
                       eps, scale, shift);
    if (k == 10) {
        AMX_DISPATCH(prec, (conv0_launch_cpl<T16, NT, 10, true>(audio, lengths, mean_rstd, N, L, T1, C, k, stride, w, b, scale, shift,
                                                               eps, do_normalize, out, out_plane, skip_padding, s)));
    } else {
        AMX_DISPATCH(prec, (conv0_launch_cpl<T16, NT, 16, true>(audio, lengths, mean_rstd, N, L, T1, C, k, stride, w, b, scale, shift,
                                                               eps, do_normalize, out, out_plane, skip_padding, s)));
    }
}

void launch_rownorm(int prec, const float* x, int64_t ldx, int64_t M, int D, const float* gamma1, const float* beta1,
                    int gelu, const float* gamma2, const float* beta2, float eps1, float eps2, void* out_p,
                    int64_t out_plane, int64_t ldp, float* out_f32, int64_t ldo, hipStream_t s) {
    dim3 grid((unsigned)((M + 3) / 4));
    if (D <= 1024) {
        AMX_DISPATCH(prec, hipLaunchKernelGGL((rownorm_kernel<T16, NT, 4>), grid, dim3(256), 0, s, x, ldx, M, D, gamma1, beta1, gelu,
                                              gamma2, beta2, eps1, eps2, (T16*)out_p, out_plane, ldp, out_f32, ldo, nullptr, nullptr, 1));
    } else {
        AMX_DISPATCH(prec, hipLaunchKernelGGL((rownorm_kernel<T16, NT, 8>), grid, dim3(256), 0, s, x, ldx, M, D, gamma1, beta1, gelu,
                                              gamma2, beta2, eps1, eps2, (T16*)out_p, out_plane, ldp, out_f32, ldo, nullptr, nullptr, 1));
    }
}

void launch_ln_rowprep(int prec, const float* x, int64_t ldx, int64_t M, int D, float eps, void* out_p, int64_t out_plane, int64_t ldp,
                       float4* rowps, float2* coef, hipStream_t s) {
    dim3 grid((unsigned)((M + 3) / 4));
    if (D <= 1024) {
        AMX_DISPATCH(prec, hipLaunchKernelGGL((ln_rowprep_kernel<T16, NT, 4>), grid, dim3(256), 0, s, x, ldx, M, D, eps, (T16*)out_p, out_plane,
                                              ldp, rowps, coef));
    } else {
        AMX_DISPATCH(prec, hipLaunchKernelGGL((ln_rowprep_kernel<T16, NT, 8>), grid, dim3(256), 0, s, x, ldx, M, D, eps, (T16*)out_p, out_plane,
                                              ldp, rowps, coef));
    }
}

void launch_ln_finalize(const float2* partial, int blocks, int64_t M, float eps, float4* rowps, float2* coef, hipStream_t s) {
    hipLaunchKernelGGL(ln_finalize_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, partial, blocks, M, eps, rowps, coef);
}

void launch_rownorm_to_packed(int prec, const float* x, int64_t ldx, int64_t M, int D, const float* gamma1, const float* beta1,
                              int gelu, const float* gamma2, const float* beta2, float eps1, float eps2, void* out_p,
                              int64_t out_plane, int64_t ldp, const int* row_off, const int* frame_len, int T_rows, hipStream_t s) {
    dim3 grid((unsigned)((M + 3) / 4));
    AMX_DISPATCH(prec, hipLaunchKernelGGL((rownorm_kernel<T16, NT>), grid, dim3(256), 0, s, x, ldx, M, D, gamma1, beta1, gelu,
                                          gamma2, beta2, eps1, eps2, (T16*)out_p, out_plane, ldp, nullptr, 0, row_off, frame_len, T_rows));
}

void launch_posconv_pack(int prec, const float* h, int N, int T, int D, int G, int pad_front, int Tpad, void* out,
                         int64_t out_plane, const int* row_off, const int* frame_len, hipStream_t s) {
    int64_t total4 = (int64_t)N * Tpad * D / 4;
    AMX_DISPATCH(prec, hipLaunchKernelGGL((posconv_pack_kernel<T16, NT>), dim3(grid_for(total4)), dim3(256), 0, s, h, N, T, D,
                                          G, pad_front, Tpad, (T16*)out, out_plane, row_off, frame_len));
}

void launch_concat(int prec, const ConcatPart* parts_dev, int n_parts, const float* logits, int64_t ld_logits, int64_t M,
                   void* out, int64_t out_plane, int64_t ldp, int kpad, hipStream_t s) {
    dim3 grid((unsigned)((M + 3) / 4));
    AMX_DISPATCH(prec, hipLaunchKernelGGL((concat_kernel<T16, NT>), grid, dim3(256), 0, s, parts_dev, n_parts, logits,
                                          ld_logits, M, (T16*)out, out_plane, ldp, kpad));
}

void launch_time_ln_pe(int prec, const float* x, int64_t M, int C, int T, const float* gamma, const float* beta, float eps,
                       const float* pe_base, void* out, int64_t out_plane, int kpad, hipStream_t s) {
    dim3 grid((unsigned)((M + 3) / 4));
    AMX_DISPATCH(prec, hipLaunchKernelGGL((time_ln_pe_kernel<T16, NT>), grid, dim3(256), 0, s, x, M, C, T, gamma, beta, eps,
                                          pe_base, (T16*)out, out_plane, kpad));
}

// keys per chunk of the time-layer attention: the whole utterance when 4 waves x (T + 2 dh) floats fit the 160 KiB of a CU
static int time_attention_chunk(int T, int dh) {
    const int room = 160 * 1024 / 16 - 2 * dh;
    return T < room ? T : room;
}
size_t time_attention_lds_bytes(int T, int dh) { return (size_t)4 * (time_attention_chunk(T, dh) + 2 * dh) * sizeof(float); }

template <typename T, int NT>
static void time_attention_launch(const float* qkv, const int* frame_len, int N, int T_frames, int C, int heads, void* out,
                                  int64_t out_plane, int kpad, hipStream_t s) {
    const int dh = C / heads;
    int dp = 1;
    while (dp < dh && dp < 64) dp <<= 1;
    const size_t lds = time_attention_lds_bytes(T_frames, dh);
    auto kernel = time_attention_kernel<T, NT>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kernel, dim3((unsigned)((T_frames + 3) / 4), heads, N), dim3(256), lds, s, qkv, frame_len, T_frames, C,
                       dh, dp, time_attention_chunk(T_frames, dh), (T*)out, out_plane, kpad);
}

void launch_time_attention(int prec, const float* qkv, const int* frame_len, int N, int T, int C, int heads, void* out,
                           int64_t out_plane, int kpad, hipStream_t s) {
    AMX_DISPATCH(prec, (time_attention_launch<T16, NT>(qkv, frame_len, N, T, C, heads, out, out_plane, kpad, s)));
}

void launch_logsoftmax_out(const OutDesc* descs_dev, int n_out, const float* logits, int64_t ld, int N, int T,
                           const int* frame_len, int log_probs, float* out, int* nonfinite, const int* row_off, hipStream_t s) {
    int64_t M = (int64_t)N * T;
    hipLaunchKernelGGL(logsoftmax_out_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, descs_dev, n_out, logits, ld,
                       N, T, frame_len, log_probs, out, nonfinite, row_off);
}

void launch_greedy_ctc(const OutDesc* descs_dev, int n_out, const float* out, const int* frame_len, int N, int T,
                       int64_t* tokens, int64_t* timesteps, int* counts, float* scores, hipStream_t s) {
    size_t lds = (size_t)(T < CTC_CHUNK ? T : CTC_CHUNK) * sizeof(int) + 256 * sizeof(int) + 256 * sizeof(float);
    hipLaunchKernelGGL(greedy_ctc_kernel, dim3(N, n_out), dim3(256), lds, s, descs_dev, out, frame_len, N, T, tokens,
                       timesteps, counts, scores);
}

void launch_greedy_ctc_emissions(const float* emissions, int64_t stride_n, int64_t stride_t, const int* frame_len, int N, int T,
                                 int C, int blank, int64_t* tokens, int64_t* timesteps, int* counts, float* scores, hipStream_t s) {
    size_t lds = (size_t)(T < CTC_CHUNK ? T : CTC_CHUNK) * sizeof(int) + 256 * sizeof(int) + 256 * sizeof(float);
    hipLaunchKernelGGL(greedy_ctc_emissions_kernel, dim3(N), dim3(256), lds, s, emissions, stride_n, stride_t, frame_len, T, C,
                       blank, tokens, timesteps, counts, scores);
}

void launch_pack_matrix(int prec, const float* src, int rows, int cols, int64_t src_row_stride, int64_t src_col_stride,
                        float scale, void* dst, int64_t dst_plane, int64_t ldd, int cols_pad, hipStream_t s) {
    int64_t total = (int64_t)rows * cols_pad;
    AMX_DISPATCH(prec, hipLaunchKernelGGL((pack_matrix_kernel<T16, NT>), dim3(grid_for(total)), dim3(256), 0, s, src, rows,
                                          cols, src_row_stride, src_col_stride, scale, (T16*)dst, dst_plane, ldd, cols_pad));
}

void launch_pack_conv_w(int prec, const float* src, int Co, int Ci, int k, float scale, void* dst, int64_t dst_plane, hipStream_t s,
                        int tap_minor_slice) {
    int64_t total = (int64_t)Co * Ci * k;
    AMX_DISPATCH(prec, hipLaunchKernelGGL((pack_conv_w_kernel<T16, NT>), dim3(grid_for(total)), dim3(256), 0, s, src, Co, Ci, k,
                                          scale, (T16*)dst, dst_plane, tap_minor_slice));
}

void launch_pack_posconv_w(int prec, const float* g, const float* v, int D, int cg, int k, float scale, float* norm_scratch, void* dst,
                           int64_t dst_plane, hipStream_t s) {
    hipLaunchKernelGGL(posconv_norm_kernel, dim3(k), dim3(256), 0, s, v, (int64_t)D * cg, k, norm_scratch);
    int64_t total = (int64_t)D * cg * k;
    AMX_DISPATCH(prec, hipLaunchKernelGGL((pack_posconv_w_kernel<T16, NT>), dim3(grid_for(total)), dim3(256), 0, s, g, v,
                                          norm_scratch, D, cg, k, scale, (T16*)dst, dst_plane));
}

void launch_compose(int prec, const float* emb, int E, const int64_t* idx, int P1, int F, float scale, float* composed_f32, void* dst,
                    int64_t dst_plane, int64_t ldd, hipStream_t s) {
    int64_t total = (int64_t)P1 * E;
    AMX_DISPATCH(prec, hipLaunchKernelGGL((compose_kernel<T16, NT>), dim3(grid_for(total)), dim3(256), 0, s, emb, E, idx, P1, F,
                                          scale, composed_f32, (T16*)dst, dst_plane, ldd));
}

// Zero fills and device copies of a forward pass as KERNELS: a pass holds no hipMemset / hipMemcpy.  A pass may be recorded into
// a HIP graph, and on this runtime (ROCm 7.2) replays of a recorded pass -- once eager passes had run between them -- left
// 0x01010101 in the non-finite frame counter that the pass's memset node zeroes (every other replay: one of the two recordings
// of a caller that alternates between two output buffers; tools/debug_range2.py, profiles/r05_memset_node_replay.log).  Found by
// the range report itself.  Two changes removed it, not separated: no memset / memcpy node in a pass (kernel nodes carry their
// arguments by value), and the graph template now lives as long as its instance (amx_api.hip).  Sizes are multiples of 4 bytes,
// pointers 4-byte aligned (callers: fp16 plane rows of 128 bytes, fp32 rows, one int counter).
namespace {
__global__ __launch_bounds__(256) void zero_fill_kernel(uint32_t* __restrict__ p, size_t words) {
    const size_t stride = (size_t)gridDim.x * 256;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool aligned = ((uintptr_t)p & 15) == 0;
    const size_t quads = aligned ? words / 4 : 0;
    for (size_t q = i; q < quads; q += stride) ((uint4*)p)[q] = make_uint4(0, 0, 0, 0);
    for (size_t w = quads * 4 + i; w < words; w += stride) p[w] = 0;
}
__global__ __launch_bounds__(256) void copy_words_kernel(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, size_t words) {
    const size_t stride = (size_t)gridDim.x * 256;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool aligned = (((uintptr_t)dst | (uintptr_t)src) & 15) == 0;
    const size_t quads = aligned ? words / 4 : 0;
    for (size_t q = i; q < quads; q += stride) ((uint4*)dst)[q] = ((const uint4*)src)[q];
    for (size_t w = quads * 4 + i; w < words; w += stride) dst[w] = src[w];
}
__global__ __launch_bounds__(256) void zero_fill_2d_kernel(unsigned char* __restrict__ base, size_t pitch, size_t width_words) {
    uint32_t* row = (uint32_t*)(base + (size_t)blockIdx.y * pitch);
    for (size_t w = (size_t)blockIdx.x * 256 + threadIdx.x; w < width_words; w += (size_t)gridDim.x * 256) row[w] = 0;
}
}  // namespace

// Developer switches that put the round-5 replay fault back (tools/r06_graph_fault.sh: which of the two changes removed it?):
// AMX_GRAPH_MEMSET_NODES=1 -- zero fills and device copies of a pass as hipMemsetAsync / hipMemcpyAsync again, i.e. memset / memcpy
// NODES in a recording.  (Constants in the product build: amx_common.h dev_switch.)
static bool graph_memset_nodes() {
    static const bool on = dev_switch("AMX_GRAPH_MEMSET_NODES");
    return on;
}

void launch_zero(void* p, size_t bytes, hipStream_t s) {
    if (bytes == 0) return;
    if (graph_memset_nodes()) {
        (void)hipMemsetAsync(p, 0, bytes, s);
        return;
    }
    const size_t words = bytes / 4;
    size_t blocks = (words / 4 + 255) / 256 + 1;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (uint32_t*)p, words);
}

// device-to-device copy of a pass as a kernel, for the same reason (bytes a multiple of 4)
void launch_copy(void* dst, const void* src, size_t bytes, hipStream_t s) {
    if (bytes == 0) return;
    if (graph_memset_nodes()) {
        (void)hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s);
        return;
    }
    const size_t words = bytes / 4;
    size_t blocks = (words / 4 + 255) / 256 + 1;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(copy_words_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (uint32_t*)dst, (const uint32_t*)src, words);
}

void launch_zero_2d(void* base, size_t pitch, size_t width_bytes, size_t rows, hipStream_t s) {
    if (width_bytes == 0 || rows == 0) return;
    if (graph_memset_nodes()) {
        (void)hipMemset2DAsync(base, pitch, 0, width_bytes, rows, s);
        return;
    }
    const size_t words = width_bytes / 4;
    size_t bx = (words + 255) / 256;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(zero_fill_2d_kernel, dim3((unsigned)bx, (unsigned)rows), dim3(256), 0, s, (unsigned char*)base, pitch, words);
}

void launch_scale_copy(const float* src, float* dst, int64_t n, float scale, hipStream_t s) {
    hipLaunchKernelGGL(scale_copy_kernel, dim3(grid_for(n)), dim3(256), 0, s, src, dst, n, scale);
}

// ---------------------------------------------------------------------------------------------------------------
// packed rows: the encoder layers of a ragged batch run on the valid frames only
// ---------------------------------------------------------------------------------------------------------------
namespace {
template <bool UNPACK>
__global__ __launch_bounds__(256) void pack_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, const int* __restrict__ row_off,
                                                        const int* __restrict__ frame_len, int T, int D) {
    const int n = blockIdx.y, t = blockIdx.x;
    if (t >= frame_len[n]) return;
    const int64_t padded_row = (int64_t)n * T + t, packed_row = (int64_t)row_off[n] + t;
    const float4* s4 = (const float4*)(src + (UNPACK ? packed_row : padded_row) * D);
    float4* d4 = (float4*)(dst + (UNPACK ? padded_row : packed_row) * D);
    for (int c = threadIdx.x; c < D / 4; c += 256) d4[c] = s4[c];
}
}  // namespace

void launch_pack_rows(const float* padded, float* packed, const int* row_off, const int* frame_len, int N, int T, int D, bool unpack,
                      hipStream_t s) {
    dim3 grid(T, N);
    if (unpack) hipLaunchKernelGGL(pack_rows_kernel<true>, grid, dim3(256), 0, s, packed, const_cast<float*>(padded), row_off, frame_len, T, D);
    else hipLaunchKernelGGL(pack_rows_kernel<false>, grid, dim3(256), 0, s, padded, packed, row_off, frame_len, T, D);
}

}  // namespace amx
