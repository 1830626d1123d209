// MFMA GEMM for gfx950:  C[M,N] = epilogue(A[M,K] . W[N,K]^T), A and W as K-contiguous 16-bit planes (1 plane: plain
// bf16/f16; 2 planes: hi/lo split, three MFMAs per product -> near-fp32 accuracy on the bf16/f16 matrix pipe).
//
// * 64-wide wavefronts, v_mfma_f32_16x16x32_{f16,bf16}; the W fragment is the first MFMA operand so that a lane ends up
//   with 4 consecutive output columns of one output row (vector epilogue loads/stores).
// * LDS tiles [rows][64 k] (128-byte rows) with the 16-byte-chunk XOR swizzle chunk ^= (row >> 1) & 7: conflict-free for
//   the ds_read_b128 fragment reads (16-lane groups) and for the ds_write_b128 staging writes (8-lane groups).
// * register-staged global->LDS pipeline: the loads of K-tile kt+1 are in flight while tile kt is multiplied.
// * A rows may overlap (lda < K): the strided 1-D convolutions of the wav2vec2 feature extractor and the grouped
//   positional convolution are implicit GEMMs over channels-last activations, no im2col buffer.
#include "amx_common.h"
#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace amx {

bool g_force_generic_gemm = false;  // developer / test switch: route every product through the generic tile kernel

namespace {

constexpr int BK = 64;

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// Epilogue shared by the GEMM kernels.  acc[ni][mi][r] is C[m_base + mi*16 + (lane&15)][n_base + ni*16 + 4*(lane>>4) + r].
template <typename T, int NT, int MI, int NI>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, int z, f32x4 (&acc)[NI][MI], int m_base, int n_base, int lane) {
    const float* bias = p.bias ? p.bias + (int64_t)z * p.zbias : nullptr;
    const float* residual = p.residual ? p.residual + (int64_t)z * p.zout : nullptr;
    float* out_f32 = p.out_f32 ? p.out_f32 + (int64_t)z * p.zout : nullptr;
    T* out_p = p.out_p ? (T*)p.out_p + (int64_t)z * p.zoutp : nullptr;
    const int D = p.H * p.dh;

#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int m = m_base + mi * 16 + (lane & 15);
        if (m >= p.M) continue;
        bool masked = false;
        int b = 0, t = 0;
        if (p.row_len || p.mode == 1) {
            int rt = p.mode == 1 ? p.T : p.rows_T;
            b = m / rt;
            t = m - b * rt;
            if (p.row_len) masked = t >= p.row_len[b];
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            const int nb = n_base + ni * 16 + 4 * (lane >> 4);
            if (nb >= p.N) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int n = nb + r;
                float x = acc[ni][mi][r] * p.scale;
                if (n < p.N) {
                    if (bias) x += bias[n];
                    if (p.act == 1) x = gelu_fast(x);
                    if (residual) x += residual[(int64_t)m * p.ldr + n];
                }
                v[r] = masked ? 0.f : x;
            }
            if (p.mode == 1) {
                // QKV scatter; nb % 4 == 0 and dh % 4 == 0 so the 4 columns share (which, head)
                int which = nb / D;
                int rem = nb - which * D;
                int hh = rem / p.dh;
                int d = rem - hh * p.dh;
                T hi[4], lo[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) split16<T, NT>(v[r], hi[r], lo[r]);
                {
                    T* dst = (T*)(which == 0 ? p.q : (which == 1 ? p.k : p.v)) + (((int64_t)b * p.H + hh) * p.Tp + t) * p.dh + d;
                    typedef typename Vec4<T>::type V4;
                    V4 hv = {hi[0], hi[1], hi[2], hi[3]};
                    *(V4*)dst = hv;
                    if (NT > 1) {
                        V4 lv = {lo[0], lo[1], lo[2], lo[3]};
                        *(V4*)(dst + p.qk_plane) = lv;
                    }
                }
                continue;
            }
            if (out_f32) {
                float* dst = out_f32 + (int64_t)m * p.ldo + nb;
                if (p.vec_ok) {
                    *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (nb + r < p.N) dst[r] = v[r];
                }
            }
            if (out_p) {
                T* dst = out_p + (int64_t)m * p.ldp + nb;
                T hi[4], lo[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) split16<T, NT>(v[r], hi[r], lo[r]);
                if (p.vec_ok) {
                    typedef typename Vec4<T>::type V4;
                    V4 hv = {hi[0], hi[1], hi[2], hi[3]};
                    *(V4*)dst = hv;
                    if (NT > 1) {
                        V4 lv = {lo[0], lo[1], lo[2], lo[3]};
                        *(V4*)(dst + p.out_plane) = lv;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (nb + r < p.N) {
                            dst[r] = hi[r];
                            if (NT > 1) dst[p.out_plane + r] = lo[r];
                        }
                    }
                }
            }
        }
    }
}

template <typename T, int NT, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void gemm_kernel(const GemmParams p) {
    constexpr int THREADS = WM * WN * 64;
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int MI = TM / 16, NI = TN / 16;
    constexpr int CA = BM * 8 / THREADS, CW = BN * 8 / THREADS;
    constexpr int ROWS_PER_PASS = THREADS / 8;
    typedef typename Vec8<T>::type V8;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sA = smem;                       // NT planes of BM*128 bytes
    unsigned char* sW = smem + NT * BM * 128;       // NT planes of BN*128 bytes

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (private 4 MiB L2 each), so the blocks that
    // share an XCD (equal id % 8) are given one contiguous chunk of a grouped tile sequence in which 64 consecutive
    // tiles (= the co-resident blocks of one XCD) form an 8 x 8 rectangle: every A / W panel slice fetched into that L2
    // is reused by 8 blocks.  Pure speed: any placement gives the same results.
    int tile_m, tile_n;
    {
        const int ntn = gridDim.x, ntm = gridDim.y;
        const int total = ntn * ntm;
        const int lin = blockIdx.x + blockIdx.y * ntn;
        const int xcd = lin & 7, q = total >> 3, r = total & 7;
        const int i = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (lin >> 3);  // bijective remap
        constexpr int GM = 8;
        const int per_group = GM * ntn;
        const int group = i / per_group;
        const int first_m = group * GM;
        const int gsize = ntm - first_m < GM ? ntm - first_m : GM;
        const int in_group = i - group * per_group;
        tile_m = first_m + in_group % gsize;
        tile_n = in_group / gsize;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int z = blockIdx.z;

    const T* A = (const T*)p.A + (int64_t)z * p.za;
    const T* W = (const T*)p.W + (int64_t)z * p.zw;

    // ---- per-thread staging addresses ----
    const int ld_row = tid >> 3, ld_c = tid & 7;
    const T* a_ptr[CA];
    const T* w_ptr[CW];
#pragma unroll
    for (int i = 0; i < CA; ++i) {
        int r = m0 + ld_row + i * ROWS_PER_PASS;
        r = r < p.M ? r : p.M - 1;
        int64_t b = r / p.rows_per_batch;
        int64_t t = r - b * p.rows_per_batch;
        a_ptr[i] = A + b * p.a_batch_stride + t * p.lda + ld_c * 8;
    }
#pragma unroll
    for (int i = 0; i < CW; ++i) {
        int r = n0 + ld_row + i * ROWS_PER_PASS;
        r = r < p.N ? r : p.N - 1;
        w_ptr[i] = W + (int64_t)r * p.ldw + ld_c * 8;
    }

    uint4 ra[NT][CA], rw[NT][CW];
    const int K = p.K;
    const int nk = (K + BK - 1) / BK;

    auto load_tile = [&](int kt) {
        int kc = kt * BK + ld_c * 8;
        bool valid = kc < K;
        int koff = valid ? kt * BK : 0;  // clamp: load something legal, select zero afterwards
        if (!valid) koff = -ld_c * 8;
#pragma unroll
        for (int pl = 0; pl < NT; ++pl) {
#pragma unroll
            for (int i = 0; i < CA; ++i) {
                uint4 v = *(const uint4*)(a_ptr[i] + (int64_t)pl * p.a_plane + koff);
                ra[pl][i] = valid ? v : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < CW; ++i) {
                uint4 v = *(const uint4*)(w_ptr[i] + (int64_t)pl * p.w_plane + koff);
                rw[pl][i] = valid ? v : make_uint4(0, 0, 0, 0);
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int pl = 0; pl < NT; ++pl) {
#pragma unroll
            for (int i = 0; i < CA; ++i) {
                int row = ld_row + i * ROWS_PER_PASS;
                *(uint4*)(sA + pl * BM * 128 + lds_off(row, ld_c)) = ra[pl][i];
            }
#pragma unroll
            for (int i = 0; i < CW; ++i) {
                int row = ld_row + i * ROWS_PER_PASS;
                *(uint4*)(sW + pl * BN * 128 + lds_off(row, ld_c)) = rw[pl][i];
            }
        }
    };

    f32x4 acc[NI][MI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    load_tile(0);
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();
        store_tile();
        __syncthreads();
        if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int c = 4 * s + (lane >> 4);
            V8 af[NT][MI], wf[NT][NI];
#pragma unroll
            for (int pl = 0; pl < NT; ++pl) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    int row = wm * TM + mi * 16 + (lane & 15);
                    af[pl][mi] = *(const V8*)(sA + pl * BM * 128 + lds_off(row, c));
                }
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) {
                    int row = wn * TN + ni * 16 + (lane & 15);
                    wf[pl][ni] = *(const V8*)(sW + pl * BN * 128 + lds_off(row, c));
                }
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    if (NT > 1) {
                        acc[ni][mi] = mfma16(wf[NT - 1][ni], af[0][mi], acc[ni][mi]);  // lo(W) * hi(A)
                        acc[ni][mi] = mfma16(wf[0][ni], af[NT - 1][mi], acc[ni][mi]);  // hi(W) * lo(A)
                    }
                    acc[ni][mi] = mfma16(wf[0][ni], af[0][mi], acc[ni][mi]);
                }
        }
    }

    // ---- epilogue ----
    gemm_epilogue<T, NT, MI, NI>(p, z, acc, m0 + wm * TM, n0 + wn * TN, lane);
}

// ---------------------------------------------------------------------------------------------------------------
// LDS-DMA variant of the tile kernel: the same fragment layout, products and epilogue as gemm_kernel, but the operand
// tiles go global -> LDS with buffer_load_dwordx4 ... lds into a ring of STAGES stages, STAGES - 1 K tiles ahead of the
// MFMAs (gemm_kernel stages one K tile through registers and exposes a memory round trip per tile, which is what short
// products -- single utterances, narrow heads -- spend their time on).  One s_barrier per K tile:
//   iteration t: wait for tile t (this wave's pieces), barrier, issue the DMA of tile t + STAGES - 1 into the stage tile
//   t - 1 was read from (every wave finished those reads before the barrier), multiply tile t.
// Two shapes, 4 waves each: 128 x 64 (waves 4 x 1, 3 stages) and 64 x 32 (waves 2 x 2, 6 stages: more workgroups and a
// deeper ring for the shortest products, which are bound by how many bytes the chip keeps in flight).
// The LDS image of a DMA is lane-linear, so the bank swizzle of lds_off() sits on the per-lane SOURCE address.
// Requires K % 64 == 0, 16-byte aligned operand rows, 32-bit byte offsets inside a tile.
// ---------------------------------------------------------------------------------------------------------------
template <int N_OUTSTANDING>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_OUTSTANDING) : "memory");
}

template <typename T, int NT, int BM, int BN, int WM, int WN, int STAGES>
__global__ __launch_bounds__(256) void gemm_dma_kernel(const GemmParams p) {
    typedef typename Vec8<T>::type V8;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    static_assert(WM * WN == 4 && BM % (32 * 1) == 0 && BN % 32 == 0, "4 waves, whole DMA pieces per wave");
    constexpr int TM = BM / WM, TN = BN / WN;
    constexpr int MI = TM / 16, NI = TN / 16;
    constexpr int APW = BM / 32, WPW = BN / 32;   // DMA pieces (8 rows x 128 B) per wave, K tile and plane
    constexpr int PLANE = (BM + BN) * 128;         // bytes of one plane of one stage: A rows then W rows, 128 B (64 k) each
    constexpr int STAGE = NT * PLANE;
    constexpr int PPT = NT * (APW + WPW);          // DMA instructions per wave and K tile
    constexpr int AHEAD = STAGES - 1;
    static_assert(AHEAD * PPT < 64, "vmcnt range");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int z = blockIdx.z;
    const T* A = (const T*)p.A + (int64_t)z * p.za;
    const T* W = (const T*)p.W + (int64_t)z * p.zw;

    // ---- DMA state: SGPR descriptors at the tile's first rows + 32-bit per-lane byte offsets ----
    const int m0c = m0 < p.M ? m0 : p.M - 1, n0c = n0 < p.N ? n0 : p.N - 1;
    const int64_t b0 = m0c / p.rows_per_batch;
    const int64_t a_tile = b0 * p.a_batch_stride + (m0c - b0 * p.rows_per_batch) * p.lda;
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(A + a_tile), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(W + (int64_t)n0c * p.ldw), 0, -1, 0x00020000);
    // fixed bounds: hipcc 7.2 silently drops the host stub of a kernel template whose called lambda captures an array
    // whose bound depends on a template parameter
    static_assert(APW <= 4 && WPW <= 2, "offset arrays");
    uint32_t a_off[4], w_off[2];
#pragma unroll
    for (int j = 0; j < APW; ++j) {
        const int row = (wave * APW + j) * 8 + (lane >> 3);
        const int lc = (lane & 7) ^ ((row >> 1) & 7);  // logical 16-byte chunk stored at physical chunk lane & 7
        int rr = m0 + row;
        rr = rr < p.M ? rr : p.M - 1;
        const int64_t b = rr / p.rows_per_batch;
        const int64_t t = rr - b * p.rows_per_batch;
        a_off[j] = (uint32_t)((b * p.a_batch_stride + t * p.lda - a_tile + lc * 8) * 2);
    }
#pragma unroll
    for (int j = 0; j < WPW; ++j) {
        const int row = (wave * WPW + j) * 8 + (lane >> 3);
        const int lc = (lane & 7) ^ ((row >> 1) & 7);
        int rn = n0 + row;
        rn = rn < p.N ? rn : p.N - 1;
        w_off[j] = (uint32_t)(((int64_t)(rn - n0c) * p.ldw + lc * 8) * 2);
    }
    const uint32_t a_plane_b = (uint32_t)(p.a_plane * 2), w_plane_b = (uint32_t)(p.w_plane * 2);
    auto stage = [&](int kt, int st) {
        unsigned char* base = smem + st * STAGE;
#pragma unroll
        for (int pl = 0; pl < NT; ++pl) {
            const uint32_t so_a = pl * a_plane_b + (uint32_t)kt * 128, so_w = pl * w_plane_b + (uint32_t)kt * 128;
            unsigned char* dst = base + pl * PLANE;
#pragma unroll
            for (int j = 0; j < APW; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_ptr_t)(dst + (wave * APW + j) * 1024), 16, a_off[j], so_a, 0, 0);
#pragma unroll
            for (int j = 0; j < WPW; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr_t)(dst + BM * 128 + (wave * WPW + j) * 1024), 16,
                                                         w_off[j], so_w, 0, 0);
        }
    };
    // waits until at most the pieces of `tiles_ahead` later K tiles are still in flight (wave-uniform run-time count)
    auto wait_tile = [](int tiles_ahead) {
        switch (tiles_ahead) {
            case 0: wait_vmcnt<0>(); break;
            case 1: wait_vmcnt<PPT>(); break;
            case 2: wait_vmcnt<(AHEAD >= 2 ? 2 : 0) * PPT>(); break;
            case 3: wait_vmcnt<(AHEAD >= 3 ? 3 : 0) * PPT>(); break;
            case 4: wait_vmcnt<(AHEAD >= 4 ? 4 : 0) * PPT>(); break;
            default: wait_vmcnt<(AHEAD >= 5 ? 5 : 0) * PPT>(); break;
        }
    };
    static_assert(AHEAD <= 6, "wait_tile covers up to 5 tiles ahead");

    f32x4 acc[NI][MI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK;
    for (int t = 0; t < AHEAD && t < nk; ++t) stage(t, t);
    for (int kt = 0; kt < nk; ++kt) {
        // tile kt has landed once only the tiles issued after it (kt + 1 .. kt + AHEAD - 1) may still be in flight
        const int later = nk - 1 - kt;
        wait_tile(later < AHEAD - 1 ? later : AHEAD - 1);
        __builtin_amdgcn_s_barrier();
        if (kt + AHEAD < nk) stage(kt + AHEAD, (kt + AHEAD) % STAGES);
        const unsigned char* sb = smem + (kt % STAGES) * STAGE;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int c = 4 * s2 + (lane >> 4);
            V8 af[NT][MI], wf[NT][NI];
#pragma unroll
            for (int pl = 0; pl < NT; ++pl) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    af[pl][mi] = *(const V8*)(sb + pl * PLANE + lds_off(wm * TM + mi * 16 + (lane & 15), c));
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    wf[pl][ni] = *(const V8*)(sb + pl * PLANE + BM * 128 + lds_off(wn * TN + ni * 16 + (lane & 15), c));
            }
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    if (NT > 1) {
                        acc[ni][mi] = mfma16(wf[NT - 1][ni], af[0][mi], acc[ni][mi]);  // lo(W) * hi(A)
                        acc[ni][mi] = mfma16(wf[0][ni], af[NT - 1][mi], acc[ni][mi]);  // hi(W) * lo(A)
                    }
                    acc[ni][mi] = mfma16(wf[0][ni], af[0][mi], acc[ni][mi]);
                }
        }
        // this wave's reads of the stage are complete before it can pass the next barrier
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    gemm_epilogue<T, NT, MI, NI>(p, z, acc, m0 + wm * TM, n0 + wn * TN, lane);
}

// Split-K fix-up: sums the `splits` raw fp32 partial slabs [M, N] of a product (slab stride `slab` floats, in slab order,
// so the result does not depend on scheduling) and runs the shared epilogue on the totals.  A wave owns a 16 x 64 patch.
template <typename T, int NT>
__global__ __launch_bounds__(256) void splitk_fixup_kernel(const GemmParams p, const float* __restrict__ ws, int splits,
                                                           int64_t slab) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m_base = (blockIdx.y * 4 + wave) * 16, n_base = blockIdx.x * 64;
    if (m_base >= p.M) return;
    f32x4 acc[4][1];
    const int m = m_base + (lane & 15);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        acc[ni][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int nb = n_base + ni * 16 + 4 * (lane >> 4);
        if (m >= p.M || nb >= p.N) continue;
        const float* src = ws + (int64_t)m * p.N + nb;
        if ((p.N & 3) == 0) {
            // eight slab reads in flight per lane (a rolled loop would pay one memory round trip per slab)
            for (int k0 = 0; k0 < splits; k0 += 8) {
                float4 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    v[j] = k0 + j < splits ? *(const float4*)(src + (k0 + j) * slab) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    acc[ni][0][0] += v[j].x; acc[ni][0][1] += v[j].y; acc[ni][0][2] += v[j].z; acc[ni][0][3] += v[j].w;
                }
            }
        } else {
            for (int k = 0; k < splits; ++k)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (nb + r < p.N) acc[ni][0][r] += src[k * slab + r];
        }
    }
    gemm_epilogue<T, NT, 1, 4>(p, 0, acc, m_base, n_base, lane);
}

// ---------------------------------------------------------------------------------------------------------------
// Ping-pong GEMM for the large products (conv layers 1-6, feature projection, QKV / out-proj / FFN, phoneme head):
//   persistent workgroups (one per CU) walk 256 x 256 output tiles; 8 waves = 2 groups x 4 waves (one wave of each
//   group on every SIMD), wave tile 128 x 64 (8 x 4 accumulator fragments of v_mfma_f32_16x16x32).
// The K dimension is consumed in "sub-steps" of 32 elements of ONE 16-bit plane: 256 A rows + 256 W rows of 64 bytes
// = 32 KiB, DMA-ed straight into a 4-slot LDS ring with buffer_load_dwordx4 ... lds (no staging registers, no
// ds_write; SGPR descriptor + 32-bit per-lane offset, so the per-lane DMA state is four registers).
// With two planes (hi/lo split operands) a 32-deep K slice takes three segments: H (hi planes of A and W: hi.hi),
// LW (lo plane of W: lo(W).hi(A)) and LA (lo plane of A: hi(W).lo(A)); the hi fragments stay in registers.
// Every wave alternates between a LOAD segment (ds_read_b128 fragment reads of sub-step u + its DMA pieces of
// sub-step u+3) and an MFMA segment (32 MFMAs on the fragments just read), one s_barrier after each.  The second
// group runs one barrier behind the first, so on each SIMD one wave is in its MFMA segment while its partner reads
// LDS / issues DMA: the matrix pipe does not wait for a load segment.
//   visibility of sub-step v: DMA issued in LOAD segment v-3; each issuing wave retires it with a counted vmcnt that
//   leaves only the pieces of sub-steps v+1, v+2 in flight -- group 0 at the end of its MFMA segment v-1, group 1 at
//   the end of its LOAD segment v-1 -- i.e. before barrier 2v-1 for both groups; first read in interval 2v.
//   Slot v%4 held sub-step v-4, last read (and waited for, lgkmcnt(0)) before barrier 2v-7; the earliest overwrite is
//   issued after barrier 2v-7.
// The LDS image of a DMA is lane-linear (wave-uniform base + 16 B x lane), so the bank-conflict swizzle is applied to
// the per-lane SOURCE address and again on the fragment read (cdna_hip_programming.md rule 21).
// Epilogue: accumulators -> per-wave LDS patch (32 x 64 fp32, XOR-swizzled) -> row-major read-back, so that bias /
// GELU / residual / row mask / 16-bit split run on consecutive columns per lane and every global store instruction
// writes whole 128 / 256-byte row segments.  The patches live in ring slots 2-3; the first two sub-steps of the NEXT
// tile are DMA-ed into slots 0-1 before the epilogue starts, so the prologue latency of a tile hides under the
// epilogue of its predecessor.
// Requires K % (128 / planes) == 0, N % 4 == 0, 16-byte aligned operand rows.
// ---------------------------------------------------------------------------------------------------------------
namespace pp {
constexpr int BN = 256, KS = 32;  // tile rows: 32 x MI (template parameter of the kernel)
constexpr int SLOT = 32768, W_OFF = 16384, NSLOT = 4;
constexpr int EPI_BASE = 2 * SLOT;        // epilogue patches: slots 2-3
constexpr int EPI_WAVE = 32 * 64 * 4;     // bytes of epilogue patch per wave: 32 rows x 64 fp32
constexpr int LDS_BYTES = SLOT * NSLOT;   // 128 KiB
// float index of 16-byte chunk `chunk` (0..15) of patch row `row`: conflict-free for the fragment writes (8 consecutive
// rows, one chunk) and for row-major reads whose 16-lane groups stay inside rows of equal row & 7
__device__ __forceinline__ int es_idx(int row, int chunk) { return row * 64 + ((chunk ^ (row & 7)) << 2); }
}  // namespace pp

// In-place MFMA (accumulator tied to its own registers).  With the builtin, hipcc gives every result a fresh register
// quad; at ~200 live registers the resulting tuple fragmentation spills into the main loop, and a scratch reload's
// s_waitcnt vmcnt(0) would drain the LDS-DMA ring.  Operands come straight from ds_read (the compiler places the
// lgkmcnt wait); independent accumulators issue back to back; the epilogue waits out the MFMA latency explicitly.
__device__ __forceinline__ void mfma16_acc(f32x4& c, f16x8 a, f16x8 b) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mfma16_acc(f32x4& c, bf16x8 a, bf16x8 b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}

// stage accumulator fragments 2q, 2q+1 (32 rows x 64 columns) of the wave block into its fp32 LDS patch
template <int MI>
__device__ __forceinline__ void pp_stage_round(float* es, f32x4 (&acc)[4][MI], int q, int lane) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
            *(f32x4*)(es + pp::es_idx(h * 16 + (lane & 15), ni * 4 + (lane >> 4))) = acc[ni][2 * q + h];
}

// ---- branch-free epilogues for interior wave blocks (all 128 x 64 outputs in range, no row mask) ----
// fp32 output (+ bias, + residual): a lane owns 4 consecutive columns; 16 lanes cover a 256-byte row segment.
// All loads of a 32-row round are issued before the first use, all stores after: no wait inside the round.
template <bool RES, int MI>
__device__ __forceinline__ void pp_epilogue_f32(const GemmParams& p, f32x4 (&acc)[4][MI], float* es, int lane, int mw, int nw,
                                                float* out_f32) {
    const int ch = lane & 15, rq = (lane >> 4) * 8;  // rows rq + i: every 16-lane group reads rows of equal row & 7
    const int n = nw + ch * 4;
    const float scale = p.scale;
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) b4 = *(const float4*)(p.bias + n);
    float* optr = out_f32 + (int64_t)(mw + rq) * p.ldo + n;
    const float* rptr = RES ? p.residual + (int64_t)(mw + rq) * p.ldr + n : nullptr;
#pragma unroll
    for (int q = 0; q < MI / 2; ++q) {
        float4 r[8];
        if (RES) {
#pragma unroll
            for (int i = 0; i < 8; ++i) r[i] = *(const float4*)(rptr + (int64_t)(q * 32 + i) * p.ldr);
        }
        pp_stage_round(es, acc, q, lane);
        f32x4 c[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = *(const f32x4*)(es + pp::es_idx(rq + i, ch));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float4 v = make_float4(fmaf(c[i][0], scale, b4.x), fmaf(c[i][1], scale, b4.y), fmaf(c[i][2], scale, b4.z),
                                   fmaf(c[i][3], scale, b4.w));
            if (RES) { v.x += r[i].x; v.y += r[i].y; v.z += r[i].z; v.w += r[i].w; }
            *(float4*)(optr + (int64_t)(q * 32 + i) * p.ldo) = v;
        }
    }
}

// 16-bit plane output (+ bias, optional GELU), or the Q / K / V scatter of the fused QKV projection (QK): a lane owns 8
// consecutive columns (one 16-byte store per plane); 8 lanes cover the 128-byte row segment of the wave block.
template <typename T, int NT, bool ACT, bool QK, int MI>
__device__ __forceinline__ void pp_epilogue_p16(const GemmParams& p, f32x4 (&acc)[4][MI], float* es, int lane, int mw, int nw) {
    typedef typename Vec8<T>::type V8;
    const int c8 = lane & 7, rs = lane >> 3;  // rows rs + 8i
    const int n = nw + c8 * 8;
    const float scale = p.scale;
    float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
    if (p.bias) { b0 = *(const float4*)(p.bias + n); b1 = *(const float4*)(p.bias + n + 4); }
    T* base;
    int64_t row_stride, plane;
    int b = 0, t = 0;
    if (QK) {
        const int D = p.H * p.dh;
        const int which = n / D;
        const int rem = n - which * D;
        const int hh = rem / p.dh, d = rem - hh * p.dh;
        base = (T*)(which == 0 ? p.q : (which == 1 ? p.k : p.v)) + (int64_t)hh * p.Tp * p.dh + d;
        row_stride = p.dh;
        plane = p.qk_plane;
        b = (mw + rs) / p.T;
        t = (mw + rs) - b * p.T;
    } else {
        base = (T*)p.out_p + (int64_t)(mw + rs) * p.ldp + n;
        row_stride = p.ldp;
        plane = p.out_plane;
    }
#pragma unroll
    for (int q = 0; q < MI / 2; ++q) {
        pp_stage_round(es, acc, q, lane);
        f32x4 c[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            c[i][0] = *(const f32x4*)(es + pp::es_idx(i * 8 + rs, 2 * c8));
            c[i][1] = *(const f32x4*)(es + pp::es_idx(i * 8 + rs, 2 * c8 + 1));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v[8] = {fmaf(c[i][0][0], scale, b0.x), fmaf(c[i][0][1], scale, b0.y), fmaf(c[i][0][2], scale, b0.z),
                          fmaf(c[i][0][3], scale, b0.w), fmaf(c[i][1][0], scale, b1.x), fmaf(c[i][1][1], scale, b1.y),
                          fmaf(c[i][1][2], scale, b1.z), fmaf(c[i][1][3], scale, b1.w)};
            V8 hv, lv;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                float x = ACT ? gelu_fast(v[r]) : v[r];
                T hi, lo = (T)0.f;
                split16<T, NT>(x, hi, lo);
                hv[r] = hi;
                lv[r] = lo;
            }
            T* dst;
            if (QK) {
                dst = base + ((int64_t)b * p.H * p.Tp + t) * row_stride;
                t += 8;  // rows advance by 8; T >= 8 here
                const bool wrap = t >= p.T;
                t -= wrap ? p.T : 0;
                b += wrap ? 1 : 0;
            } else {
                dst = base + (int64_t)(q * 32 + i * 8) * row_stride;
            }
            *(V8*)dst = hv;
            if (NT > 1) *(V8*)(dst + plane) = lv;
        }
    }
}

// edge blocks (M / N tails), row masks, combined fp32 + plane outputs: every feature, runtime flags
template <typename T, int NT, int MI>
__device__ __forceinline__ void pp_epilogue_generic(const GemmParams& p, f32x4 (&acc)[4][MI], float* es, int lane, int mw, int nw,
                                                    float* out_f32) {
    typedef typename Vec4<T>::type V4;
    const int D = p.H * p.dh;
    const float scale = p.scale;
    const int ch = lane & 15, rq = (lane >> 4) * 8;
    const int n = nw + ch * 4;
    const bool n_ok = n < p.N;  // N % 4 == 0: all four columns or none
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias && n_ok) bias4 = *(const float4*)(p.bias + n);
    int64_t qk_col = 0;
    T* qk_base = nullptr;
    if (p.mode == 1 && n_ok) {
        const int which = n / D;
        const int rem = n - which * D;
        const int hh = rem / p.dh, d = rem - hh * p.dh;
        qk_base = (T*)(which == 0 ? p.q : (which == 1 ? p.k : p.v));
        qk_col = (int64_t)hh * p.Tp * p.dh + d;
    }
    const bool need_bt = p.row_len || p.mode == 1;
    const int rt = p.mode == 1 ? p.T : p.rows_T;
#pragma unroll
    for (int q = 0; q < MI / 2; ++q) {
        pp_stage_round(es, acc, q, lane);
        // LDS operations of one wave complete in order: the reads below see the writes above
        int b = 0, t = 0;
        if (need_bt) {
            const int mfirst = mw + q * 32 + rq;
            b = mfirst / rt;
            t = mfirst - b * rt;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = mw + q * 32 + rq + i;
            const f32x4 c = *(const f32x4*)(es + pp::es_idx(rq + i, ch));
            if (m < p.M && n_ok) {
                float v[4] = {fmaf(c[0], scale, bias4.x), fmaf(c[1], scale, bias4.y), fmaf(c[2], scale, bias4.z),
                              fmaf(c[3], scale, bias4.w)};
                if (p.act == 1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = gelu_fast(v[r]);
                }
                if (p.residual) {
                    const float4 rr = *(const float4*)(p.residual + (int64_t)m * p.ldr + n);
                    v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
                }
                if (p.row_len && t >= p.row_len[b]) v[0] = v[1] = v[2] = v[3] = 0.f;
                if (p.mode == 1) {
                    T hi[4], lo[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) split16<T, NT>(v[r], hi[r], lo[r]);
                    T* dst = qk_base + ((int64_t)b * p.H * p.Tp + t) * p.dh + qk_col;
                    V4 hv = {hi[0], hi[1], hi[2], hi[3]};
                    *(V4*)dst = hv;
                    if (NT > 1) {
                        V4 lv = {lo[0], lo[1], lo[2], lo[3]};
                        *(V4*)(dst + p.qk_plane) = lv;
                    }
                } else {
                    if (out_f32) *(float4*)(out_f32 + (int64_t)m * p.ldo + n) = make_float4(v[0], v[1], v[2], v[3]);
                    if (p.out_p) {
                        T hi[4], lo[4];
#pragma unroll
                        for (int r = 0; r < 4; ++r) split16<T, NT>(v[r], hi[r], lo[r]);
                        T* dst = (T*)p.out_p + (int64_t)m * p.ldp + n;
                        V4 hv = {hi[0], hi[1], hi[2], hi[3]};
                        *(V4*)dst = hv;
                        if (NT > 1) {
                            V4 lv = {lo[0], lo[1], lo[2], lo[3]};
                            *(V4*)(dst + p.out_plane) = lv;
                        }
                    }
                }
            }
            if (need_bt) {
                t += 1;
                if (t >= rt) { t = 0; ++b; }
            }
        }
    }
}

// MI = accumulator fragments per wave along M: 8 -> the 256 x 256 tile described above; 4 -> a 128 x 256 tile (wave tile
// 64 x 64, one A piece per wave and sub-step, 16 MFMAs per segment) for products whose 256-row tiles cannot fill the chip.
template <typename T, int NT, int MI>
__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(const GemmParams p) {
    constexpr int BMK = MI * 32;   // tile rows
    constexpr int HALF = MI * 16;  // rows of a wave group
    constexpr int APW = MI / 4;    // A pieces (16 rows x 64 B) per wave and sub-step
    typedef typename Vec8<T>::type V8;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wc = wave & 3;
    const int ntn = (p.N + pp::BN - 1) / pp::BN, ntm = (p.M + BMK - 1) / BMK;
    const int tiles = ntn * ntm;
    // split-K (p.splits > 1, set by launch_gemm for products with too few tiles to fill the chip): a work unit is
    // (tile, K chunk of p.K elements); chunk ks reads A / W columns [ks * p.K, (ks + 1) * p.K) and writes its raw fp32
    // partial tile to slab ks of the workspace p.out_f32 points to (the fix-up kernel reduces and runs the epilogue)
    const int total = tiles * p.splits;

    // ---- fragment read offsets (bytes inside a slot) ----
    const int rd_chunk = ((lane >> 4) ^ ((lane >> 2) & 2)) << 4;
    const int a_rd = (grp * HALF + (lane & 15)) * 64 + rd_chunk;
    const int w_rd = pp::W_OFF + (wc * 64 + (lane & 15)) * 64 + rd_chunk;
    const uint32_t a_plane_b = (uint32_t)(p.a_plane * 2), w_plane_b = (uint32_t)(p.w_plane * 2);

    // ---- per-tile DMA state: this wave fills pieces APW*wave.. (16 rows x 64 B each) of the A part and 2*wave, 2*wave+1
    // of the W part of a slot; addresses are (wave-uniform tile base in an SGPR buffer descriptor) + (32-bit per-lane offset)
    int m0 = 0, n0 = 0, ks = 0;
    __amdgpu_buffer_rsrc_t a_rsrc, w_rsrc;
    uint32_t a_off[2], w_off[2];
    auto setup_tile = [&](int unit) {
        ks = unit / tiles;
        const int i = unit - ks * tiles;
        const int total = tiles;
        // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (private 4 MiB L2 each), so the tile
        // sequence numbers that share an XCD (equal i % 8) are given one contiguous chunk of a grouped tile sequence in
        // which the 32 co-resident tiles of an XCD form an 8 (M) x 4 (N) rectangle.  Pure speed: any placement gives
        // the same results.
        const int xcd = i & 7, q = total >> 3, r = total & 7;
        const int j = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (i >> 3);  // bijective remap
        constexpr int GM = 8;
        const int per_group = GM * ntn;
        const int group = j / per_group;
        const int first_m = group * GM;
        const int gsize = ntm - first_m < GM ? ntm - first_m : GM;
        const int in_group = j - group * per_group;
        m0 = (first_m + in_group % gsize) * BMK;
        n0 = (in_group / gsize) * pp::BN;
        const int m0c = m0 < p.M ? m0 : p.M - 1, n0c = n0 < p.N ? n0 : p.N - 1;
        const int64_t b0 = m0c / p.rows_per_batch;
        const int64_t a_tile = b0 * p.a_batch_stride + (m0c - b0 * p.rows_per_batch) * p.lda;  // element offset of row m0
        const int64_t k_first = (int64_t)ks * p.K;
        a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((const T*)p.A + a_tile + k_first), 0, -1, 0x00020000);
        w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((const T*)p.W + (int64_t)n0c * p.ldw + k_first), 0, -1, 0x00020000);
#pragma unroll
        for (int jj = 0; jj < APW; ++jj) {
            const int row = (wave * APW + jj) * 16 + ((tid & 63) >> 2);
            const int lc = (tid & 3) ^ ((row >> 2) & 2);  // logical 16-byte chunk stored at physical chunk lane & 3
            int rr = m0 + row;
            rr = rr < p.M ? rr : p.M - 1;
            const int64_t b = rr / p.rows_per_batch;
            const int64_t t = rr - b * p.rows_per_batch;
            a_off[jj] = (uint32_t)((b * p.a_batch_stride + t * p.lda - a_tile + lc * 8) * 2);
        }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int row = (wave * 2 + jj) * 16 + ((tid & 63) >> 2);
            const int lc = (tid & 3) ^ ((row >> 2) & 2);
            int rn = n0 + row;
            rn = rn < p.N ? rn : p.N - 1;
            w_off[jj] = (uint32_t)(((int64_t)(rn - n0c) * p.ldw + lc * 8) * 2);
        }
    };

    f32x4 acc[4][MI];
    V8 fa[NT][MI], fw[NT][4];

    // DMA of the A part and / or W part of (plane, k-offset) into a ring slot
    auto stage = [&](int slot, int plane, bool do_a, bool do_w, int koff) {
        unsigned char* dst = smem + slot * pp::SLOT + wave * 2048;
        if (do_a) {
            const uint32_t so = plane * a_plane_b + (uint32_t)koff * 2;
            unsigned char* dst_a = smem + slot * pp::SLOT + wave * (APW * 1024);
#pragma unroll
            for (int j = 0; j < APW; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_ptr_t)(dst_a + j * 1024), 16, a_off[j], so, 0, 0);
        }
        if (do_w) {
            const uint32_t so = plane * w_plane_b + (uint32_t)koff * 2;
#pragma unroll
            for (int j = 0; j < 2; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr_t)(dst + pp::W_OFF + j * 1024), 16, w_off[j], so, 0, 0);
        }
    };
    // the first two sub-steps of a tile (slots 0 and 1; none of them touches the epilogue patches in slots 2-3)
    auto stage_head = [&]() {
        if constexpr (NT == 1) {
            stage(0, 0, true, true, 0);
            stage(1, 0, true, true, pp::KS);
        } else {
            stage(0, 0, true, true, 0);   // H(0)
            stage(1, 1, false, true, 0);  // LW(0)
        }
    };

    // s_waitcnt vmcnt(n) for a wave-uniform run-time n (the instruction takes an immediate)
    auto wait_dma = [](int keep) {
        switch (keep) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        }
    };
#ifdef AMX_PP_STAMP
    // developer diagnostic (tools/gemm_bench.hip, -DAMX_PP_STAMP): cycles per phase of the LOAD / MFMA segments, summed
    // over the main loops in scalar registers; lane 0 of waves 0 and 4 writes them to p.stamps
    unsigned long long st_load = 0, st_bar1 = 0, st_mfma = 0, st_vm = 0, st_bar2 = 0, st_prev = 0, st_begin_rt = 0, st_loop = 0;
    auto stamp = []() -> unsigned long long {
        unsigned long long t;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        __builtin_amdgcn_sched_barrier(0);
        return t;
    };
    {
        unsigned long long t;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        st_begin_rt = t;
    }
    const unsigned long long st_begin = stamp();
#endif
    // One LOAD segment + one MFMA segment.  Compile-time code: read slot RS (A and / or W fragments into plane DPL of
    // the fragment registers), product PROD (0: W.A on plane 0, 1: lo(W).hi(A), 2: hi(W).lo(A)), the DMA that refills,
    // three segments ahead, the same parts of slot SS from plane SPL, and CNT2 = DMA pieces of the segment two ahead.
    auto segment = [&](auto code, bool stage_ok, int stage_koff, bool next2_ok) {
        constexpr int C = decltype(code)::value;
        constexpr int RS = C & 3, RA = (C >> 2) & 1, RW = (C >> 3) & 1, DPL = (C >> 4) & 1, PROD = (C >> 5) & 3,
                      SS = (C >> 7) & 3, SPL = (C >> 9) & 1, CNT2 = (C >> 10) & 7;
        constexpr int CNT3 = APW * RA + 2 * RW;
        // DMA pieces that may stay in flight past this segment's wait: those of sub-steps u+2 and u+3
        const int keep = (stage_ok ? CNT3 : 0) + (next2_ok ? CNT2 : 0);
        // ---------------- LOAD segment ----------------
#ifdef AMX_PP_STAMP
        const unsigned long long t0 = stamp();
        if (st_prev) st_bar2 += t0 - st_prev;
#endif
        const unsigned char* s = smem + RS * pp::SLOT;
        if (RA) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) fa[DPL][mi] = *(const V8*)(s + a_rd + mi * 1024);
        }
        if (RW) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) fw[DPL][ni] = *(const V8*)(s + w_rd + ni * 1024);
        }
        if (stage_ok) stage(SS, SPL, RA, RW, stage_koff);
        if (grp == 1) wait_dma(keep);  // group 1 publishes sub-step u+1 with the barrier that ends its LOAD segment
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef AMX_PP_STAMP
        const unsigned long long t1 = stamp();
        st_load += t1 - t0;
#endif
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#ifdef AMX_PP_STAMP
        const unsigned long long t2 = stamp();
        st_bar1 += t2 - t1;
#endif
        // ---------------- MFMA segment ----------------
        __builtin_amdgcn_s_setprio(1);
        constexpr int PW = PROD == 1 ? NT - 1 : 0, PA = PROD == 2 ? NT - 1 : 0;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) mfma16_acc(acc[ni][mi], fw[PW][ni], fa[PA][mi]);
        __builtin_amdgcn_s_setprio(0);
#ifdef AMX_PP_STAMP
        const unsigned long long t3 = stamp();
        st_mfma += t3 - t2;
#endif
        if (grp == 0) wait_dma(keep);  // group 0 publishes sub-step u+1 with the barrier that ends its MFMA segment
#ifdef AMX_PP_STAMP
        const unsigned long long t4 = stamp();
        st_vm += t4 - t3;
        st_prev = t4;
#endif
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
#define PP_CODE(RS, RA, RW, DPL, PROD, SS, SPL, CNT2) \
    std::integral_constant<int, (RS) | ((RA) << 2) | ((RW) << 3) | ((DPL) << 4) | ((PROD) << 5) | ((SS) << 7) | ((SPL) << 9) | ((CNT2) << 10)> {}

    int it = blockIdx.x;
    setup_tile(it);
    stage_head();
    for (;;) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < MI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifdef AMX_PP_STAMP
        const unsigned long long st_l0 = stamp();
        st_prev = 0;
#endif
        if constexpr (NT == 1) {
            // segments = 32-deep K slices; slice u lives in slot u % 4
            const int nseg = p.K / pp::KS;
            stage(2, 0, true, true, 2 * pp::KS);
            wait_dma(2 * (APW + 2));  // slice 0 has landed
            __builtin_amdgcn_s_barrier();
            if (grp == 1) __builtin_amdgcn_s_barrier();  // group 1 runs one barrier behind group 0
            for (int u = 0; u < nseg; u += 4) {
                segment(PP_CODE(0, 1, 1, 0, 0, 3, 0, APW + 2), u + 3 < nseg, (u + 3) * pp::KS, u + 2 < nseg);
                segment(PP_CODE(1, 1, 1, 0, 0, 0, 0, APW + 2), u + 4 < nseg, (u + 4) * pp::KS, u + 3 < nseg);
                segment(PP_CODE(2, 1, 1, 0, 0, 1, 0, APW + 2), u + 5 < nseg, (u + 5) * pp::KS, u + 4 < nseg);
                segment(PP_CODE(3, 1, 1, 0, 0, 2, 0, APW + 2), u + 6 < nseg, (u + 6) * pp::KS, u + 5 < nseg);
            }
        } else {
            // hi planes of slice k in slot 2*(k%2), lo planes in slot 2*(k%2)+1
            const int nk = p.K / pp::KS;
            stage(1, 1, true, false, 0);  // LA(0)
            wait_dma(2 + APW);            // H(0) has landed
            __builtin_amdgcn_s_barrier();
            if (grp == 1) __builtin_amdgcn_s_barrier();
            for (int k = 0; k < nk; k += 2) {  // nk is even
                const bool ok2 = k + 2 < nk;
                segment(PP_CODE(0, 1, 1, 0, 0, 2, 0, APW), true, (k + 1) * pp::KS, true);
                segment(PP_CODE(1, 0, 1, 1, 1, 3, 1, APW + 2), true, (k + 1) * pp::KS, true);
                segment(PP_CODE(1, 1, 0, 1, 2, 3, 1, 2), true, (k + 1) * pp::KS, true);
                segment(PP_CODE(2, 1, 1, 0, 0, 0, 0, APW), ok2, (k + 2) * pp::KS, true);
                segment(PP_CODE(3, 0, 1, 1, 1, 1, 1, APW + 2), ok2, (k + 2) * pp::KS, ok2);
                segment(PP_CODE(3, 1, 0, 1, 2, 1, 1, 2), ok2, (k + 2) * pp::KS, ok2);
            }
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();  // realign the groups: every LDS read and DMA of the ring is complete
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // MFMA results -> VALU / LDS readers (the MFMAs are inline asm)
#ifdef AMX_PP_STAMP
        st_loop += stamp() - st_l0;
#endif

        // ---- next tile: set up its DMA state and start its first two sub-steps under this tile's epilogue ----
        const int mw = m0 + grp * HALF, nw = n0 + wc * 64;
        float* out_f32 = p.out_f32 ? p.out_f32 + (int64_t)ks * p.split_out : nullptr;
        const int next = it + gridDim.x;
        const bool has_next = next < total;
        if (has_next) {
            setup_tile(next);
            stage_head();
        }

        // ---------------------------------------- epilogue ----------------------------------------
#ifndef AMX_ABLATE_NO_EPI
        // make the lane id opaque so that no epilogue address arithmetic is hoisted above the main loop (register pressure)
        asm volatile("" : "+v"(lane));
        if (nw < p.N && mw < p.M) {
            float* es = (float*)(smem + pp::EPI_BASE + wave * pp::EPI_WAVE);
            bool done = false;
            if (mw + HALF <= p.M && nw + 64 <= p.N && !p.row_len && p.vec_ok) {
                // interior block: branch-free epilogues
                if (p.mode == 1) {
                    if (p.T >= 8 && p.dh % 8 == 0 && p.qk_plane % 8 == 0 && !(((uintptr_t)p.q | (uintptr_t)p.k | (uintptr_t)p.v) & 15)) {
                        pp_epilogue_p16<T, NT, false, true, MI>(p, acc, es, lane, mw, nw);
                        done = true;
                    }
                } else if (p.out_f32 && !p.out_p && p.act == 0) {
                    if (p.residual) pp_epilogue_f32<true, MI>(p, acc, es, lane, mw, nw, out_f32);
                    else pp_epilogue_f32<false, MI>(p, acc, es, lane, mw, nw, out_f32);
                    done = true;
                } else if (p.out_p && !p.out_f32 && !p.residual && p.ldp % 8 == 0 && p.out_plane % 8 == 0 && !((uintptr_t)p.out_p & 15)) {
                    if (p.act == 1) pp_epilogue_p16<T, NT, true, false, MI>(p, acc, es, lane, mw, nw);
                    else pp_epilogue_p16<T, NT, false, false, MI>(p, acc, es, lane, mw, nw);
                    done = true;
                }
            }
            if (!done) pp_epilogue_generic<T, NT, MI>(p, acc, es, lane, mw, nw, out_f32);
        }
#else
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < MI; ++j) asm volatile("" ::"v"(acc[i][j]));
#endif
        if (!has_next) break;
        it = next;
        // every wave is done with its patch (its LDS reads have returned) before slot 2 is refilled
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
#undef PP_CODE
#ifdef AMX_PP_STAMP
    if (p.stamps && (wave == 0 || wave == 4) && (tid & 63) == 0) {
        const unsigned long long st_end = stamp();
        unsigned long long rt;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt)::"memory");
        unsigned long long* o = p.stamps + ((int64_t)blockIdx.x * 2 + (wave >> 2)) * 10;
        o[0] = st_load; o[1] = st_bar1; o[2] = st_mfma; o[3] = st_vm; o[4] = st_bar2; o[5] = st_loop;
        o[6] = rt - st_begin_rt; o[7] = 1; o[8] = st_end - st_begin; o[9] = (unsigned long long)((total - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x);
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// Row-complete ping-pong GEMM with fused LayerNorm + GELU (conv layers of the feature extractor, N = 512):
//   out planes = GELU(LayerNorm_N(A.W^T * scale + bias; gamma, beta, eps))
// Same pipeline as gemm_pp_kernel (4-slot LDS-DMA ring, two wave groups in anti-phase, three segments per K slice with
// two planes), but a 128 x 512 tile so that one workgroup owns whole output rows: wave tile 64 x 128 (4 x 8 accumulator
// fragments), slot = 128 A rows + 512 W rows of 64 B = 40 KiB, ring = the whole 160 KiB LDS.  Per wave and sub-step: 1 DMA
// piece of A, 4 of W.  The LayerNorm statistics are reduced over the 4 lane groups of a wave (ds_bpermute) and over the
// 4 waves that share a row block (LDS scratch + barrier), two-pass (mean, then centred variance) like the reference's
// fp32 LayerNorm; the fp32 pre-normalisation tensor never reaches HBM.
// ---------------------------------------------------------------------------------------------------------------
namespace ppw {
constexpr int BM = 128, BN = 512, KS = 32;
constexpr int A_BYTES = BM * 64, W_BYTES = BN * 64, SLOT = A_BYTES + W_BYTES, W_OFF = A_BYTES;  // 8 + 32 KiB
constexpr int LDS_BYTES = 4 * SLOT;                  // 160 KiB
constexpr int EPI_BASE = 2 * SLOT;                   // patches: slots 2-3 (80 KiB), 8 waves x 8 KiB
constexpr int RED_BASE = EPI_BASE + 8 * pp::EPI_WAVE;  // 2 x [2 groups][4 waves][64 rows] fp32 = 4 KiB
constexpr int AJ = 1, WJ = 4;                        // DMA pieces per wave and sub-step
}  // namespace ppw

template <typename T, int NT>
__global__ __launch_bounds__(512, 2) void gemm_ln_kernel(const GemmParams p) {
    typedef typename Vec8<T>::type V8;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wc = wave & 3;
    const int total = (p.M + ppw::BM - 1) / ppw::BM;  // N == BN: tiles along M only

    const int rd_chunk = ((lane >> 4) ^ ((lane >> 2) & 2)) << 4;
    const int a_rd = (grp * 64 + (lane & 15)) * 64 + rd_chunk;
    const int w_rd = ppw::W_OFF + (wc * 128 + (lane & 15)) * 64 + rd_chunk;
    const uint32_t a_plane_b = (uint32_t)(p.a_plane * 2), w_plane_b = (uint32_t)(p.w_plane * 2);

    int m0 = 0;
    __amdgpu_buffer_rsrc_t a_rsrc;
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, -1, 0x00020000);
    uint32_t a_off, w_off[ppw::WJ];
#pragma unroll
    for (int j = 0; j < ppw::WJ; ++j) {
        const int row = (wave * ppw::WJ + j) * 16 + (lane >> 2);
        const int lc = (lane & 3) ^ ((row >> 2) & 2);
        w_off[j] = (uint32_t)(((int64_t)row * p.ldw + lc * 8) * 2);
    }
    auto setup_tile = [&](int i) {
        m0 = i * ppw::BM;  // consecutive workgroups take consecutive row blocks: overlapping conv windows share lines
        const int64_t b0 = m0 / p.rows_per_batch;
        const int64_t a_tile = b0 * p.a_batch_stride + (m0 - b0 * p.rows_per_batch) * p.lda;
        a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((const T*)p.A + a_tile), 0, -1, 0x00020000);
        const int row = wave * 16 + ((tid & 63) >> 2);
        const int lc = (tid & 3) ^ ((row >> 2) & 2);
        int rr = m0 + row;
        rr = rr < p.M ? rr : p.M - 1;
        const int64_t b = rr / p.rows_per_batch;
        const int64_t t = rr - b * p.rows_per_batch;
        a_off = (uint32_t)((b * p.a_batch_stride + t * p.lda - a_tile + lc * 8) * 2);
    };

    f32x4 acc[8][4];  // [ni][mi]
    V8 fa[NT][4], fw[NT][8];

    auto stage = [&](int slot, int plane, bool do_a, bool do_w, int koff) {
        unsigned char* dst = smem + slot * ppw::SLOT;
        if (do_a)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_ptr_t)(dst + wave * 1024), 16, a_off,
                                                     plane * a_plane_b + (uint32_t)koff * 2, 0, 0);
        if (do_w) {
            const uint32_t so = plane * w_plane_b + (uint32_t)koff * 2;
#pragma unroll
            for (int j = 0; j < ppw::WJ; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr_t)(dst + ppw::W_OFF + (wave * ppw::WJ + j) * 1024), 16,
                                                         w_off[j], so, 0, 0);
        }
    };
    auto stage_head = [&]() {
        if constexpr (NT == 1) {
            stage(0, 0, true, true, 0);
            stage(1, 0, true, true, ppw::KS);
        } else {
            stage(0, 0, true, true, 0);   // H(0)
            stage(1, 1, false, true, 0);  // LW(0)
        }
    };
    auto wait_dma = [](int keep) {
        switch (keep) {
            case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
            case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
            case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        }
    };
    // code: RS | RA<<2 | RW<<3 | DPL<<4 | PROD<<5 | SS<<7 | SPL<<9 | RA2<<10 | RW2<<11 (parts staged by the segment two ahead)
    auto segment = [&](auto code, bool stage_ok, int stage_koff, bool next2_ok) {
        constexpr int C = decltype(code)::value;
        constexpr int RS = C & 3, RA = (C >> 2) & 1, RW = (C >> 3) & 1, DPL = (C >> 4) & 1, PROD = (C >> 5) & 3,
                      SS = (C >> 7) & 3, SPL = (C >> 9) & 1, RA2 = (C >> 10) & 1, RW2 = (C >> 11) & 1;
        constexpr int CNT3 = ppw::AJ * RA + ppw::WJ * RW, CNT2 = ppw::AJ * RA2 + ppw::WJ * RW2;
        const int keep = (stage_ok ? CNT3 : 0) + (next2_ok ? CNT2 : 0);
        const unsigned char* s = smem + RS * ppw::SLOT;
        if (RA) {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) fa[DPL][mi] = *(const V8*)(s + a_rd + mi * 1024);
        }
        if (RW) {
#pragma unroll
            for (int ni = 0; ni < 8; ++ni) fw[DPL][ni] = *(const V8*)(s + w_rd + ni * 1024);
        }
        if (stage_ok) stage(SS, SPL, RA, RW, stage_koff);
        if (grp == 1) wait_dma(keep);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        constexpr int PW = PROD == 1 ? NT - 1 : 0, PA = PROD == 2 ? NT - 1 : 0;
#pragma unroll
        for (int ni = 0; ni < 8; ++ni)
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) mfma16_acc(acc[ni][mi], fw[PW][ni], fa[PA][mi]);
        __builtin_amdgcn_s_setprio(0);
        if (grp == 0) wait_dma(keep);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
#define PPW_CODE(RS, RA, RW, DPL, PROD, SS, SPL, RA2, RW2) \
    std::integral_constant<int, (RS) | ((RA) << 2) | ((RW) << 3) | ((DPL) << 4) | ((PROD) << 5) | ((SS) << 7) | ((SPL) << 9) | ((RA2) << 10) | ((RW2) << 11)> {}

    int it = blockIdx.x;
    setup_tile(it);
    stage_head();
    for (;;) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (NT == 1) {
            const int nseg = p.K / ppw::KS;
            stage(2, 0, true, true, 2 * ppw::KS);
            asm volatile("s_waitcnt vmcnt(10)" ::: "memory");  // slice 0 has landed (slices 1, 2 in flight)
            __builtin_amdgcn_s_barrier();
            if (grp == 1) __builtin_amdgcn_s_barrier();
            for (int u = 0; u < nseg; u += 4) {
                segment(PPW_CODE(0, 1, 1, 0, 0, 3, 0, 1, 1), u + 3 < nseg, (u + 3) * ppw::KS, u + 2 < nseg);
                segment(PPW_CODE(1, 1, 1, 0, 0, 0, 0, 1, 1), u + 4 < nseg, (u + 4) * ppw::KS, u + 3 < nseg);
                segment(PPW_CODE(2, 1, 1, 0, 0, 1, 0, 1, 1), u + 5 < nseg, (u + 5) * ppw::KS, u + 4 < nseg);
                segment(PPW_CODE(3, 1, 1, 0, 0, 2, 0, 1, 1), u + 6 < nseg, (u + 6) * ppw::KS, u + 5 < nseg);
            }
        } else {
            const int nk = p.K / ppw::KS;
            stage(1, 1, true, false, 0);                      // LA(0): 1 piece
            asm volatile("s_waitcnt vmcnt(5)" ::: "memory");  // H(0) has landed (LW(0): 4 pieces, LA(0): 1 in flight)
            __builtin_amdgcn_s_barrier();
            if (grp == 1) __builtin_amdgcn_s_barrier();
            for (int k = 0; k < nk; k += 2) {  // nk is even
                const bool ok2 = k + 2 < nk;
                // H: two ahead = LA (A only); LW: two ahead = H (A + W); LA: two ahead = LW (W only)
                segment(PPW_CODE(0, 1, 1, 0, 0, 2, 0, 1, 0), true, (k + 1) * ppw::KS, true);
                segment(PPW_CODE(1, 0, 1, 1, 1, 3, 1, 1, 1), true, (k + 1) * ppw::KS, true);
                segment(PPW_CODE(1, 1, 0, 1, 2, 3, 1, 0, 1), true, (k + 1) * ppw::KS, true);
                segment(PPW_CODE(2, 1, 1, 0, 0, 0, 0, 1, 0), ok2, (k + 2) * ppw::KS, true);
                segment(PPW_CODE(3, 0, 1, 1, 1, 1, 1, 1, 1), ok2, (k + 2) * ppw::KS, ok2);
                segment(PPW_CODE(3, 1, 0, 1, 2, 1, 1, 0, 1), ok2, (k + 2) * ppw::KS, ok2);
            }
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");

        const int mw = m0 + grp * 64, nw = wc * 128;
        const int next = it + gridDim.x;
        const bool has_next = next < total;
        if (has_next) {
            setup_tile(next);
            stage_head();
        }

        // ---------------- epilogue: bias, LayerNorm over the 512 columns of every row, GELU, planes ----------------
        asm volatile("" : "+v"(lane));
        {
            const int lr = lane & 15, lg = lane >> 4;
            const float scale = p.scale;
            float* red = (float*)(smem + ppw::RED_BASE);  // [2 passes][2 groups][4 waves][64 rows]
            // v = acc * scale + bias
#pragma unroll
            for (int ni = 0; ni < 8; ++ni) {
                const float4 b4 = p.bias ? *(const float4*)(p.bias + nw + ni * 16 + 4 * lg) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    acc[ni][mi][0] = fmaf(acc[ni][mi][0], scale, b4.x);
                    acc[ni][mi][1] = fmaf(acc[ni][mi][1], scale, b4.y);
                    acc[ni][mi][2] = fmaf(acc[ni][mi][2], scale, b4.z);
                    acc[ni][mi][3] = fmaf(acc[ni][mi][3], scale, b4.w);
                }
            }
            const float inv_n = 1.0f / (float)ppw::BN;
            float mean[4], rstd[4];
            // pass 0: mean; pass 1: centred variance
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                float part[4];
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    float s = 0.f;
#pragma unroll
                    for (int ni = 0; ni < 8; ++ni)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float x = pass == 0 ? acc[ni][mi][r] : (acc[ni][mi][r] - mean[mi]);
                            s += pass == 0 ? x : x * x;
                        }
                    s += __shfl_xor(s, 16);
                    s += __shfl_xor(s, 32);
                    part[mi] = s;
                }
                float* mine = red + ((pass * 2 + grp) * 4 + wc) * 64;
                if (lg == 0) {
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) mine[mi * 16 + lr] = part[mi];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                const float* all = red + (pass * 2 + grp) * 4 * 64;
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    const float tot = (all[0 * 64 + mi * 16 + lr] + all[1 * 64 + mi * 16 + lr]) +
                                      (all[2 * 64 + mi * 16 + lr] + all[3 * 64 + mi * 16 + lr]);
                    if (pass == 0) mean[mi] = tot * inv_n;
                    else rstd[mi] = 1.0f / sqrtf(tot * inv_n + p.ln_eps);
                }
            }
            // normalise + affine + GELU in the accumulator layout, then patch -> row-major -> planes
#pragma unroll
            for (int ni = 0; ni < 8; ++ni) {
                const float4 g4 = *(const float4*)(p.ln_gamma + nw + ni * 16 + 4 * lg);
                const float4 e4 = *(const float4*)(p.ln_beta + nw + ni * 16 + 4 * lg);
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) {
                    acc[ni][mi][0] = gelu_fast(fmaf((acc[ni][mi][0] - mean[mi]) * rstd[mi], g4.x, e4.x));
                    acc[ni][mi][1] = gelu_fast(fmaf((acc[ni][mi][1] - mean[mi]) * rstd[mi], g4.y, e4.y));
                    acc[ni][mi][2] = gelu_fast(fmaf((acc[ni][mi][2] - mean[mi]) * rstd[mi], g4.z, e4.z));
                    acc[ni][mi][3] = gelu_fast(fmaf((acc[ni][mi][3] - mean[mi]) * rstd[mi], g4.w, e4.w));
                }
            }
            float* es = (float*)(smem + ppw::EPI_BASE + wave * pp::EPI_WAVE);
            const int c8 = lane & 7, rs = lane >> 3;
            // 4 rounds: row halves (mi 0-1 / 2-3) x column halves (ni 0-3 / 4-7), 32 rows x 64 columns each
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int mh = q >> 1, nh = q & 1;
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int ni = 0; ni < 4; ++ni)
                        *(f32x4*)(es + pp::es_idx(h * 16 + lr, ni * 4 + lg)) = acc[nh * 4 + ni][mh * 2 + h];
                f32x4 c[4][2];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    c[i][0] = *(const f32x4*)(es + pp::es_idx(i * 8 + rs, 2 * c8));
                    c[i][1] = *(const f32x4*)(es + pp::es_idx(i * 8 + rs, 2 * c8 + 1));
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = mw + mh * 32 + i * 8 + rs;
                    V8 hv, lv;
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        T hi, lo = (T)0.f;
                        split16<T, NT>(c[i][r >> 2][r & 3], hi, lo);
                        hv[r] = hi;
                        lv[r] = lo;
                    }
                    if (m < p.M) {
                        T* dst = (T*)p.out_p + (int64_t)m * p.ldp + nw + nh * 64 + c8 * 8;
                        *(V8*)dst = hv;
                        if (NT > 1) *(V8*)(dst + p.out_plane) = lv;
                    }
                }
            }
        }
        if (!has_next) break;
        it = next;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
#undef PPW_CODE
}

bool ln_eligible(int NT, const GemmParams& p) {
    if (g_force_generic_gemm) return false;
    if (!p.ln_gamma || !p.ln_beta || p.act != 1 || !p.out_p || p.out_f32 || p.residual || p.row_len || p.mode != 0) return false;
    if (p.N != ppw::BN || p.K % (128 / NT) != 0 || p.M < 1024) return false;
    if (p.lda % 8 || p.ldw % 8 || p.a_plane % 8 || p.w_plane % 8 || p.a_batch_stride % 8) return false;
    if (((uintptr_t)p.A & 15) || ((uintptr_t)p.W & 15)) return false;
    if (p.M > p.rows_per_batch && (p.rows_per_batch < 256 || p.a_batch_stride < (p.rows_per_batch - 1) * p.lda)) return false;
    {
        const int64_t a_span = (NT > 1 ? p.a_plane : 0) + (p.M > p.rows_per_batch ? p.a_batch_stride : 0) + 256 * p.lda + p.K;
        const int64_t w_span = (NT > 1 ? p.w_plane : 0) + 512 * p.ldw + p.K;
        if (a_span < 0 || w_span < 0 || a_span * 2 >= (int64_t)0xFFFFFF00 || w_span * 2 >= (int64_t)0xFFFFFF00) return false;
    }
    if (p.bias && ((uintptr_t)p.bias & 15)) return false;
    if (((uintptr_t)p.ln_gamma & 15) || ((uintptr_t)p.ln_beta & 15)) return false;
    if (p.ldp % 8 || p.out_plane % 8 || ((uintptr_t)p.out_p & 15)) return false;
    return true;
}

template <typename T, int NT>
void launch_gemm_ln(const GemmParams& p, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_ln_kernel<T, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, ppw::LDS_BYTES);
        attr_set = true;
    }
    const int tiles = (p.M + ppw::BM - 1) / ppw::BM;
    int cus = 256;
    {
        int dev = 0;
        hipDeviceProp_t prop;
        static int cached = 0;
        if (!cached && hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cached = prop.multiProcessorCount;
        if (cached > 0) cus = cached;
    }
    dim3 grid(tiles < cus ? tiles : cus, 1, 1);
    hipLaunchKernelGGL((gemm_ln_kernel<T, NT>), grid, dim3(512), ppw::LDS_BYTES, stream, p);
}

// below this many rows the 128 x 128 tile kernel with K chunks on grid.z is as fast (tools/gemm_bench `small`)
constexpr int PP_MIN_ROWS = 384;

bool pp_eligible(int NT, const GemmParams& p) {
    // eligibility: whole sub-step groups, aligned operand rows and vector epilogue, enough rows to fill the chip
    if (g_force_generic_gemm) return false;
    if (p.K % (128 / NT) != 0 || p.N < 256 || p.N % 4 != 0 || p.M < PP_MIN_ROWS) return false;
    if (p.lda % 8 || p.ldw % 8 || p.a_plane % 8 || p.w_plane % 8 || p.a_batch_stride % 8) return false;
    if (((uintptr_t)p.A & 15) || ((uintptr_t)p.W & 15)) return false;
    // DMA addressing: 32-bit byte offsets from the tile's first row; rows of a tile ascend in memory
    if (p.M > p.rows_per_batch && (p.rows_per_batch < 256 || p.a_batch_stride < (p.rows_per_batch - 1) * p.lda)) return false;
    {
        const int64_t a_span = (NT > 1 ? p.a_plane : 0) + (p.M > p.rows_per_batch ? p.a_batch_stride : 0) + 256 * p.lda + p.K;
        const int64_t w_span = (NT > 1 ? p.w_plane : 0) + 256 * p.ldw + p.K;
        if (a_span < 0 || w_span < 0 || a_span * 2 >= (int64_t)0xFFFFFF00 || w_span * 2 >= (int64_t)0xFFFFFF00) return false;
    }
    if (p.bias && ((uintptr_t)p.bias & 15)) return false;
    if (p.out_f32 && (p.ldo % 4 || ((uintptr_t)p.out_f32 & 15))) return false;
    if (p.residual && (p.ldr % 4 || ((uintptr_t)p.residual & 15))) return false;
    if (p.out_p && (p.ldp % 4 || p.out_plane % 4 || ((uintptr_t)p.out_p & 7))) return false;
    if (p.mode == 1) {
        const int D = p.H * p.dh;
        if (D % 64 || p.dh % 4 || p.qk_plane % 4 || p.N != 3 * D) return false;
        if (((uintptr_t)p.q & 7) || ((uintptr_t)p.k & 7) || ((uintptr_t)p.v & 7)) return false;
    } else if (p.row_len && p.rows_T <= 0) {
        return false;
    }
    return true;
}

int device_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
        cus -= cus % 8;  // the tile order assumes sequence numbers i and i + grid share an XCD
        if (cus < 8) cus = 8;
    }
    return cus;
}

// Number of K chunks for a product of `tiles` output tiles: 1 when no workspace was given or the tiles alone occupy more
// than `max_tiles` CUs; otherwise the largest divisor of K / granule that neither over-subscribes the chip, nor makes chunks
// shorter than `min_chunk`, nor overflows the workspace.  (Thresholds from tools/gemm_bench `small`: the fix-up launch and
// the partial slabs cost 10-15 us, so a split only pays when it removes more main-loop time than that.)
int choose_splits(const GemmParams& p, int tiles, int max_tiles, int granule, int min_chunk) {
    if (!p.splitk_ws || p.K % granule || p.K < 2 * min_chunk || tiles > max_tiles) return 1;
    int64_t smax = device_cus() / tiles;
    smax = std::min<int64_t>(smax, p.K / min_chunk);
    smax = std::min<int64_t>(smax, p.splitk_ws_elems / ((int64_t)p.M * p.N));
    smax = std::min<int64_t>(smax, 32);
    const int n_g = p.K / granule;
    int best = 1;
    for (int sp = 2; sp <= smax; ++sp)
        if (n_g % sp == 0) best = sp;
    return best;
}

// the kernel-side view of one K chunk: raw fp32 partials into the workspace, no epilogue features
GemmParams split_view(const GemmParams& p, int splits) {
    GemmParams q = p;
    q.K = p.K / splits;
    q.splits = splits;
    q.split_out = (int64_t)p.M * p.N;
    q.out_f32 = p.splitk_ws;
    q.ldo = p.N;
    q.out_p = nullptr;
    q.bias = nullptr;
    q.scale = 1.f;
    q.act = 0;
    q.residual = nullptr;
    q.row_len = nullptr;
    q.mode = 0;
    q.ln_gamma = q.ln_beta = nullptr;
    q.vec_ok = (p.N % 4 == 0) ? 1 : 0;
    return q;
}

template <typename T, int NT>
void launch_fixup(const GemmParams& p, int splits, hipStream_t stream) {
    dim3 grid((p.N + 63) / 64, (p.M + 63) / 64);
    hipLaunchKernelGGL((splitk_fixup_kernel<T, NT>), grid, dim3(256), 0, stream, p, (const float*)p.splitk_ws, splits,
                       (int64_t)p.M * p.N);
}

template <typename T, int NT, int MI>
void launch_pp_tiles(const GemmParams& p, int splits, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_pp_kernel<T, NT, MI>, hipFuncAttributeMaxDynamicSharedMemorySize, pp::LDS_BYTES);
        attr_set = true;
    }
    const int tiles = ((p.N + pp::BN - 1) / pp::BN) * ((p.M + MI * 32 - 1) / (MI * 32));
    const int cus = device_cus();
    const int units = tiles * splits;
    dim3 grid(units < cus ? units : cus, 1, 1);  // persistent: one 128-KiB-LDS workgroup per CU
    if (splits > 1) {
        hipLaunchKernelGGL((gemm_pp_kernel<T, NT, MI>), grid, dim3(512), pp::LDS_BYTES, stream, split_view(p, splits));
        launch_fixup<T, NT>(p, splits, stream);
    } else {
        hipLaunchKernelGGL((gemm_pp_kernel<T, NT, MI>), grid, dim3(512), pp::LDS_BYTES, stream, p);
    }
}

// Tile height (256 or 128 rows) and number of K chunks of a ping-pong product, by a cost model in units of one 32-deep MFMA
// segment pair per K element (~ 0.0166 us; fitted to tools/gemm_bench `small` on MI355X):
//   rounds of the persistent grid x (main loop of a work unit + its prologue / epilogue) + fix-up launch and slab traffic.
// Config-2-sized products (>= one full round of 256-row tiles) always come out as (256 rows, 1 chunk).
void pp_plan(int NT, const GemmParams& p, int* mi_out, int* splits_out) {
    const int cus = device_cus();
    const double loop = (double)p.K * (NT > 1 ? 3 : 1);  // main loop of a 256-row tile
    double best = 1e30;
    *mi_out = 8;
    *splits_out = 1;
    for (int mi = 8; mi >= 4; mi -= 4) {
        const int tiles = ((p.N + pp::BN - 1) / pp::BN) * ((p.M + mi * 32 - 1) / (mi * 32));
        const double per_unit = 480.0 + 60.0 * mi;  // prologue + epilogue of a work unit
        int64_t smax = 1;
        if (p.splitk_ws && p.K % 128 == 0) {
            smax = std::min<int64_t>(p.K / 256, p.splitk_ws_elems / ((int64_t)p.M * p.N));
            smax = std::min<int64_t>(smax, 32);
        }
        for (int sp = 1; sp <= smax; ++sp) {
            if ((p.K / 128) % sp) continue;
            const int64_t units = (int64_t)tiles * sp;
            double cost = (double)((units + cus - 1) / cus) * (loop * mi / 8.0 / sp + per_unit);
            if (sp > 1) cost += 700.0 + (double)sp * p.M * p.N * 8.0 / 4.0e6 / 0.0166;  // fix-up launch + slab write / read at 4 TB/s
            if (cost < best * 0.97) {  // prefer the earlier candidate (taller tile, fewer chunks) on near ties
                best = cost;
                *mi_out = mi;
                *splits_out = sp;
            }
        }
    }
}

template <typename T, int NT>
bool launch_gemm_pp(const GemmParams& p, hipStream_t stream) {
    if (!pp_eligible(NT, p)) return false;
    int mi, splits;
    pp_plan(NT, p, &mi, &splits);
    if (mi == 8) launch_pp_tiles<T, NT, 8>(p, splits, stream);
    else launch_pp_tiles<T, NT, 4>(p, splits, stream);
    return true;
}

bool dma_tile_eligible(int NT, const GemmParams& p) {
    static const bool off = getenv("AMX_NO_DMA_TILE") && atoi(getenv("AMX_NO_DMA_TILE")) != 0;  // developer A/B switch
    if (off || p.K % BK != 0) return false;
    if (p.lda % 8 || p.ldw % 8 || p.a_plane % 8 || p.w_plane % 8 || p.a_batch_stride % 8 || p.za % 8 || p.zw % 8) return false;
    if (((uintptr_t)p.A & 15) || ((uintptr_t)p.W & 15)) return false;
    // rows of a tile ascend in memory and stay inside 32-bit byte offsets from the tile's first row
    if (p.M > p.rows_per_batch && p.a_batch_stride < (p.rows_per_batch - 1) * p.lda) return false;
    const int64_t batches_per_tile = p.M > p.rows_per_batch ? (128 + p.rows_per_batch - 1) / p.rows_per_batch + 1 : 0;
    const int64_t a_span = (NT > 1 ? p.a_plane : 0) + batches_per_tile * p.a_batch_stride + 128 * p.lda + p.K;
    const int64_t w_span = (NT > 1 ? p.w_plane : 0) + 64 * p.ldw + p.K;
    if (a_span < 0 || w_span < 0 || a_span * 2 >= (int64_t)0xFFFFFF00 || w_span * 2 >= (int64_t)0xFFFFFF00) return false;
    return true;
}

// the LDS-DMA kernel holds one workgroup per CU (144 KiB ring): it serves the short products, big grids keep the
// register-staged kernel (several workgroups per CU).  Returns the tile shape: 0 = not used, 1 = 128 x 64, 2 = 64 x 32.
int dma_tile_shape(int NT, const GemmParams& p, int zdim) {
    if (!dma_tile_eligible(NT, p)) return 0;
    const int64_t big = (int64_t)((p.N + 63) / 64) * ((p.M + 127) / 128) * zdim;
    if (big > device_cus()) return 0;
    return big * 2 <= device_cus() ? 2 : 1;
}

template <typename T, int NT, int BM, int BN, int WM, int WN, int STAGES>
void launch_gemm_dma(const GemmParams& q, int zdim, hipStream_t stream) {
    constexpr int lds = STAGES * NT * (BM + BN) * 128;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gemm_dma_kernel<T, NT, BM, BN, WM, WN, STAGES>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    dim3 grid((q.N + BN - 1) / BN, (q.M + BM - 1) / BM, zdim);
    hipLaunchKernelGGL((gemm_dma_kernel<T, NT, BM, BN, WM, WN, STAGES>), grid, dim3(256), lds, stream, q);
}

template <typename T, int NT>
void launch_gemm_dma_shape(int shape, const GemmParams& q, int zdim, hipStream_t stream) {
    if (shape == 2) launch_gemm_dma<T, NT, 64, 32, 2, 2, 6>(q, zdim, stream);
    else launch_gemm_dma<T, NT, 128, 64, 4, 1, 3>(q, zdim, stream);
}

template <typename T, int NT>
void launch_gemm_t(const GemmParams& p, hipStream_t stream) {
    // below ~768 rows a product that fits one round of LDS-DMA tiles is faster there than on 128-row ping-pong tiles
    // (tools/geometry_sweep.py: 1 x 10 s 5.5 -> 4.0 ms); everything else the ping-pong kernel accepts goes to it
    const int shape = dma_tile_shape(NT, p, 1);
    if (!(shape && p.M < 768) && launch_gemm_pp<T, NT>(p, stream)) return;
    if (shape) {
        // the ring hides the memory latency, so the K loop is only cut where it is long (K = 4096) and the grid small
        const int bm = shape == 2 ? 64 : 128, bn = shape == 2 ? 32 : 64;
        const int tiles = ((p.N + bn - 1) / bn) * ((p.M + bm - 1) / bm);
        const int splits = p.K >= 4096 ? choose_splits(p, tiles, device_cus() / 2, BK, 16 * BK) : 1;
        GemmParams q = p;
        if (splits > 1) {
            q = split_view(p, splits);
            q.za = q.zw = q.K;
            q.zout = q.split_out;
            q.zbias = q.zoutp = 0;
        }
        launch_gemm_dma_shape<T, NT>(shape, q, splits, stream);
        if (splits > 1) launch_fixup<T, NT>(p, splits, stream);
        return;
    }
    // narrow outputs (grouped pos-conv, small classifier heads) use the 128x64 tile
    const bool narrow = p.N <= 64;
    const int BM = 128, BN = narrow ? 64 : 128;
    const int tiles = ((p.N + BN - 1) / BN) * ((p.M + BM - 1) / BM);
    // short products whose shape the LDS-DMA kernel rejects: the K loop of a register-staged tile is latency-bound, so cut it
    const int splits = choose_splits(p, tiles, device_cus() / 4, BK, 2 * BK);
    GemmParams q = p;
    if (splits > 1) {
        q = split_view(p, splits);
        q.za = q.zw = q.K;              // grid.z = K chunk: operand pointers advance by the chunk length,
        q.zout = q.split_out;           // the fp32 output by one slab
        q.zbias = q.zoutp = 0;
    }
    dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM, splits);
    const size_t lds = (size_t)NT * (BM + BN) * 128;
    if (narrow) hipLaunchKernelGGL((gemm_kernel<T, NT, 128, 64, 4, 1>), grid, dim3(256), lds, stream, q);
    else hipLaunchKernelGGL((gemm_kernel<T, NT, 128, 128, 2, 2>), grid, dim3(256), lds, stream, q);
    if (splits > 1) launch_fixup<T, NT>(p, splits, stream);
}

template <typename T, int NT>
void launch_gemm_z(const GemmParams& p, int zdim, hipStream_t stream) {
    const int shape = dma_tile_shape(NT, p, zdim);
    if (shape) {
        launch_gemm_dma_shape<T, NT>(shape, p, zdim, stream);
        return;
    }
    constexpr int BM = 128, BN = 64;
    dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM, zdim);
    size_t lds = (size_t)NT * (BM + BN) * 128;
    hipLaunchKernelGGL((gemm_kernel<T, NT, BM, BN, 4, 1>), grid, dim3(256), lds, stream, p);
}

}  // namespace

static GemmParams with_vec_flag(const GemmParams& in) {
    GemmParams p = in;
    p.splits = 1;
    p.split_out = 0;
    p.vec_ok = (p.N % 4 == 0) && (!p.out_f32 || (p.ldo % 4 == 0 && p.zout % 4 == 0 && ((uintptr_t)p.out_f32 & 15) == 0)) &&
               (!p.out_p || (p.ldp % 4 == 0 && p.zoutp % 4 == 0 && p.out_plane % 4 == 0 && ((uintptr_t)p.out_p & 7) == 0));
    return p;
}

bool gemm_fuses_ln(int prec, const GemmParams& p_in) { return ln_eligible(prec_planes(prec), with_vec_flag(p_in)); }

bool gemm_uses_pp(int prec, const GemmParams& p_in) {
    const GemmParams p = with_vec_flag(p_in);
    const int NT = prec_planes(prec);
    return pp_eligible(NT, p) && !(p.M < 768 && dma_tile_shape(NT, p, 1));  // the routing of launch_gemm_t
}

void launch_gemm(int prec, const GemmParams& p_in, hipStream_t stream) {
    const GemmParams p = with_vec_flag(p_in);
    if (p.ln_gamma) {
        // fused LayerNorm + GELU: only the row-complete kernel implements it (callers check gemm_fuses_ln first)
        switch (prec) {
            case PREC_BF16: launch_gemm_ln<bf16, 1>(p, stream); break;
            case PREC_F16: launch_gemm_ln<f16, 1>(p, stream); break;
            case PREC_BF16X3: launch_gemm_ln<bf16, 2>(p, stream); break;
            default: launch_gemm_ln<f16, 2>(p, stream); break;
        }
        return;
    }
    switch (prec) {
        case PREC_BF16: launch_gemm_t<bf16, 1>(p, stream); break;
        case PREC_F16: launch_gemm_t<f16, 1>(p, stream); break;
        case PREC_BF16X3: launch_gemm_t<bf16, 2>(p, stream); break;
        default: launch_gemm_t<f16, 2>(p, stream); break;
    }
}

void launch_gemm_grouped(int prec, const GemmParams& p_in, int groups, hipStream_t stream) {
    const GemmParams p = with_vec_flag(p_in);
    switch (prec) {
        case PREC_BF16: launch_gemm_z<bf16, 1>(p, groups, stream); break;
        case PREC_F16: launch_gemm_z<f16, 1>(p, groups, stream); break;
        case PREC_BF16X3: launch_gemm_z<bf16, 2>(p, groups, stream); break;
        default: launch_gemm_z<f16, 2>(p, groups, stream); break;
    }
}

}  // namespace amx
