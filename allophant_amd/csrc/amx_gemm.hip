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

namespace amx {

namespace {

constexpr int BK = 64;

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

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
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
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
    const float* bias = p.bias ? p.bias + (int64_t)z * p.zbias : nullptr;
    const float* residual = p.residual ? p.residual + (int64_t)z * p.zout : nullptr;
    float* out_f32 = p.out_f32 ? p.out_f32 + (int64_t)z * p.zout : nullptr;
    T* out_p = p.out_p ? (T*)p.out_p + (int64_t)z * p.zoutp : nullptr;
    const int D = p.H * p.dh;

#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int m = m0 + wm * TM + mi * 16 + (lane & 15);
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
            const int nb = n0 + wn * TN + ni * 16 + 4 * (lane >> 4);
            if (nb >= p.N) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int n = nb + r;
                float x = acc[ni][mi][r] * p.scale;
                if (n < p.N) {
                    if (bias) x += bias[n];
                    if (p.act == 1) x = gelu_erf(x);
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
                if (which < 2) {
                    T* dst = (T*)(which == 0 ? p.q : p.k) + (((int64_t)b * p.H + hh) * p.Tp + t) * p.dh + d;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        dst[r] = hi[r];
                        if (NT > 1) dst[p.qk_plane + r] = lo[r];
                    }
                } else {
                    T* dst = (T*)p.vt + (((int64_t)b * p.H + hh) * p.dh + d) * p.Tp + vt_perm(t);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        dst[(int64_t)r * p.Tp] = hi[r];
                        if (NT > 1) dst[p.vt_plane + (int64_t)r * p.Tp] = lo[r];
                    }
                }
                continue;
            }
            if (out_f32) {
                float* dst = out_f32 + (int64_t)m * p.ldo + nb;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (nb + r < p.N) dst[r] = v[r];
            }
            if (out_p) {
                T* dst = out_p + (int64_t)m * p.ldp + nb;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (nb + r < p.N) {
                        T hi, lo;
                        split16<T, NT>(v[r], hi, lo);
                        dst[r] = hi;
                        if (NT > 1) dst[p.out_plane + r] = lo;
                    }
                }
            }
        }
    }
}

template <typename T, int NT>
void launch_gemm_t(const GemmParams& p, hipStream_t stream) {
    int zdim = 1;
    // narrow outputs (grouped pos-conv, small classifier heads) use the 128x64 tile
    if (p.N <= 64) {
        constexpr int BM = 128, BN = 64;
        dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM, zdim);
        size_t lds = (size_t)NT * (BM + BN) * 128;
        hipLaunchKernelGGL((gemm_kernel<T, NT, BM, BN, 4, 1>), grid, dim3(256), lds, stream, p);
    } else {
        constexpr int BM = 128, BN = 128;
        dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM, zdim);
        size_t lds = (size_t)NT * (BM + BN) * 128;
        hipLaunchKernelGGL((gemm_kernel<T, NT, BM, BN, 2, 2>), grid, dim3(256), lds, stream, p);
    }
}

template <typename T, int NT>
void launch_gemm_z(const GemmParams& p, int zdim, hipStream_t stream) {
    constexpr int BM = 128, BN = 64;
    dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM, zdim);
    size_t lds = (size_t)NT * (BM + BN) * 128;
    hipLaunchKernelGGL((gemm_kernel<T, NT, BM, BN, 4, 1>), grid, dim3(256), lds, stream, p);
}

}  // namespace

void launch_gemm(int prec, const GemmParams& p, hipStream_t stream) {
    switch (prec) {
        case PREC_BF16: launch_gemm_t<bf16, 1>(p, stream); break;
        case PREC_F16: launch_gemm_t<f16, 1>(p, stream); break;
        case PREC_BF16X3: launch_gemm_t<bf16, 2>(p, stream); break;
        default: launch_gemm_t<f16, 2>(p, stream); break;
    }
}

void launch_gemm_grouped(int prec, const GemmParams& p, int groups, hipStream_t stream) {
    switch (prec) {
        case PREC_BF16: launch_gemm_z<bf16, 1>(p, groups, stream); break;
        case PREC_F16: launch_gemm_z<f16, 1>(p, groups, stream); break;
        case PREC_BF16X3: launch_gemm_z<bf16, 2>(p, groups, stream); break;
        default: launch_gemm_z<f16, 2>(p, groups, stream); break;
    }
}

}  // namespace amx
