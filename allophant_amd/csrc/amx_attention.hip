// Flash-style multi-head self-attention with key-padding mask for gfx950 (head_dim 64), never materialising T x T.
//
// Per workgroup: one (utterance, head) and 128 queries (4 waves x 32).  K/V tiles of 64 keys are staged through LDS
// (register-prefetched), scores are computed transposed, S^T = K.Q^T with v_mfma_f32_32x32x16, so that a lane owns one
// query column: the online-softmax row statistics are lane-local (one cross-half exchange), and the exponentiated
// accumulator registers feed the second product O^T = V^T.P directly as its B operand (accumulator-as-operand order,
// cdna_hip_programming.md section 3) -- no LDS round trip for P.  V is consumed from a transposed image [dh][Tp] written
// by the QKV projection epilogue with keys permuted inside groups of 16 (vt_perm) so each V fragment is one 16-byte read.
// Masked keys (t' >= frame_len[n]) get -inf before the softmax; key tiles past the utterance end are skipped (their
// probabilities are exactly 0 in the reference too: finfo.min bias underflows exp to 0).
#include "amx_common.h"

namespace amx {

namespace {

constexpr int DH = 64;
constexpr int KT = 64;   // keys per tile
constexpr int QB = 128;  // queries per workgroup

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <typename T, int NT>
__global__ __launch_bounds__(256) void attn_kernel(const AttnParams p) {
    typedef typename Vec8<T>::type V8;
    typedef typename Vec4<T>::type V4;
    __shared__ __attribute__((aligned(16))) unsigned char sK[NT][KT * 128];
    __shared__ __attribute__((aligned(16))) unsigned char sV[NT][DH * 128];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hh = lane >> 5, lq = lane & 31;
    const int nh = blockIdx.y;
    const int n = nh / p.H, h = nh % p.H;
    const int q_base = blockIdx.x * QB + wave * 32;
    const int query = q_base + lq;
    int klen = p.frame_len[n];
    klen = klen < 1 ? 1 : (klen > p.T ? p.T : klen);
    const int nkt = (klen + KT - 1) / KT;

    const T* Qb = (const T*)p.q + (int64_t)nh * p.Tp * DH;
    const T* Kb = (const T*)p.k + (int64_t)nh * p.Tp * DH;
    const T* Vb = (const T*)p.vt + (int64_t)nh * DH * p.Tp;

    // Q fragments (B operand): lane (query, hh) holds Q[query][16ks + 8hh + j]
    V8 qf[NT][4];
    {
        int qr = query < p.Tp ? query : p.Tp - 1;
#pragma unroll
        for (int pl = 0; pl < NT; ++pl)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                qf[pl][ks] = *(const V8*)(Qb + (int64_t)pl * p.qk_plane + (int64_t)qr * DH + ks * 16 + 8 * hh);
    }

    // staging: K tile = 64 rows x 128 B contiguous; Vt tile = 64 rows (d) x 128 B at row stride Tp
    const int ld_row = tid >> 3, ld_c = tid & 7;
    uint4 rk[NT][2], rv[NT][2];
    auto load_tile = [&](int kt) {
        const int kb = kt * KT;
#pragma unroll
        for (int pl = 0; pl < NT; ++pl)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int row = ld_row + 32 * i;
                rk[pl][i] = *(const uint4*)(Kb + (int64_t)pl * p.qk_plane + (int64_t)(kb + row) * DH + ld_c * 8);
                rv[pl][i] = *(const uint4*)(Vb + (int64_t)pl * p.vt_plane + (int64_t)row * p.Tp + kb + ld_c * 8);
            }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int pl = 0; pl < NT; ++pl)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int row = ld_row + 32 * i;
                *(uint4*)(&sK[pl][lds_off(row, ld_c)]) = rk[pl][i];
                *(uint4*)(&sV[pl][lds_off(row, ld_c)]) = rv[pl][i];
            }
    };

    f32x16 O[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { O[0][r] = 0.f; O[1][r] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;
    const float LOG2E = 1.44269504088896340736f;

    load_tile(0);
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();
        store_tile();
        __syncthreads();
        if (kt + 1 < nkt) load_tile(kt + 1);

        // ---- S^T = K . Q^T : X[c][r] = score(key = kb + 32c + (r&3) + 8(r>>2) + 4hh, query) ----
        f32x16 X[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int r = 0; r < 16; ++r) X[c][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                V8 kf = *(const V8*)(&sK[0][lds_off(c * 32 + lq, 2 * ks + hh)]);
                if (NT > 1) {
                    V8 kl = *(const V8*)(&sK[NT - 1][lds_off(c * 32 + lq, 2 * ks + hh)]);
                    X[c] = mfma32(kl, qf[0][ks], X[c]);
                    X[c] = mfma32(kf, qf[NT - 1][ks], X[c]);
                }
                X[c] = mfma32(kf, qf[0][ks], X[c]);
            }
        }
        const int kb = kt * KT;
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int key = kb + 32 * c + (r & 3) + 8 * (r >> 2) + 4 * hh;
                float s = key < klen ? X[c][r] : -INFINITY;
                X[c][r] = s;
                mx = fmaxf(mx, s);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float e = __builtin_amdgcn_exp2f((X[c][r] - m_new) * LOG2E);
                X[c][r] = e;
                psum += e;
            }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int r = 0; r < 16; ++r) { O[0][r] *= alpha; O[1][r] *= alpha; }

        // ---- O^T += V^T . P : P's accumulator registers are the B operand ----
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                V8 ph, pl_;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    T hi, lo = (T)0.f;
                    split16<T, NT>(X[c][8 * s2 + j], hi, lo);
                    ph[j] = hi;
                    if (NT > 1) pl_[j] = lo;
                }
                const int chunk = 2 * (2 * c + s2) + hh;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    V8 vf = *(const V8*)(&sV[0][lds_off(dt * 32 + lq, chunk)]);
                    if (NT > 1) {
                        V8 vl = *(const V8*)(&sV[NT - 1][lds_off(dt * 32 + lq, chunk)]);
                        O[dt] = mfma32(vl, ph, O[dt]);
                        O[dt] = mfma32(vf, pl_, O[dt]);
                    }
                    O[dt] = mfma32(vf, ph, O[dt]);
                }
            }
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    if (query < p.T) {
        T* dst = (T*)p.out + ((int64_t)n * p.T + query) * (p.H * DH) + h * DH;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                V4 hv, lv;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    T hi, lo = (T)0.f;
                    split16<T, NT>(O[dt][4 * g + j] * inv, hi, lo);
                    hv[j] = hi;
                    lv[j] = lo;
                }
                int d0 = dt * 32 + 8 * g + 4 * hh;
                *(V4*)(dst + d0) = hv;
                if (NT > 1) *(V4*)(dst + p.out_plane + d0) = lv;
            }
    }
}

}  // namespace

void launch_attention(int prec, const AttnParams& p, hipStream_t stream) {
    dim3 grid((p.T + QB - 1) / QB, p.N * p.H);
    switch (prec) {
        case PREC_BF16: hipLaunchKernelGGL((attn_kernel<bf16, 1>), grid, dim3(256), 0, stream, p); break;
        case PREC_F16: hipLaunchKernelGGL((attn_kernel<f16, 1>), grid, dim3(256), 0, stream, p); break;
        case PREC_BF16X3: hipLaunchKernelGGL((attn_kernel<bf16, 2>), grid, dim3(256), 0, stream, p); break;
        default: hipLaunchKernelGGL((attn_kernel<f16, 2>), grid, dim3(256), 0, stream, p); break;
    }
}

}  // namespace amx
