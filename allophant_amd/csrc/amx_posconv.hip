// Grouped positional convolution of the wav2vec 2.0 encoder prologue for gfx950 (reference: HF Wav2Vec2PositionalConvEmbedding
// as called from allophant/network/acoustic_model.py:837-853):
//     h[n, t, g*64 + co] += gelu( b[g*64 + co] + sum_{tap, ci} W[g][co][tap][ci] * x[n, t + tap - pad, g*64 + ci] )
// with 64 channels per group and up to 128 taps.  As an implicit GEMM (M = frames, N = 64, K = taps * 64) the A operand
// of output frame t is the window of frames t .. t + taps - 1: consecutive frames share all but one row of it, and the
// generic tile kernel re-reads every frame once per tap from L2 (128 x).  Here a workgroup keeps the WINDOW of its 256
// output frames resident in LDS -- 256 + taps - 1 rows of 128 bytes per plane, DMA-ed once -- and streams only the weights:
// per tap one 64 x 64 slice (8 KiB per plane) through a 4-stage LDS-DMA ring, two taps per barrier interval.  The A fragments of tap j are the fragment
// reads of tap 0 shifted down by j rows.
//   * 8 waves, wave w owns output frames 32w .. 32w+31 (2 x 4 accumulator fragments of v_mfma_f32_16x16x32; W is the first
//     operand, so a lane ends up with 4 consecutive output channels of one frame);
//   * bank swizzle: 16-byte chunk ^= (row >> 1) & 7 on the DMA source address and on the fragment reads; 16 consecutive
//     rows give 16 distinct (row parity, chunk) pairs whatever the first row, so the shifted reads stay conflict-free;
//   * ring protocol per pair of taps (j, j+1): this wave's pieces of both have landed (vmcnt(0): nothing younger is in
//     flight) and its LDS reads of the previous pair have returned (lgkmcnt(0)) -> s_barrier -> DMA of taps j+2, j+3 into the
//     two stages the previous pair was read from -> multiply taps j and j+1.  (A read left in flight across the barrier
//     could see the refill: the race found in the attention ring.)  One barrier per pair: with one per tap the waves sat
//     parked at the barrier for 48 % of their cycles (profiles/r02_wave_state_pmc.txt).
// Requires hidden / groups == 64 and an even number of taps <= 128; other shapes stay on the implicit-GEMM path.
#include "amx_common.h"

namespace amx {

namespace {

constexpr int PC_ROWS = 256;                        // output frames per workgroup
constexpr int PC_CG = 64;                           // channels per group
constexpr int PC_MAX_TAPS = 128;
constexpr int PC_WIN = PC_ROWS + PC_MAX_TAPS;       // window rows reserved per plane (256 + taps - 1 used)
constexpr int PC_STAGES = 4;
constexpr int PC_WTAP = PC_CG * 128;                // bytes of one tap's weights of one plane: 64 rows x 128 B

template <typename T, int NT>
__global__ __launch_bounds__(512, 2) void posconv_window_kernel(const T* __restrict__ image, int64_t image_plane,
                                                                const T* __restrict__ weights, int64_t w_plane, int64_t ldw,
                                                                const float* __restrict__ bias, float* __restrict__ h,
                                                                int N, int Tn, int Tpad, int D, int taps, float scale,
                                                                const int* __restrict__ row_off, const int* __restrict__ frame_len) {
    typedef typename Vec8<T>::type V8;
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* s_win = smem;                                  // [NT][PC_WIN][128 B]
    unsigned char* s_w = smem + NT * PC_WIN * 128;                // [PC_STAGES][NT][64][128 B]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_per_utt = (Tn + PC_ROWS - 1) / PC_ROWS;
    const int n = blockIdx.x / tiles_per_utt, t0 = (blockIdx.x % tiles_per_utt) * PC_ROWS;
    const int g = blockIdx.y, G = gridDim.y;
    // packed rows (ragged batches): utterance n owns rows row_off[n] .. + frame_len[n] of h; frame blocks beyond its end do
    // not exist
    const int t_end = row_off ? frame_len[n] : Tn;
    if (t0 >= t_end) return;
    const int64_t h_row0 = row_off ? (int64_t)row_off[n] : (int64_t)n * Tn;

    // ---- window DMA: rows t0 .. t0 + 255 + taps - 1 of image [g][n][Tpad][64], one contiguous block per plane ----
    const int win_rows = PC_ROWS + taps - 1;
    const int win_pieces = (win_rows + 7) / 8;  // pieces of 8 rows x 128 B
    const T* img = image + (((int64_t)g * N + n) * Tpad) * PC_CG;
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)img, 0, -1, 0x00020000);
    const uint32_t a_plane_b = (uint32_t)(image_plane * 2);
    for (int pc = wave; pc < win_pieces; pc += 8) {
        const int row = pc * 8 + (lane >> 3);
        int src_row = t0 + row;
        src_row = src_row < Tpad ? src_row : Tpad - 1;  // rows past the utterance's image feed frames t >= Tn only
        const int lc = (lane & 7) ^ ((row >> 1) & 7);
        const uint32_t off = (uint32_t)(src_row * 128 + lc * 16);
#pragma unroll
        for (int pl = 0; pl < NT; ++pl)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (lds_ptr_t)(s_win + (pl * PC_WIN + pc * 8) * 128), 16, off,
                                                     pl * a_plane_b, 0, 0);
    }
    // ---- weight ring: this wave moves piece `wave` (8 output channels x 128 B) of every tap and plane ----
    const T* wg = weights + (int64_t)g * PC_CG * ldw;
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)wg, 0, -1, 0x00020000);
    const uint32_t w_plane_b = (uint32_t)(w_plane * 2);
    uint32_t w_off;
    {
        const int row = wave * 8 + (lane >> 3);
        const int lc = (lane & 7) ^ ((row >> 1) & 7);
        w_off = (uint32_t)((int64_t)row * ldw * 2 + lc * 16);
    }
    auto stage_w = [&](int tap) {
        unsigned char* dst = s_w + (tap % PC_STAGES) * (NT * PC_WTAP) + wave * 1024;
#pragma unroll
        for (int pl = 0; pl < NT; ++pl)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_ptr_t)(dst + pl * PC_WTAP), 16, w_off,
                                                     pl * w_plane_b + (uint32_t)tap * 128, 0, 0);
    };
    stage_w(0);
    stage_w(1);

    // ---- fragment read offsets ----
    // A fragment (mi, kk) of tap j: window row 32 wave + 16 mi + (lane & 15) + j, logical chunk 4 kk + (lane >> 4)
    // W fragment (ni, kk): ring row 16 ni + (lane & 15), same chunk
    const int frag_chunk = lane >> 4;
    const int a_row0 = wave * 32 + (lane & 15);
    int w_rd[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const int row = ni * 16 + (lane & 15);
        w_rd[ni] = row * 128 + ((frag_chunk ^ ((row >> 1) & 7)) << 4);  // chunk of kk = 0; kk = 1: ^ 64
    }

    f32x4 acc[4][2];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) acc[ni][mi] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto multiply_tap = [&](int j) {
        const unsigned char* sw = s_w + (j % PC_STAGES) * (NT * PC_WTAP);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            V8 af[NT][2], wf[NT][4];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int row = a_row0 + mi * 16 + j;
                const int off = row * 128 + ((((kk << 2) | frag_chunk) ^ ((row >> 1) & 7)) << 4);
#pragma unroll
                for (int pl = 0; pl < NT; ++pl) af[pl][mi] = *(const V8*)(s_win + pl * PC_WIN * 128 + off);
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int pl = 0; pl < NT; ++pl) wf[pl][ni] = *(const V8*)(sw + pl * PC_WTAP + (w_rd[ni] ^ (kk << 6)));
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    if (NT > 1) {
                        acc[ni][mi] = mfma16(wf[NT - 1][ni], af[0][mi], acc[ni][mi]);  // lo(W) * hi(x)
                        acc[ni][mi] = mfma16(wf[0][ni], af[NT - 1][mi], acc[ni][mi]);  // hi(W) * lo(x)
                    }
                    acc[ni][mi] = mfma16(wf[0][ni], af[0][mi], acc[ni][mi]);
                }
        }
    };
    for (int j = 0; j < taps; j += 2) {  // taps is even
        // taps j and j + 1 (and, for j = 0, the window) have landed: nothing younger has been issued yet
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");  // no LDS read of this pair is scheduled above the barrier
        if (j + 2 < taps) { stage_w(j + 2); stage_w(j + 3); }
        multiply_tap(j);
        multiply_tap(j + 1);
    }

    // ---- epilogue: h += gelu(acc * scale + bias) (scale: the power of two the packed weights were divided by); acc[ni][mi][r] is frame 16 mi + (lane & 15), channel 16 ni + 4 (lane >> 4) + r ----
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int t = t0 + wave * 32 + mi * 16 + (lane & 15);
        if (t >= t_end) continue;
        float* row = h + (h_row0 + t) * D + g * PC_CG;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int c = ni * 16 + 4 * (lane >> 4);
            const float4 b4 = *(const float4*)(bias + g * PC_CG + c);
            float4 r4 = *(const float4*)(row + c);
            const f32x2 g01 = gelu_fast2(f32x2{fmaf(acc[ni][mi][0], scale, b4.x), fmaf(acc[ni][mi][1], scale, b4.y)});
            const f32x2 g23 = gelu_fast2(f32x2{fmaf(acc[ni][mi][2], scale, b4.z), fmaf(acc[ni][mi][3], scale, b4.w)});
            r4.x += g01[0];
            r4.y += g01[1];
            r4.z += g23[0];
            r4.w += g23[1];
            *(float4*)(row + c) = r4;
        }
    }
    (void)G;
}

template <typename T, int NT>
void launch_posconv_t(const void* image, int64_t image_plane, const void* weights, int64_t w_plane, int64_t ldw,
                      const float* bias, float scale, float* h, int N, int Tn, int Tpad, int D, int G, int taps, const int* row_off,
                      const int* frame_len, hipStream_t s) {
    constexpr int lds = NT * PC_WIN * 128 + PC_STAGES * NT * PC_WTAP;
    static OncePerDevice attr;
    if (attr.first())
        (void)hipFuncSetAttribute((const void*)posconv_window_kernel<T, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int tiles_per_utt = (Tn + PC_ROWS - 1) / PC_ROWS;
    dim3 grid((unsigned)(N * tiles_per_utt), (unsigned)G);
    hipLaunchKernelGGL((posconv_window_kernel<T, NT>), grid, dim3(512), lds, s, (const T*)image, image_plane, (const T*)weights,
                       w_plane, ldw, bias, h, N, Tn, Tpad, D, taps, scale, row_off, frame_len);
}

}  // namespace

bool posconv_window_eligible(int D, int G, int taps, int N, int Tn, int Tpad, int64_t image_plane) {
    if (G < 1 || D % G || D / G != PC_CG || taps < 2 || taps % 2 || taps > PC_MAX_TAPS || Tpad < Tn + taps - 1) return false;
    // 32-bit byte offsets inside one (group, utterance) image and one group's weights; the plane offset rides in soffset
    if ((int64_t)Tpad * 128 >= (int64_t)0x7FFFFFFF || image_plane * 2 >= (int64_t)0xFFFFFF00) return false;
    if ((int64_t)PC_CG * taps * PC_CG * 2 * PC_CG >= (int64_t)0x7FFFFFFF) return false;
    if (N < 1 || Tn < 1) return false;
    // one workgroup per 256 frames of an utterance and group: with too few of them (4 x 10 s: 128 on 256 CUs) the implicit
    // GEMM on smaller tiles fills the chip better (measured: 0.15 ms against 0.18 ms; 1 x 60 s with 192 workgroups: 0.13
    // against 0.21 ms the other way)
    static int cus_of[MAX_DEVICES] = {};
    int& cus = cus_of[current_device()];
    if (!cus) {
        hipDeviceProp_t prop;
        cus = hipGetDeviceProperties(&prop, current_device()) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int64_t workgroups = (int64_t)N * ((Tn + PC_ROWS - 1) / PC_ROWS) * G;
    return workgroups * 8 >= (int64_t)cus * 5;
}

void launch_posconv_window(int prec, const void* image, int64_t image_plane, const void* weights, int64_t w_plane, int64_t ldw,
                           const float* bias, float scale, float* h, int N, int Tn, int Tpad, int D, int G, int taps, const int* row_off,
                           const int* frame_len, hipStream_t s) {
    switch (prec) {
        case PREC_BF16: launch_posconv_t<bf16, 1>(image, image_plane, weights, w_plane, ldw, bias, scale, h, N, Tn, Tpad, D, G, taps, row_off, frame_len, s); break;
        case PREC_F16: launch_posconv_t<f16, 1>(image, image_plane, weights, w_plane, ldw, bias, scale, h, N, Tn, Tpad, D, G, taps, row_off, frame_len, s); break;
        case PREC_BF16X3: launch_posconv_t<bf16, 2>(image, image_plane, weights, w_plane, ldw, bias, scale, h, N, Tn, Tpad, D, G, taps, row_off, frame_len, s); break;
        default: launch_posconv_t<f16, 2>(image, image_plane, weights, w_plane, ldw, bias, scale, h, N, Tn, Tpad, D, G, taps, row_off, frame_len, s); break;
    }
}

}  // namespace amx
