// Internal declarations shared by the HIP translation units of liballophant_amx (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#ifdef AMX_DEVELOPER
#include <cstdlib>
#endif

namespace amx {

// Developer A/B switches.  The PRODUCT library reads no environment variable on the compute path (the one variable of the
// library is AMX_RCCL_LIBRARY in amx_dist.hip, which names a file, not a behaviour): these helpers are constants there and
// the switch names do not even reach the binary.  `make DEVELOPER=1` (-DAMX_DEVELOPER; tools/ab_build.sh, the A/B scripts under
// tools/) builds the library in which they read the environment.
#ifdef AMX_DEVELOPER
inline bool dev_switch(const char* name) { const char* v = getenv(name); return v && atoi(v) != 0; }
inline int dev_int(const char* name, int otherwise) { const char* v = getenv(name); return v ? atoi(v) : otherwise; }
#else
constexpr bool dev_switch(const char*) { return false; }
constexpr int dev_int(const char*, int otherwise) { return otherwise; }
#endif

typedef _Float16 f16;
typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// Precision modes (mirrors AMX_PREC_* in include/allophant_amx.h).
//   *_X3: every GEMM-shaped product runs as hi*hi + lo*hi + hi*lo on two 16-bit planes per operand
//         (x = hi + lo, hi = rn16(x), lo = rn16(x - hi)), fp32 accumulate: ~2^-21 (f16) / 2^-16 (bf16) relative.
enum Precision { PREC_BF16 = 0, PREC_F16 = 1, PREC_BF16X3 = 2, PREC_F16X3 = 3 };
inline int prec_planes(int p) { return p >= 2 ? 2 : 1; }

// Interleaved planes (the layout of every GEMM operand in the two-plane modes, except the positional-convolution image):
// the hi and lo values of 32 consecutive K elements share one 128-byte line, [hi x 32 | lo x 32], so that the operand DMA of
// the ping-pong GEMM fetches whole lines (tools/dma_row_probe.hip: 128-byte row segments move 1.4x faster through L2 -> LDS
// than the 64-byte segments of separate planes).  An operand is described as before by (pointer, plane distance, row stride)
// in LOGICAL elements; plane == PLANE_IL says that it is interleaved: logical element offset o (row * ld + column, ld % 32
// == 0) then lives at pidx(o) and its lo value PLANE_IL elements behind -- "lo = hi + plane" holds in both layouts.
constexpr int64_t PLANE_IL = 32;
#ifdef __HIPCC__
// LDS-DMA (buffer_load_dwordx4 ... lds: 16 bytes per lane, 1 KiB per wave instruction) written as inline asm.  With the builtin
// (__builtin_amdgcn_raw_ptr_buffer_load_lds) the compiler tracks the transfer as a store to LDS at an address it cannot analyse and
// puts s_waitcnt vmcnt(0) in front of the next LDS read with a memory operand -- i.e. in front of the first fragment read behind
// every prefetch (which then waits for the tile just REQUESTED: the ring's run-ahead is gone) and in front of every LDS read of an
// epilogue that alternates LDS reads and global stores (each store is waited for before the next read).  Kernels that read LDS
// through plain C++ loads / ds_read builtins use this form; the hand-placed counted waits (s_waitcnt vmcnt(n) + barrier) are what
// orders the transfers against the reads.  (Kernels whose LDS reads are inline asm too never had the problem.)
typedef unsigned dma_rsrc_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ dma_rsrc_t dma_rsrc(const void* base) {  // raw buffer, no bounds (num_records = -1), like make_buffer_rsrc(base, 0, -1, 0x20000)
    const uint64_t a = (uint64_t)base;
    dma_rsrc_t r = {(unsigned)a, (unsigned)(a >> 32) & 0xffffu, 0xffffffffu, 0x00020000u};
    return r;
}
__device__ __forceinline__ void dma16(dma_rsrc_t rsrc, const void* lds_dst, uint32_t voff, uint32_t soff) {
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
    const uint32_t lds_addr = (uint32_t)(uintptr_t)(lds_ptr_t)lds_dst;
    // (m0 carries the LDS address of the transfer; naming it as clobbered draws -Winline-asm "reserved register": the compiler keeps
    // nothing live in m0 across statements on gfx9 -- it loads m0 immediately in front of each use)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
#pragma clang diagnostic pop
}
#endif

#ifdef __HIPCC__
// s_waitcnt lgkmcnt(0) -- every LDS read of the wave has returned -- through the BUILTIN (vmcnt / expcnt fields all-ones: no wait on
// them), i.e. as an instruction the compiler's wait-count pass sees.  An inline-asm wait is opaque to that pass: behind it the pass
// still believes the fragment reads of a K slice to be in flight and, depending on nothing one can control from the source, either
// adds ONE lgkmcnt(0) behind the barrier or re-waits for every fragment inside the MFMA segment -- 12 s_waitcnt between the 96
// MFMAs of a slice, ~ 3.5 % of every product (round 6: the conv kernel had carried them since round 3, the ping-pong kernel picked
// them up as soon as its epilogues changed).  The empty asm keeps the "memory" clobber the asm form had.
__device__ __forceinline__ void lgkm_wait0() {
    __builtin_amdgcn_s_waitcnt(0xC07F);
    asm volatile("" ::: "memory");
}
#endif

__host__ __device__ __forceinline__ int64_t pidx(int64_t o, bool il) { return il ? (((o >> 5) << 6) | (o & 31)) : o; }
template <int NT>
__host__ __device__ __forceinline__ bool plane_is_il(int64_t plane) { return NT > 1 && plane == PLANE_IL; }

template <typename T> struct Vec8;
template <> struct Vec8<f16> { typedef f16x8 type; };
template <> struct Vec8<bf16> { typedef bf16x8 type; };
template <typename T> struct Vec2;
template <> struct Vec2<f16> { typedef f16x2 type; };
template <> struct Vec2<bf16> { typedef bf16x2 type; };
template <typename T> struct Vec4;
template <> struct Vec4<f16> { typedef f16x4 type; };
template <> struct Vec4<bf16> { typedef bf16x4 type; };

__device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// hi/lo split of an fp32 value onto 16-bit planes: x ~= hi + lo with hi = rn16(x), lo = rn16(x - hi).
// The two empty asm statements pin `x` and `hi` to one register each: without them hipcc (fp-contract=fast) may fold the
// producer of x into the residual, or convert x twice with differently fused inputs, so that the stored hi and the hi the
// residual was taken against disagree at rounding ties -- observed as rare sign-flipped lo (error = 1 ulp16 of x).
template <typename T, int NT>
__device__ __forceinline__ void split16(float x, T& hi, T& lo) {
    if (NT > 1) asm volatile("" : "+v"(x));
    hi = (T)x;
    if (NT > 1) {
        asm volatile("" : "+v"(hi));
        lo = (T)(x - (float)hi);
    }
}

// low plane of a split pair: lo = f16(x - hi).  x - float(hi) is exact in fp32 (hi is x rounded to 11 bits), so the fused form
// fma(float(hi), -1, x) rounded once to f16 is the same value -- and one v_fma_mixlo / mixhi_f16 per value instead of a convert
// back, a subtract and a share of a packed convert.  (bf16 has no mixed-precision fma: the three-step form.)
template <typename T>
__device__ __forceinline__ typename Vec2<T>::type residual2(f32x2 x, typename Vec2<T>::type hi) {
    typedef typename Vec2<T>::type V2;
    if constexpr (std::is_same<T, f16>::value) {
        const unsigned h = __builtin_bit_cast(unsigned, hi);
        unsigned lo;
        asm("v_fma_mixlo_f16 %0, %1, %3, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(h), "v"(x[0]), "s"(-1.0f));
        asm("v_fma_mixhi_f16 %0, %1, %3, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(h), "v"(x[1]), "s"(-1.0f));
        return __builtin_bit_cast(V2, lo);
    } else {
        const f32x2 back = {(float)hi[0], (float)hi[1]};
        return __builtin_convertvector(x - back, V2);
    }
}

// the same for two values at once: packed converts (v_cvt_pk_f16_f32 / v_cvt_pk_bf16_f32) and one v_pk_add_f32 for the
// residuals; `y` and `hi` are pinned to one register (pair) each like in split16
template <typename T, int NT>
__device__ __forceinline__ void split16x2(f32x2 y, typename Vec2<T>::type& hi, typename Vec2<T>::type& lo) {
    typedef typename Vec2<T>::type V2;
    if (NT > 1) asm volatile("" : "+v"(y));
    hi = __builtin_convertvector(y, V2);
    if (NT > 1) {
        asm volatile("" : "+v"(hi));
        // (fp16: one v_fma_mixlo / mixhi_f16 per value -- residual2 -- instead of two converts back, a packed subtract and a packed
        // convert: the same bits, 3 instead of 5 instructions per pair in every epilogue that writes planes; round 6)
        lo = residual2<T>(y, hi);
    }
}

// exact (erf) GELU as  gelu(x) = max(x, 0) - |x| * 2^-(|x| Q(|x|) + 1),  where  |x| Q(|x|) = -log2(erfc(|x| / sqrt 2))  and Q
// is a degree-6 minimax fit on [0, 5.7] (weighted by the sensitivity |x|^2 erfc; beyond 5.7 the correction is < 1e-8 and
// |x| is clamped).  One quarter-rate instruction (v_exp_f32) and 8 FMAs that pack two values per v_pk_fma_f32: the
// Abramowitz-Stegun 7.1.26 form used before needed v_rcp_f32 + v_exp_f32 and 13 unpacked operations, and the FFN1 / conv
// epilogues evaluate 2.6e9 GELUs per config-2 step.  Max abs error of the GELU value against float64 over [-9, 9]:
// 2.8e-7 (tools/gelu_probe.hip measures it on the device), of which 2.4e-7 is the rounding of the fp32 result itself.
constexpr float GELU_AMAX = 5.7f;
constexpr float GELU_Q0 = 1.1511269331183631f, GELU_Q1 = 0.4590439022614392f, GELU_Q2 = 0.052948624174104675f,
                GELU_Q3 = -0.00767084535295501f, GELU_Q4 = 0.0005758184767923562f, GELU_Q5 = 1.2789958507917868e-05f,
                GELU_Q6 = -4.277928111285899e-06f;
__device__ __forceinline__ float gelu_fast(float x) {
    const float ax = fminf(fabsf(x), GELU_AMAX);
    float q = fmaf(ax, GELU_Q6, GELU_Q5);
    q = fmaf(ax, q, GELU_Q4);
    q = fmaf(ax, q, GELU_Q3);
    q = fmaf(ax, q, GELU_Q2);
    q = fmaf(ax, q, GELU_Q1);
    q = fmaf(ax, q, GELU_Q0);
    const float e = __builtin_amdgcn_exp2f(-fmaf(ax, q, 1.0f));
    return fmaf(-ax, e, fmaxf(x, 0.f));
}
// two values at once on the packed fp32 pipe
__device__ __forceinline__ f32x2 gelu_fast2(f32x2 x) {
    const f32x2 ax = {fminf(fabsf(x[0]), GELU_AMAX), fminf(fabsf(x[1]), GELU_AMAX)};
    f32x2 q = ax * f32x2{GELU_Q6, GELU_Q6} + f32x2{GELU_Q5, GELU_Q5};
    q = ax * q + f32x2{GELU_Q4, GELU_Q4};
    q = ax * q + f32x2{GELU_Q3, GELU_Q3};
    q = ax * q + f32x2{GELU_Q2, GELU_Q2};
    q = ax * q + f32x2{GELU_Q1, GELU_Q1};
    q = ax * q + f32x2{GELU_Q0, GELU_Q0};
    const f32x2 arg = ax * q + f32x2{1.0f, 1.0f};
    const f32x2 e = {__builtin_amdgcn_exp2f(-arg[0]), __builtin_amdgcn_exp2f(-arg[1])};
    const f32x2 pos = {fmaxf(x[0], 0.f), fmaxf(x[1], 0.f)};
    return pos - ax * e;
}

// 64-lane reductions on the DPP cross-lane network (no LDS crossbar round trips): butterfly inside each row of 16 lanes
// (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror), then the four row totals via v_readlane.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float lane_value(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_move<0xB1>(v);
    v += dpp_move<0x4E>(v);
    v += dpp_move<0x141>(v);
    v += dpp_move<0x140>(v);
    return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_move<0xB1>(v));
    v = fmaxf(v, dpp_move<0x4E>(v));
    v = fmaxf(v, dpp_move<0x141>(v));
    v = fmaxf(v, dpp_move<0x140>(v));
    return fmaxf(fmaxf(lane_value(v, 0), lane_value(v, 16)), fmaxf(lane_value(v, 32), lane_value(v, 48)));
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) and the CU count are properties of a device, not of the process: a
// process may hold handles on several GPUs, so "done once" flags are kept per device id.
constexpr int MAX_DEVICES = 64;
inline int current_device() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= MAX_DEVICES) d = 0;
    return d;
}
struct OncePerDevice {
    bool done[MAX_DEVICES] = {};
    // true exactly once per device (the caller then sets the attribute)
    bool first() {
        const int d = current_device();
        if (done[d]) return false;
        done[d] = true;
        return true;
    }
};

// ---------------------------------------------------------------------------------------------------------------
// GEMM:  C[M,N] = epilogue( A[M,K] . W[N,K]^T )      A and W are K-contiguous 16-bit planes
// ---------------------------------------------------------------------------------------------------------------
constexpr int GEMM_LN_TILE_ROWS = 128;  // tile height of the row-complete conv kernel (GemmParams.tile_list)

struct GemmParams {
    // A operand: row r lives at A + (r / rows_per_batch) * a_batch_stride + (r % rows_per_batch) * lda  (elements).
    // Overlapping rows (lda < K) express the strided 1-D convolutions as implicit GEMMs over channels-last input.
    // Strides and offsets of A, W and out_p are LOGICAL elements in both plane layouts (plane == PLANE_IL: see pidx()).
    const void* A;
    int64_t a_plane;  // distance hi plane -> lo plane (elements); PLANE_IL: interleaved
    int64_t lda;
    int64_t rows_per_batch;
    int64_t a_batch_stride;
    const void* W;  // [N, ldw]
    int64_t w_plane;
    int64_t ldw;
    int M, N, K;  // K % 8 == 0
    // grid.z batching (grouped positional convolution): pointer advances per z (elements)
    int64_t za, zw, zbias, zout, zoutp;
    // epilogue: v = acc*scale + bias[n]; v = act(v); v += residual[m,n]; if row masked: v = 0
    float scale;
    const float* bias;
    int act;  // 0 none, 1 exact GELU
    const float* residual;
    int64_t ldr;
    const int* row_len;  // optional [batch]: row r = b*rows_T + t is zeroed when t >= row_len[b]
    int rows_T;
    float* out_f32;
    int64_t ldo;
    void* out_p;  // 16-bit planes
    int64_t out_plane;
    int64_t ldp;
    // QKV scatter epilogue (mode 1): columns [0,D) -> Q[b,h,t,dh], [D,2D) -> K[b,h,t,dh], [2D,3D) -> V[b,h,t,dh]
    int mode;
    int vec_ok;  // set by launch_gemm: N % 4 == 0 and every output row stride is a multiple of 4 elements
    void* q;
    void* k;
    void* v;
    int64_t qk_plane;  // plane distance of q, k and v
    int T, Tp, H, dh;
    int dhp;  // elements per row of q / k / v (>= dh: 64 or 128; the scatter never writes columns [dh, dhp), which stay zero)
    // fused LayerNorm over the N outputs of a row + GELU (gemm_fuses_ln(); the row-complete 128 x 512 kernel, N == 512)
    const float* ln_gamma;
    const float* ln_beta;
    float ln_eps;
    // split-K: products whose tile count cannot fill the chip are cut along K into `splits` chunks; every chunk writes a
    // raw fp32 partial [M, N] into a slab of `splitk_ws` and a fix-up kernel reduces the slabs (fixed order, so results are
    // deterministic) and applies the epilogue.  The caller only provides the workspace; launch_gemm decides.
    float* splitk_ws;
    int64_t splitk_ws_elems;  // capacity in floats
    int splits;               // internal (kernel view): number of K chunks, K = chunk length
    int64_t split_out;        // internal: distance between partial slabs (elements)
    // ragged batches (row-complete conv kernel only): the 128-row tiles to compute, ascending (the tiles that lie wholly inside
    // the padding of one batch item are left out: neither computed nor stored); null: every tile
    const int* tile_list;  // tile index = first row / GEMM_LN_TILE_ROWS
    int n_tiles;
    // set by a caller that will run the split-K fix-up itself (launch_fixup_rownorm: fused with the LayerNorm that follows the
    // product): launch_gemm then writes the partial slabs only.  Only with gemm_planned_splits(...) > 1.
    int defer_fixup;
    // implicit-GEMM convolutions on the row-complete kernel (gemm_ln_tap_minor_slice()): K walks the taps INSIDE a channel
    // slice -- slice s of W (packed by launch_pack_conv_w(..., tap_minor_slice)) holds tap (a_taps - s % a_taps) % a_taps (order
    // 0, 2, 1 for three taps) of channel slice s / a_taps, and the A operand's slice s starts tap * a_tap_stride + (s / a_taps) *
    // slice elements into the row.  With k = 3,
    // s = 2 the last tap of output row r is the first tap of row r + 1: in tap-major order the two fetches of that input row lie
    // a third of the K loop apart (32 slices x 80 KiB x 32 workgroups per XCD: far beyond the 4 MiB L2, so every input row was
    // fetched 1.5 times from beyond L2); tap-minor they are adjacent slices and the second one hits L2.  0 / 1: plain K order.
    int a_taps;
    int64_t a_tap_stride;
    unsigned long long* stamps;  // developer diagnostic (-DAMX_PP_STAMP builds of tools/gemm_bench.hip), else null
    // ---- LayerNorm folded into the products around it (pre-LN encoder layers; ping-pong kernel only: gemm_ln_fold_ok()) ----
    // LN(x) . W^T + b  =  rstd * ((x - p) . (gamma (.) W)^T) - rstd * (mu - p) * c + d   with c = W gamma, d = W beta + b, for
    // ANY per-row pivot p.  PRODUCER (a residual product: out-projection, FFN2; ln_partial != null): besides the fp32 stream
    // v = acc * scale + bias + residual it writes the planes out_p of u = (v - p_m) * s_m -- pivot and power-of-two scale of row m
    // from ln_rowps, what the row's PREVIOUS statistics gave -- and, per row and 64-column block, (sum, sum of squares about the
    // block's own mean) of v - p_m into ln_partial[m * (N / 64) + block]: fixed slots, no atomics, so results stay bitwise
    // reproducible.  launch_ln_finalize() merges the blocks (Chan's update: no E[x^2] - E[x]^2 cancellation however far the pivot
    // is from the mean) into row_coef[m] = (rstd / s_m, -rstd * (mu - p_m)) and moves ln_rowps on.  CONSUMER (QKV, FFN1;
    // row_coef != null): A = the planes of u, W = gamma (.) W packed at amx_create, bias = d, and the epilogue computes
    // v = alpha_m * (acc * scale) + beta_m * col_c[n] + bias[n].
    const float4* ln_rowps;   // producer: [M] (pivot, scale) of the planes the row holds NOW, (pivot, scale) they are written under next
    float2* ln_partial;       // producer: [M][N / 64]
    // producer, two-plane modes: the residual is read from the stream's own planes (out_p, in place, under the first pair of
    // ln_rowps) instead of the fp32 rows, and out_f32 may be null: between the products of a layer the stream then exists as planes
    // only (22 significant bits about the row's pivot: emulated in tests/diagnostics/emulate_ln_fold.py, `fold_planes`) and the
    // fp32 copy is written where something reads it -- 65 MB less per product at config 2, the tail of these kernels being a pure
    // HBM burst (profiles/r06_per_product_durations.log)
    int ln_res_planes;
    const float2* row_coef;   // consumer: [M] (alpha, beta)
    const float* col_c;       // consumer: [N]
};

extern bool g_force_generic_gemm;
void launch_gemm(int prec, const GemmParams& p, hipStream_t stream);
// true when launch_gemm routes this product to the 256x256 ping-pong kernel (else: generic tile kernel)
bool gemm_uses_pp(int prec, const GemmParams& p);
// true when a product with ln_gamma / ln_beta set can run on the row-complete kernel with fused LayerNorm + GELU
bool gemm_fuses_ln(int prec, const GemmParams& p);
// K elements per slice when that product will run on the whole-line kernel that understands a_taps (0: it will not -- the
// caller then passes the tap-major weights and a_taps = 0)
int gemm_ln_tap_minor_slice(int prec, const GemmParams& p);
// K chunks launch_gemm will cut this product into (1: no split-K, no fix-up launch); `p` as it will be passed to launch_gemm,
// split-K workspace included
int gemm_planned_splits(int prec, const GemmParams& p);
// the fix-up of a product launched with defer_fixup, fused with the LayerNorm of the rows it completes: p.out_f32 (the
// residual stream, N = row width <= 1024, N % 4 == 0) gets scale * sum of slabs + bias + residual, and LayerNorm(gamma, beta,
// eps) of those rows goes to the planes out_p (may be null) and to out_ln (fp32, may be null)
bool fixup_rownorm_eligible(const GemmParams& p);
void launch_fixup_rownorm(int prec, const GemmParams& p, int splits, const float* gamma, const float* beta, float eps, void* out_p,
                          int64_t out_plane, int64_t ldp, float* out_ln, int64_t ldo_ln, hipStream_t stream);
// true when launch_gemm runs this product -- a producer (ln_partial set) or a consumer (row_coef set) of the LayerNorm fold -- on
// the ping-pong kernel in one piece (no K chunks), the only place the fold's epilogues exist
bool gemm_ln_fold_ok(int prec, const GemmParams& p);
// same kernel with grid.z = groups (per-group pointer advances za/zw/zbias/zout/zoutp); requires N <= 64
void launch_gemm_grouped(int prec, const GemmParams& p, int groups, hipStream_t stream);

// ---------------------------------------------------------------------------------------------------------------
// attention
// ---------------------------------------------------------------------------------------------------------------
struct AttnParams {
    const void* q;  // [N,H,Tp,dh] planes, already scaled by dh^-0.5 * log2(e)
    const void* k;  // [N,H,Tp,dh]
    const void* v;  // [N,H,Tp,dh]; rows [T, Tp) of all three must be finite (zero)
    int64_t qk_plane;
    void* out;  // [N*T, D] planes, column h*dh + d
    int64_t out_plane;
    const int* frame_len;  // [N] valid keys per utterance
    int N, H, T, Tp, dh;
    int dhp;  // elements per Q / K / V row: 64 for dh <= 64, 128 for dh in (64, 128]; columns [dh, dhp) hold zeros
    // packed rows (null: padded layout as described above).  With row_off the utterances lie back to back: Q / K / V are
    // [H, Tp, dh] planes in which utterance n owns rows row_off[n] .. row_off[n] + frame_len[n] (Tp = rows per head, at
    // least 64 finite rows beyond the last utterance), `out` is [sum(frame_len), D] and only valid queries are computed
    const int* row_off;
    const int* order;  // packed rows only, may be null: utterance indices, longest first (workgroups are dispatched in this order)
    unsigned long long* stamps;  // developer diagnostic (-DAMX_ATTN_STAMP builds of tools/attn_bench.hip), else null
};
void launch_attention(int prec, const AttnParams& p, hipStream_t stream);

// ---------------------------------------------------------------------------------------------------------------
// row-wise / elementwise kernels (amx_rowops.hip)
// ---------------------------------------------------------------------------------------------------------------
void launch_audio_stats(const float* audio, const int64_t* lengths, int N, int64_t L, double* partial, float* mean_rstd,
                        int do_normalize, hipStream_t s);
// conv layer 0 (C_in = 1) + LayerNorm(C) + GELU, fused; writes planes [N*T1, C]
void launch_conv0(int prec, const float* audio, const int64_t* lengths, const float* mean_rstd, int N, int64_t L, int T1,
                  int C, int k, int stride, const float* w /*[C,k]*/, const float* b, const float* gamma,
                  const float* beta, float eps, int do_normalize, void* out, int64_t out_plane, int skip_padding, hipStream_t s,
                  const double* mfma_stats = nullptr, float w_scale = 1.f);
// (skip_padding: frame blocks that start beyond an utterance's own frames are not computed -- ragged batches)
// (mfma_stats: device table of CONV0_MFMA_STATS doubles for the matrix-pipe form of the kernel, conv0_mfma_eligible shapes:
// the mean of the rows [w_c, b_c] over the channels (11) and their covariance (11 x 11, row-major), computed in fp64 at
// amx_create; w_scale: the power of two the fp16 weight planes are built under.  Null: the VALU kernel.)
constexpr int CONV0_MFMA_STATS = 11 + 121;
bool conv0_mfma_eligible(int C, int k, int stride);
// the group-norm feature extractor (feat_extract_norm = "group"): conv layer 0 + GroupNorm(C groups) over the T1 frames of the
// padded length + GELU.  Two passes over the audio (the k-tap conv is recomputed, never stored): per-(utterance, channel)
// fp64 statistics into `partial` (conv0_groupnorm_partial_bytes) -> scale / shift [N, C] -> the conv0 kernel with the affine
// form y = conv * scale + shift.  gamma / beta: the GroupNorm affine parameters [C]
size_t conv0_groupnorm_partial_bytes(int N, int T1, int C);
// dynamic LDS the conv-0 kernels request for this first-layer geometry (must stay <= 64 KiB: checked by amx_create)
size_t conv0_window_lds_bytes(int k, int stride, int group_norm);
void launch_conv0_groupnorm(int prec, const float* audio, const int64_t* lengths, const float* mean_rstd, int N, int64_t L, int T1,
                            int C, int k, int stride, const float* w, const float* b, const float* gamma, const float* beta,
                            float eps, int do_normalize, double* partial, float* scale, float* shift, void* out, int64_t out_plane,
                            int skip_padding, hipStream_t s, float mfma_w_scale = 0.f);
// (mfma_w_scale > 0 and a conv0_mfma_eligible shape: GroupNorm statistics from the utterance's 10 x 10 sample covariance in fp64
// -- no second evaluation of the convolution -- and the apply pass on the matrix pipe; the power of two is that of the fp16
// weight planes)
// rows of the padded [N, T, D] fp32 matrix <-> rows of the packed [sum(frame_len), D] matrix (utterance n at row_off[n]);
// only rows t < frame_len[n] move
void launch_pack_rows(const float* padded, float* packed, const int* row_off, const int* frame_len, int N, int T, int D, bool unpack,
                      hipStream_t s);
// rows of width D: [LN1 -> GELU] (if gamma1) then [LN2] (if gamma2); outputs planes and/or f32
void launch_rownorm(int prec, const float* x, int64_t ldx, int64_t M, int D, const float* gamma1, const float* beta1,
                    int gelu, const float* gamma2, const float* beta2, float eps1, float eps2, void* out_p,
                    int64_t out_plane, int64_t ldp, float* out_f32, int64_t ldo, hipStream_t s);
// LayerNorm fold (GemmParams.ln_partial / row_coef): the FIRST norm of the encoder stack, from the fp32 stream itself: exact row
// statistics (two passes over the row in registers, like launch_rownorm) -> planes of u = (x - mu) * s with s the power of two
// that puts 16 / sigma ... 8 / sigma into it, rowps[m] = (mu, s, mu, s), coef[m] = (rstd / s, 0)
void launch_ln_rowprep(int prec, const float* x, int64_t ldx, int64_t M, int D, float eps, void* out_p, int64_t out_plane, int64_t ldp,
                       float4* rowps, float2* coef, hipStream_t s);
// merges the per-block statistics a producer left in `partial` [M][blocks] (of v - pivot, 64 columns per block) into
// coef[m] = (rstd / s_m, -rstd * (mu - p_m)) for the consumer of those planes, then moves rowps[m] on: the pair the producer wrote
// under becomes "what the planes hold now", (mu, scale of the new rstd) what the next producer writes under
void launch_ln_finalize(const float2* partial, int blocks, int64_t M, float eps, float4* rowps, float2* coef, hipStream_t s);
// the same with the rows of a ragged batch gathered on the way out: input row n * T_rows + t -> plane row row_off[n] + t,
// frames t >= frame_len[n] dropped (the feature projection of a ragged batch then runs on the valid frames only)
void launch_rownorm_to_packed(int prec, const float* x, int64_t ldx, int64_t M, int D, const float* gamma1, const float* beta1,
                              int gelu, const float* gamma2, const float* beta2, float eps1, float eps2, void* out_p,
                              int64_t out_plane, int64_t ldp, const int* row_off, const int* frame_len, int T_rows, hipStream_t s);
// grouped, zero-padded 16-bit image of the (masked) projected features for the positional convolution; row_off / frame_len
// (may be null): h holds the packed rows of a ragged batch
void launch_posconv_pack(int prec, const float* h, int N, int T, int D, int G, int pad_front, int Tpad, void* out,
                         int64_t out_plane, const int* row_off, const int* frame_len, hipStream_t s);

// window-resident grouped positional convolution (amx_posconv.hip): h[n, t, g*64 + co] += gelu(bias + conv) from the padded
// image [G][N][Tpad][64] and the weights [G][64][taps * 64]; eligible when hidden / groups == 64 and taps <= 128
bool posconv_window_eligible(int D, int G, int taps, int N, int Tn, int Tpad, int64_t image_plane);
// (row_off / frame_len, may be null: h holds the packed rows of a ragged batch; frame blocks beyond an utterance are skipped)
void launch_posconv_window(int prec, const void* image, int64_t image_plane, const void* weights, int64_t w_plane, int64_t ldw,
                           const float* bias, float scale, float* h, int N, int Tn, int Tpad, int D, int G, int taps, const int* row_off,
                           const int* frame_len, hipStream_t s);

struct ConcatPart {
    int type;     // 0: fp32 hidden rows -> planes; 1: softmax over logits columns
    int src_col;  // column offset in the logits buffer (type 1)
    int width;    // number of source columns
    int dst_col;
    const float* src;  // type 0: [M, width] fp32 (ld = width)
};
void launch_concat(int prec, const ConcatPart* parts_dev, int n_parts, const float* logits, int64_t ld_logits, int64_t M,
                   void* out, int64_t out_plane, int64_t ldp, int kpad, hipStream_t s);

// time-layer classifier heads (ProjectingMultiheadAttention): LayerNorm + sinusoidal positions -> planes [M, kpad];
// key-masked fp32 attention over the frames of each utterance on qkv [M, 3C] -> planes [M, kpad]
void launch_time_ln_pe(int prec, const float* x, int64_t M, int C, int T, const float* gamma, const float* beta, float eps,
                       const float* pe_base /*[C] or null*/, void* out, int64_t out_plane, int kpad, hipStream_t s);
size_t time_attention_lds_bytes(int T, int dh);
void launch_time_attention(int prec, const float* qkv, const int* frame_len, int N, int T, int C, int heads, void* out,
                           int64_t out_plane, int kpad, hipStream_t s);

struct OutDesc {
    int col;          // column offset in the logits buffer
    int C;            // classes incl. blank
    int64_t prefix;   // classes of all earlier output blocks: the [T,N,C] block starts at T * N * prefix floats
                      // (geometry-independent, so the device table only changes with the inventory)
};
// `nonfinite` (device counter, may be null): incremented once per valid frame whose logits hold a NaN or an infinity
// `row_off` (may be null): the logits rows are the packed rows of a ragged batch (utterance n at row_off[n])
void launch_logsoftmax_out(const OutDesc* descs_dev, int n_out, const float* logits, int64_t ld, int N, int T,
                           const int* frame_len, int log_probs, float* out, int* nonfinite, const int* row_off, hipStream_t s);
void launch_greedy_ctc(const OutDesc* descs_dev, int n_out, const float* out, const int* frame_len, int N, int T,
                       int64_t* tokens, int64_t* timesteps, int* counts, float* scores, hipStream_t s);
// the same decoder over one [N, T, C] emission tensor with element strides (stride_n, stride_t, 1)
void launch_greedy_ctc_emissions(const float* emissions, int64_t stride_n, int64_t stride_t, const int* frame_len, int N, int T,
                                 int C, int blank, int64_t* tokens, int64_t* timesteps, int* counts, float* scores, hipStream_t s);

// weight packing helpers (device side; run once at amx_create / amx_set_inventory)
void launch_pack_matrix(int prec, const float* src, int rows, int cols, int64_t src_row_stride, int64_t src_col_stride,
                        float scale, void* dst, int64_t dst_plane, int64_t ldd, int cols_pad, hipStream_t s);
// (`scale`: the weights are multiplied by it before the 16-bit split -- a per-tensor power of two chosen at amx_create so
// that the largest weight lands in [4096, 8192): the lo plane of an fp16 pair is only a normal number, i.e. only carries its
// full 11 bits, for |w| >= 0.25, and the hi plane underflows below 6e-5; the product's epilogue multiplies by 1 / scale)
void launch_pack_conv_w(int prec, const float* src /*[Co,Ci,k]*/, int Co, int Ci, int k, float scale, void* dst, int64_t dst_plane,
                        hipStream_t s, int tap_minor_slice = 0);
// (tap_minor_slice = S > 0: K order [channel slice of S][tap][S channels] instead of [tap][channel] -- GemmParams.a_taps)
void launch_pack_posconv_w(int prec, const float* g /*[k]*/, const float* v /*[D,cg,k]*/, int D, int cg, int k, float scale,
                           float* norm_scratch, void* dst /*[G][cg][k*cg]*/, int64_t dst_plane, hipStream_t s);
void launch_compose(int prec, const float* emb, int E, const int64_t* idx /*[P+1, F] absolute rows, row 0 = blank*/,
                    int P1, int F, float scale, float* composed_f32 /*[P1,E]*/, void* dst, int64_t dst_plane, int64_t ldd,
                    hipStream_t s);
void launch_scale_copy(const float* src, float* dst, int64_t n, float scale, hipStream_t s);
// zero fills as kernels (a forward pass holds no hipMemset: see amx_rowops.hip); sizes / pitch multiples of 4 bytes
void launch_zero(void* p, size_t bytes, hipStream_t s);
void launch_zero_2d(void* base, size_t pitch, size_t width_bytes, size_t rows, hipStream_t s);
void launch_copy(void* dst, const void* src, size_t bytes, hipStream_t s);

}  // namespace amx
