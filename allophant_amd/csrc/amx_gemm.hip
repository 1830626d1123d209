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

// device code: the kernels live in three include files of this translation unit
#include "amx_gemm_tile.inc"  // gemm_epilogue, gemm_kernel, gemm_dma_kernel, splitk_fixup_kernel
#include "amx_gemm_pp.inc"    // gemm_pp_kernel and its epilogues
#include "amx_gemm_ln.inc"    // gemm_ln_kernel

// ---------------------------------------------------------------------------------------------------------------
// host side: eligibility, the plan (tile height / K chunks / kernel) of a product, launches
// ---------------------------------------------------------------------------------------------------------------
int device_cus() {
    static int per_device[MAX_DEVICES] = {};
    const int dev = current_device();
    int& cus = per_device[dev];
    if (!cus) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        // developer switch: size the persistent grids for a CU-masked stream (tools/two_stream_probe.py)
        if (dev_int("AMX_FORCE_CUS", 0) > 0) cus = dev_int("AMX_FORCE_CUS", 0);
        if (cus <= 0) cus = 256;
        cus -= cus % 8;  // the tile order assumes sequence numbers i and i + grid share an XCD
        if (cus < 8) cus = 8;
    }
    return cus;
}

bool ln_eligible(int NT, const GemmParams& p) {
    if (g_force_generic_gemm) return false;
    if (!p.ln_gamma || !p.ln_beta || p.act != 1 || !p.out_p || p.out_f32 || p.residual || p.row_len || p.mode != 0) return false;
    if (p.N != ppw::BN || p.K % (128 / NT) != 0 || p.M < 1024) return false;
    if (p.lda % 8 || p.ldw % 8 || p.a_plane % 8 || p.w_plane % 8 || p.a_batch_stride % 8) return false;
    if (((uintptr_t)p.A & 15) || ((uintptr_t)p.W & 15)) return false;
    if (p.M > p.rows_per_batch && (p.rows_per_batch < 256 || p.a_batch_stride < (p.rows_per_batch - 1) * p.lda)) return false;
    {
        const bool a_il = NT > 1 && p.a_plane == PLANE_IL, w_il = NT > 1 && p.w_plane == PLANE_IL;
        if ((a_il && (p.lda % 32 || p.a_batch_stride % 32)) || (w_il && p.ldw % 32)) return false;
        const int64_t a_span = (a_il ? 2 : 1) * ((p.M > p.rows_per_batch ? p.a_batch_stride : 0) + 256 * p.lda + p.K) + (NT > 1 ? p.a_plane : 0);
        const int64_t w_span = (w_il ? 2 : 1) * (512 * p.ldw + p.K) + (NT > 1 ? p.w_plane : 0);
        if (a_span < 0 || w_span < 0 || a_span * 2 >= (int64_t)0xFFFFFF00 || w_span * 2 >= (int64_t)0xFFFFFF00) return false;
    }
    if (p.bias && ((uintptr_t)p.bias & 15)) return false;
    if (((uintptr_t)p.ln_gamma & 15) || ((uintptr_t)p.ln_beta & 15)) return false;
    if (p.ldp % 8 || p.out_plane % 8 || ((uintptr_t)p.out_p & 15)) return false;
    return true;
}

// the whole-line kernel (gemm_ln_il_kernel) takes the product: the one that understands the tap-minor K order
bool ln_uses_il(int NT, const GemmParams& p) {
    static const bool plain_loop = dev_switch("AMX_LN_SEGMENT_LOOP");  // developer A/B
    const bool layout_ok = NT == 1 || (p.a_plane == PLANE_IL && p.w_plane == PLANE_IL && p.out_plane == PLANE_IL);
    return !plain_loop && layout_ok && p.K % (128 / NT) == 0;
}

template <typename T, int NT>
void launch_gemm_ln(const GemmParams& p, hipStream_t stream) {
    static OncePerDevice attr;
    if (attr.first())
        (void)hipFuncSetAttribute((const void*)gemm_ln_kernel<T, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, ppw::LDS_BYTES);
    const int tiles = p.tile_list ? p.n_tiles : (p.M + ppw::BM - 1) / ppw::BM;
    if (tiles <= 0) return;
    const int cus = device_cus();
    dim3 grid(tiles < cus ? tiles : cus, 1, 1);
    {
        // whole-line operand DMA, one segment pair per K slice (two planes: interleaved operands only)
        if (ln_uses_il(NT, p)) {
            static OncePerDevice attr_il;
            if (attr_il.first())
                (void)hipFuncSetAttribute((const void*)gemm_ln_il_kernel<T, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, ppi::LDS_BYTES);
            hipLaunchKernelGGL((gemm_ln_il_kernel<T, NT>), grid, dim3(512), ppi::LDS_BYTES, stream, p);
            return;
        }
    }
    hipLaunchKernelGGL((gemm_ln_kernel<T, NT>), grid, dim3(512), ppw::LDS_BYTES, stream, p);
}

// below this many rows the 128 x 128 tile kernel with K chunks on grid.z is as fast (tools/gemm_bench `small`)
constexpr int PP_MIN_ROWS = 384;

bool pp_eligible(int NT, const GemmParams& p) {
    // eligibility: whole sub-step groups, aligned operand rows and vector epilogue, enough rows to fill the chip
    if (g_force_generic_gemm) return false;
    if (p.K % (128 / NT) != 0 || p.N < 256 || p.N % 4 != 0 || p.M < PP_MIN_ROWS) return false;
    if (p.lda % 8 || p.ldw % 8 || p.a_plane % 8 || p.w_plane % 8 || p.a_batch_stride % 8) return false;
    // two planes: the slice-per-phase loop reads interleaved operands only (block-aligned rows)
    if (NT > 1 && (p.a_plane != PLANE_IL || p.w_plane != PLANE_IL || p.lda % 32 || p.ldw % 32 || p.a_batch_stride % 32)) return false;
    if (((uintptr_t)p.A & 15) || ((uintptr_t)p.W & 15)) return false;
    // DMA addressing: 32-bit byte offsets from the tile's first row; rows of a tile ascend in memory
    if (p.M > p.rows_per_batch && (p.rows_per_batch < 256 || p.a_batch_stride < (p.rows_per_batch - 1) * p.lda)) return false;
    {
        // (interleaved planes: every logical offset doubles and the lo value is 64 bytes behind)
        const int64_t a_span = NT * ((p.M > p.rows_per_batch ? p.a_batch_stride : 0) + 256 * p.lda + p.K) + 64;
        const int64_t w_span = NT * (256 * p.ldw + p.K) + 64;
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

// LayerNorm fold (GemmParams.row_coef / ln_partial; callers checked gemm_ln_fold_ok): its own kernel instances, one piece, no K chunks
template <typename T, int NT, int MI, int NI, int FOLD>
void launch_pp_fold(const GemmParams& p, hipStream_t stream) {
    static OncePerDevice attr;
    constexpr int lds = pp::lds_bytes(NT, MI, NI);
    if (attr.first())
        (void)hipFuncSetAttribute((const void*)gemm_pp_kernel<T, NT, MI, NI, FOLD>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int tiles = ((p.N + NI * 64 - 1) / (NI * 64)) * ((p.M + MI * 32 - 1) / (MI * 32));
    const int cus = device_cus();
    hipLaunchKernelGGL((gemm_pp_kernel<T, NT, MI, NI, FOLD>), dim3(tiles < cus ? tiles : cus, 1, 1), dim3(512), lds, stream, p);
}

template <typename T, int NT, int MI, int NI = 4>
void launch_pp_tiles(const GemmParams& p_in, int splits, hipStream_t stream) {
    // developer timing switches (WRONG results: the fold's products run as plain products, to price its epilogues one side at a time)
    static const bool producer_plain = dev_switch("AMX_FOLD_PRODUCER_PLAIN"), consumer_plain = dev_switch("AMX_FOLD_CONSUMER_PLAIN");
    GemmParams p = p_in;
    if (producer_plain && p.ln_partial) { p.ln_partial = nullptr; p.ln_rowps = nullptr; p.out_p = nullptr; p.ln_res_planes = 0; }
    if (consumer_plain && p.row_coef) { p.row_coef = nullptr; p.col_c = nullptr; }
    if (p.ln_partial) {
        if constexpr (NI == 4) launch_pp_fold<T, NT, MI, 4, 2>(p, stream);
        return;
    }
    if (p.row_coef) {
        launch_pp_fold<T, NT, MI, NI, 1>(p, stream);
        return;
    }
    static OncePerDevice attr;
    constexpr int lds = pp::lds_bytes(NT, MI, NI);
    if (attr.first())
        (void)hipFuncSetAttribute((const void*)gemm_pp_kernel<T, NT, MI, NI>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int tiles = ((p.N + NI * 64 - 1) / (NI * 64)) * ((p.M + MI * 32 - 1) / (MI * 32));
    const int cus = device_cus();
    const int units = tiles * splits;
    dim3 grid(units < cus ? units : cus, 1, 1);  // persistent: one 112-160-KiB-LDS workgroup per CU
    if (splits > 1) {
        hipLaunchKernelGGL((gemm_pp_kernel<T, NT, MI, NI>), grid, dim3(512), lds, stream, split_view(p, splits));
        if (!p.defer_fixup) launch_fixup<T, NT>(p, splits, stream);
    } else {
        hipLaunchKernelGGL((gemm_pp_kernel<T, NT, MI, NI>), grid, dim3(512), lds, stream, p);
    }
}

// Tile height (256 or 128 rows) and number of K chunks of a ping-pong product, by a cost model in units of one 32-deep MFMA
// segment pair per K element (~ 0.0166 us; fitted to tools/gemm_bench `small` on MI355X):
//   rounds of the persistent grid x (main loop of a work unit + its prologue / epilogue) + fix-up launch and slab traffic.
// Config-2-sized products (>= one full round of 256-row tiles) always come out as (256 rows, 1 chunk).
// NI (W fragments per wave: 4 = 256-column tiles, 3 = 192-column tiles, see gemm_pp_kernel) is part of the plan: the narrower
// tile costs 3/4 of the matrix work of a unit but moves 7/8 (256 rows) or 5/6 (128 rows) of its operand bytes and takes the
// generic epilogue, priced as + 6 % on the loop and + 10 % on the per-unit overhead.
void pp_plan(int NT, const GemmParams& p, int* mi_out, int* splits_out, int* ni_out) {
    const int cus = device_cus();
    const double loop = (double)p.K * (NT > 1 ? 3 : 1);  // main loop of a 256 x 256 tile
    static const bool no_narrow = dev_switch("AMX_NO_NARROW_TILES");  // developer A/B switch: 256-column tiles only
    static const int force_ni = dev_int("AMX_PP_FORCE_NI", 0), force_mi = dev_int("AMX_PP_FORCE_MI", 0);  // developer: one tile shape
    double best = 1e30;
    *mi_out = 8;
    *splits_out = 1;
    *ni_out = 4;
    for (int mi = 8; mi >= 4; mi -= 4)
        for (int ni = 4; ni >= (no_narrow ? 4 : 3); --ni) {
            if ((force_ni && ni != force_ni) || (force_mi && mi != force_mi)) continue;
            const int tiles = ((p.N + ni * 64 - 1) / (ni * 64)) * ((p.M + mi * 32 - 1) / (mi * 32));
            const double per_unit = (480.0 + 60.0 * mi) * (ni == 4 ? 1.0 : 1.1);  // prologue + epilogue of a work unit
            const double unit_loop = loop * mi / 8.0 * (ni == 4 ? 1.0 : 0.75 * 1.06);
            int64_t smax = 1;
            if (p.splitk_ws && p.K % 128 == 0) {
                smax = std::min<int64_t>(p.K / 256, p.splitk_ws_elems / ((int64_t)p.M * p.N));
                smax = std::min<int64_t>(smax, 32);
                // (a product too large for even one slab -- M x N beyond the workspace -- is still planned: until round 5 this
                // came out as 0 candidates, i.e. always 256 x 256 tiles; QKV of 16 x 10 s ran 164 instead of 145 us for it)
                smax = std::max<int64_t>(smax, 1);
            }
            for (int sp = 1; sp <= smax; ++sp) {
                if ((p.K / 128) % sp) continue;
                const int64_t units = (int64_t)tiles * sp;
                double cost = (double)((units + cus - 1) / cus) * (unit_loop / sp + per_unit);
                if (sp > 1) cost += 700.0 + (double)sp * p.M * p.N * 8.0 / 4.0e6 / 0.0166;  // fix-up launch + slab write / read at 4 TB/s
                if (cost < best * 0.97) {  // prefer the earlier candidate (taller, wider tile, fewer chunks) on near ties
                    best = cost;
                    *mi_out = mi;
                    *splits_out = sp;
                    *ni_out = ni;
                }
            }
        }
}

template <typename T, int NT>
bool launch_gemm_pp(const GemmParams& p, hipStream_t stream) {
    if (!pp_eligible(NT, p)) return false;
    int mi, splits, ni;
    pp_plan(NT, p, &mi, &splits, &ni);
    if (mi == 8 && ni == 4) launch_pp_tiles<T, NT, 8, 4>(p, splits, stream);
    else if (mi == 8) launch_pp_tiles<T, NT, 8, 3>(p, splits, stream);
    else if (ni == 4) launch_pp_tiles<T, NT, 4, 4>(p, splits, stream);
    else launch_pp_tiles<T, NT, 4, 3>(p, splits, stream);
    return true;
}

bool dma_tile_eligible(int NT, const GemmParams& p) {
    static const bool off = dev_switch("AMX_NO_DMA_TILE");  // developer A/B switch
    if (off || p.K % BK != 0) return false;
    if (p.lda % 8 || p.ldw % 8 || p.a_plane % 8 || p.w_plane % 8 || p.a_batch_stride % 8 || p.za % 8 || p.zw % 8) return false;
    if (((uintptr_t)p.A & 15) || ((uintptr_t)p.W & 15)) return false;
    // rows of a tile ascend in memory and stay inside 32-bit byte offsets from the tile's first row
    if (p.M > p.rows_per_batch && p.a_batch_stride < (p.rows_per_batch - 1) * p.lda) return false;
    const int64_t batches_per_tile = p.M > p.rows_per_batch ? (128 + p.rows_per_batch - 1) / p.rows_per_batch + 1 : 0;
    const bool a_il = NT > 1 && p.a_plane == PLANE_IL, w_il = NT > 1 && p.w_plane == PLANE_IL;
    if ((a_il && (p.lda % 32 || p.a_batch_stride % 32 || p.za % 32)) || (w_il && (p.ldw % 32 || p.zw % 32))) return false;
    const int64_t a_span = (a_il ? 2 : 1) * (batches_per_tile * p.a_batch_stride + 128 * p.lda + p.K) + (NT > 1 ? p.a_plane : 0);
    const int64_t w_span = (w_il ? 2 : 1) * (64 * p.ldw + p.K) + (NT > 1 ? p.w_plane : 0);
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
    dim3 grid((q.N + BN - 1) / BN, (q.M + BM - 1) / BM, zdim);
    if constexpr (NT == 2) {
        if (q.a_plane == PLANE_IL && q.w_plane == PLANE_IL) {  // interleaved operands: whole-line pieces
            static OncePerDevice attr_il;
            if (attr_il.first())
                (void)hipFuncSetAttribute((const void*)gemm_dma_kernel<T, NT, BM, BN, WM, WN, STAGES, true>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            hipLaunchKernelGGL((gemm_dma_kernel<T, NT, BM, BN, WM, WN, STAGES, true>), grid, dim3(256), lds, stream, q);
            return;
        }
    }
    static OncePerDevice attr;
    if (attr.first())
        (void)hipFuncSetAttribute((const void*)gemm_dma_kernel<T, NT, BM, BN, WM, WN, STAGES>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((gemm_dma_kernel<T, NT, BM, BN, WM, WN, STAGES>), grid, dim3(256), lds, stream, q);
}

template <typename T, int NT>
void launch_gemm_dma_shape(int shape, const GemmParams& q, int zdim, hipStream_t stream) {
    if (shape == 2) launch_gemm_dma<T, NT, 64, 32, 2, 2, 6>(q, zdim, stream);
    else launch_gemm_dma<T, NT, 128, 64, 4, 1, 3>(q, zdim, stream);
}

// A product that fits one round of LDS-DMA tiles runs there instead of on the ping-pong kernel when it is short (below
// ~768 rows: tools/geometry_sweep.py, 1 x 10 s 5.5 -> 4.0 ms) or when its 128 x 256 ping-pong tiles would occupy less than
// half of the CUs (N = 1024 products of a few thousand rows: 64 tiles on 256 CUs) while 128 x 64 tiles fill them.
// AMX_DMA_MAX_ROWS (developer switch) overrides the row threshold of the second rule.
int dma_preferred_shape(int NT, const GemmParams& p) {
    static const int max_rows = dev_int("AMX_DMA_MAX_ROWS", 4096);
    const int shape = dma_tile_shape(NT, p, 1);
    if (!shape) return 0;
    if (p.M < 768) return shape;
    const int pp_tiles = ((p.N + pp::BN - 1) / pp::BN) * ((p.M + 127) / 128);
    // (K = 4096 products stay on the ping-pong kernel: its K chunks beat a 128-sub-step loop per tile -- tools/gemm_bench
    // small, M = 1996: out-proj 40.6 -> 29.5 us on DMA tiles, FFN2 66.1 -> 74.8 us)
    if (p.M < max_rows && p.K <= 2048 && pp_tiles * 2 <= device_cus()) return shape;
    return 0;
}

template <typename T, int NT>
void launch_gemm_t(const GemmParams& p, hipStream_t stream) {
    if (!dma_preferred_shape(NT, p) && launch_gemm_pp<T, NT>(p, stream)) return;
    const int shape = dma_tile_shape(NT, p, 1);  // preferred, or the ping-pong kernel rejected the product
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
        if (splits > 1 && !p.defer_fixup) launch_fixup<T, NT>(p, splits, stream);
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
    if (splits > 1 && !p.defer_fixup) launch_fixup<T, NT>(p, splits, stream);
}

// the K chunks launch_gemm_t will use: the same decisions, without the launches
int planned_splits(int NT, const GemmParams& p) {
    if (!dma_preferred_shape(NT, p) && pp_eligible(NT, p)) {
        int mi, splits, ni;
        pp_plan(NT, p, &mi, &splits, &ni);
        return splits;
    }
    const int shape = dma_tile_shape(NT, p, 1);
    if (shape) {
        const int bm = shape == 2 ? 64 : 128, bn = shape == 2 ? 32 : 64;
        const int tiles = ((p.N + bn - 1) / bn) * ((p.M + bm - 1) / bm);
        return p.K >= 4096 ? choose_splits(p, tiles, device_cus() / 2, BK, 16 * BK) : 1;
    }
    const int BN = p.N <= 64 ? 64 : 128;
    const int tiles = ((p.N + BN - 1) / BN) * ((p.M + 127) / 128);
    return choose_splits(p, tiles, device_cus() / 4, BK, 2 * BK);
}

template <typename T, int NT>
void launch_fixup_rownorm_t(const GemmParams& p, int splits, const float* gamma, const float* beta, float eps, void* out_p,
                            int64_t out_plane, int64_t ldp, float* out_ln, int64_t ldo_ln, hipStream_t stream) {
    dim3 grid((unsigned)((p.M + 3) / 4));
    hipLaunchKernelGGL((splitk_fixup_rownorm_kernel<T, NT>), grid, dim3(256), 0, stream, p, (const float*)p.splitk_ws, splits,
                       (int64_t)p.M * p.N, gamma, beta, eps, (T*)out_p, out_plane, ldp, out_ln, ldo_ln);
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

int gemm_ln_tap_minor_slice(int prec, const GemmParams& p_in) {
    static const bool tap_major = dev_switch("AMX_LN_TAP_MAJOR");  // developer A/B switch
    const int NT = prec_planes(prec);
    const GemmParams p = with_vec_flag(p_in);
    if (tap_major || !ln_eligible(NT, p) || !ln_uses_il(NT, p)) return 0;
    return NT == 2 ? 32 : 64;
}

bool gemm_uses_pp(int prec, const GemmParams& p_in) {
    const GemmParams p = with_vec_flag(p_in);
    const int NT = prec_planes(prec);
    return pp_eligible(NT, p) && !dma_preferred_shape(NT, p);  // the routing of launch_gemm_t
}

bool gemm_ln_fold_ok(int prec, const GemmParams& p_in) {
    const GemmParams p = with_vec_flag(p_in);
    const int NT = prec_planes(prec);
    if (!p.vec_ok || !pp_eligible(NT, p) || dma_preferred_shape(NT, p)) return false;
    int mi, splits, ni;
    pp_plan(NT, p, &mi, &splits, &ni);
    if (splits != 1) return false;
    if (p.ln_partial) {
        // producer: fp32 stream + residual, planes of the new rows, whole 64-column blocks, 256-column tiles
        if (ni != 4 || p.N % 64 || p.N > 2048 || !p.out_p || !p.ln_rowps || p.act || p.mode || p.row_len || p.row_coef) return false;
        // the stream: fp32 rows in and out, or (two planes) the planes themselves as the residual and fp32 out only on request
        if (p.ln_res_planes ? NT != 2 : (!p.out_f32 || !p.residual)) return false;
        if (NT == 2 && (p.out_plane != PLANE_IL || p.ldp % 32)) return false;
        if (p.ldp % 8 || ((uintptr_t)p.out_p & 15) || ((uintptr_t)p.ln_partial & 7) || ((uintptr_t)p.ln_rowps & 15)) return false;
    }
    if (p.row_coef) {
        if (!p.col_c || ((uintptr_t)p.col_c & 15) || ((uintptr_t)p.row_coef & 7) || p.residual || p.row_len) return false;
        if (p.mode != 1 && (!p.out_p || p.out_f32)) return false;
    }
    return true;
}

int gemm_planned_splits(int prec, const GemmParams& p_in) {
    const GemmParams p = with_vec_flag(p_in);
    if (p.ln_gamma) return 1;
    return planned_splits(prec_planes(prec), p);
}

bool fixup_rownorm_eligible(const GemmParams& p) {
    return p.splitk_ws && p.out_f32 && !p.out_p && p.act == 0 && p.mode == 0 && !p.row_len && !p.ln_gamma && p.N % 4 == 0 &&
           p.N <= 1024 && p.ldo % 4 == 0 && (!p.residual || p.ldr % 4 == 0) && !((uintptr_t)p.out_f32 & 15) &&
           !((uintptr_t)p.residual & 15) && !((uintptr_t)p.bias & 15) && p.zout == 0;
}

void launch_fixup_rownorm(int prec, const GemmParams& p, int splits, const float* gamma, const float* beta, float eps, void* out_p,
                          int64_t out_plane, int64_t ldp, float* out_ln, int64_t ldo_ln, hipStream_t stream) {
    switch (prec) {
        case PREC_BF16: launch_fixup_rownorm_t<bf16, 1>(p, splits, gamma, beta, eps, out_p, out_plane, ldp, out_ln, ldo_ln, stream); break;
        case PREC_F16: launch_fixup_rownorm_t<f16, 1>(p, splits, gamma, beta, eps, out_p, out_plane, ldp, out_ln, ldo_ln, stream); break;
        case PREC_BF16X3: launch_fixup_rownorm_t<bf16, 2>(p, splits, gamma, beta, eps, out_p, out_plane, ldp, out_ln, ldo_ln, stream); break;
        default: launch_fixup_rownorm_t<f16, 2>(p, splits, gamma, beta, eps, out_p, out_plane, ldp, out_ln, ldo_ln, stream); break;
    }
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
