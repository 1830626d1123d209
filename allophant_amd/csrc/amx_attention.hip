// Flash-style multi-head self-attention with key-padding mask for gfx950 (head_dim 64; other head dimensions up to 128 on a wider
// instance of the first kernel: NDH), never materialising T x T.
//
// Per workgroup: one (utterance, head) and WAVES x 32 queries.  Q, K and V arrive row-major [N, H, Tp, 64] as 16-bit
// planes from the QKV projection epilogue; Q already carries dh^-0.5 * log2(e), so the softmax runs on v_exp_f32
// (exp2) with no extra multiply.
//   * K / V tiles of 64 keys are DMA-ed straight into a double-buffered LDS ring (buffer_load ... lds, one 1-KiB piece
//     = 8 rows x 128 B per wave-instruction, issued one tile ahead, one s_barrier per tile).  The DMA image is
//     lane-linear, so the bank swizzles live on the per-lane SOURCE address and on the reads (rule 21):
//       K rows: 16-byte chunk ^= (row >> 1) & 7    -> conflict-free ds_read_b128 fragment reads (16 rows, one chunk)
//       V rows: 16-byte chunk ^= 4 * ((row >> 1) & 1) -> conflict-free ds_read_b64_tr_b16 (4 rows x 64 B per half-wave)
//   * scores are computed transposed, S^T = K.Q^T with v_mfma_f32_32x32x16, so a lane owns one query column: the
//     online-softmax statistics are lane-local (one cross-half exchange per tile) and the exponentiated accumulator
//     registers feed O^T = V^T.P directly as the B operand (accumulator-as-operand order, cdna_hip_programming.md
//     section 3); the matching V^T fragments (4 consecutive keys of one d per 64-bit half) come from the row-major V
//     tile through the transposing LDS read ds_read_b64_tr_b16.
//   * online softmax with a deferred maximum: the score accumulators start at -m_run (no subtraction pass), the first
//     tile sets m_run to its maximum, and afterwards m_run (with the O / l rescale) only moves when some query's tile
//     maximum exceeds it by more than 2^8; probabilities then stay <= 256, exact in the hi/lo planes.
//   * softmax on v_max3 trees, packed adds and packed 16-bit converts (v_cvt_pk_*), the hi/lo split of P two values
//     at a time.  (Measured: explicit double-buffering of the LDS fragment reads, with the tile cut into two 32-key
//     halves to stay inside the 128-VGPR budget of 4 waves per SIMD, was slower -- 4 waves per SIMD already hide the
//     LDS latency.)
// Masked keys (t' >= frame_len[n]) get -inf before the softmax; key tiles past the utterance end are skipped (their
// probabilities are exactly 0 in the reference too: finfo.min bias underflows exp to 0).
#include "amx_common.h"
#include <cstdlib>
#include <type_traits>

namespace amx {

namespace {

constexpr int DH = 64;
constexpr float DEFER_THR = 8.0f;  // log2 units

// The value of the lane 32 away, combined with this lane's, in the vector ALU: v_permlane32_swap of a register with a copy of
// itself leaves the lower half's values in one register and the upper half's in the other, on every lane.  __shfl_xor(x, 32)
// is a ds_bpermute, an LDS round trip (> 100 cycles) in the middle of the softmax's dependent chain.  max and + commute: the
// same bits as the shuffle form.
__device__ __forceinline__ void both_halves(float x, float& lower, float& upper) {
    const unsigned b = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane32_swap(b, b, false, false);
    const unsigned r0 = r[0], r1 = r[1];  // (scalars first: __builtin_bit_cast of the ELEMENT r[1] reads element 0 with this clang)
    lower = __builtin_bit_cast(float, r0);
    upper = __builtin_bit_cast(float, r1);
}
// (The max trees stay plain fmaxf.  Written as inline-asm v_max3_f32 they save the canonicalising v_max x, x the compiler puts in
// front of MFMA outputs, but the compiler does not place the MFMA -> VALU-read wait states in front of an asm statement that reads
// an accumulator: the scores of a block would be read on timing luck.)
__device__ __forceinline__ float max_halves(float x) {
    float lo, up;
    both_halves(x, lo, up);
    return fmaxf(lo, up);
}
__device__ __forceinline__ float sum_halves(float x) {
    float lo, up;
    both_halves(x, lo, up);
    return lo + up;
}
// waves per SIMD the register allocation must allow: two 8-wave workgroups or four 4-wave / 32-key workgroups per CU -> 4;
// two 4-wave / 64-key workgroups -> 2
#ifndef AMX_ATTN_OCC
#define AMX_ATTN_OCC ((NDH == 1 && KS == 1 && (WAVES == 8 || KT == 32)) ? 4 : 2)
#endif
// KT = keys per tile (64; a 32-key instance with four 4-wave workgroups per CU was 5 % slower at 32 x 10 s)
// PACKED: the packed-row layout of a ragged batch (AttnParams.row_off), a compile-time variant so that the padded kernel
// keeps its code
// KS = 2 (round 4, short batches): the key tiles of a query block are split over TWO halves of the workgroup -- waves [0, WAVES)
// take the first half of the utterance's key tiles, waves [WAVES, 2 WAVES) the second half, each half with a K / V ring of its
// own -- and the halves' (maximum, row sum, O) are merged through LDS in a fixed order at the end.  When the grid is one
// workgroup per CU or less anyway (4 x 10 s: 256 workgroups of 128 queries), the serial key loop of a wave IS the launch; two
// waves per SIMD halve it.  Every query's result is that of one wave with the same tiles in two groups: not bitwise the KS = 1
// result (different rescale points), same gate.
// TSTORE (round 5): the outputs leave through a per-wave LDS patch as whole lines.  A lane owns a query COLUMN of O^T and four-element
// groups of its 64 values, so direct stores are 8 bytes per lane -- 16 instructions per wave, each touching 32 rows in 16-byte
// pieces: 512 partial-line write requests for 8 KiB.  The patch (the idle K / V ring behind the last tile barrier; 32 rows x 256
// bytes per wave, 16-byte chunks XOR-swizzled by the row) is read back row-major: 8 stores of 16 bytes per lane, whole 128-byte
// lines.  Same values, same addresses: bitwise the direct form.
// NDH (round 6): head dimensions other than 64.  Q / K / V rows are DHP = 64 NDH elements wide -- AttnParams.dh valid columns, the
// rest zero (the QKV scatter never writes them, the buffers are zero-filled): the zero columns add nothing to Q.K^T, and the
// columns of O beyond dh are simply not stored.  NDH = 1 serves dh <= 64, NDH = 2 dh in (64, 128] (XLS-R 1B / 2B: 80 / 120): a K / V
// tile is then two [KT x 64] sub-tiles side by side in LDS, each in the layout of the dh = 64 kernel (same swizzles, same fragment
// addresses), the score chain runs over both and O holds four 32-column blocks -- 64 more fragment and 32 more accumulator
// registers, hence two waves per SIMD.  dh != 64 stores through the masked direct path (TSTORE's line patches assume 64 columns).
// KQN: 16-column steps of the score chain that hold real columns (ceil(dh / 16); default: the whole padded row) and, with them, the
// 32-column blocks of O that are kept ((KQN + 1) / 2).  The steps and blocks beyond are products with the zero padding -- leaving
// them out changes no bit (x + 0 in the same order) and, at dh = 80 in a 128-wide row (XLS-R 1B), saves 3 of 8 score steps and 1 of
// 4 output blocks.
template <typename T, int NT, int WAVES, int KT, bool PACKED, int KS = 1, bool TSTORE = true, int NDH = 1, int KQN = 4 * NDH>
__global__ __launch_bounds__(WAVES * KS * 64, AMX_ATTN_OCC) void attn_kernel(const AttnParams p) {
    static_assert(NDH == 1 || (KS == 1 && !TSTORE), "the wide form has neither the key split nor the line-patch stores");
    static_assert(KQN >= 1 && KQN <= 4 * NDH && (KQN == 4 * NDH || !TSTORE), "real 16-column steps of the padded row");
    constexpr int OBN = (KQN + 1) / 2;  // 32-column blocks of O that hold real columns
    constexpr int DHP = 64 * NDH;      // padded head dimension = elements per Q / K / V row
    constexpr int ROWB = DHP * 2;      // bytes per row of one plane
    constexpr int SUB = KT * 128;      // bytes of one [KT x 64] sub-tile of one plane
    constexpr int TILE = SUB * NDH;    // bytes of one K or V tile of one plane
    constexpr int NC = KT / 32;        // 32-key blocks per tile
    typedef typename Vec8<T>::type V8;
    typedef typename Vec4<T>::type V4;
    typedef short s16x4 __attribute__((__vector_size__(8)));
    typedef __attribute__((address_space(3))) s16x4* lds_s4_t;
    constexpr int QB = WAVES * 32;
    constexpr int STAGE = NT * 2 * TILE;  // [plane][K tile | V tile]
    constexpr int PPW = (KT / 8) / WAVES;  // DMA pieces (8 rows x 128 B) per wave, per tile and plane (K and V each)
    static_assert(PPW >= 1 && PPW * WAVES * 8 == KT, "whole DMA pieces per wave");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = KS == 1 ? wave_all : wave_all % WAVES;  // position among the query waves
    const int kh = KS == 1 ? 0 : wave_all / WAVES;           // key half of this wave
    unsigned char* const ring = smem + kh * 2 * STAGE;
    const int hh = lane >> 5, lq = lane & 31;
    // XCD-aware order (1-D grid): workgroups are dealt round-robin over the 8 XCDs, so the query blocks of one (utterance,
    // head) are given consecutive slots of ONE XCD: they run side by side and the K / V tiles one of them pulls into that
    // XCD's L2 serve the others (with the plain (query block, head) grid every query block re-read K and V from HBM, and
    // the kernel was bound by that traffic).  Pure speed: any placement gives the same results.
    const int qblocks = (p.T + QB - 1) / QB;
    const int slot = blockIdx.x >> 3;
    const int nh = (slot / qblocks) * 8 + (blockIdx.x & 7);
    const int qblock = slot % qblocks;
    if (nh >= p.N * p.H) return;
    // a ragged batch is walked longest utterance first, so that the long workgroups do not start last
    const int h = nh % p.H;
    const int n = PACKED && p.order ? p.order[nh / p.H] : nh / p.H;
    const int q_base = qblock * QB + wave * 32;
    const int query = q_base + lq;
    int klen = p.frame_len[n];
    klen = klen < 1 ? 1 : (klen > p.T ? p.T : klen);
    const int nkt = (klen + KT - 1) / KT;
    // KS == 2: this wave's share of the key tiles -- tiles kt0 .. kt0 + my_tiles - 1; both halves run half_tiles loop steps
    const int half_tiles = (nkt + KS - 1) / KS;
    const int kt0 = kh * half_tiles;
    const int my_tiles = nkt - kt0 < 0 ? 0 : (nkt - kt0 < half_tiles ? nkt - kt0 : half_tiles);
    // packed rows: the utterance starts at row row_off[n] of every head's [Tp, 64] block; query blocks past its end do
    // not exist (in the padded layout they are computed like the reference computes them: the rows feed later kernels)
    constexpr bool packed = PACKED;
    const int roff = packed ? p.row_off[n] : 0;
    if (packed && qblock * QB >= klen) return;
    const int64_t first = packed ? ((int64_t)h * p.Tp + roff) * DHP : (int64_t)nh * p.Tp * DHP;  // element offset of row 0
    const int q_rows = packed ? p.Tp - roff : p.Tp;  // rows that may be read from `first` on

    const T* Qb = (const T*)p.q + first;
    const dma_rsrc_t k_rsrc = dma_rsrc((const T*)p.k + first), v_rsrc = dma_rsrc((const T*)p.v + first);
    const uint32_t plane_b = (uint32_t)(p.qk_plane * 2);

    // Q fragments (B operand): lane (query, hh) holds Q[query][16ks + 8hh + j]
    V8 qf[NT][KQN];
    {
        const int qr = query < q_rows ? query : q_rows - 1;
#pragma unroll
        for (int pl = 0; pl < NT; ++pl)
#pragma unroll
            for (int ks = 0; ks < KQN; ++ks)
                qf[pl][ks] = *(const V8*)(Qb + (int64_t)pl * p.qk_plane + (int64_t)qr * DHP + ks * 16 + 8 * hh);
    }

    // ---- DMA: this wave moves pieces wave, wave + WAVES, ... (8 rows x 128 B) of every K and V tile plane ----
    // piece parity == wave parity (WAVES is even), so the swizzled source chunk is a per-lane constant
    // (a piece is 8 rows x 128 B of ONE sub-tile: its rows are ROWB bytes apart in memory, its 128 bytes at dhh * 128 into the row)
    const uint32_t voff_k = (uint32_t)((lane >> 3) * ROWB + (((lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7)) << 4));
    const uint32_t voff_v = (uint32_t)((lane >> 3) * ROWB + (((lane & 7) ^ (4 * ((lane >> 4) & 1))) << 4));
    auto stage = [&](int kt, int st) {
#pragma unroll
        for (int pl = 0; pl < NT; ++pl)
#pragma unroll
            for (int dhh = 0; dhh < NDH; ++dhh)
#pragma unroll
                for (int j = 0; j < PPW; ++j) {
                    const int piece = wave + WAVES * j;
                    const uint32_t so = (uint32_t)kt * (KT * ROWB) + pl * plane_b + piece * (8 * ROWB) + dhh * 128;
                    unsigned char* dst = ring + st * STAGE + pl * 2 * TILE + dhh * SUB + piece * 1024;
                    dma16(k_rsrc, dst, voff_k, so);  // (inline asm, not the builtin: amx_common.h)
                    dma16(v_rsrc, dst + TILE, voff_v, so);
                }
    };

    // ---- LDS read addresses ----
    // K fragment (c, ks): row 32c + lq, chunk (2ks + hh) ^ ((lq >> 1) & 7)
    // (2ks + hh) ^ s == (hh ^ s) ^ 2ks: one base register, the other three by a constant XOR at the use
    const int kaddr0 = lq * 128 + ((hh ^ ((lq >> 1) & 7)) << 4);
    // V^T fragment half (c, s, g, dt): 16-lane group = 16 d columns x 4 keys; lane 4q + pp of the group addresses key row
    // 32c + 16s + 8g + 4hh + q, columns 32dt + 16*((lane >> 4) & 1) + 4pp .. +3   (chunk bit 2 swizzled by (q >> 1) & 1)
    int vaddr[2];
    {
        const int q4 = (lane & 15) >> 2, pp = lane & 3;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
            vaddr[dt] = TILE + (4 * hh + q4) * 128 + 64 * (dt ^ ((q4 >> 1) & 1)) + 32 * ((lane >> 4) & 1) + 8 * pp;
    }

    f32x16 O[OBN];
#pragma unroll
    for (int b = 0; b < OBN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) O[b][r] = 0.f;
    float m_run = 0.f, l_run = 0.f;  // scores are kept relative to m_run; the first tile sets it

    if (my_tiles > 0) stage(kt0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // the Q fragments have arrived (the wait above); a use the compiler can see, so that ITS wait for those loads sits here and not
    // at their first use inside the key loop -- it does not see the DMA transfers (inline asm), so a "vmcnt(0) for the Q loads" in
    // the loop would wait for every tile in flight, every iteration
#pragma unroll
    for (int pl = 0; pl < NT; ++pl)
#pragma unroll
        for (int ks = 0; ks < KQN; ++ks) asm volatile("" ::"v"(qf[pl][ks]));

#ifdef AMX_ATTN_STAMP
    // developer diagnostic (tools/attn_bench.hip): cycles per phase of a key tile, summed in scalar registers
    unsigned long long st_s = 0, st_max = 0, st_exp = 0, st_pv = 0, st_wait = 0, st_bar = 0;
    auto stamp = []() -> unsigned long long {
        unsigned long long t;
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        __builtin_amdgcn_sched_barrier(0);
        return t;
    };
    unsigned long long st_rt0;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_rt0)::"memory");
    const unsigned long long st_begin = stamp();
#define ATTN_STAMP(var, prev) { const unsigned long long now_ = stamp(); var += now_ - prev; prev = now_; }
#else
#define ATTN_STAMP(var, prev)
#endif
    // `i`: loop step of this wave's half, tile kt0 + i (a half with fewer tiles than the other idles through its last step)
    auto tile = [&](int i, auto stc) {
        constexpr int ST = decltype(stc)::value;
        const int kt = kt0 + i;
#ifdef AMX_ATTN_STAMP
        unsigned long long st_prev = stamp();
#endif
        if (i + 1 < my_tiles) stage(kt + 1, ST ^ 1);
        const unsigned char* sb = ring + ST * STAGE;
        if (KS == 1 || i < my_tiles) {

        // ---- S^T = K . Q^T : X[c][r] = score(key = kb + 32c + (r&3) + 8(r>>2) + 4hh, query), log2 units ----
        f32x16 X[NC];
        const float neg_m = -m_run;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
#pragma unroll
            for (int r = 0; r < 16; ++r) X[c][r] = neg_m;
#pragma unroll
            for (int kq = 0; kq < KQN; ++kq) {
                const int ks = kq & 3, dsub = (kq >> 2) * SUB;  // K step inside its [KT x 64] sub-tile
#ifdef AMX_ATTN_ABL_NOLDS  // developer ablation (wrong results): one K fragment read per tile instead of 16
                const V8 kf = *(const V8*)(sb + kaddr0);
                if (NT > 1) {
                    const V8 kl = *(const V8*)(sb + 2 * TILE + kaddr0);
#else
                const V8 kf = *(const V8*)(sb + dsub + c * 4096 + (kaddr0 ^ (ks << 5)));
                if (NT > 1) {
                    const V8 kl = *(const V8*)(sb + 2 * TILE + dsub + c * 4096 + (kaddr0 ^ (ks << 5)));
#endif
#ifndef AMX_ATTN_ABL_NOCROSS  // developer ablation (results lose the lo planes): a third of the MFMAs, same loads
                    X[c] = mfma32(kl, qf[0][kq], X[c]);
                    X[c] = mfma32(kf, qf[NT - 1][kq], X[c]);
#else
                    asm volatile("" ::"v"(kl));
#endif
                }
                X[c] = mfma32(kf, qf[0][kq], X[c]);
            }
        }
        ATTN_STAMP(st_s, st_prev)
        const int kb = kt * KT;
        if (kb + KT > klen) {  // only the last tile holds masked keys (wave-uniform branch)
            const int rem = klen - kb - 4 * hh;  // keys of this lane's rows left in the utterance
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) X[c][r] = 32 * c + (r & 3) + 8 * (r >> 2) < rem ? X[c][r] : -INFINITY;
        }
        // scores are relative to the running maximum (the accumulators start at -m_run): tile maximum by v_max3 trees
        float mx;
        if constexpr (NC == 2) {
            float t0 = fmaxf(fmaxf(X[0][0], X[0][1]), X[0][2]), t1 = fmaxf(fmaxf(X[1][0], X[1][1]), X[1][2]);
#pragma unroll
            for (int r = 3; r + 1 < 16; r += 2) {
                t0 = fmaxf(fmaxf(t0, X[0][r]), X[0][r + 1]);
                t1 = fmaxf(fmaxf(t1, X[1][r]), X[1][r + 1]);
            }
            mx = fmaxf(fmaxf(t0, t1), fmaxf(X[0][15], X[NC - 1][15]));
        } else {
            float t0 = fmaxf(fmaxf(X[0][0], X[0][1]), X[0][2]), t1 = fmaxf(fmaxf(X[0][3], X[0][4]), X[0][5]);
#pragma unroll
            for (int r = 6; r + 3 < 16; r += 4) {
                t0 = fmaxf(fmaxf(t0, X[0][r]), X[0][r + 1]);
                t1 = fmaxf(fmaxf(t1, X[0][r + 2]), X[0][r + 3]);
            }
            mx = fmaxf(fmaxf(t0, t1), fmaxf(X[0][14], X[0][15]));
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        // wave-uniform; both lane halves of a query see the same mx.  The first tile always takes its own maximum (m_run
        // starts at 0, not at the scores' level); afterwards the maximum only moves when a tile exceeds it by 2^THR.
        if (i == 0 || !__all(mx <= DEFER_THR)) {
            const float d = i == 0 ? mx : fmaxf(mx, 0.f);
            const float alpha = __builtin_amdgcn_exp2f(-d);
            m_run += d;
            l_run *= alpha;
#pragma unroll
            for (int b = 0; b < OBN; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) O[b][r] *= alpha;
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) X[c][r] -= d;
        }
        ATTN_STAMP(st_max, st_prev)
        f32x2 ps = {0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
#ifdef AMX_ATTN_ABL_NOEXP  // developer ablation (wrong results): no transcendental
                const f32x2 e = {X[c][r] * X[c][r], X[c][r + 1] * X[c][r + 1]};
#else
                const f32x2 e = {__builtin_amdgcn_exp2f(X[c][r]), __builtin_amdgcn_exp2f(X[c][r + 1])};
#endif
                X[c][r] = e[0];
                X[c][r + 1] = e[1];
                ps += e;  // v_pk_add_f32
            }
        l_run += ps[0] + ps[1];
        ATTN_STAMP(st_exp, st_prev)

        // ---- O^T += V^T . P : P's accumulator registers are the B operand ----
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                // hi/lo planes of 8 probabilities, two at a time on the packed converts (exp2 results are plain register
                // values, so the single-value pinning of split16 is not needed here)
                typedef typename Vec2<T>::type V2;
                union { V2 h[4]; V8 v; } ph, pl_;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x2 x = {X[c][8 * s2 + 2 * j], X[c][8 * s2 + 2 * j + 1]};
                    const V2 hi = __builtin_convertvector(x, V2);
                    ph.h[j] = hi;
                    if (NT > 1) {
                        const f32x2 back = {(float)hi[0], (float)hi[1]};
                        pl_.h[j] = __builtin_convertvector(x - back, V2);
                    }
                }
#ifdef AMX_ATTN_ABL_NOLDS
                const int koff = 0;
#else
                const int koff = c * 4096 + s2 * 2048;  // key rows 32c + 16s2 (+ 8g)
#endif
#pragma unroll
                for (int ob = 0; ob < OBN; ++ob) {
                    const int dt = ob & 1, vsub = (ob >> 1) * SUB + koff;  // 32-column block `dt` of V sub-tile ob / 2
                    union { s16x4 h[2]; V8 v; } vf, vl;
                    vf.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t)(sb + vsub + vaddr[dt]));
                    vf.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t)(sb + vsub + 1024 + vaddr[dt]));
                    if (NT > 1) {
                        vl.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t)(sb + 2 * TILE + vsub + vaddr[dt]));
                        vl.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t)(sb + 2 * TILE + vsub + 1024 + vaddr[dt]));
#ifndef AMX_ATTN_ABL_NOCROSS
                        O[ob] = mfma32(vl.v, ph.v, O[ob]);
                        O[ob] = mfma32(vf.v, pl_.v, O[ob]);
#else
                        asm volatile("" ::"v"(vl.v), "v"(pl_.v));
#endif
                    }
                    O[ob] = mfma32(vf.v, ph.v, O[ob]);
                }
            }
        // tile kt+1 has landed (this wave's pieces) and this wave's LDS reads of tile kt have RETURNED: the first thing any
        // wave does after the barrier is to DMA tile kt+2 over tile kt, so a read still in flight at the barrier could see
        // the new tile (cdna_hip_programming.md, "restage a buffer one phase after its last ds_read only when an lgkmcnt
        // before the barrier retired those reads").  Without the lgkmcnt(0) hipcc leaves the V reads of the last P.V group
        // in flight across the barrier: one 32-query block in ~1000 launches came out with a few keys of the wrong tile
        // (4 x 60 s batches, tools/stress_repro.py).
        ATTN_STAMP(st_pv, st_prev)
        }  // (KS == 2: a step beyond this half's tiles only keeps the barrier count)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        ATTN_STAMP(st_wait, st_prev)
        __builtin_amdgcn_s_barrier();
        ATTN_STAMP(st_bar, st_prev)
    };

    for (int i = 0; i < half_tiles; i += 2) {
        tile(i, std::integral_constant<int, 0>{});
        if (i + 1 < half_tiles) tile(i + 1, std::integral_constant<int, 1>{});
    }
    if constexpr (KS == 2) {
        // merge the two key halves: the second half hands (m_run, l_run, O) over through LDS (the ring is idle: every wave is
        // behind the last tile barrier), the first half folds them in -- max of the maxima, both sides rescaled to it
        float* patch = (float*)(smem + wave * (34 * 64 * 4));
        if (kh == 1) {
            patch[lane] = my_tiles > 0 ? m_run : -INFINITY;  // a half without tiles contributes nothing
            patch[64 + lane] = l_run;
#pragma unroll
            for (int r = 0; r < 16; ++r) { patch[(2 + r) * 64 + lane] = O[0][r]; patch[(18 + r) * 64 + lane] = O[1][r]; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kh == 1) return;
        const float m1 = patch[lane], l1 = patch[64 + lane];
        const float m = fmaxf(m_run, m1);
        const float a0 = __builtin_amdgcn_exp2f(m_run - m), a1 = __builtin_amdgcn_exp2f(m1 - m);
        l_run = l_run * a0 + l1 * a1;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            O[0][r] = O[0][r] * a0 + patch[(2 + r) * 64 + lane] * a1;
            O[1][r] = O[1][r] * a0 + patch[(18 + r) * 64 + lane] * a1;
        }
    }

#ifdef AMX_ATTN_STAMP
    const unsigned long long st_loop_end = stamp();
#endif
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    if (NDH > 1 || p.dh != DH) {
        // head dimensions other than 64: only the dh real columns of the padded O leave, four at a time (dh % 8 == 0: a quad lies
        // wholly inside or wholly outside), each quad mapped through pidx() -- heads need not start on a 32-column block
        if (query < (packed ? klen : p.T)) {
            const bool o_il = plane_is_il<NT>(p.out_plane);
            const int64_t col0 = ((packed ? (int64_t)roff : (int64_t)n * p.T) + query) * ((int64_t)p.H * p.dh) + (int64_t)h * p.dh;
#pragma unroll
            for (int ob = 0; ob < OBN; ++ob)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int d0 = ob * 32 + 8 * g + 4 * hh;
                    if (d0 < p.dh) {
                        V4 hv, lv;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            T hi, lo = (T)0.f;
                            split16<T, NT>(O[ob][4 * g + j] * inv, hi, lo);
                            hv[j] = hi;
                            lv[j] = lo;
                        }
                        T* dst = (T*)p.out + pidx(col0 + d0, o_il);
                        *(V4*)dst = hv;
                        if (NT > 1) *(V4*)(dst + p.out_plane) = lv;
                    }
                }
        }
#ifdef AMX_ATTN_STAMP
        goto attn_stamp_out;
#else
        return;
#endif
    }
    if constexpr (TSTORE && NDH == 1) {
        const bool o_il = plane_is_il<NT>(p.out_plane);
        if (NT == 1 || o_il) {  // (separate hi / lo planes -- non-wav2vec 2.0 widths -- keep the direct stores below)
            constexpr int ROW = NT == 2 ? 256 : 128, CH = ROW / 16;  // bytes / 16-byte chunks of one query's head segment
            unsigned char* patch = smem + (KS == 2 ? 40 * 1024 : 0) + wave * (32 * ROW);
            const int sw = lq & (CH - 1);
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
                    // byte offset of the hi quad inside the segment: interleaved planes [hi x 32 | lo x 32] per 32 values
                    const int o = NT == 2 ? dt * 128 + (8 * g + 4 * hh) * 2 : (dt * 32 + 8 * g + 4 * hh) * 2;
                    *(V4*)(patch + lq * ROW + ((((o >> 4) ^ sw) << 4) | (o & 8))) = hv;
                    if (NT > 1) *(V4*)(patch + lq * ROW + (((((o >> 4) + 4) ^ sw) << 4) | (o & 8))) = lv;
                }
            // (LDS operations of one wave complete in order: the reads below see the writes above)
            const int limit = packed ? klen : p.T;
            const int64_t row0 = packed ? (int64_t)roff : (int64_t)n * p.T;
#pragma unroll
            for (int q = 0; q < 32 * CH / 64; ++q) {
                const int id = q * 64 + lane, row = id / CH, c = id % CH;
                const uint4 v = *(const uint4*)(patch + row * ROW + ((c ^ (row & (CH - 1))) << 4));
                const int qr = q_base + row;
                if (qr < limit) {
                    T* dst = (T*)p.out + pidx((row0 + qr) * (p.H * DH) + h * DH, o_il);
                    *(uint4*)((unsigned char*)dst + c * 16) = v;
                }
            }
#ifdef AMX_ATTN_STAMP
            goto attn_stamp_out;
#else
            return;
#endif
        }
    }
    if (NDH == 1 && query < (packed ? klen : p.T)) {
        // (the output feeds the out-projection GEMM: interleaved planes in the two-plane modes, amx_common.h pidx())
        const bool o_il = plane_is_il<NT>(p.out_plane);
        T* dst = (T*)p.out + pidx(((packed ? (int64_t)roff : (int64_t)n * p.T) + query) * (p.H * DH) + h * DH, o_il);
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
                const int d0 = (dt * 32 << (o_il ? 1 : 0)) + 8 * g + 4 * hh;
                *(V4*)(dst + d0) = hv;
                if (NT > 1) *(V4*)(dst + p.out_plane + d0) = lv;
            }
    }
#ifdef AMX_ATTN_STAMP
attn_stamp_out:
    if (p.stamps && lane == 0) {
        const unsigned long long st_end = stamp();
        unsigned long long st_rt1;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_rt1)::"memory");
        unsigned long long* o = p.stamps + ((int64_t)blockIdx.x * WAVES + wave) * 12;
        o[0] = st_s; o[1] = st_max; o[2] = st_exp; o[3] = st_pv; o[4] = st_wait; o[5] = st_bar;
        o[6] = st_loop_end - st_begin; o[7] = st_end - st_loop_end; o[8] = (unsigned long long)nkt; o[9] = 1;
        o[10] = st_rt0; o[11] = st_rt1;  // 100 MHz wall clock at start / end of the wave
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// attn2_kernel (round 4): the same arithmetic on 64 queries per wave, for long key loops.
//
// attn_kernel gives a wave 32 queries and runs four waves per SIMD: every wave reads the whole K and V tile from LDS for its
// 32 queries.  Here a wave owns TWO 32-query sub-blocks: every K fragment and every V^T fragment it reads from LDS feeds both
// (half the LDS read bytes per query), and the DMA pieces a wave issues per tile serve twice as many queries.  Two waves per
// SIMD (<= 256 VGPRs: 64 of Q fragments, 64 of accumulators, 32 of scores).  The online softmax steps in 32-key blocks (one
// score accumulator set per sub-block) with the same deferred maximum, so the results are not bitwise those of attn_kernel;
// both are gated against the oracle.  Measured per launch, f16x3 (tools/attn_bench.hip, profiles/r04_attention_experiments.log):
//   8 x 60 s (47 key tiles per item)   attn_kernel 849-870 us   <4 waves, 2 slots> 752   <8 waves, 4 slots, persistent> 790
//   32 x 10 s (8 key tiles per item)   attn_kernel 114-122 us   <4 waves, 2 slots> 136   <8 waves, 4 slots, persistent> 146
// i.e. it pays from ~15 key tiles per item (launch_attn_any), where the longer prologue (Q fragments of 64 queries) and the
// output stores of an item are small beside its key loop.  What its anatomy shows (cycle stamps, same log): the two waves of a
// SIMD run in lockstep -- score MFMAs together (pipe-bound), exponentials together (VALU-bound), P.V together -- so matrix
// and vector work never overlap: 10.2 k cycles per 64-key tile = 6.1 k MFMA + 4.2 k VALU.  Three attempts to overlap them
// were built and measured slower, all for the same reason (the 256-VGPR budget): wave groups in anti-phase with a barrier
// per segment (tools/experiments/attn3_pingpong.inc: a wave alone in its MFMA segment exposes the LDS fragment latency, and
// prefetching fragments into registers spills), and the two sub-blocks of a wave pipelined against each other (sub-block 1
// re-reads the fragments: 24-30 spilled registers, 965 us).
// WAVES = 8, STAGES = 4: one 512-query workgroup per CU (ring 128 KiB), tiles three ahead.  WAVES = 4, STAGES = 2: two independent
// 256-query workgroups per CU (64 KiB each, one tile ahead) -- their phases drift apart, so one's prologue / epilogue sits under
// the other's key loop; the form for short key loops (10 s utterances: 8 tiles per item).
template <typename T, int NT, bool PACKED, int WAVES, int STAGES>
__global__ __launch_bounds__(WAVES * 64, WAVES == 8 ? 1 : 2) void attn2_kernel(const AttnParams p, int total_items) {
    constexpr int KT = 64, SUB = 2;
    constexpr int TILE = KT * 128;        // bytes of one K or V tile of one plane
    constexpr int STAGE = NT * 2 * TILE;  // [plane][K tile | V tile]
    constexpr int QB = WAVES * 32 * SUB;  // queries per item
    constexpr int PPW = 8 / WAVES;        // DMA pieces (8 rows x 128 B) per wave, tile, plane and operand
    static_assert(STAGES == 2 || STAGES == 4, "ring depth");
    typedef typename Vec8<T>::type V8;
    typedef typename Vec4<T>::type V4;
    typedef typename Vec2<T>::type V2;
    typedef short s16x4 __attribute__((__vector_size__(8)));
    typedef __attribute__((address_space(3))) s16x4* lds_s4_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hh = lane >> 5, lq = lane & 31;
    const int qblocks = (p.T + QB - 1) / QB;
    const uint32_t plane_b = (uint32_t)(p.qk_plane * 2);

    // per-lane DMA source offsets and LDS read addresses: those of attn_kernel (WAVES == 8: one piece per wave, tile and plane)
    const uint32_t voff_k = (uint32_t)((lane >> 3) * 128 + (((lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7)) << 4));
    const uint32_t voff_v = (uint32_t)((lane >> 3) * 128 + (((lane & 7) ^ (4 * ((lane >> 4) & 1))) << 4));
    const int kaddr0 = lq * 128 + ((hh ^ ((lq >> 1) & 7)) << 4);
    int vaddr[2];
    {
        const int q4 = (lane & 15) >> 2, pp = lane & 3;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
            vaddr[dt] = TILE + (4 * hh + q4) * 128 + 64 * (dt ^ ((q4 >> 1) & 1)) + 32 * ((lane >> 4) & 1) + 8 * pp;
    }

    // item -> (utterance, head, query block): consecutive slots of one XCD are the query blocks of one (utterance, head)
    struct Item { int n, h, qblock, klen, roff, nkt; int64_t first; bool live; };
    auto decode = [&](int item) {
        Item it;
        const int slot = item >> 3;
        const int nh = (slot / qblocks) * 8 + (item & 7);
        it.qblock = slot % qblocks;
        it.live = item < total_items && nh < p.N * p.H;
        const int nhc = it.live ? nh : 0;
        it.h = nhc % p.H;
        it.n = PACKED && p.order ? p.order[nhc / p.H] : nhc / p.H;
        int klen = p.frame_len[it.n];
        klen = klen < 1 ? 1 : (klen > p.T ? p.T : klen);
        it.klen = klen;
        it.nkt = (klen + KT - 1) / KT;
        it.roff = PACKED ? p.row_off[it.n] : 0;
        if (PACKED && it.qblock * QB >= klen) it.live = false;  // query blocks past a packed utterance's end do not exist
        it.first = PACKED ? ((int64_t)it.h * p.Tp + it.roff) * DH : (int64_t)nhc * p.Tp * DH;
        return it;
    };
    auto stage = [&](const Item& it, int kt) {
        const dma_rsrc_t k_rsrc = dma_rsrc((const T*)p.k + it.first), v_rsrc = dma_rsrc((const T*)p.v + it.first);
#pragma unroll
        for (int pl = 0; pl < NT; ++pl)
#pragma unroll
            for (int j = 0; j < PPW; ++j) {
                const int piece = wave + WAVES * j;  // piece parity == wave parity: the swizzled source chunk is a per-lane constant
                const uint32_t so = (uint32_t)kt * TILE + pl * plane_b + piece * 1024;
                unsigned char* dst = smem + (kt & (STAGES - 1)) * STAGE + pl * 2 * TILE + piece * 1024;
                dma16(k_rsrc, dst, voff_k, so);  // (inline asm, not the builtin: amx_common.h)
                dma16(v_rsrc, dst + TILE, voff_v, so);
            }
    };
    auto stage_head = [&](const Item& it) {
        if (!it.live) return;
        stage(it, 0);
        if (STAGES > 2) {
            if (it.nkt > 1) stage(it, 1);
            if (it.nkt > 2) stage(it, 2);
        }
    };

    Item cur = decode(blockIdx.x);
    stage_head(cur);
    for (int item = blockIdx.x; item < total_items; item += gridDim.x) {
        const Item nxt = decode(item + gridDim.x);
        if (!cur.live) {  // (wave-uniform: nothing was staged for it)
            cur = nxt;
            stage_head(cur);
            continue;
        }
        const int klen = cur.klen, nkt = cur.nkt;
        const int q_rows = PACKED ? p.Tp - cur.roff : p.Tp;
        const T* Qb = (const T*)p.q + cur.first;
        const int q_base = cur.qblock * QB + wave * 32 * SUB;
#ifdef AMX_ATTN_STAMP
        // developer diagnostic (tools/attn_bench.hip): wall clock (100 MHz) at the start of the item, of its key loop, at the end of
        // the loop and of the item; shader cycles of the loop
        unsigned long long a2_t0, a2_t1, a2_t2, a2_t3, a2_c0, a2_c1;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(a2_t0)::"memory");
        // cycles per phase: [0] DMA issue, [1] scores (K reads + MFMA issue), [2] mask / max / exp, [3] P split + V reads + P.V issue,
        // [4] wait for the tile, [5] barrier
        unsigned long long a2_ph[6] = {0, 0, 0, 0, 0, 0}, a2_prev = 0;
#define A2_STAMP(var) { unsigned long long now_; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory"); __builtin_amdgcn_sched_barrier(0); var += now_ - a2_prev; a2_prev = now_; }
#else
#define A2_STAMP(var)
#endif

        // Q fragments (B operand) of both sub-blocks: lane (query, hh) holds Q[query][16 ks + 8 hh + j]
        V8 qf[SUB][NT][4];
#pragma unroll
        for (int sb = 0; sb < SUB; ++sb) {
            const int query = q_base + 32 * sb + lq;
            const int qr = query < q_rows ? query : q_rows - 1;
#pragma unroll
            for (int pl = 0; pl < NT; ++pl)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
                    qf[sb][pl][ks] = *(const V8*)(Qb + (int64_t)pl * p.qk_plane + (int64_t)qr * DH + ks * 16 + 8 * hh);
        }
        f32x16 O[SUB][2];
        float m_run[SUB], l_run[SUB];
#pragma unroll
        for (int sb = 0; sb < SUB; ++sb) {
            m_run[sb] = 0.f;
            l_run[sb] = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { O[sb][0][r] = 0.f; O[sb][1][r] = 0.f; }
        }
        bool first_block = true;
        // a use of the Q fragments the compiler can see, in front of the key loop: its wait for those loads sits here (see attn_kernel)
        // and not at their first use inside the loop, where "vmcnt(0)" would also wait for the tiles in flight, every iteration
#pragma unroll
        for (int sb = 0; sb < SUB; ++sb)
#pragma unroll
            for (int pl = 0; pl < NT; ++pl)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) asm volatile("" ::"v"(qf[sb][pl][ks]));
#ifdef AMX_ATTN_STAMP
        asm volatile("s_waitcnt vmcnt(0)\n\ts_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(a2_t1), "=s"(a2_c0)::"memory");
        a2_prev = a2_c0;
#endif

        for (int kt = 0; kt < nkt; ++kt) {
            // tile kt has landed (this wave's pieces; later tiles may stay in flight), this wave's LDS reads of tile kt - 1 have
            // returned; behind the barrier that holds for every wave, so slot (kt + 3) & 3 == (kt - 1) & 3 may be refilled
            const int ahead = nkt - 1 - kt;
            if (STAGES == 4 && ahead >= 2) {  // (WAVES == 8: NT * 2 DMA instructions per wave and tile)
                if (NT == 2) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            } else if (STAGES == 4 && ahead == 1) {
                if (NT == 2) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            }
            A2_STAMP(a2_ph[4])
            __builtin_amdgcn_s_barrier();
            A2_STAMP(a2_ph[5])
            if (kt + STAGES - 1 < nkt) stage(cur, kt + STAGES - 1);
            A2_STAMP(a2_ph[0])
            const unsigned char* sbuf = smem + (kt & (STAGES - 1)) * STAGE;
            const int kb = kt * KT;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if (kb + 32 * c >= klen) break;  // a 32-key block wholly beyond the utterance (wave-uniform)
                // ---- S^T = K . Q^T for both sub-blocks: X[sb][r] = score(key kb + 32c + (r&3) + 8(r>>2) + 4hh, query) ----
                f32x16 X[SUB];
#pragma unroll
                for (int sb = 0; sb < SUB; ++sb) {
                    const float neg_m = -m_run[sb];
#pragma unroll
                    for (int r = 0; r < 16; ++r) X[sb][r] = neg_m;
                }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const V8 kf = *(const V8*)(sbuf + c * 4096 + (kaddr0 ^ (ks << 5)));
                    if (NT > 1) {
                        const V8 kl = *(const V8*)(sbuf + 2 * TILE + c * 4096 + (kaddr0 ^ (ks << 5)));
#pragma unroll
                        for (int sb = 0; sb < SUB; ++sb) {
                            X[sb] = mfma32(kl, qf[sb][0][ks], X[sb]);
                            X[sb] = mfma32(kf, qf[sb][NT - 1][ks], X[sb]);
                        }
                    }
#pragma unroll
                    for (int sb = 0; sb < SUB; ++sb) X[sb] = mfma32(kf, qf[sb][0][ks], X[sb]);
                }
                A2_STAMP(a2_ph[1])
                const bool tail = kb + 32 * c + 32 > klen;  // the block holds masked keys (wave-uniform)
                const int rem = klen - kb - 32 * c - 4 * hh;
#pragma unroll
                for (int sb = 0; sb < SUB; ++sb) {
                    if (tail) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) X[sb][r] = (r & 3) + 8 * (r >> 2) < rem ? X[sb][r] : -INFINITY;
                    }
                    float t0 = fmaxf(fmaxf(X[sb][0], X[sb][1]), X[sb][2]), t1 = fmaxf(fmaxf(X[sb][3], X[sb][4]), X[sb][5]);
#pragma unroll
                    for (int r = 6; r + 3 < 16; r += 4) {
                        t0 = fmaxf(fmaxf(t0, X[sb][r]), X[sb][r + 1]);
                        t1 = fmaxf(fmaxf(t1, X[sb][r + 2]), X[sb][r + 3]);
                    }
                    const float mx = max_halves(fmaxf(fmaxf(t0, t1), fmaxf(X[sb][14], X[sb][15])));
                    // the first block takes its own maximum; afterwards the maximum only moves when a block exceeds it by 2^THR
                    if (first_block || !__all(mx <= DEFER_THR)) {
                        const float d = first_block ? mx : fmaxf(mx, 0.f);
                        const float alpha = __builtin_amdgcn_exp2f(-d);
                        m_run[sb] += d;
                        l_run[sb] *= alpha;
#pragma unroll
                        for (int r = 0; r < 16; ++r) { O[sb][0][r] *= alpha; O[sb][1][r] *= alpha; X[sb][r] -= d; }
                    }
                    f32x2 ps = {0.f, 0.f};
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const f32x2 e = {__builtin_amdgcn_exp2f(X[sb][r]), __builtin_amdgcn_exp2f(X[sb][r + 1])};
                        X[sb][r] = e[0];
                        X[sb][r + 1] = e[1];
                        ps += e;
                    }
                    l_run[sb] += ps[0] + ps[1];
                }
                first_block = false;
                A2_STAMP(a2_ph[2])
                // ---- O^T += V^T . P : every V^T fragment feeds both sub-blocks ----
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    union { V2 h[4]; V8 v; } ph[SUB], pl_[SUB];
#pragma unroll
                    for (int sb = 0; sb < SUB; ++sb)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const f32x2 x = {X[sb][8 * s2 + 2 * j], X[sb][8 * s2 + 2 * j + 1]};
                            const V2 hi = __builtin_convertvector(x, V2);
                            ph[sb].h[j] = hi;
                            if (NT > 1) pl_[sb].h[j] = residual2<T>(x, hi);
                        }
                    const int koff = c * 4096 + s2 * 2048;  // key rows 32c + 16 s2 (+ 8g)
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        union { s16x4 h[2]; V8 v; } vf, vl;
                        vf.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t)(sbuf + koff + vaddr[dt]));
                        vf.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t)(sbuf + koff + 1024 + vaddr[dt]));
                        if (NT > 1) {
                            vl.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t)(sbuf + 2 * TILE + koff + vaddr[dt]));
                            vl.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4_t)(sbuf + 2 * TILE + koff + 1024 + vaddr[dt]));
#pragma unroll
                            for (int sb = 0; sb < SUB; ++sb) {
                                O[sb][dt] = mfma32(vl.v, ph[sb].v, O[sb][dt]);
                                O[sb][dt] = mfma32(vf.v, pl_[sb].v, O[sb][dt]);
                            }
                        }
#pragma unroll
                        for (int sb = 0; sb < SUB; ++sb) O[sb][dt] = mfma32(vf.v, ph[sb].v, O[sb][dt]);
                    }
                }
                A2_STAMP(a2_ph[3])
            }
        }
#ifdef AMX_ATTN_STAMP
        asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(a2_t2), "=s"(a2_c1)::"memory");
#endif
        // every wave is done with the ring before the next item's first tiles overwrite its first slots
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        stage_head(nxt);  // ... in flight under the output stores below

        const bool o_il = plane_is_il<NT>(p.out_plane);
#pragma unroll
        for (int sb = 0; sb < SUB; ++sb) {
            const int query = q_base + 32 * sb + lq;
            const float l_tot = sum_halves(l_run[sb]);
            const float inv = 1.0f / l_tot;
            // Round 5: 16-byte stores.  A lane owns a query column of O^T and the 4-element groups 8g + 4hh + {0..3} of its 64 values
            // (hh = which half of the wave), so the natural stores are 8 bytes per lane, 16 contiguous bytes per row and
            // instruction.  v_permlane32_swap exchanges, per pair of groups (g, g + 1), the upper half's group g with the lower
            // half's group g + 1: afterwards the lower lanes hold values 8g .. 8g + 7 and the upper lanes 8g + 8 .. 8g + 15 of their
            // row -- one 16-byte store per pair, 32 contiguous bytes per row and instruction, half the write requests, no LDS
            // round trip (cdna_hip_programming.md T21).  Same values to the same addresses.  Both halves of a query's lane pair
            // take the branch together (same query).
            if (query < (PACKED ? klen : p.T)) {
                T* dst = (T*)p.out + pidx(((PACKED ? (int64_t)cur.roff : (int64_t)cur.n * p.T) + query) * (p.H * DH) + cur.h * DH, o_il);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int g = 0; g < 4; g += 2) {
                        uint2 h2[2], l2[2];
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            V4 hv, lv;
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                T hi, lo = (T)0.f;
                                split16<T, NT>(O[sb][dt][4 * (g + e) + j] * inv, hi, lo);
                                hv[j] = hi;
                                lv[j] = lo;
                            }
                            h2[e] = __builtin_bit_cast(uint2, hv);
                            l2[e] = __builtin_bit_cast(uint2, lv);
                        }
                        auto swap2 = [](uint2& a, uint2& b2) {  // vdst = group g, src = group g + 1
                            auto rx = __builtin_amdgcn_permlane32_swap(a.x, b2.x, false, false);
                            auto ry = __builtin_amdgcn_permlane32_swap(a.y, b2.y, false, false);
                            a.x = rx[0]; b2.x = rx[1];
                            a.y = ry[0]; b2.y = ry[1];
                        };
                        swap2(h2[0], h2[1]);
                        const int d0 = (dt * 32 << (o_il ? 1 : 0)) + 8 * g + 8 * hh;
                        *(uint4*)(dst + d0) = make_uint4(h2[0].x, h2[0].y, h2[1].x, h2[1].y);
                        if (NT > 1) {
                            swap2(l2[0], l2[1]);
                            *(uint4*)(dst + p.out_plane + d0) = make_uint4(l2[0].x, l2[0].y, l2[1].x, l2[1].y);
                        }
                    }
            }
        }
#ifdef AMX_ATTN_STAMP
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(a2_t3)::"memory");
        if (p.stamps && lane == 0) {
            unsigned long long* o = p.stamps + ((int64_t)item * WAVES + wave) * 8;
            o[0] = a2_t0; o[1] = a2_t1; o[2] = a2_t2; o[3] = a2_t3; o[4] = a2_c1 - a2_c0; o[5] = (unsigned long long)nkt; o[6] = 1;
            unsigned long long* ph = p.stamps + ((int64_t)total_items * WAVES) * 8 + ((int64_t)item * WAVES + wave) * 6;
            for (int i = 0; i < 6; ++i) ph[i] = a2_ph[i];
        }
#endif
        cur = nxt;
    }
}

template <typename T, int NT, bool PACKED, int WAVES, int STAGES>
void launch_attn2_layout(const AttnParams& p, int cus, hipStream_t stream) {
    static_assert(WAVES == 8 || STAGES == 2, "the counted waits of the 4-slot ring assume one DMA piece per wave");
    constexpr int lds = STAGES * NT * 2 * 64 * 128;
    static OncePerDevice attr;
    if (attr.first())
        (void)hipFuncSetAttribute((const void*)attn2_kernel<T, NT, PACKED, WAVES, STAGES>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int qb = WAVES * 64;
    const int qblocks = (p.T + qb - 1) / qb;
    const int items = 8 * ((p.N * p.H + 7) / 8) * qblocks;
    int grid = (cus - cus % 8) * (WAVES == 8 ? 1 : 2);  // the item order assumes that workgroups i and i + grid share an XCD
    if (grid < 8) grid = 8;
    if (grid > items) grid = items;
    // WAVES == 4: one item per workgroup -- the hardware dispatcher hands them out as workgroups retire, which balances the two
    // workgroups of a CU (the younger one loses the issue arbitration and runs ~20 % slower: with a static split of the items
    // it sets the span).  AMX_ATTN2_PERSISTENT=1: developer A/B switch
    static const bool persistent4 = dev_switch("AMX_ATTN2_PERSISTENT");
    if (WAVES == 4 && !persistent4) grid = items;
    hipLaunchKernelGGL((attn2_kernel<T, NT, PACKED, WAVES, STAGES>), dim3((unsigned)grid), dim3(WAVES * 64), lds, stream, p, items);
}

template <typename T, int NT, int WAVES, int STAGES>
void launch_attn2(const AttnParams& p, int cus, hipStream_t stream) {
    if (p.row_off) launch_attn2_layout<T, NT, true, WAVES, STAGES>(p, cus, stream);
    else launch_attn2_layout<T, NT, false, WAVES, STAGES>(p, cus, stream);
}

template <typename T, int NT, int WAVES, int KT, bool PACKED, int KS = 1>
void launch_attn_layout(const AttnParams& p, hipStream_t stream) {
#ifdef AMX_ATTN_ABL_ONE_WG  // developer ablation: pad the LDS request so that only one workgroup fits a CU
    constexpr int lds = 100 * 1024;
#else
    constexpr int lds = KS * 2 * NT * 2 * KT * 128;  // (KS == 2: at least the 4 x 8.5 KiB of the merge patches)
#endif
    const int qblocks = (p.T + WAVES * 32 - 1) / (WAVES * 32);
    dim3 grid((unsigned)(8 * ((p.N * p.H + 7) / 8) * qblocks));
    static const bool narrow = dev_switch("AMX_ATTN_NARROW_STORES");  // developer A/B switch: 8-byte stores straight from the registers
    if (narrow) {
        static OncePerDevice attr_n;
        if (attr_n.first())
            (void)hipFuncSetAttribute((const void*)attn_kernel<T, NT, WAVES, KT, PACKED, KS, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      lds);
        hipLaunchKernelGGL((attn_kernel<T, NT, WAVES, KT, PACKED, KS, false>), grid, dim3(WAVES * KS * 64), lds, stream, p);
        return;
    }
    static OncePerDevice attr;
    if (attr.first())
        (void)hipFuncSetAttribute((const void*)attn_kernel<T, NT, WAVES, KT, PACKED, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((attn_kernel<T, NT, WAVES, KT, PACKED, KS>), grid, dim3(WAVES * KS * 64), lds, stream, p);
}

template <typename T, int NT, int WAVES, int KT, int KS = 1>
void launch_attn(const AttnParams& p, hipStream_t stream) {
    if (p.row_off) launch_attn_layout<T, NT, WAVES, KT, true, KS>(p, stream);
    else launch_attn_layout<T, NT, WAVES, KT, false, KS>(p, stream);
}

}  // namespace

// head dimensions other than 64 (AttnParams.dh; rows padded to dhp = 64 or 128): 8-wave workgroups, masked direct stores
template <typename T, int NT, bool PACKED, int NDH, int KQN = 4 * NDH>
void launch_attn_other_dh(const AttnParams& p, hipStream_t stream) {
    constexpr int WAVES = 8, KT = 64;
    constexpr int lds = 2 * NT * 2 * KT * 128 * NDH;
    static OncePerDevice attr;
    if (attr.first())
        (void)hipFuncSetAttribute((const void*)attn_kernel<T, NT, WAVES, KT, PACKED, 1, false, NDH, KQN>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int qblocks = (p.T + WAVES * 32 - 1) / (WAVES * 32);
    dim3 grid((unsigned)(8 * ((p.N * p.H + 7) / 8) * qblocks));
    hipLaunchKernelGGL((attn_kernel<T, NT, WAVES, KT, PACKED, 1, false, NDH, KQN>), grid, dim3(WAVES * 64), lds, stream, p);
}

// 128-wide rows: the instance whose score chain / output blocks stop at the real columns (80: XLS-R 1B; 96; otherwise all 128)
template <typename T, int NT, bool PACKED>
void launch_attn_wide(const AttnParams& p, hipStream_t stream) {
    static const bool whole = dev_switch("AMX_ATTN_WHOLE_ROW");  // developer A/B switch: every padded column, as until round 6
    if (!whole && p.dh <= 80) launch_attn_other_dh<T, NT, PACKED, 2, 5>(p, stream);
    else if (!whole && p.dh <= 96) launch_attn_other_dh<T, NT, PACKED, 2, 6>(p, stream);
    else launch_attn_other_dh<T, NT, PACKED, 2>(p, stream);
}

template <typename T, int NT>
void launch_attn_any(const AttnParams& p, hipStream_t stream) {
    if (p.dh != DH) {
        if (p.dhp > 64) {
            if (p.row_off) launch_attn_wide<T, NT, true>(p, stream);
            else launch_attn_wide<T, NT, false>(p, stream);
        } else {
            if (p.row_off) launch_attn_other_dh<T, NT, true, 1>(p, stream);
            else launch_attn_other_dh<T, NT, false, 1>(p, stream);
        }
        return;
    }
    // Short batches (e.g. 4 x 10 s: 64 (utterance, head) pairs x 2 blocks of 256 queries) leave most CUs without a
    // workgroup while the busy ones run two waves per SIMD: with 128-query workgroups of 4 waves the same waves spread over
    // twice as many CUs, one per SIMD.  Each query's arithmetic is identical in both forms (bitwise equal outputs).
    static int cus_of[MAX_DEVICES] = {};
    int& cus = cus_of[current_device()];
    if (!cus) {
        hipDeviceProp_t prop;
        cus = hipGetDeviceProperties(&prop, current_device()) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int64_t wg8 = (int64_t)p.N * p.H * ((p.T + 255) / 256);
    static const int force = dev_int("AMX_ATTN_WAVES", 0);  // developer A/B switch
    {
        // 64 queries per wave (attn2_kernel, 256-query workgroups) for long key loops on a full chip (AMX_ATTN_V2 = 0 / 1 forces
        // the choice: developer A/B switch)
        static const int v2 = dev_int("AMX_ATTN_V2", -1);
        // (crossover measured at 12-15 key tiles per item with the chip full: T = 749 123 -> 147 us, T = 999 235 -> 221,
        // T = 1499 246 -> 234, T = 1999 640 -> 558, T = 2999 849 -> 752; one utterance alone keeps the 32-query waves)
        const bool fits = wg8 * 2 >= 3 * (int64_t)cus && p.T >= 960;
        if (v2 == 1 || (v2 < 0 && fits && !force)) {
            launch_attn2<T, NT, 4, 2>(p, cus, stream);
            return;
        }
    }
    const bool small = force ? force == 4 : wg8 * 2 <= cus;
    // ... and when even the 128-query workgroups are at most one per CU, the key tiles of a query block are split over two wave
    // groups (KS = 2): two waves per SIMD instead of one, half the serial key loop (AMX_ATTN_KSPLIT=0: developer A/B switch)
    static const bool no_ksplit = dev_int("AMX_ATTN_KSPLIT", 1) == 0;
    const int64_t wg4 = (int64_t)p.N * p.H * ((p.T + 127) / 128);
    // (tools/attn_bench.hip, per launch: 4 x 10 s 23.0 -> 21.2 us, 1 x 10 s 19.2 -> 17.4, 2 x 20 s 38.7 -> 34.4; 1 x 3 s -- three key
    // tiles -- 10.3 -> 10.9: from six tiles on)
    if (small && !no_ksplit && !force && wg4 <= cus && p.T >= 384) launch_attn<T, NT, 4, 64, 2>(p, stream);
    else if (small) launch_attn<T, NT, 4, 64>(p, stream);
    else launch_attn<T, NT, 8, 64>(p, stream);
}

void launch_attention(int prec, const AttnParams& p, hipStream_t stream) {
    switch (prec) {
        case PREC_BF16: launch_attn_any<bf16, 1>(p, stream); break;
        case PREC_F16: launch_attn_any<f16, 1>(p, stream); break;
        case PREC_BF16X3: launch_attn_any<bf16, 2>(p, stream); break;
        default: launch_attn_any<f16, 2>(p, stream); break;
    }
}

}  // namespace amx
