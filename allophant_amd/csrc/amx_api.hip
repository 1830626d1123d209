// C ABI of liballophant_amx (see include/allophant_amx.h): model construction from a reference state_dict, inventory
// composition, and the forward pass orchestration (kernel sequence on one HIP stream).
#include "../../include/allophant_amx.h"
#include "amx_common.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <initializer_list>
#include <map>
#include <cstdlib>
#include <string>
#include <vector>

using namespace amx;

namespace {

thread_local std::string g_create_error;

struct Layer {
    float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
    void *wqkv, *wo, *w1, *w2;
    float *bqkv, *bo, *b1, *b2;
    float r_qkv = 1.f, r_o = 1.f, r_1 = 1.f, r_2 = 1.f;  // 1 / (power of two the packed planes were multiplied by)
    // pre-LN layers: the LayerNorm in front of QKV / FFN1 is folded into those products (GemmParams.row_coef): wqkv / w1 hold
    // gamma (.) W, bqkv / b1 hold d = W beta + b, and c = W gamma (row sums of the folded weights) is what the mean term multiplies
    float *c_qkv = nullptr, *c_1 = nullptr;
};

// One GEMM (or GEMM pair for the composition head) of the hierarchical projection
struct HeadStep {
    std::vector<int> classes;   // classes computed by this step (stacked rows of W), evaluation order
    bool direct_output;         // A operand = final LayerNorm planes (single "OUTPUT" dependency)
    std::vector<ConcatPart> parts;  // otherwise: concatenation recipe (src pointers filled per forward)
    std::vector<int> part_dep;      // per part: class index (>=0) or output code (<0)
    int K, Kpad;                // input width
    void* W;                    // planes [rows, Kpad]
    float* bias;                // [rows]
    int rows;                   // sum of out_features
    bool composed;              // rows == embedding_size, followed by the composition product
    ConcatPart* parts_dev;
    std::vector<ConcatPart> parts_uploaded;  // what parts_dev currently holds (re-uploaded only when it changes)
    // time layer (ProjectingMultiheadAttention, acoustic_model.py:237-268): W / bias are its input_projection and the
    // step continues LayerNorm(+positions) -> in_proj -> key-masked attention over time -> out_proj
    int time_heads = 0;         // 0: plain linear classifier
    int Cpad = 0;               // rows rounded up to 8 (K of the in_proj / out_proj products)
    float *tl_g = nullptr, *tl_b = nullptr, *tl_bin = nullptr, *tl_bout = nullptr, *tl_pe = nullptr;
    void *tl_win = nullptr, *tl_wout = nullptr;
    float r_w = 1.f, r_tin = 1.f, r_tout = 1.f;  // reciprocal pack scales of W, tl_win, tl_wout
};

}  // namespace

struct amx_handle_s {
    int device = 0;
    amx_config cfg{};
    std::vector<amx_class_desc> classes;
    std::vector<int> order;
    int prec = AMX_PREC_F16X3, NT = 2;
    // two-plane modes: GEMM operands as interleaved planes (amx_common.h pidx(): hi and lo of 32 K elements share a 128-byte
    // line, which is what the ping-pong GEMM's whole-line operand DMA needs).  Needs 32-element row blocks: conv_dim and ffn
    // multiples of 32 (every wav2vec 2.0 shape); other shapes keep separate planes and run on the generic tile kernels.
    bool il = true;
    std::string err;
    std::vector<void*> allocs;
    int64_t weight_bytes = 0, ws_bytes = 0;

    // weights
    float *c0_w = nullptr, *c0_b = nullptr;
    double* c0_stats = nullptr;  // conv layer 0 on the matrix pipe: fp64 mean / covariance of the rows [w_c, b_c] (amx_common.h)
    float c0_wscale = 1.f;       // power of two its fp16 weight planes are built under
    // the variant of the wav2vec 2.0 encoder (amx_config ABI 4)
    bool gn = false;         // feat_extract_norm == "group": GroupNorm over time behind conv layer 0, no norm behind layers 1..
    bool stable = true;      // do_stable_layer_norm: pre-LN layers + final LayerNorm; false: post-LN layers
    bool masked = true;      // the model is called with the attention mask of the lengths; false: attention_mask=None
    float* conv_b[AMX_MAX_CONV] = {};
    float* conv_g[AMX_MAX_CONV] = {};
    float* conv_be[AMX_MAX_CONV] = {};
    void* conv_w[AMX_MAX_CONV] = {};
    void* conv_w_tm[AMX_MAX_CONV] = {};  // the same weights in tap-minor K order (GemmParams.a_taps), for the row-complete kernel
    int conv_tm_slice = 0;               // channels per slice of that order (0: no such copy)
    // Range of the 16-bit planes: every weight tensor is packed times a power of two that puts its largest element into
    // [4096, 8192) (exact, so results do not change) and the product's epilogue multiplies by the reciprocal kept here.  An
    // fp16 lo plane only carries its 11 bits while it is a normal number (|w| >= 0.25) and the hi plane underflows below
    // 6e-5: with unscaled planes the accuracy of the split modes would depend on the magnitude of the checkpoint's weights.
    float conv_r[AMX_MAX_CONV] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
    float fp_r = 1.f, pos_r = 1.f;
    std::vector<float> emb_host;  // composition embedding table (host copy: scale of a composed inventory matrix)
    int* nonfinite = nullptr;     // device counter of valid frames with non-finite logits in the last forward pass
    float *fp_g = nullptr, *fp_b = nullptr, *fp_bias = nullptr;
    void* fp_w = nullptr;
    void* pos_w = nullptr;
    float* pos_b = nullptr;
    std::vector<Layer> layers;
    float *fln_g = nullptr, *fln_b = nullptr;
    float *unit_g = nullptr, *zero_b = nullptr;  // [hidden] ones / zeros
    std::vector<HeadStep> steps;
    float* emb = nullptr;  // composition embedding table [rows, E]
    int emb_rows = 0;
    int composed_class = -1;
    std::vector<bool> need_hidden;

    // inventories: every distinct `target_feature_indices` matrix seen gets its own composed phoneme matrix and its own
    // device output tables, so switching between them (the per-language loop of run.py:742-753) neither recomputes nor
    // overwrites anything a forward pass still in flight may be reading.  Models without a composition layer own one
    // implicit entry (P1 = 0).
    struct Inventory {
        std::vector<int64_t> key;  // phones, features, tfi..., offsets...
        int P1 = 0;                // phones + blank
        float r_composed = 1.f;    // reciprocal pack scale of composed_w
        void* composed_w = nullptr;
        float* composed_f32 = nullptr;
        int64_t* idx_dev = nullptr;
        OutDesc *out_unique_dev = nullptr, *out_all_dev = nullptr;
        uint64_t last_use = 0;
        uint64_t generation = 0;   // unique per install: a cached layout of an evicted entry never matches its successor
    };
    static constexpr int INV_CAP = 16;
    std::vector<Inventory> inventories;
    int inv = -1;  // current entry, -1 = composition model without an inventory yet
    uint64_t inv_clock = 0;
    int P1 = 0;    // of the current inventory
    float r_composed = 1.f;
    void* composed_w = nullptr;
    float* composed_f32 = nullptr;

    // logits layout
    std::vector<int> col, width;  // per class
    int ld_logits = 0;
    std::vector<amx_output_desc> outputs;  // relative to N, T of the last layout computation
    std::vector<OutDesc> out_unique, out_all;
    OutDesc *out_unique_dev = nullptr, *out_all_dev = nullptr;  // of the current inventory
    int layout_inv = -2;  // inventory slot the host-side layout was computed for ...
    uint64_t layout_gen = 0, inv_gen = 0;  // ... and the install generation of that slot: slots are reused after eviction
    int layout_N = -1;
    int64_t layout_T = -1;

    // workspace
    struct WS {
        void* p = nullptr;
        size_t bytes = 0;
    };
    std::map<std::string, WS> ws;
    // pinned host staging of the per-utterance lengths: a small ring of slots, each guarded by an event recorded behind
    // its H2D copies, so that a call never waits for the previous forward pass (the host keeps running ahead of the GPU)
    static constexpr int PIN_SLOTS = 4;
    int64_t* h_lengths_pinned[PIN_SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    int* h_frames_pinned[PIN_SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    int* h_rowoff_pinned[PIN_SLOTS] = {nullptr, nullptr, nullptr, nullptr};  // packed-row offsets of the utterances (N + 1)
    int* h_tiles_pinned[PIN_SLOTS] = {nullptr, nullptr, nullptr, nullptr};  // tile lists of the conv layers of a ragged batch
    size_t tiles_cap[PIN_SLOTS] = {0, 0, 0, 0};                                // capacity of each, in ints
    hipEvent_t pin_event[PIN_SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    bool pin_busy[PIN_SLOTS] = {false, false, false, false};
    int pin_next = 0;
    int pinned_cap = 0;
    // per-kernel-class HIP event timing (AMX_FLAG_TIMING)
    struct Span { int cls; hipEvent_t a, b; };
    std::vector<Span> spans;
    std::vector<hipEvent_t> event_pool;
    bool timing = false;
    hipStream_t timing_stream = nullptr;
    // HIP graphs of whole forward passes (region "enqueue" of amx_forward: every memset, kernel and device copy of a pass;
    // the small pinned H2D uploads stay eager in front of it).  A pass is captured when the same key -- buffers, geometry,
    // lengths, flags, inventory, workspace generation -- recurs among the last few passes, and replayed while it recurs: one
    // hipGraphLaunch instead of ~185 launches.  Cached LRU; dropped when a workspace buffer moves.
    struct GraphEntry {
        std::vector<int64_t> key;
        hipGraphExec_t exec = nullptr;
        hipGraph_t graph = nullptr;  // the template stays alive as long as its instance (destroyed together)
        uint64_t last_use = 0;
        hipStream_t last_stream = nullptr;  // where it was launched last: waited for before the instance is destroyed
    };
    static constexpr int GRAPH_CAP = 32;  // (a length-sorted corpus on a grid of batch geometries cycles through two dozen of them)
    std::vector<GraphEntry> graphs;
    std::vector<std::vector<int64_t>> graph_seen;  // keys of the last few eager passes (a caller that alternates between two
                                                   // output buffers -- an overlapped gather holds one -- repeats with period 2)
    static constexpr int GRAPH_SEEN = 8;
    hipStream_t capture_stream = nullptr;  // capture never runs on the caller's stream (which may be the null stream)
    bool graph_broken = false;             // a capture failed once: stay eager
    uint64_t ws_gen = 0, graph_clock = 0;
    int64_t graph_captures = 0, graph_replays = 0;
    // range report (AMX_ERANGE) without a host synchronisation: every pass copies its non-finite frame count into a pinned
    // slot behind itself; the NEXT amx_forward / amx_synchronize reads the slots whose event has completed
    struct RangeSlot {
        int* host = nullptr;
        hipEvent_t ev = nullptr;
        bool pending = false, cont = false;
        uint64_t pass_id = 0;  // which amx_forward of this handle the reading belongs to (amx_pass_info: AMX_PASS_INFO_ID)
    };
    static constexpr int RANGE_SLOTS = 8;
    RangeSlot range[RANGE_SLOTS];
    int range_next = 0;
    int64_t range_carry = 0;  // count of a slot that had to be reused before any call had read it
    std::string range_carry_ids;  // ... and the passes it belonged to
    uint64_t pass_counter = 0;    // passes issued so far (the id of the last one)
    // the slices of one over-long batch (AMX_FLAG_CONTINUE) share ONE running device counter: the reading of a slice includes its
    // predecessors'.  What the last poll saw of a chain whose later slices were not ready yet, so that the next poll counts the
    // difference only (round-5 advisor finding: a chain read across two polls used to be counted twice)
    int range_chain_seen = 0;
    bool range_chain_open = false;
    // last forward geometry
    int last_N = 0;
    bool last_fold = false;  // amx_pass_info: the last pass ran its encoder layers with the LayerNorm fold ...
    int last_packed = 0, last_graph = 0;  // ... on packed rows (1; 2: from the feature projection on); eager / recorded / replayed
    int64_t last_rows = 0;
    bool qkv_dirty = false;
    int64_t last_L = 0, last_T = 0;
    bool last_keep = false;
    // the last forward pass ran its heads on the packed rows of a ragged batch (packed_early): hidden states kept for
    // OUTPUT_i classifiers, the final hidden state and the logits hold sum(frames) rows, utterance n at last_rowoff[n]
    bool last_packed_rows = false;
    std::vector<int> last_rowoff, last_frames;
};

namespace {

#define HIPCHK(h, expr)                                                                             \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) {                                                                     \
            (h)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                           \
            return AMX_EHIP;                                                                        \
        }                                                                                           \
    } while (0)

int fail(amx_handle h, int code, const std::string& msg) {
    if (h) h->err = msg;
    g_create_error = msg;
    return code;
}

void* dev_alloc(amx_handle h, size_t bytes, bool weight = true) {
    void* p = nullptr;
    if (bytes == 0) bytes = 16;
    if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    h->allocs.push_back(p);
    if (weight) h->weight_bytes += (int64_t)bytes;
    return p;
}

int ws_get(amx_handle h, const char* name, size_t bytes, void** out, bool zero_on_grow = false) {
    auto& w = h->ws[name];
    if (w.bytes < bytes) {
        if (w.p) {
            HIPCHK(h, hipDeviceSynchronize());
            HIPCHK(h, hipFree(w.p));
            h->ws_bytes -= (int64_t)w.bytes;
            w.p = nullptr;
            w.bytes = 0;
        }
        // grow geometrically: a corpus of ragged batches would otherwise reallocate (and synchronise) every few calls
        size_t want = bytes + bytes / 4 + 256;
        if (hipMalloc(&w.p, want) != hipSuccess) return fail(h, AMX_ENOMEM, std::string("workspace allocation failed: ") + name);
        w.bytes = want;
        h->ws_bytes += (int64_t)want;
        ++h->ws_gen;  // a buffer moved: every captured graph holds stale pointers
        if (zero_on_grow) HIPCHK(h, hipMemset(w.p, 0, want));
    }
    *out = w.p;
    return AMX_OK;
}

int round_up(int x, int m) { return (x + m - 1) / m * m; }

// brackets one kernel launch with HIP events on the launch stream when timing is enabled
struct Timed {
    amx_handle h;
    hipEvent_t b = nullptr;
    Timed(amx_handle h_, int cls) : h(h_) {
        if (!h->timing) return;
        hipEvent_t a = nullptr;
        auto get = [&](hipEvent_t& e) {
            if (!h->event_pool.empty()) { e = h->event_pool.back(); h->event_pool.pop_back(); }
            else if (hipEventCreate(&e) != hipSuccess) e = nullptr;
        };
        get(a); get(b);
        if (!a || !b) { b = nullptr; return; }
        (void)hipEventRecord(a, h->timing_stream);
        h->spans.push_back({cls, a, b});
    }
    ~Timed() { if (b) (void)hipEventRecord(b, h->timing_stream); }
};
inline int gemm_class(int prec, const GemmParams& g) { return gemm_uses_pp(prec, g) ? AMX_KC_GEMM_PP : AMX_KC_GEMM_TILE; }

struct TensorMap {
    std::map<std::string, const amx_tensor*> m;
    const amx_tensor* get(const std::string& k) const {
        auto it = m.find(k);
        return it == m.end() ? nullptr : it->second;
    }
};

const std::string AM = "_acoustic_model._model.";
const std::string PROJ = "_projection._layers.";

// evaluation order of the classifier graph = order in which AttributeGraph.sort() yields nodes
// (reference attribute_graph.py:124-199): depth-first post-order, roots in index order, edges in listed order.
int evaluation_order(const std::vector<amx_class_desc>& cls, std::vector<int>& order, std::string& err) {
    int n = (int)cls.size();
    std::vector<int> state(n, 0);
    order.clear();
    for (int root = 0; root < n; ++root) {
        if (state[root]) continue;
        std::vector<std::pair<int, int>> stack;
        stack.push_back({root, 0});
        state[root] = 1;
        while (!stack.empty()) {
            auto [node, edge] = stack.back();
            stack.pop_back();
            std::vector<int> deps;
            for (int i = 0; i < cls[node].n_deps; ++i)
                if (cls[node].deps[i] >= 0) deps.push_back(cls[node].deps[i]);
            if (edge < (int)deps.size()) {
                stack.push_back({node, edge + 1});
                int t = deps[edge];
                if (t >= n) { err = "dependency index out of range"; return AMX_EINVAL; }
                if (state[t] == 1) { err = std::string("Dependency cycle detected at ") + cls[t].name; return AMX_EINVAL; }
                if (state[t] == 0) { state[t] = 1; stack.push_back({t, 0}); }
            } else {
                state[node] = 2;
                order.push_back(node);
            }
        }
    }
    return AMX_OK;
}

}  // namespace

static void free_inventory(amx_handle_s::Inventory& e);
static void drop_graphs(amx_handle h);
static int install_inventory(amx_handle h, std::vector<int64_t>&& key, const std::vector<int64_t>& idx, int P1, int features,
                             hipStream_t s);

// =================================================================================================================
// creation
// =================================================================================================================
static int upload_f32(amx_handle h, const TensorMap& tm, const std::string& key, int64_t numel, float** out, float scale = 1.f,
                      float* staging = nullptr) {
    const amx_tensor* t = tm.get(key);
    if (!t) return fail(h, AMX_EINVAL, "missing tensor in state_dict: " + key);
    if (t->numel != numel)
        return fail(h, AMX_EINVAL, "tensor " + key + " has " + std::to_string(t->numel) + " elements, expected " + std::to_string(numel));
    float* d = (float*)dev_alloc(h, (size_t)numel * 4);
    if (!d) return fail(h, AMX_ENOMEM, "device allocation failed for " + key);
    if (scale == 1.f) {
        HIPCHK(h, hipMemcpy(d, t->data, (size_t)numel * 4, hipMemcpyHostToDevice));
    } else {
        HIPCHK(h, hipMemcpy(staging, t->data, (size_t)numel * 4, hipMemcpyHostToDevice));
        launch_scale_copy(staging, d, numel, scale, 0);
        HIPCHK(h, hipDeviceSynchronize());
    }
    *out = d;
    return AMX_OK;
}

// uploads a [rows, cols] fp32 matrix and packs it as planes at row offset `row0` of dst [*, ldd]
static int pack_linear(amx_handle h, const TensorMap& tm, const std::string& key, int rows, int cols, float scale, void* dst,
                       int64_t plane, int64_t ldd, int row0, int cols_pad, float* staging) {
    const amx_tensor* t = tm.get(key);
    if (!t) return fail(h, AMX_EINVAL, "missing tensor in state_dict: " + key);
    if (t->numel != (int64_t)rows * cols)
        return fail(h, AMX_EINVAL, "tensor " + key + " has " + std::to_string(t->numel) + " elements, expected " +
                                       std::to_string((int64_t)rows * cols));
    HIPCHK(h, hipMemcpy(staging, t->data, (size_t)t->numel * 4, hipMemcpyHostToDevice));
    launch_pack_matrix(h->prec, staging, rows, cols, cols, 1, scale, (char*)dst + (size_t)row0 * ldd * 2 * (plane == PLANE_IL ? 2 : 1), plane, ldd,
                       cols_pad, 0);
    HIPCHK(h, hipDeviceSynchronize());
    return AMX_OK;
}

// the same from host data (weights amx_create folds before packing)
static int pack_linear_host(amx_handle h, const float* data, int rows, int cols, float scale, void* dst, int64_t plane, int64_t ldd,
                            int row0, int cols_pad, float* staging) {
    HIPCHK(h, hipMemcpy(staging, data, (size_t)rows * cols * 4, hipMemcpyHostToDevice));
    launch_pack_matrix(h->prec, staging, rows, cols, cols, 1, scale, (char*)dst + (size_t)row0 * ldd * 2 * (plane == PLANE_IL ? 2 : 1), plane, ldd,
                       cols_pad, 0);
    HIPCHK(h, hipDeviceSynchronize());
    return AMX_OK;
}

// LayerNorm(gamma, beta) folded into the Linear(W [rows, cols], b) behind it:  LN(x) W^T + b = rstd ((x - p) . (gamma (.) W)^T)
// - rstd (mu - p) c + d  with  c = W gamma,  d = W beta + b  (fp64 sums; `pre` multiplies everything: the attention's query scale).
// folded: gamma (.) W * pre in fp32 (one rounding at 2^-24, below the 2^-22 of the planes); c from those values, as the product sees them
static void fold_layer_norm(const float* W, const float* b, const float* gamma, const float* beta, int rows, int cols, float pre,
                            float* folded, float* c, float* d) {
    for (int r = 0; r < rows; ++r) {
        double cs = 0.0, ds = 0.0;
        const float* w = W + (size_t)r * cols;
        float* f = folded + (size_t)r * cols;
        for (int k = 0; k < cols; ++k) {
            f[k] = w[k] * gamma[k] * pre;
            cs += (double)f[k];
            ds += (double)w[k] * (double)beta[k];
        }
        c[r] = (float)cs;
        d[r] = (float)((ds + (double)b[r]) * (double)pre);
    }
}

// power of two that puts the largest |w * pre| of the listed tensors into [4096, 8192); 1 for empty / zero / non-finite data
static bool pack_scale_off() {  // AMX_NO_PACK_SCALE=1: developer A/B switch (unscaled planes, as before round 3)
    static const bool off = dev_switch("AMX_NO_PACK_SCALE");
    return off;
}
static float pow2_for(float m) {
    if (pack_scale_off() || !(m > 0.f) || !std::isfinite(m)) return 1.f;
    return std::ldexp(1.f, 12 - std::ilogb(m));
}
static float pack_scale(std::initializer_list<std::pair<const amx_tensor*, float>> tensors) {
    float m = 0.f;
    for (auto& tp : tensors) {
        if (!tp.first) continue;
        const float* d = tp.first->data;
        float mt = 0.f;
        for (int64_t i = 0; i < tp.first->numel; ++i) mt = std::max(mt, std::fabs(d[i]));
        m = std::max(m, mt * std::fabs(tp.second));
    }
    return pow2_for(m);
}

// plane distance of a GEMM operand / row-stride granule of its padded K: interleaved (PLANE_IL, 32) or separate planes
// (a separate plane is never exactly PLANE_IL elements away: tiny matrices get a distance of 64, and every plane buffer is
// allocated with PLANE_SLACK bytes to spare for it)
constexpr size_t PLANE_SLACK = 256;
static int64_t pln(amx_handle h, int64_t separate) { return h->il ? PLANE_IL : std::max<int64_t>(separate, 64); }
static int kalign(amx_handle h) { return h->il ? 32 : 8; }

static void* alloc_planes(amx_handle h, int64_t elems_per_plane) {
    return dev_alloc(h, (size_t)elems_per_plane * 2 * h->NT + PLANE_SLACK);
}

extern "C" int amx_create(amx_handle* out, int device, const amx_config* cfg, const amx_class_desc* classes, int n_classes,
                          const amx_tensor* tensors, int n_tensors) {
    if (!out || !cfg || !classes || !tensors) return fail(nullptr, AMX_EINVAL, "null argument");
    *out = nullptr;
    if (cfg->abi_version != AMX_ABI_VERSION) return fail(nullptr, AMX_EINVAL, "ABI version mismatch");
    if (cfg->n_conv < 2 || cfg->n_conv > AMX_MAX_CONV) return fail(nullptr, AMX_EINVAL, "n_conv out of range");
    if (cfg->heads < 1 || cfg->hidden % cfg->heads != 0 || (cfg->hidden / cfg->heads) % 8 || cfg->hidden / cfg->heads > 128)
        return fail(nullptr, AMX_EINVAL, "head_dim (hidden / heads) must be a multiple of 8 and at most 128");
    if (cfg->hidden > 2048 || cfg->hidden % 8 || cfg->conv_dim > 1024 || cfg->conv_dim % 8)
        return fail(nullptr, AMX_EINVAL, "hidden must be a multiple of 8 and <= 2048, conv_dim a multiple of 8 and <= 1024");
    if (cfg->conv_dim >= 64 && cfg->conv_dim % 64) return fail(nullptr, AMX_EINVAL, "conv_dim must be < 64 or a multiple of 64");
    if (cfg->conv_kernel[0] > 16) return fail(nullptr, AMX_EINVAL, "first conv kernel must be <= 16");
    for (int i = 0; i < cfg->n_conv; ++i)
        if (cfg->conv_kernel[i] < 1 || cfg->conv_stride[i] < 1) return fail(nullptr, AMX_EINVAL, "conv kernels and strides must be positive");
    if (conv0_window_lds_bytes(cfg->conv_kernel[0], cfg->conv_stride[0], cfg->feat_extract_norm == AMX_NORM_GROUP) > 64 * 1024)
        return fail(nullptr, AMX_EINVAL, "first conv stride too large for the LDS window of the conv-0 kernels (stride <= 16 with the "
                                         "group-norm extractor, <= 127 otherwise)");
    if (cfg->hidden % cfg->pos_groups != 0 || (cfg->hidden / cfg->pos_groups) % 8 || cfg->hidden / cfg->pos_groups > 128)
        return fail(nullptr, AMX_EINVAL, "hidden / pos_groups must be a multiple of 8 and <= 128");
    if (cfg->ffn % 8) return fail(nullptr, AMX_EINVAL, "ffn must be a multiple of 8");
    if (cfg->precision < 0 || cfg->precision > 3) return fail(nullptr, AMX_EINVAL, "unknown precision");
    if (cfg->feat_extract_norm != AMX_NORM_LAYER && cfg->feat_extract_norm != AMX_NORM_GROUP)
        return fail(nullptr, AMX_EINVAL, "`feat_extract_norm` has to be one of ['group', 'layer']");
    if (cfg->embedding_size % 8) return fail(nullptr, AMX_EINVAL, "embedding_size must be a multiple of 8");
    if (n_classes < 1) return fail(nullptr, AMX_EINVAL, "Each model needs at least one classifier");

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return fail(nullptr, AMX_EHIP, "no HIP device available: liballophant_amx has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(nullptr, AMX_EINVAL, "device index out of range");
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, AMX_EHIP, "hipSetDevice failed");

    amx_handle h = new amx_handle_s();
    h->device = device;
    h->cfg = *cfg;
    h->prec = cfg->precision;
    h->NT = prec_planes(cfg->precision);
    h->gn = cfg->feat_extract_norm == AMX_NORM_GROUP;
    h->stable = cfg->stable_layer_norm != 0;
    h->masked = cfg->use_attention_mask != 0;
    {
        static const bool plain = dev_switch("AMX_PLAIN_PLANES");  // developer A/B switch
        // (hidden too -- round 6: the residual-stream planes [M, hidden] are GEMM operands like the others; a hidden size off the
        // 32-element blocks -- 240 = 2 heads of 120 -- ran interleaved planes through kernels that assume block-aligned rows)
        h->il = h->NT > 1 && !plain && cfg->conv_dim % 32 == 0 && cfg->ffn % 32 == 0 && cfg->hidden % 32 == 0;
    }
    h->classes.assign(classes, classes + n_classes);
    auto bail = [&](int code) {
        g_create_error = h->err;
        amx_destroy(h);
        return code;
    };

    // ---- validate the classifier graph (mirrors the ValueErrors of acoustic_model.py:353-466) ----
    bool uses_output = false;
    for (int i = 0; i < n_classes; ++i) {
        const auto& c = h->classes[i];
        for (int j = 0; j < i; ++j)
            if (!strncmp(c.name, h->classes[j].name, AMX_NAME_LEN)) { h->err = "Dependencies contain duplicate keys"; return bail(AMX_EINVAL); }
        if (!strncmp(c.name, "OUTPUT", 6)) { h->err = "'OUTPUT' is a reserved keyword"; return bail(AMX_EINVAL); }
        if (c.n_deps < 1 || c.n_deps > AMX_MAX_DEPS) { h->err = "Each class projection requires a dependency"; return bail(AMX_EINVAL); }
        if (c.out_features < 1) { h->err = "classifier without outputs"; return bail(AMX_EINVAL); }
        if (c.time_heads < 0 || (c.time_heads > 0 && c.out_features % c.time_heads)) {
            h->err = "embed_dim must be divisible by num_heads";  // nn.MultiheadAttention's assertion
            return bail(AMX_EINVAL);
        }
        if (c.time_heads > 0 && c.time_positional && (c.out_features & 1)) {
            h->err = "sinusoidal position embeddings need an even number of classifier outputs";
            return bail(AMX_EINVAL);
        }
        for (int d = 0; d < c.n_deps; ++d) {
            if (c.deps[d] < 0) {
                uses_output = true;
                if (c.deps[d] < -1 && -2 - c.deps[d] > cfg->layers) { h->err = "OUTPUT_i exceeds the number of encoder layers"; return bail(AMX_EINVAL); }
            } else if (c.deps[d] >= n_classes) { h->err = "unknown dependency"; return bail(AMX_EINVAL); }
        }
    }
    if (!uses_output) { h->err = "At least one of the input layers requires 'OUTPUT' as a dependency"; return bail(AMX_EINVAL); }
    {
        int rc = evaluation_order(h->classes, h->order, h->err);
        if (rc) return bail(rc);
    }

    TensorMap tm;
    int64_t max_numel = 0;
    for (int i = 0; i < n_tensors; ++i) {
        tm.m[tensors[i].name] = &tensors[i];
        max_numel = std::max(max_numel, tensors[i].numel);
    }
    // (the fused, LayerNorm-folded Q / K / V matrix is packed from one host buffer of 3 hidden^2 floats: larger than any single
    // tensor when ffn < 3 hidden)
    max_numel = std::max<int64_t>(max_numel, (int64_t)3 * cfg->hidden * cfg->hidden);
    float* staging = nullptr;
    if (hipMalloc(&staging, (size_t)max_numel * 4 + 16) != hipSuccess) { h->err = "staging allocation failed"; return bail(AMX_ENOMEM); }
    struct StagingGuard {
        float* p;
        ~StagingGuard() { (void)hipFree(p); }
    } guard{staging};

    const int C = cfg->conv_dim, D = cfg->hidden, F = cfg->ffn;
    const float eps = cfg->eps;
    (void)eps;
    int rc;
#define TRY(x) do { rc = (x); if (rc) return bail(rc); } while (0)

    // ---- feature extractor ----
    {
        std::string p = AM + "feature_extractor.conv_layers.0.";
        TRY(upload_f32(h, tm, p + "conv.weight", (int64_t)C * cfg->conv_kernel[0], &h->c0_w));
        int c_in = 1;
        for (int i = 0; i < cfg->n_conv; ++i) {
            p = AM + "feature_extractor.conv_layers." + std::to_string(i) + ".";
            int k = cfg->conv_kernel[i];
            if (cfg->conv_bias) {
                TRY(upload_f32(h, tm, p + "conv.bias", C, &h->conv_b[i]));
            } else {
                // config.conv_bias = False (wav2vec2-base / -large): nn.Conv1d(bias=False) -- a zero bias for the kernels
                h->conv_b[i] = (float*)dev_alloc(h, (size_t)C * 4);
                if (!h->conv_b[i] || hipMemset(h->conv_b[i], 0, (size_t)C * 4) != hipSuccess) { h->err = "device allocation failed"; return bail(AMX_ENOMEM); }
            }
            // "layer": LayerNorm(C) behind every conv layer; "group": GroupNorm(C, C) behind layer 0 only -- the same key names
            // (`layer_norm.{weight,bias}`, transformers Wav2Vec2GroupNormConvLayer / Wav2Vec2NoLayerNormConvLayer)
            if (!h->gn || i == 0) {
                TRY(upload_f32(h, tm, p + "layer_norm.weight", C, &h->conv_g[i]));
                TRY(upload_f32(h, tm, p + "layer_norm.bias", C, &h->conv_be[i]));
            }
            if (i > 0) {
                const amx_tensor* t = tm.get(p + "conv.weight");
                if (!t || t->numel != (int64_t)C * c_in * k) { h->err = "missing or mis-shaped tensor " + p + "conv.weight"; return bail(AMX_EINVAL); }
                int64_t plane = (int64_t)C * c_in * k;
                h->conv_w[i] = alloc_planes(h, plane);
                if (!h->conv_w[i]) { h->err = "device allocation failed"; return bail(AMX_ENOMEM); }
                if (hipMemcpy(staging, t->data, (size_t)t->numel * 4, hipMemcpyHostToDevice) != hipSuccess) { h->err = "H2D failed"; return bail(AMX_EHIP); }
                const float ps = pack_scale({{t, 1.f}});
                h->conv_r[i] = 1.f / ps;
                launch_pack_conv_w(h->prec, staging, C, c_in, k, ps, h->conv_w[i], pln(h, plane), 0);
                // layers the row-complete kernel may take (LayerNorm variant, conv_dim 512, k > 1 taps): a tap-minor copy too
                const int tm_slice = h->NT == 2 ? 32 : 64;
                if (!h->gn && i < cfg->n_conv - 1 && k > 1 && C == 512 && c_in % tm_slice == 0) {
                    h->conv_w_tm[i] = alloc_planes(h, plane);
                    if (!h->conv_w_tm[i]) { h->err = "device allocation failed"; return bail(AMX_ENOMEM); }
                    launch_pack_conv_w(h->prec, staging, C, c_in, k, ps, h->conv_w_tm[i], pln(h, plane), 0, tm_slice);
                    h->conv_tm_slice = tm_slice;
                }
                if (hipDeviceSynchronize() != hipSuccess) { h->err = "pack_conv_w failed"; return bail(AMX_EHIP); }
            }
            c_in = C;
        }
        // conv layer 0 on the matrix pipe (LayerNorm variant, k = 10, C = 512): the LayerNorm statistics of a frame are a linear
        // and a quadratic form of its 10 samples -- mean_c(w_c . x + b_c) = m . [x; 1], var_c = [x; 1]^T G [x; 1] with m / G the mean
        // / covariance of the rows [w_c, b_c] over the channels -- tabulated here in fp64
        if (conv0_mfma_eligible(C, cfg->conv_kernel[0], cfg->conv_stride[0]))
            h->c0_wscale = pack_scale({{tm.get(AM + "feature_extractor.conv_layers.0.conv.weight"), 1.f}});
        if (!h->gn && conv0_mfma_eligible(C, cfg->conv_kernel[0], cfg->conv_stride[0])) {
            const amx_tensor* wt = tm.get(AM + "feature_extractor.conv_layers.0.conv.weight");
            const amx_tensor* bt = cfg->conv_bias ? tm.get(AM + "feature_extractor.conv_layers.0.conv.bias") : nullptr;
            const int k0 = cfg->conv_kernel[0], dim = k0 + 1;
            std::vector<double> table((size_t)dim + (size_t)dim * dim, 0.0), row(dim);
            for (int ch = 0; ch < C; ++ch)
                for (int j = 0; j < dim; ++j) table[j] += (j < k0 ? (double)wt->data[(size_t)ch * k0 + j] : (bt ? (double)bt->data[ch] : 0.0)) / C;
            for (int ch = 0; ch < C; ++ch) {
                for (int j = 0; j < dim; ++j) row[j] = (j < k0 ? (double)wt->data[(size_t)ch * k0 + j] : (bt ? (double)bt->data[ch] : 0.0)) - table[j];
                for (int a = 0; a < dim; ++a)
                    for (int b2 = 0; b2 < dim; ++b2) table[(size_t)dim + (size_t)a * dim + b2] += row[a] * row[b2] / C;
            }
            h->c0_stats = (double*)dev_alloc(h, table.size() * 8);
            if (!h->c0_stats || hipMemcpy(h->c0_stats, table.data(), table.size() * 8, hipMemcpyHostToDevice) != hipSuccess) {
                h->err = "device allocation failed";
                return bail(AMX_ENOMEM);
            }
            h->c0_wscale = pack_scale({{wt, 1.f}});
        }
    }
    // ---- feature projection ----
    {
        std::string p = AM + "feature_projection.";
        TRY(upload_f32(h, tm, p + "layer_norm.weight", C, &h->fp_g));
        TRY(upload_f32(h, tm, p + "layer_norm.bias", C, &h->fp_b));
        TRY(upload_f32(h, tm, p + "projection.bias", D, &h->fp_bias));
        h->fp_w = alloc_planes(h, (int64_t)D * C);
        if (!h->fp_w) { h->err = "device allocation failed"; return bail(AMX_ENOMEM); }
        const float ps = pack_scale({{tm.get(p + "projection.weight"), 1.f}});
        h->fp_r = 1.f / ps;
        TRY(pack_linear(h, tm, p + "projection.weight", D, C, ps, h->fp_w, pln(h, (int64_t)D * C), C, 0, C, staging));
    }
    // ---- positional conv (weight-norm folded) ----
    {
        std::string p = AM + "encoder.pos_conv_embed.conv.";
        const int cg = D / cfg->pos_groups, k = cfg->pos_kernel;
        const amx_tensor* g = tm.get(p + "parametrizations.weight.original0");
        const amx_tensor* v = tm.get(p + "parametrizations.weight.original1");
        if (!g) g = tm.get(p + "weight_g");
        if (!v) v = tm.get(p + "weight_v");
        if (!g || !v || g->numel != k || v->numel != (int64_t)D * cg * k) { h->err = "missing or mis-shaped positional conv weights"; return bail(AMX_EINVAL); }
        TRY(upload_f32(h, tm, p + "bias", D, &h->pos_b));
        float* gd = (float*)dev_alloc(h, (size_t)k * 4 * 2);
        if (!gd) { h->err = "device allocation failed"; return bail(AMX_ENOMEM); }
        if (hipMemcpy(gd, g->data, (size_t)k * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(staging, v->data, (size_t)v->numel * 4, hipMemcpyHostToDevice) != hipSuccess) { h->err = "H2D failed"; return bail(AMX_EHIP); }
        int64_t plane = (int64_t)D * cg * k;
        h->pos_w = alloc_planes(h, plane);
        if (!h->pos_w) { h->err = "device allocation failed"; return bail(AMX_ENOMEM); }
        float ps = 1.f;
        {   // largest folded weight |v * g[tap] / ||v[:, :, tap]|||, on the host (the device folds them while packing)
            std::vector<double> sq(k, 0.0);
            std::vector<float> mx(k, 0.f);
            const int64_t rows = (int64_t)D * cg;
            for (int64_t r = 0; r < rows; ++r)
                for (int tap = 0; tap < k; ++tap) {
                    const float x = v->data[r * k + tap];
                    sq[tap] += (double)x * x;
                    mx[tap] = std::max(mx[tap], std::fabs(x));
                }
            float m = 0.f;
            for (int tap = 0; tap < k; ++tap)
                if (sq[tap] > 0.0) m = std::max(m, (float)(mx[tap] * std::fabs(g->data[tap]) / std::sqrt(sq[tap])));
            ps = pow2_for(m);
        }
        h->pos_r = 1.f / ps;
        launch_pack_posconv_w(h->prec, gd, staging, D, cg, k, ps, gd + k, h->pos_w, plane, 0);
        if (hipDeviceSynchronize() != hipSuccess) { h->err = "pack_posconv_w failed"; return bail(AMX_EHIP); }
    }
    // ---- encoder layers ----
    h->layers.resize(cfg->layers);
    // folded into W_q / b_q: dh^-0.5 and log2(e) -- the attention softmax runs in base 2 (v_exp_f32)
    const float qscale = 1.44269504088896340736f / sqrtf((float)(D / cfg->heads));
    for (int l = 0; l < cfg->layers; ++l) {
        Layer& ly = h->layers[l];
        std::string p = AM + "encoder.layers." + std::to_string(l) + ".";
        TRY(upload_f32(h, tm, p + "layer_norm.weight", D, &ly.ln1_g));
        TRY(upload_f32(h, tm, p + "layer_norm.bias", D, &ly.ln1_b));
        TRY(upload_f32(h, tm, p + "final_layer_norm.weight", D, &ly.ln2_g));
        TRY(upload_f32(h, tm, p + "final_layer_norm.bias", D, &ly.ln2_b));
        ly.wqkv = alloc_planes(h, (int64_t)3 * D * D);
        ly.wo = alloc_planes(h, (int64_t)D * D);
        ly.w1 = alloc_planes(h, (int64_t)F * D);
        ly.w2 = alloc_planes(h, (int64_t)D * F);
        ly.bqkv = (float*)dev_alloc(h, (size_t)3 * D * 4);
        if (!ly.wqkv || !ly.wo || !ly.w1 || !ly.w2 || !ly.bqkv) { h->err = "device allocation failed"; return bail(AMX_ENOMEM); }
        const char* names[3] = {"q_proj", "k_proj", "v_proj"};
        if (h->stable) {
            // pre-LN layer: `layer_norm` folded into Q / K / V, `final_layer_norm` into FFN1 (see fold_layer_norm)
            const amx_tensor *g1 = tm.get(p + "layer_norm.weight"), *be1 = tm.get(p + "layer_norm.bias");
            const amx_tensor *g2 = tm.get(p + "final_layer_norm.weight"), *be2 = tm.get(p + "final_layer_norm.bias");
            std::vector<float> folded((size_t)std::max(3 * D, F) * D), cvec(std::max(3 * D, F)), dvec(std::max(3 * D, F));
            for (int j = 0; j < 3; ++j) {
                const amx_tensor* w = tm.get(p + "attention." + names[j] + ".weight");
                const amx_tensor* b = tm.get(p + "attention." + names[j] + ".bias");
                if (!w || w->numel != (int64_t)D * D || !b || b->numel != D) { h->err = "missing or mis-shaped tensor " + p + "attention." + names[j]; return bail(AMX_EINVAL); }
                fold_layer_norm(w->data, b->data, g1->data, be1->data, D, D, j == 0 ? qscale : 1.f, folded.data() + (size_t)j * D * D,
                                cvec.data() + j * D, dvec.data() + j * D);
            }
            amx_tensor ft{};
            ft.data = folded.data(); ft.numel = (int64_t)3 * D * D;
            const float ps_qkv = pack_scale({{&ft, 1.f}});
            ly.r_qkv = 1.f / ps_qkv;
            TRY(pack_linear_host(h, folded.data(), 3 * D, D, ps_qkv, ly.wqkv, pln(h, (int64_t)3 * D * D), D, 0, D, staging));
            ly.c_qkv = (float*)dev_alloc(h, (size_t)3 * D * 4);
            if (!ly.c_qkv || hipMemcpy(ly.c_qkv, cvec.data(), (size_t)3 * D * 4, hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(ly.bqkv, dvec.data(), (size_t)3 * D * 4, hipMemcpyHostToDevice) != hipSuccess) { h->err = "device allocation failed"; return bail(AMX_ENOMEM); }
            const amx_tensor* w1 = tm.get(p + "feed_forward.intermediate_dense.weight");
            const amx_tensor* b1 = tm.get(p + "feed_forward.intermediate_dense.bias");
            if (!w1 || w1->numel != (int64_t)F * D || !b1 || b1->numel != F) { h->err = "missing or mis-shaped tensor " + p + "feed_forward.intermediate_dense"; return bail(AMX_EINVAL); }
            fold_layer_norm(w1->data, b1->data, g2->data, be2->data, F, D, 1.f, folded.data(), cvec.data(), dvec.data());
            ft.numel = (int64_t)F * D;
            const float ps_1 = pack_scale({{&ft, 1.f}});
            ly.r_1 = 1.f / ps_1;
            TRY(pack_linear_host(h, folded.data(), F, D, ps_1, ly.w1, pln(h, (int64_t)F * D), D, 0, D, staging));
            ly.c_1 = (float*)dev_alloc(h, (size_t)F * 4);
            ly.b1 = (float*)dev_alloc(h, (size_t)F * 4);
            if (!ly.c_1 || !ly.b1 || hipMemcpy(ly.c_1, cvec.data(), (size_t)F * 4, hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(ly.b1, dvec.data(), (size_t)F * 4, hipMemcpyHostToDevice) != hipSuccess) { h->err = "device allocation failed"; return bail(AMX_ENOMEM); }
            const float ps_o = pack_scale({{tm.get(p + "attention.out_proj.weight"), 1.f}});
            const float ps_2 = pack_scale({{tm.get(p + "feed_forward.output_dense.weight"), 1.f}});
            ly.r_o = 1.f / ps_o; ly.r_2 = 1.f / ps_2;
            TRY(pack_linear(h, tm, p + "attention.out_proj.weight", D, D, ps_o, ly.wo, pln(h, (int64_t)D * D), D, 0, D, staging));
            TRY(upload_f32(h, tm, p + "attention.out_proj.bias", D, &ly.bo));
            TRY(pack_linear(h, tm, p + "feed_forward.output_dense.weight", D, F, ps_2, ly.w2, pln(h, (int64_t)D * F), F, 0, F, staging));
            TRY(upload_f32(h, tm, p + "feed_forward.output_dense.bias", D, &ly.b2));
            continue;
        }
        const float ps_qkv = pack_scale({{tm.get(p + "attention.q_proj.weight"), qscale}, {tm.get(p + "attention.k_proj.weight"), 1.f},
                                         {tm.get(p + "attention.v_proj.weight"), 1.f}});
        ly.r_qkv = 1.f / ps_qkv;
        for (int j = 0; j < 3; ++j) {
            float sc = j == 0 ? qscale : 1.f;
            TRY(pack_linear(h, tm, p + "attention." + names[j] + ".weight", D, D, sc * ps_qkv, ly.wqkv, pln(h, (int64_t)3 * D * D), D, j * D, D, staging));
            const amx_tensor* b = tm.get(p + "attention." + names[j] + ".bias");
            if (!b || b->numel != D) { h->err = "missing tensor " + p + "attention." + names[j] + ".bias"; return bail(AMX_EINVAL); }
            if (hipMemcpy(staging, b->data, (size_t)D * 4, hipMemcpyHostToDevice) != hipSuccess) { h->err = "H2D failed"; return bail(AMX_EHIP); }
            launch_scale_copy(staging, ly.bqkv + j * D, D, sc, 0);
            if (hipDeviceSynchronize() != hipSuccess) { h->err = "scale_copy failed"; return bail(AMX_EHIP); }
        }
        const float ps_o = pack_scale({{tm.get(p + "attention.out_proj.weight"), 1.f}});
        const float ps_1 = pack_scale({{tm.get(p + "feed_forward.intermediate_dense.weight"), 1.f}});
        const float ps_2 = pack_scale({{tm.get(p + "feed_forward.output_dense.weight"), 1.f}});
        ly.r_o = 1.f / ps_o; ly.r_1 = 1.f / ps_1; ly.r_2 = 1.f / ps_2;
        TRY(pack_linear(h, tm, p + "attention.out_proj.weight", D, D, ps_o, ly.wo, pln(h, (int64_t)D * D), D, 0, D, staging));
        TRY(upload_f32(h, tm, p + "attention.out_proj.bias", D, &ly.bo));
        TRY(pack_linear(h, tm, p + "feed_forward.intermediate_dense.weight", F, D, ps_1, ly.w1, pln(h, (int64_t)F * D), D, 0, D, staging));
        TRY(upload_f32(h, tm, p + "feed_forward.intermediate_dense.bias", F, &ly.b1));
        TRY(pack_linear(h, tm, p + "feed_forward.output_dense.weight", D, F, ps_2, ly.w2, pln(h, (int64_t)D * F), F, 0, F, staging));
        TRY(upload_f32(h, tm, p + "feed_forward.output_dense.bias", D, &ly.b2));
    }
    {   // the affine part of a folded LayerNorm lives in the weights: its row pass (short batches) normalises with (1, 0)
        std::vector<float> ones(D, 1.f);
        h->unit_g = (float*)dev_alloc(h, (size_t)D * 4);
        h->zero_b = (float*)dev_alloc(h, (size_t)D * 4);
        if (!h->unit_g || !h->zero_b || hipMemcpy(h->unit_g, ones.data(), (size_t)D * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemset(h->zero_b, 0, (size_t)D * 4) != hipSuccess) { h->err = "device allocation failed"; return bail(AMX_ENOMEM); }
    }
    TRY(upload_f32(h, tm, AM + "encoder.layer_norm.weight", D, &h->fln_g));
    TRY(upload_f32(h, tm, AM + "encoder.layer_norm.bias", D, &h->fln_b));

    // ---- hierarchical projection plan ----
    h->need_hidden.assign(cfg->layers + 1, false);
    const bool blanks = cfg->dependency_blanks != 0;
    auto dep_width = [&](int dep) -> int {
        if (dep < 0) return D;
        return h->classes[dep].size + (blanks ? 1 : 0);
    };
    for (size_t oi = 0; oi < h->order.size(); ++oi) {
        int ci = h->order[oi];
        const amx_class_desc& c = h->classes[ci];
        bool composed = cfg->embedding_size > 0 && !strcmp(c.name, "phoneme");
        if (composed) {
            if (c.out_features != cfg->embedding_size) { h->err = "phoneme out_features must equal embedding_size"; return bail(AMX_EINVAL); }
            h->composed_class = ci;
        }
        bool direct = c.n_deps == 1 && c.deps[0] == AMX_DEP_OUTPUT;
        int K = 0;
        for (int d = 0; d < c.n_deps; ++d) {
            K += dep_width(c.deps[d]);
            if (c.deps[d] < -1) h->need_hidden[-2 - c.deps[d]] = true;
        }
        // stack onto the previous step when both read the final LayerNorm output directly and neither is composed
        if (direct && !composed && c.time_heads == 0 && !h->steps.empty() && h->steps.back().direct_output &&
            !h->steps.back().composed && h->steps.back().time_heads == 0) {
            h->steps.back().classes.push_back(ci);
            h->steps.back().rows += c.out_features;
            continue;
        }
        HeadStep st{};
        st.classes = {ci};
        st.direct_output = direct;
        st.K = K;
        st.Kpad = round_up(K, kalign(h));
        st.rows = c.out_features;
        st.composed = composed;
        st.parts_dev = nullptr;
        st.time_heads = c.time_heads;
        st.Cpad = round_up(c.out_features, kalign(h));
        if (!direct) {
            int colp = 0;
            for (int d = 0; d < c.n_deps; ++d) {
                ConcatPart pt{};
                int dep = c.deps[d];
                pt.type = dep < 0 ? 0 : 1;
                pt.width = dep_width(dep);
                pt.dst_col = colp;
                pt.src_col = 0;
                pt.src = nullptr;
                colp += pt.width;
                st.parts.push_back(pt);
                st.part_dep.push_back(dep);
            }
        }
        h->steps.push_back(st);
    }
    for (auto& st : h->steps) {
        st.W = alloc_planes(h, (int64_t)st.rows * st.Kpad);
        st.bias = (float*)dev_alloc(h, (size_t)st.rows * 4);
        if (!st.W || !st.bias) { h->err = "device allocation failed"; return bail(AMX_ENOMEM); }
        int row0 = 0;
        float ps_w = 1.f;
        {   // one scale for the stacked rows of the step
            float m = 0.f;
            for (int ci : st.classes) {
                std::string key = PROJ + h->classes[ci].name + "._time_distributed_layer." + (st.time_heads > 0 ? "input_projection." : "") + "weight";
                const amx_tensor* t = tm.get(key);
                if (t) for (int64_t i = 0; i < t->numel; ++i) m = std::max(m, std::fabs(t->data[i]));
            }
            ps_w = pow2_for(m);
        }
        st.r_w = 1.f / ps_w;
        for (int ci : st.classes) {
            const amx_class_desc& c = h->classes[ci];
            std::string p = PROJ + c.name + "._time_distributed_layer.";
            if (st.time_heads > 0) {
                // module tree of ProjectingMultiheadAttention: input_projection, layer_norm, attention (nn.MultiheadAttention)
                const int Co = c.out_features;
                const int64_t plane_in = (int64_t)3 * Co * st.Cpad, plane_out = (int64_t)Co * st.Cpad;
                st.tl_win = alloc_planes(h, plane_in);
                st.tl_wout = alloc_planes(h, plane_out);
                if (!st.tl_win || !st.tl_wout) { h->err = "device allocation failed"; return bail(AMX_ENOMEM); }
                const float ps_in = pack_scale({{tm.get(p + "attention.in_proj_weight"), 1.f}});
                const float ps_out = pack_scale({{tm.get(p + "attention.out_proj.weight"), 1.f}});
                st.r_tin = 1.f / ps_in; st.r_tout = 1.f / ps_out;
                TRY(pack_linear(h, tm, p + "attention.in_proj_weight", 3 * Co, Co, ps_in, st.tl_win, pln(h, plane_in), st.Cpad, 0, st.Cpad, staging));
                TRY(pack_linear(h, tm, p + "attention.out_proj.weight", Co, Co, ps_out, st.tl_wout, pln(h, plane_out), st.Cpad, 0, st.Cpad, staging));
                TRY(upload_f32(h, tm, p + "attention.in_proj_bias", 3 * Co, &st.tl_bin));
                TRY(upload_f32(h, tm, p + "attention.out_proj.bias", Co, &st.tl_bout));
                TRY(upload_f32(h, tm, p + "layer_norm.weight", Co, &st.tl_g));
                TRY(upload_f32(h, tm, p + "layer_norm.bias", Co, &st.tl_b));
                if (c.time_positional) {
                    // SinusoidalPositionEmbeddings (acoustic_model.py:34-69): base_k = exp(-2k ln(1e4) / size) for the
                    // column pair (2k, 2k+1)
                    std::vector<float> base(Co);
                    // fp32 like upstream: arange(0, size, 2) * float(-ln(1e4) / size), then exp
                    const float step = (float)(-(std::log(10000.0) / Co));
                    for (int col = 0; col < Co; ++col) {
                        volatile float arg = (float)(col & ~1) * step;
                        base[col] = (float)std::exp((double)arg);
                    }
                    st.tl_pe = (float*)dev_alloc(h, (size_t)Co * 4);
                    if (!st.tl_pe) { h->err = "device allocation failed"; return bail(AMX_ENOMEM); }
                    if (hipMemcpy(st.tl_pe, base.data(), (size_t)Co * 4, hipMemcpyHostToDevice) != hipSuccess) { h->err = "H2D failed"; return bail(AMX_EHIP); }
                }
                p += "input_projection.";
            }
            TRY(pack_linear(h, tm, p + "weight", c.out_features, st.K, ps_w, st.W, pln(h, (int64_t)st.rows * st.Kpad), st.Kpad, row0, st.Kpad, staging));
            const amx_tensor* b = tm.get(p + "bias");
            if (!b || b->numel != c.out_features) { h->err = "missing tensor " + p + "bias"; return bail(AMX_EINVAL); }
            if (hipMemcpy(st.bias + row0, b->data, (size_t)c.out_features * 4, hipMemcpyHostToDevice) != hipSuccess) { h->err = "H2D failed"; return bail(AMX_EHIP); }
            row0 += c.out_features;
        }
        if (!st.parts.empty()) {
            st.parts_dev = (ConcatPart*)dev_alloc(h, st.parts.size() * sizeof(ConcatPart));
            if (!st.parts_dev) { h->err = "device allocation failed"; return bail(AMX_ENOMEM); }
        }
    }
    if (h->composed_class >= 0) {
        std::string key = PROJ + "phoneme._composition_layer._attribute_embeddings.weight";
        const amx_tensor* t = tm.get(key);
        if (!t || t->numel % cfg->embedding_size) { h->err = "missing tensor " + key; return bail(AMX_EINVAL); }
        h->emb_rows = (int)(t->numel / cfg->embedding_size);
        h->emb_host.assign(t->data, t->data + t->numel);
        TRY(upload_f32(h, tm, key, t->numel, &h->emb));
    }
#undef TRY
    h->nonfinite = (int*)dev_alloc(h, 16);
    if (!h->nonfinite || hipMemset(h->nonfinite, 0, 16) != hipSuccess) { h->err = "device allocation failed"; return bail(AMX_ENOMEM); }
    if (h->composed_class < 0) {
        // no composition layer: one implicit inventory (fixed output widths)
        int rc2 = install_inventory(h, std::vector<int64_t>{}, std::vector<int64_t>{}, 0, 0, 0);
        if (rc2) return bail(rc2);
    }
    if (hipDeviceSynchronize() != hipSuccess) { h->err = "device synchronisation failed after packing"; return bail(AMX_EHIP); }
    *out = h;
    return AMX_OK;
}

extern "C" int amx_destroy(amx_handle h) {
    if (!h) return AMX_OK;
    // teardown is best effort: a failing release cannot be reported more usefully than by carrying on
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    for (void* p : h->allocs) (void)hipFree(p);
    for (auto& kv : h->ws)
        if (kv.second.p) (void)hipFree(kv.second.p);
    for (auto& e : h->inventories) free_inventory(e);
    for (int i = 0; i < amx_handle_s::PIN_SLOTS; ++i) {
        if (h->h_lengths_pinned[i]) (void)hipHostFree(h->h_lengths_pinned[i]);
        if (h->h_frames_pinned[i]) (void)hipHostFree(h->h_frames_pinned[i]);
        if (h->h_rowoff_pinned[i]) (void)hipHostFree(h->h_rowoff_pinned[i]);
        if (h->h_tiles_pinned[i]) (void)hipHostFree(h->h_tiles_pinned[i]);
        if (h->pin_event[i]) (void)hipEventDestroy(h->pin_event[i]);
    }
    drop_graphs(h);
    if (h->capture_stream) (void)hipStreamDestroy(h->capture_stream);
    for (auto& sl : h->range) {
        if (sl.host) (void)hipHostFree(sl.host);
        if (sl.ev) (void)hipEventDestroy(sl.ev);
    }
    for (auto& sp : h->spans) { (void)hipEventDestroy(sp.a); (void)hipEventDestroy(sp.b); }
    for (auto e : h->event_pool) (void)hipEventDestroy(e);
    delete h;
    return AMX_OK;
}

extern "C" const char* amx_last_error(amx_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

extern "C" int64_t amx_device_bytes(amx_handle h) { return h ? h->weight_bytes + h->ws_bytes : 0; }

// =================================================================================================================
// inventory
// =================================================================================================================
// (column, width) of every class in the logits buffer and the (column, classes, class prefix) tables of the outputs for an
// inventory of P1 - 1 phones; geometry-independent
static void build_tables(amx_handle h, int P1, std::vector<int>& col, std::vector<int>& width, int& ld_logits,
                         std::vector<OutDesc>& uniq, std::vector<OutDesc>& all) {
    const int nc = (int)h->classes.size();
    col.assign(nc, 0);
    width.assign(nc, 0);
    int colp = 0;
    for (int ci : h->order) {
        col[ci] = colp;
        width[ci] = ci == h->composed_class ? P1 : h->classes[ci].out_features;
        colp += width[ci];
    }
    ld_logits = round_up(colp, 4);
    uniq.clear();
    all.clear();
    int64_t prefix = 0;
    for (int ci : h->order) {
        OutDesc od{col[ci], width[ci], prefix};
        uniq.push_back(od);
        if (!strcmp(h->classes[ci].name, "phoneme") && h->cfg.allophone_layer) all.push_back(od);  // "phone" alias
        all.push_back(od);
        prefix += width[ci];
    }
}

static void free_inventory(amx_handle_s::Inventory& e) {
    if (e.composed_w) (void)hipFree(e.composed_w);
    if (e.composed_f32) (void)hipFree(e.composed_f32);
    if (e.idx_dev) (void)hipFree(e.idx_dev);
    if (e.out_unique_dev) (void)hipFree(e.out_unique_dev);
    if (e.out_all_dev) (void)hipFree(e.out_all_dev);
    e = amx_handle_s::Inventory{};
}

static void select_inventory(amx_handle h, int i) {
    auto& e = h->inventories[i];
    e.last_use = ++h->inv_clock;
    h->inv = i;
    h->P1 = e.P1;
    h->r_composed = e.r_composed;
    h->composed_w = e.composed_w;
    h->composed_f32 = e.composed_f32;
    h->out_unique_dev = e.out_unique_dev;
    h->out_all_dev = e.out_all_dev;
}

// builds a new cache entry on `s`: every buffer is fresh, so nothing in flight can be reading it
static int install_inventory(amx_handle h, std::vector<int64_t>&& key, const std::vector<int64_t>& idx, int P1, int features,
                             hipStream_t s) {
    int slot = -1;
    if ((int)h->inventories.size() < amx_handle_s::INV_CAP) {
        h->inventories.emplace_back();
        slot = (int)h->inventories.size() - 1;
    } else {
        // evict the least recently used entry; a forward pass still in flight may read its buffers
        for (int i = 0; i < (int)h->inventories.size(); ++i)
            if (i != h->inv && (slot < 0 || h->inventories[i].last_use < h->inventories[slot].last_use)) slot = i;
        HIPCHK(h, hipDeviceSynchronize());
        free_inventory(h->inventories[slot]);
    }
    auto& e = h->inventories[slot];
    e.P1 = P1;
    const int E = h->cfg.embedding_size;
    std::vector<int> col, width;
    int ld;
    std::vector<OutDesc> uniq, all;
    build_tables(h, P1, col, width, ld, uniq, all);
    auto fail_free = [&](int code, const char* msg) {
        free_inventory(e);
        if (slot == (int)h->inventories.size() - 1) h->inventories.pop_back();
        return fail(h, code, msg);
    };
    if (hipMalloc((void**)&e.out_unique_dev, uniq.size() * sizeof(OutDesc)) != hipSuccess ||
        hipMalloc((void**)&e.out_all_dev, all.size() * sizeof(OutDesc)) != hipSuccess)
        return fail_free(AMX_ENOMEM, "inventory table allocation failed");
    if (hipMemcpyAsync(e.out_unique_dev, uniq.data(), uniq.size() * sizeof(OutDesc), hipMemcpyHostToDevice, s) != hipSuccess ||
        hipMemcpyAsync(e.out_all_dev, all.data(), all.size() * sizeof(OutDesc), hipMemcpyHostToDevice, s) != hipSuccess)
        return fail_free(AMX_EHIP, "inventory table upload failed");
    if (P1 > 0) {
        if (hipMalloc((void**)&e.idx_dev, idx.size() * 8) != hipSuccess ||
            hipMalloc(&e.composed_w, (size_t)P1 * round_up(E, kalign(h)) * 2 * h->NT + PLANE_SLACK) != hipSuccess ||
            hipMalloc((void**)&e.composed_f32, (size_t)P1 * E * 4) != hipSuccess)
            return fail_free(AMX_ENOMEM, "inventory allocation failed");
        if (hipMemcpyAsync(e.idx_dev, idx.data(), idx.size() * 8, hipMemcpyHostToDevice, s) != hipSuccess)
            return fail_free(AMX_EHIP, "inventory upload failed");
        {   // largest entry of the composed matrix (sums of embedding rows), on the host
            float m = 0.f;
            for (int p = 0; p < P1; ++p)
                for (int col = 0; col < E; ++col) {
                    float acc = 0.f;
                    for (int f = 0; f < features; ++f) {
                        const int64_t r = idx[(size_t)p * features + f];
                        if (r >= 0) acc += h->emb_host[(size_t)r * E + col];
                    }
                    m = std::max(m, std::fabs(acc));
                }
            const float ps = pow2_for(m);
            e.r_composed = 1.f / ps;
            const int Eld = round_up(E, kalign(h));  // row stride of the planes; the pad columns are never read (K = E)
            if (Eld != E && hipMemsetAsync(e.composed_w, 0, (size_t)P1 * Eld * 2 * h->NT, s) != hipSuccess)
                return fail_free(AMX_EHIP, "inventory upload failed");
            launch_compose(h->prec, h->emb, E, e.idx_dev, P1, features, ps, e.composed_f32, e.composed_w, pln(h, (int64_t)P1 * Eld), Eld, s);
        }
        if (hipGetLastError() != hipSuccess) return fail_free(AMX_EHIP, "compose kernel launch failed");
    }
    // the uploads above read pageable host vectors that die with this call; a new inventory is a rare event (the cache
    // serves repeats), so the call simply waits for them instead of relying on the runtime's staging behaviour
    if (hipStreamSynchronize(s) != hipSuccess) return fail_free(AMX_EHIP, "inventory upload failed");
    e.key = std::move(key);
    e.generation = ++h->inv_gen;
    select_inventory(h, slot);
    return AMX_OK;
}

extern "C" int amx_set_inventory(amx_handle h, const int64_t* tfi, int phones, int features, const int64_t* offsets,
                                 void* stream) {
    if (!h) return AMX_EINVAL;
    if (h->composed_class < 0) return fail(h, AMX_ESTATE, "model has no embedding composition layer");
    if (!tfi || !offsets || phones < 1 || features < 1) return fail(h, AMX_EINVAL, "bad inventory arguments");
    HIPCHK(h, hipSetDevice(h->device));
    std::vector<int64_t> key;
    key.reserve(2 + (size_t)phones * features + features);
    key.push_back(phones);
    key.push_back(features);
    key.insert(key.end(), tfi, tfi + (size_t)phones * features);
    key.insert(key.end(), offsets, offsets + features);
    for (int i = 0; i < (int)h->inventories.size(); ++i)
        if (h->inventories[i].key == key) {
            select_inventory(h, i);
            return AMX_OK;
        }
    const int P1 = phones + 1;
    std::vector<int64_t> idx((size_t)P1 * features, -1);
    idx[0] = 0;  // blank embedding = row 0 (acoustic_model.py:226-228)
    for (int p = 0; p < phones; ++p)
        for (int f = 0; f < features; ++f) {
            int64_t r = tfi[(size_t)p * features + f] + offsets[f];
            if (r < 0 || r >= h->emb_rows) return fail(h, AMX_EINVAL, "composition feature index out of range of the embedding table");
            idx[(size_t)(p + 1) * features + f] = r;
        }
    return install_inventory(h, std::move(key), idx, P1, features, (hipStream_t)stream);
}

// =================================================================================================================
// layout
// =================================================================================================================
// frontend.py:192-203 applied per conv layer (acoustic_model.py:832-835) with floor division; 0 as soon as a layer's input
// is shorter than its kernel (the reference's formula goes non-positive there)
static int64_t frames_of(const amx_config& c, int64_t len) {
    for (int i = 0; i < c.n_conv; ++i) {
        if (len < c.conv_kernel[i]) return 0;
        len = (len - c.conv_kernel[i]) / c.conv_stride[i] + 1;
    }
    return len;
}

// Largest N for which a batch padded to L samples stays inside the 32-bit offsets of the kernels: in the two-plane modes
// the lo plane of an activation is addressed as (32-bit byte offset of the hi plane element) + (plane bytes), so every
// activation plane plus the largest offset inside a tile must stay below 4 GiB; row indices are 32-bit.
static int64_t max_utterances_for(const amx_config& c, int NT, int64_t L) {
    int64_t Ts[AMX_MAX_CONV + 1];
    Ts[0] = L;
    for (int i = 0; i < c.n_conv; ++i) {
        if (Ts[i] < c.conv_kernel[i]) return 0;
        Ts[i + 1] = (Ts[i] - c.conv_kernel[i]) / c.conv_stride[i] + 1;
    }
    const int64_t T = Ts[c.n_conv], Tp = (T + 63) / 64 * 64;
    const int64_t C = c.conv_dim, D = c.hidden, F = c.ffn, H = c.heads;
    const int64_t wide = std::max<int64_t>(F, 3 * D);
    const int64_t LIMIT = ((int64_t)1 << 32) - ((int64_t)64 << 20);  // 4 GiB minus slack for the offsets inside a tile
    int64_t best = INT32_MAX / std::max<int64_t>(Ts[1], 1);           // 32-bit row indices of the first conv output
    best = std::min(best, (int64_t)INT32_MAX / std::max<int64_t>(T, 1));
    best = std::min(best, ((int64_t)INT32_MAX * 4 / wide) / std::max<int64_t>(T, 1));
    // one plane of every activation stays below 4 GiB in the single-plane modes too: the buffer loads of those kernels use
    // the same 32-bit byte offsets (NT only decides whether a second plane follows)
    (void)NT;
    // conv activations [N * Ts[i], C]: plane + one utterance (tiles crossing an utterance boundary)
    for (int i = 1; i < c.n_conv; ++i) best = std::min(best, LIMIT / (Ts[i] * C * 2) - 1);
    best = std::min(best, LIMIT / (T * wide * 2));        // FFN activation / fused QKV rows
    best = std::min(best, LIMIT / (H * Tp * (D / H > 64 ? 128 : 64) * 2));     // Q / K / V planes (rows padded to 64 / 128 columns)
    best = std::min(best, LIMIT / ((T + c.pos_kernel) * D * 2));  // padded image of the positional convolution
    return std::max<int64_t>(best, 0);
}

extern "C" int64_t amx_max_utterances(amx_handle h, int64_t L) {
    if (!h || L < 1) return 0;
    return max_utterances_for(h->cfg, h->NT, L);
}

static int compute_layout(amx_handle h, int N, int64_t L) {
    if (N < 1 || L < 1) return fail(h, AMX_EINVAL, "empty batch");
    int64_t T = L;
    for (int i = 0; i < h->cfg.n_conv; ++i) {
        if (T < h->cfg.conv_kernel[i]) return fail(h, AMX_EINVAL, "utterances are shorter than the receptive field of the feature extractor");
        T = (T - h->cfg.conv_kernel[i]) / h->cfg.conv_stride[i] + 1;
    }
    if (T > 1 << 20) return fail(h, AMX_EINVAL, "utterance too long");
    if (h->inv < 0)
        return fail(h, AMX_ESTATE, "composition model needs amx_set_inventory before prediction (the training inventory is a non-persistent buffer upstream)");
    if (h->layout_N == N && h->layout_T == T && h->layout_inv == h->inv && h->layout_gen == h->inventories[h->inv].generation)
        return AMX_OK;
    build_tables(h, h->P1, h->col, h->width, h->ld_logits, h->out_unique, h->out_all);
    h->outputs.clear();
    int64_t off = 0;
    for (int ci : h->order) {
        const amx_class_desc& c = h->classes[ci];
        bool is_phoneme = !strcmp(c.name, "phoneme");
        if (is_phoneme && h->cfg.allophone_layer) {
            amx_output_desc d{};
            strncpy(d.name, "phone", AMX_NAME_LEN - 1);
            d.classes = h->width[ci];
            d.offset = off;
            h->outputs.push_back(d);
        }
        amx_output_desc d{};
        strncpy(d.name, c.name, AMX_NAME_LEN - 1);
        d.classes = h->width[ci];
        d.offset = off;
        h->outputs.push_back(d);
        off += (int64_t)T * N * h->width[ci];
    }
    h->layout_N = N;
    h->layout_T = T;
    h->layout_inv = h->inv;
    h->layout_gen = h->inventories[h->inv].generation;
    return AMX_OK;
}

static int64_t layout_total(amx_handle h, int N, int64_t T) {
    int64_t tot = 0;
    for (auto& o : h->out_unique) tot += (int64_t)T * N * o.C;
    return tot;
}

extern "C" int amx_output_layout(amx_handle h, int N, int64_t L, amx_output_desc* descs, int* n_outputs, int64_t* T,
                                 int64_t* total) {
    if (!h) return AMX_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    int rc = compute_layout(h, N, L);
    if (rc) return rc;
    if (n_outputs) *n_outputs = (int)h->outputs.size();
    if (T) *T = h->layout_T;
    if (total) *total = layout_total(h, N, h->layout_T);
    if (descs) memcpy(descs, h->outputs.data(), h->outputs.size() * sizeof(amx_output_desc));
    return AMX_OK;
}

// =================================================================================================================
// forward
// =================================================================================================================
static void drop_graphs(amx_handle h) {
    for (auto& g : h->graphs) {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (g.graph) (void)hipGraphDestroy(g.graph);
    }
    h->graphs.clear();
    h->graph_seen.clear();
}

// Range report of EARLIER forward passes: reads the pinned count of every pass whose trailing event has completed (`wait`:
// of every pass issued -- the caller has just synchronised or accepts to).  Returns AMX_ERANGE when one of them counted valid
// frames with non-finite logits; the report is consumed by the call that returns it.
static int range_poll(amx_handle h, bool wait) {
    // (one handle = one stream, include/allophant_amx.h: passes complete in issue order, so the slots are read oldest first and the
    // first one that is not ready ends the poll)
    int64_t total = 0;
    std::string readings;
    for (int i = 0; i < amx_handle_s::RANGE_SLOTS; ++i) {
        auto& sl = h->range[(h->range_next + i) % amx_handle_s::RANGE_SLOTS];  // oldest first
        if (!sl.pending) continue;
        if (wait) {
            HIPCHK(h, hipEventSynchronize(sl.ev));
        } else {
            const hipError_t q = hipEventQuery(sl.ev);
            if (q == hipErrorNotReady) break;
            if (q != hipSuccess) { (void)hipGetLastError(); break; }
        }
        sl.pending = false;
        // a continuation slice's reading is the chain's running count: what it adds is the difference to its predecessor's reading
        const int reading = *sl.host;
        const int added = sl.cont && h->range_chain_open ? reading - h->range_chain_seen : reading;
        h->range_chain_seen = reading;
        h->range_chain_open = true;
        if (added > 0) {
            total += added;
            readings += (readings.empty() ? "" : ", ") + std::string("pass #") + std::to_string(sl.pass_id) + ": " + std::to_string(added);
        }
    }
    if (h->range_carry > 0) {
        total += h->range_carry;
        readings += (readings.empty() ? "" : ", ") + h->range_carry_ids;
    }
    h->range_carry = 0;
    h->range_carry_ids.clear();
    if (total > 0)
        return fail(h, AMX_ERANGE, std::to_string(total) + " valid frame(s) of an EARLIER forward pass hold non-finite logits: an activation left the "
                                   "range of the 16-bit planes (fp16: |x| <= 65504) or the input was not finite; the outputs of that pass "
                                   "are not usable.  The bf16 planes (precision bf16x3) have the range of fp32; AMX_FLAG_NO_RANGE_CHECK "
                                   "turns this report off (frames per offending pass -- passes are numbered from 1 per handle, the one being "
                                   "issued now would be #" + std::to_string(h->pass_counter + 1) + " -- " + readings + ")");
    return AMX_OK;
}

// enqueues the count of the pass just issued on `s` into the next pinned slot
static int range_record(amx_handle h, bool cont, hipStream_t s) {
    auto& sl = h->range[h->range_next];
    if (!sl.host) {
        HIPCHK(h, hipHostMalloc((void**)&sl.host, 16));
        HIPCHK(h, hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming));
    }
    if (sl.pending) {
        // RANGE_SLOTS passes ago and never read since (the host ran that far ahead): wait for that pass and carry its count
        // into the next report instead of losing it
        HIPCHK(h, hipEventSynchronize(sl.ev));
        const int reading = *sl.host;
        const int added = sl.cont && h->range_chain_open ? reading - h->range_chain_seen : reading;
        h->range_chain_seen = reading;
        h->range_chain_open = true;
        if (added > 0) {
            h->range_carry += added;
            h->range_carry_ids += (h->range_carry_ids.empty() ? "" : ", ") + std::string("pass #") + std::to_string(sl.pass_id) + ": " + std::to_string(added);
        }
        sl.pending = false;
    }
    HIPCHK(h, hipMemcpyAsync(sl.host, h->nonfinite, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipEventRecord(sl.ev, s));
    sl.pending = true;
    sl.cont = cont;
    sl.pass_id = h->pass_counter;
    h->range_next = (h->range_next + 1) % amx_handle_s::RANGE_SLOTS;
    return AMX_OK;
}


namespace {

// split-K workspace (fp32 partial slabs of products too small to fill the chip; see launch_gemm): the partials of one product never
// exceed CUs x 256 x 256 floats
constexpr size_t SPLITK_BYTES = (size_t)72 << 20;

// Everything amx_forward decides before it launches anything -- the PLAN of a pass: geometry, row layout, which optional forms the
// pass takes (packed rows, skipped conv tiles, the LayerNorm fold, the stream kept in planes), every buffer, the products of an
// encoder layer.  A plain value: plan_pass() fills it (and is the only place that may allocate, upload or synchronise),
// enqueue_pass() reads it and issues nothing but kernels -- eagerly on the caller's stream, or once on the capture stream when
// the pass is recorded into a HIP graph.  (Round 6: until then "plan + enqueue" was a lambda inside one 800-line function.)
struct PassPlan {
    amx_handle h = nullptr;
    // the call
    uint32_t flags = 0;
    int N = 0;
    int64_t L = 0;
    const float* d_audio = nullptr;
    float* d_out = nullptr;
    int64_t total = 0;
    // model shorthands
    int NT = 2, prec = 0, C = 0, D = 0, F = 0, H = 0, dh = 64, dhp = 64, E = 0, Eld = 0, cg = 0;
    // geometry
    int64_t Ts[AMX_MAX_CONV + 1] = {};
    int T = 0, Tp = 0, TpTot = 0, Tpad = 0;
    int64_t M = 0, Mp = 0, Mrows = 0, Mh = 0, rows1 = 0, rows2 = 0, xp_plane = 0, qk_plane = 0;
    size_t qkv_bytes = 0;
    // the forms the pass takes
    bool keep = false, masked = true, stable = true, blanks = true;
    bool packed = false, packed_early = false, ragged = false, window_ok = false, needs_qkv_zero = false;
    bool fold = false, stream_in_planes = false;
    // buffers (workspace of the handle: valid until a later plan grows one of them)
    void *d_len = nullptr, *d_frames = nullptr, *d_rowoff = nullptr, *d_partial = nullptr, *d_stats = nullptr, *actA = nullptr, *actB = nullptr,
         *preln = nullptr, *hbuf = nullptr, *xp = nullptr, *hg = nullptr, *qb = nullptr, *kb = nullptr, *vtb = nullptr, *ao = nullptr, *ff = nullptr,
         *hfin = nullptr, *logits = nullptr, *hpk = nullptr, *splitk = nullptr, *gn_partial = nullptr, *gn_scale = nullptr, *gn_shift = nullptr,
         *ebuf = nullptr, *cat = nullptr, *tl_x = nullptr, *tl_p = nullptr, *tl_qkv = nullptr, *ln_rowps = nullptr, *ln_coef = nullptr,
         *ln_partial = nullptr;
    const int* d_frames_enc = nullptr;
    int* d_tiles = nullptr;
    size_t tile_first[AMX_MAX_CONV + 1] = {};
    std::vector<float*> saved;   // per layer: where hidden state l is published (null: nobody reads it)
    float* conv_dbg = nullptr;
    float* hfin_rows = nullptr;
    float* stream = nullptr;     // the residual stream of the layers [Mrows, D]
    // host copies the tail of amx_forward keeps for amx_debug_fetch (packed_early only)
    std::vector<int> rowoff_host, frames_host;

    int64_t plane(int64_t separate) const { return h->il ? PLANE_IL : std::max<int64_t>(separate, 64); }  // pln(h, .)
    GemmParams with_ws(GemmParams g) const {
        g.splitk_ws = (float*)splitk;
        g.splitk_ws_elems = splitk ? (int64_t)(SPLITK_BYTES / 4) : 0;
        return g;
    }
    // ---- the products of an encoder layer ----
    GemmParams qkv_params(const Layer& ly) const {
        GemmParams g{};
        g.A = xp; g.a_plane = xp_plane; g.lda = D; g.rows_per_batch = Mrows;
        g.W = ly.wqkv; g.w_plane = plane((int64_t)3 * D * D); g.ldw = D;
        g.M = (int)Mrows; g.N = 3 * D; g.K = D;
        g.scale = ly.r_qkv; g.bias = ly.bqkv;
        g.mode = 1; g.q = qb; g.k = kb; g.v = vtb;
        g.qk_plane = qk_plane;
        // packed rows: one "utterance" of Mp rows, so the scatter writes row m of head hh to [hh][m][:]
        g.T = packed ? (int)std::max<int64_t>(Mp, 8) : T; g.Tp = packed ? TpTot : Tp; g.H = H; g.dh = dh; g.dhp = dhp;
        return with_ws(g);
    }
    GemmParams oproj_params(const Layer& ly) const {
        GemmParams g{};
        g.A = ao; g.a_plane = xp_plane; g.lda = D; g.rows_per_batch = Mrows;
        g.W = ly.wo; g.w_plane = plane((int64_t)D * D); g.ldw = D;
        g.M = (int)Mrows; g.N = D; g.K = D;
        g.scale = ly.r_o; g.bias = ly.bo;
        g.residual = stream; g.ldr = D; g.out_f32 = stream; g.ldo = D;
        return with_ws(g);
    }
    GemmParams ffn1_params(const Layer& ly) const {
        GemmParams g{};
        g.A = xp; g.a_plane = xp_plane; g.lda = D; g.rows_per_batch = Mrows;
        g.W = ly.w1; g.w_plane = plane((int64_t)F * D); g.ldw = D;
        g.M = (int)Mrows; g.N = F; g.K = D;
        g.scale = ly.r_1; g.bias = ly.b1; g.act = 1;
        g.out_p = ff; g.out_plane = plane(Mrows * F); g.ldp = F;
        return with_ws(g);
    }
    GemmParams ffn2_params(const Layer& ly) const {
        GemmParams g{};
        g.A = ff; g.a_plane = plane(Mrows * F); g.lda = F; g.rows_per_batch = Mrows;
        g.W = ly.w2; g.w_plane = plane((int64_t)D * F); g.ldw = F;
        g.M = (int)Mrows; g.N = D; g.K = F;
        g.scale = ly.r_2; g.bias = ly.b2;
        g.residual = stream; g.ldr = D; g.out_f32 = stream; g.ldo = D;
        return with_ws(g);
    }
    // LayerNorm fold (pre-LN layers; GemmParams.ln_partial / row_coef): the out-projection and FFN2 leave the planes and the row
    // statistics of the stream they have just updated, QKV and FFN1 apply the normalisation in their epilogues -- no LayerNorm
    // pass between the products of a layer.  Only where all four products run on the ping-pong kernel in one piece (batches
    // from a few thousand frames: every benchmark configuration); short batches keep the row pass -- with (1, 0) as its affine
    // part, since the weights hold gamma and beta either way -- fused with the split-K fix-up as before.
    GemmParams as_consumer(GemmParams g, const float* col_c) const {
        g.row_coef = (const float2*)ln_coef; g.col_c = col_c;
        return g;
    }
    // Two-plane modes (stream_in_planes): between the products of a layer the stream lives in its planes only -- a producer reads its
    // residual from them and writes fp32 rows only where something reads those (`f32`: a published hidden state, the last layer:
    // the final LayerNorm and the unpacking read fp32)
    GemmParams as_producer(GemmParams g, bool f32 = true) const {
        g.ln_partial = (float2*)ln_partial; g.ln_rowps = (const float4*)ln_rowps;
        g.out_p = xp; g.out_plane = xp_plane; g.ldp = D;
        if (stream_in_planes) {
            g.ln_res_planes = 1;
            g.residual = nullptr;
            if (!f32) g.out_f32 = nullptr;
        }
        return g;
    }
};

}  // namespace

// The plan region of a forward pass (see PassPlan): argument and geometry checks, pinned uploads of lengths / frame counts / row
// offsets / tile lists, every workspace buffer (ws_get may allocate, free and synchronise), the concatenation recipes of dependent
// classifiers, and the decisions that depend on the lengths.
static int plan_pass(amx_handle h, const float* audio, const int64_t* lengths, int N, int64_t L, float* out, int64_t* out_lengths,
                     uint32_t flags, hipStream_t s, PassPlan& P) {
    int rc;
    const amx_config& c = h->cfg;
    const int NT = h->NT, prec = h->prec;
    const int C = c.conv_dim, D = c.hidden, F = c.ffn, H = c.heads;
    const bool keep = (flags & AMX_FLAG_KEEP_HIDDEN) != 0;
    h->timing = (flags & AMX_FLAG_TIMING) != 0;
    h->timing_stream = s;

    int64_t Ts[AMX_MAX_CONV + 1];
    Ts[0] = L;
    for (int i = 0; i < c.n_conv; ++i) Ts[i + 1] = (Ts[i] - c.conv_kernel[i]) / c.conv_stride[i] + 1;
    const int T = (int)Ts[c.n_conv];
    const int64_t M = (int64_t)N * T;
    {
        const int64_t nmax = max_utterances_for(c, NT, L);
        if (N > nmax)
            return fail(h, AMX_EINVAL, "batch too large for the 32-bit offsets of the kernels: at most " + std::to_string(nmax) +
                                           " utterances of " + std::to_string(L) + " samples per call (amx_max_utterances); split the batch");
    }
    const int Tp = round_up(T, 64);

    int64_t maxlen = 0;
    for (int n = 0; n < N; ++n) {
        if (lengths[n] < 1 || lengths[n] > L) return fail(h, AMX_EINVAL, "lengths must lie in [1, L]");
        maxlen = std::max(maxlen, lengths[n]);
    }
    if (maxlen != L && !(flags & AMX_FLAG_PADDED))
        return fail(h, AMX_EINVAL, "the batch must be padded to exactly max(lengths) (reference utils.py:62-63, acoustic_model.py:765-767)");

    // ---- pinned host staging of lengths (ring of event-guarded slots: no stream synchronisation on the hot path) ----
    if (h->pinned_cap < N) {
        HIPCHK(h, hipStreamSynchronize(s));
        for (int i = 0; i < amx_handle_s::PIN_SLOTS; ++i) {
            if (h->h_lengths_pinned[i]) {
                (void)hipHostFree(h->h_lengths_pinned[i]);
                (void)hipHostFree(h->h_frames_pinned[i]);
                (void)hipHostFree(h->h_rowoff_pinned[i]);
            }
            HIPCHK(h, hipHostMalloc((void**)&h->h_lengths_pinned[i], (size_t)N * 8));
            HIPCHK(h, hipHostMalloc((void**)&h->h_frames_pinned[i], (size_t)N * 4));
            HIPCHK(h, hipHostMalloc((void**)&h->h_rowoff_pinned[i], (size_t)(3 * N + 1) * 4));  // offsets, the order, encoder frames
            if (!h->pin_event[i]) HIPCHK(h, hipEventCreateWithFlags(&h->pin_event[i], hipEventDisableTiming));
            h->pin_busy[i] = false;
        }
        h->pinned_cap = N;
    }
    const int slot = h->pin_next;
    h->pin_next = (h->pin_next + 1) % amx_handle_s::PIN_SLOTS;
    if (h->pin_busy[slot]) HIPCHK(h, hipEventSynchronize(h->pin_event[slot]));  // its copies ran PIN_SLOTS calls ago
    int64_t* pin_len = h->h_lengths_pinned[slot];
    int* pin_frames = h->h_frames_pinned[slot];
    int* pin_rowoff = h->h_rowoff_pinned[slot];
    std::vector<int> conv_rows((size_t)c.n_conv * N);  // [conv layer][n]: valid output rows
    int64_t Mp = 0;  // valid frames of the batch = rows of the packed layout
    for (int n = 0; n < N; ++n) {
        pin_len[n] = lengths[n];
        {
            int64_t len_i = lengths[n];  // valid output rows of every conv layer (floor arithmetic of frontend.py:192-203)
            for (int i = 0; i < c.n_conv; ++i) {
                len_i = len_i < c.conv_kernel[i] ? 0 : (len_i - c.conv_kernel[i]) / c.conv_stride[i] + 1;
                conv_rows[(size_t)i * N + n] = (int)len_i;
            }
        }
        int64_t f = frames_of(c, lengths[n]);
        if (f < 1) return fail(h, AMX_EINVAL, "utterance shorter than the receptive field of the feature extractor");
        pin_frames[n] = (int)f;
        pin_rowoff[n] = (int)Mp;
        Mp += f;
        if (out_lengths) out_lengths[n] = f;
    }
    pin_rowoff[N] = (int)Mp;
    {   // utterances by descending length (ties by index): the order the packed attention dispatches them in
        int* order = pin_rowoff + N + 1;
        for (int n = 0; n < N; ++n) order[n] = n;
        std::stable_sort(order, order + N, [&](int a, int b) { return pin_frames[a] > pin_frames[b]; });
    }
    // The frames the ENCODER treats as valid: the utterance's own, or -- a model whose preprocessor has
    // return_attention_mask = False is called with attention_mask=None (acoustic_model.py:814,842-846) -- every frame of the
    // padded length: nothing is zeroed, every key is attended to.  `Predictions.lengths` are the utterance's own either way.
    const bool masked = h->masked;
    for (int n = 0; n < N; ++n) pin_rowoff[2 * N + 1 + n] = masked ? pin_frames[n] : T;
    if (!masked) Mp = M;  // no padding as far as the encoder is concerned: nothing to pack, nothing to skip

    // ---- workspace ----
    void *d_len, *d_frames, *d_partial, *d_stats, *actA, *actB, *preln, *hbuf, *xp, *hg, *qb, *kb, *vtb, *ao, *ff, *hfin, *logits;
    const int cg = D / c.pos_groups;
    const int Tpad = T + c.pos_kernel;
    const int64_t rows1 = (int64_t)N * Ts[1], rows2 = (int64_t)N * Ts[2];
#define WS(name, bytes, ptr) do { if ((rc = ws_get(h, name, (size_t)(bytes), &ptr))) return rc; } while (0)
    // split-K workspace (fp32 partial slabs of products too small to fill the chip; see launch_gemm): the partials of one
    // product never exceed CUs x 256 x 256 floats
    void* splitk = nullptr;
    static const bool no_splitk = dev_switch("AMX_NO_SPLITK");  // developer A/B switch
    if (!no_splitk) WS("splitk", SPLITK_BYTES, splitk);
    WS("len", (size_t)N * 8, d_len);
    WS("frames", (size_t)N * 4, d_frames);
    void* d_rowoff;
    WS("rowoff", (size_t)(3 * N + 1) * 4, d_rowoff);
    const int* d_frames_enc = (const int*)d_rowoff + 2 * N + 1;
    WS("partial", (size_t)N * 64 * 3 * 8, d_partial);
    WS("stats", (size_t)N * 2 * 4, d_stats);
    WS("actA", (size_t)rows1 * C * 2 * NT + PLANE_SLACK, actA);
    WS("actB", (size_t)rows2 * C * 2 * NT + PLANE_SLACK, actB);
    WS("preln", (size_t)rows2 * C * 4, preln);
    WS("h", (size_t)M * D * 4, hbuf);
    WS("xp", (size_t)M * D * 2 * NT + PLANE_SLACK, xp);
    WS("hg", (size_t)N * Tpad * D * 2 * NT, hg);
    // head dimension and the width of a Q / K / V row: 64 columns, 128 for heads wider than that (the columns beyond dh stay zero:
    // the buffers are zero-filled when they are created and the QKV scatter writes the dh real columns only)
    const int dh = D / H, dhp = dh > 64 ? 128 : 64;
    const size_t qkv_bytes = (size_t)N * H * Tp * dhp * 2 * NT;
    if ((rc = ws_get(h, "q", qkv_bytes, &qb, true))) return rc;
    if ((rc = ws_get(h, "k", qkv_bytes, &kb, true))) return rc;
    if ((rc = ws_get(h, "vt", qkv_bytes, &vtb, true))) return rc;
    WS("ao", (size_t)M * D * 2 * NT + PLANE_SLACK, ao);
    WS("ff", (size_t)M * F * 2 * NT + PLANE_SLACK, ff);
    WS("hfin", (size_t)M * D * 4, hfin);
    WS("logits", (size_t)M * h->ld_logits * 4, logits);
    // Packed rows: a ragged batch runs its encoder layers on the valid frames only (rows of utterance n at row_off[n], all
    // utterances back to back): every kernel of a layer is row-wise except the attention, which takes the offsets.  A row's
    // arithmetic does not depend on its position, so the valid frames come out as in the padded layout (the same bits when
    // the products pick the same kernels for the smaller row count, within rounding of the K-chunk order otherwise).  Used
    // when at least a tenth of the padded rows are padding (not under AMX_FLAG_KEEP_HIDDEN); AMX_FLAG_NO_PACK /
    // AMX_NO_PACKED_ROWS=1 keep the padded layout.
    static const bool no_pack_env = dev_switch("AMX_NO_PACKED_ROWS");
    const bool any_hidden = keep;  // the debug capture wants every hidden state in the padded layout, padding included
    const int TpTot = round_up((int)Mp, 64) + 64;  // rows per head of the packed Q / K / V planes
    bool packed = !no_pack_env && !(flags & AMX_FLAG_NO_PACK) && !any_hidden && D % 4 == 0 && Mp * 10 <= M * 9 &&
                  (int64_t)TpTot <= (int64_t)N * Tp;
    // Packed from the feature projection on ("early"): the last conv layer's LayerNorm pass gathers the valid frames, so the
    // feature projection, the positional convolution (window kernel: skips the frame blocks beyond an utterance), the final
    // LayerNorm, the classifier heads and the log-softmax read packed rows too and nothing is packed or unpacked in between.
    // Needs the window kernel (the grouped-GEMM form of the positional convolution addresses padded rows) and no time-layer
    // head (its attention walks (utterance, frame) pairs); otherwise the rows are packed after the positional convolution
    // and unpacked before the final LayerNorm, as in round 2.
    static const bool no_window = dev_switch("AMX_NO_POSCONV_WINDOW");  // developer A/B switch
    static const bool late_pack = dev_switch("AMX_PACK_LATE");          // developer A/B switch
    const bool window_ok = !no_window && posconv_window_eligible(D, c.pos_groups, c.pos_kernel, N, T, Tpad, (int64_t)N * Tpad * D);
    bool any_time_layer = false;
    for (auto& st : h->steps) any_time_layer |= st.time_heads > 0;
    // (the post-LN encoder has no LayerNorm pass between its last layer and the heads to unpack behind: it packs early or not at all)
    if (!h->stable && !(window_ok && !any_time_layer && !late_pack)) packed = false;
    const bool packed_early = packed && window_ok && !any_time_layer && !late_pack;
    void* hpk = nullptr;
    if (packed) WS("h_packed", (size_t)Mp * D * 4, hpk);
    const float* d_audio = audio;
    float* d_out = out;
    const int64_t total = layout_total(h, N, T);
    if (flags & AMX_FLAG_HOST_IO) {
        void *a, *o;
        WS("audio_host_io", (size_t)N * L * 4, a);
        WS("out_host_io", (size_t)total * 4, o);
        HIPCHK(h, hipMemcpyAsync(a, audio, (size_t)N * L * 4, hipMemcpyHostToDevice, s));
        d_audio = (const float*)a;
        d_out = (float*)o;
    }
    std::vector<float*> saved(c.layers + 1, nullptr);
    for (int l = 0; l < c.layers; ++l)
        if (keep || h->need_hidden[l]) {
            void* p;
            std::string nm = "hid" + std::to_string(l);
            WS(nm.c_str(), (size_t)M * D * 4, p);
            saved[l] = (float*)p;
        }
    saved[c.layers] = (float*)hfin;
    float* conv_dbg = nullptr;
    if (keep) {
        void* p;
        WS("conv_dbg", (size_t)M * C * 4, p);
        conv_dbg = (float*)p;
    }

    HIPCHK(h, hipMemcpyAsync(d_len, pin_len, (size_t)N * 8, hipMemcpyHostToDevice, s));
    HIPCHK(h, hipMemcpyAsync(d_frames, pin_frames, (size_t)N * 4, hipMemcpyHostToDevice, s));
    HIPCHK(h, hipMemcpyAsync(d_rowoff, pin_rowoff, (size_t)(3 * N + 1) * 4, hipMemcpyHostToDevice, s));
    // Ragged batch: the conv stack skips what lies wholly in an utterance's padding (conv0: frame blocks; the row-complete
    // layers 1..n-2: 128-row tiles).  A valid frame of any layer only reads valid frames of the layer below, and the rows
    // left unwritten (stale, possibly non-finite) stay inside padded rows until the feature projection zeroes those.
    // (only from a tenth of padding on -- the threshold of the packed rows: below it the padded frames are simply computed, and
    // NOTHING of the pass depends on the lengths by value any more -- lengths, frame counts and masks are device buffers the plan
    // refreshes -- so a recording of this (N, L) geometry serves every batch of that geometry: see the key below)
    const bool ragged = Mp < M && Mp * 10 <= M * 9 && !keep && !no_pack_env && !(flags & AMX_FLAG_NO_PACK);
    // per conv layer i >= 1: the ascending list of its 128-row output tiles that hold a row some utterance owns
    int* d_tiles = nullptr;
    size_t tile_first[AMX_MAX_CONV + 1] = {0};
    if (ragged) {
        size_t total_tiles = 0;
        constexpr int64_t TR = GEMM_LN_TILE_ROWS;
        for (int i = 1; i < c.n_conv; ++i) total_tiles += (size_t)((N * Ts[i + 1] + TR - 1) / TR);
        if (h->tiles_cap[slot] < total_tiles) {  // the slot is idle: its event was waited for above
            if (h->h_tiles_pinned[slot]) (void)hipHostFree(h->h_tiles_pinned[slot]);
            h->h_tiles_pinned[slot] = nullptr;
            h->tiles_cap[slot] = 0;
            HIPCHK(h, hipHostMalloc((void**)&h->h_tiles_pinned[slot], (total_tiles + total_tiles / 4 + 64) * 4));
            h->tiles_cap[slot] = total_tiles + total_tiles / 4 + 64;
        }
        int* list = h->h_tiles_pinned[slot];
        size_t count = 0;
        for (int i = 1; i < c.n_conv; ++i) {
            tile_first[i] = count;
            const int64_t rpb = Ts[i + 1], rows = N * rpb;
            const int* valid = conv_rows.data() + (size_t)i * N;
            for (int64_t first = 0; first < rows; first += TR) {
                const int64_t last = std::min(first + TR, rows) - 1;
                const int64_t b0 = first / rpb, b1 = last / rpb;
                if (b0 != b1 || first - b0 * rpb < valid[b0]) list[count++] = (int)(first / TR);
            }
        }
        tile_first[c.n_conv] = count;
        void* p;
        WS("conv_tiles", std::max<size_t>(count, 1) * 4, p);
        d_tiles = (int*)p;
        HIPCHK(h, hipMemcpyAsync(d_tiles, list, count * 4, hipMemcpyHostToDevice, s));
    }
    HIPCHK(h, hipEventRecord(h->pin_event[slot], s));
    h->pin_busy[slot] = true;

    // ---- the rest of the plan: every buffer and host-side table the pass needs exists before anything is enqueued, so that
    // the enqueue region below issues nothing but memsets, kernels and device copies (which is what a HIP graph may hold) ----
    void *gn_partial = nullptr, *gn_scale = nullptr, *gn_shift = nullptr;
    if (h->gn) {
        WS("gn_partial", conv0_groupnorm_partial_bytes(N, (int)Ts[1], C), gn_partial);
        WS("gn_scale", (size_t)N * C * 4, gn_scale);
        WS("gn_shift", (size_t)N * C * 4, gn_shift);
    }
    const bool stable = h->stable;
    const int64_t Mh = packed_early ? Mp : M;  // rows of the final LayerNorm and of the classifier heads
    // hidden_states[layers]: the final LayerNorm's fp32 rows; post-LN encoder: the stream as the last layer left it (kept in
    // a buffer of its own only for the debug capture)
    float* const hfin_rows = (stable || keep) ? (float*)hfin : (float*)(packed ? hpk : hbuf);
    saved[c.layers] = hfin_rows;
    const int E = c.embedding_size;
    void *ebuf = nullptr, *cat = nullptr;
    int kcat = 0;
    for (auto& st : h->steps) kcat = std::max(kcat, st.direct_output ? 0 : st.Kpad);
    const int Eld = round_up(E, kalign(h));  // row stride of the embedding planes (pad columns are never read: K = E)
    if (E > 0) WS("e", (size_t)M * Eld * 2 * NT + PLANE_SLACK, ebuf);
    if (kcat > 0) WS("cat", (size_t)M * kcat * 2 * NT + PLANE_SLACK, cat);
    void *tl_x = nullptr, *tl_p = nullptr, *tl_qkv = nullptr;
    {
        int cmax = 0;
        for (auto& st : h->steps)
            if (st.time_heads > 0) {
                cmax = std::max(cmax, st.Cpad);
                if (st.rows / st.time_heads > 4096)
                    return fail(h, AMX_EINVAL, "head_dim of a time-layer classifier beyond 4096");
            }
        if (cmax > 0) {
            WS("tl_x", (size_t)M * cmax * 4, tl_x);
            WS("tl_p", (size_t)M * cmax * 2 * NT + PLANE_SLACK, tl_p);
            WS("tl_qkv", (size_t)M * cmax * 3 * 4, tl_qkv);
        }
    }
    const bool blanks = c.dependency_blanks != 0;
    for (auto& st : h->steps) {
        if (st.direct_output) continue;
        // concatenation recipe of a classifier with dependencies: source pointers / logit columns of this pass
        for (size_t i = 0; i < st.parts.size(); ++i) {
            int dep = st.part_dep[i];
            if (dep < 0) {
                st.parts[i].src = dep == AMX_DEP_OUTPUT ? hfin_rows : saved[-2 - dep];
            } else {
                st.parts[i].src_col = h->col[dep] + (blanks ? 0 : 1);
                // a composed dependency has an inventory-dependent width; the classifier was trained on the training
                // inventory, so the widths must agree
                int w = h->width[dep] - (blanks ? 0 : 1);
                if (w != st.parts[i].width)
                    return fail(h, AMX_EINVAL, "inventory size does not match the input width of a dependent classifier");
            }
        }
        if (st.parts_uploaded.size() != st.parts.size() ||
            memcmp(st.parts_uploaded.data(), st.parts.data(), st.parts.size() * sizeof(ConcatPart)) != 0) {
            // the recipe only changes with the workspace pointers or the inventory; st.parts is pageable host memory,
            // so this (rare) upload completes before the call goes on.  A captured graph never holds it: parts_dev is
            // read by the concat kernel, and a changed recipe comes with a changed workspace generation or inventory.
            HIPCHK(h, hipMemcpyAsync(st.parts_dev, st.parts.data(), st.parts.size() * sizeof(ConcatPart), hipMemcpyHostToDevice, s));
            HIPCHK(h, hipStreamSynchronize(s));
            st.parts_uploaded = st.parts;
        }
    }
    const bool needs_qkv_zero = !packed && (h->last_N != N || h->last_T != T || h->qkv_dirty);
    const int64_t Mrows = packed ? Mp : M;  // rows the layers work on
    const int64_t xp_plane = pln(h, Mrows * D);
    const int64_t qk_plane = packed ? (int64_t)H * TpTot * dhp : (int64_t)N * H * Tp * dhp;
    float* const stream = (float*)(packed ? hpk : hbuf);  // the residual stream of the layers [Mrows, D]
    // ---- the plan as a value ----
    P.h = h;
    P.flags = flags;
    P.N = N;
    P.L = L;
    P.d_audio = d_audio;
    P.d_out = d_out;
    P.total = total;
    P.NT = NT;
    P.prec = prec;
    P.C = C;
    P.D = D;
    P.F = F;
    P.H = H;
    P.T = T;
    P.Tp = Tp;
    P.TpTot = TpTot;
    P.Tpad = Tpad;
    P.cg = cg;
    P.dh = dh;
    P.dhp = dhp;
    P.E = E;
    P.Eld = Eld;
    P.M = M;
    P.Mp = Mp;
    P.Mrows = Mrows;
    P.Mh = Mh;
    P.rows1 = rows1;
    P.rows2 = rows2;
    P.xp_plane = xp_plane;
    P.qk_plane = qk_plane;
    P.qkv_bytes = qkv_bytes;
    P.keep = keep;
    P.masked = masked;
    P.stable = stable;
    P.packed = packed;
    P.packed_early = packed_early;
    P.ragged = ragged;
    P.window_ok = window_ok;
    P.needs_qkv_zero = needs_qkv_zero;
    P.blanks = blanks;
    P.d_len = d_len;
    P.d_frames = d_frames;
    P.d_rowoff = d_rowoff;
    P.d_partial = d_partial;
    P.d_stats = d_stats;
    P.actA = actA;
    P.actB = actB;
    P.preln = preln;
    P.hbuf = hbuf;
    P.xp = xp;
    P.hg = hg;
    P.qb = qb;
    P.kb = kb;
    P.vtb = vtb;
    P.ao = ao;
    P.ff = ff;
    P.hfin = hfin;
    P.logits = logits;
    P.hpk = hpk;
    P.splitk = splitk;
    P.gn_partial = gn_partial;
    P.gn_scale = gn_scale;
    P.gn_shift = gn_shift;
    P.ebuf = ebuf;
    P.cat = cat;
    P.tl_x = tl_x;
    P.tl_p = tl_p;
    P.tl_qkv = tl_qkv;
    P.d_frames_enc = d_frames_enc;
    P.d_tiles = d_tiles;
    P.conv_dbg = conv_dbg;
    P.hfin_rows = hfin_rows;
    P.stream = stream;
    for (int i = 0; i <= AMX_MAX_CONV; ++i) { P.Ts[i] = i <= c.n_conv ? Ts[i] : 0; P.tile_first[i] = tile_first[i]; }
    P.saved = saved;
    if (packed_early) {
        P.rowoff_host.assign(pin_rowoff, pin_rowoff + N);
        P.frames_host.assign(pin_frames, pin_frames + N);
    }
    // LayerNorm fold: decided on the products as they will be launched (AMX_NO_LN_FOLD / AMX_FOLD_F32_STREAM: developer A/B switches)
    static const bool no_ln_fold = dev_switch("AMX_NO_LN_FOLD");
    static const bool f32_stream = dev_switch("AMX_FOLD_F32_STREAM");
    P.stream_in_planes = NT == 2 && !f32_stream;
    if (stable && !no_ln_fold && c.layers > 0 && D % 64 == 0 && D <= 2048) {
        WS("ln_rowps", (size_t)Mrows * 16, P.ln_rowps);
        WS("ln_coef", (size_t)Mrows * 8, P.ln_coef);
        WS("ln_partial", (size_t)Mrows * (D / 64) * 8, P.ln_partial);
        const Layer& ly = h->layers[0];
        P.fold = gemm_ln_fold_ok(prec, P.as_consumer(P.qkv_params(ly), ly.c_qkv)) && gemm_ln_fold_ok(prec, P.as_producer(P.oproj_params(ly))) &&
                 gemm_ln_fold_ok(prec, P.as_consumer(P.ffn1_params(ly), ly.c_1)) && gemm_ln_fold_ok(prec, P.as_producer(P.ffn2_params(ly)));
    }
    ++h->pass_counter;
    h->last_fold = P.fold;
    h->last_packed = packed_early ? 2 : (packed ? 1 : 0);
    h->last_rows = Mrows;
    h->last_graph = 0;
#undef WS
    return AMX_OK;
}

// The enqueue region: the whole pass on stream `s` -- kernels only (zero fills and device copies are kernels too: amx_rowops.hip,
// launch_zero), nothing that allocates, uploads or synchronises.
static int enqueue_pass(amx_handle h, const PassPlan& P, hipStream_t s) {
    const amx_config& c = h->cfg;
    const auto flags = P.flags;
    const auto N = P.N;
    const auto L = P.L;
    const auto d_audio = P.d_audio;
    const auto d_out = P.d_out;
    const auto total = P.total;
    const auto NT = P.NT;
    const auto prec = P.prec;
    const auto C = P.C;
    const auto D = P.D;
    const auto F = P.F;
    const auto H = P.H;
    const auto T = P.T;
    const auto Tp = P.Tp;
    const auto TpTot = P.TpTot;
    const auto Tpad = P.Tpad;
    const auto cg = P.cg;
    const auto dh = P.dh;
    const auto dhp = P.dhp;
    const auto E = P.E;
    const auto Eld = P.Eld;
    const auto M = P.M;
    const auto Mp = P.Mp;
    const auto Mrows = P.Mrows;
    const auto Mh = P.Mh;
    const auto rows1 = P.rows1;
    const auto rows2 = P.rows2;
    const auto xp_plane = P.xp_plane;
    const auto qk_plane = P.qk_plane;
    const auto qkv_bytes = P.qkv_bytes;
    const auto keep = P.keep;
    const auto masked = P.masked;
    const auto stable = P.stable;
    const auto packed = P.packed;
    const auto packed_early = P.packed_early;
    const auto ragged = P.ragged;
    const auto window_ok = P.window_ok;
    const auto needs_qkv_zero = P.needs_qkv_zero;
    const auto blanks = P.blanks;
    const auto d_len = P.d_len;
    const auto d_frames = P.d_frames;
    const auto d_rowoff = P.d_rowoff;
    const auto d_partial = P.d_partial;
    const auto d_stats = P.d_stats;
    const auto actA = P.actA;
    const auto actB = P.actB;
    const auto preln = P.preln;
    void* hbuf = P.hbuf;  // (local: the layers switch it to the packed stream and back)
    const auto xp = P.xp;
    const auto hg = P.hg;
    const auto qb = P.qb;
    const auto kb = P.kb;
    const auto vtb = P.vtb;
    const auto ao = P.ao;
    const auto ff = P.ff;
    const auto hfin = P.hfin;
    const auto logits = P.logits;
    const auto hpk = P.hpk;
    const auto splitk = P.splitk;
    const auto gn_partial = P.gn_partial;
    const auto gn_scale = P.gn_scale;
    const auto gn_shift = P.gn_shift;
    const auto ebuf = P.ebuf;
    const auto cat = P.cat;
    const auto tl_x = P.tl_x;
    const auto tl_p = P.tl_p;
    const auto tl_qkv = P.tl_qkv;
    const auto d_frames_enc = P.d_frames_enc;
    const auto d_tiles = P.d_tiles;
    const auto conv_dbg = P.conv_dbg;
    const auto hfin_rows = P.hfin_rows;
    const auto stream = P.stream;
    const auto& Ts = P.Ts;
    const auto& tile_first = P.tile_first;
    const auto& saved = P.saved;
    const bool fold = P.fold, stream_in_planes = P.stream_in_planes;
    void* const ln_rowps = P.ln_rowps; void* const ln_coef = P.ln_coef; void* const ln_partial = P.ln_partial;
    (void)F; (void)rows2; (void)total; (void)Mh; (void)E; (void)hfin; (void)stream; (void)blanks; (void)ln_coef; (void)stream_in_planes;
    (void)ff; (void)splitk; (void)hfin_rows; (void)qk_plane; (void)dh; (void)Tp; (void)cg; (void)Eld;
    auto run_gemm = [&](int precision, GemmParams& g, hipStream_t stream_) { g = P.with_ws(g); launch_gemm(precision, g, stream_); };
    // K/V/Q padding rows [T, Tp) must stay finite: re-zero when the geometry changes
    // (zero fills inside a pass are kernels, never memsets: amx_rowops.hip, launch_zero)
    if (needs_qkv_zero) {
        launch_zero(qb, qkv_bytes, s);
        launch_zero(kb, qkv_bytes, s);
        launch_zero(vtb, qkv_bytes, s);
    }
    if (packed) {
        // rows [Mp, TpTot) of every head are read (never used) by the last key tile and query block: keep them finite
        const size_t row_b = (size_t)dhp * 2, tail = (size_t)(TpTot - Mp) * row_b, pitch = (size_t)TpTot * row_b;
        for (void* buf : {qb, kb, vtb})  // the planes lie back to back: NT * H blocks of TpTot rows
            launch_zero_2d((char*)buf + (size_t)Mp * row_b, pitch, tail, (size_t)NT * H, s);
    }
    // amx_check_finite reports on THIS forward pass; the later slices of one over-long batch (AMX_FLAG_CONTINUE) add up
    if (!(flags & AMX_FLAG_CONTINUE)) launch_zero(h->nonfinite, 4, s);
    // ---- input normalisation statistics + conv layer 0 (fused norm + conv + LN + GELU) ----
    { Timed t_(h, AMX_KC_OTHER); launch_audio_stats(d_audio, (const int64_t*)d_len, N, L, (double*)d_partial, (float*)d_stats, c.do_normalize, s); }
    if (h->gn) {
        // group-norm variant: GroupNorm statistics over ALL frames of the padded length (upstream normalises the padded batch
        // tensor), then conv + affine + GELU; frame blocks in an utterance's padding may still be skipped in the second pass
        Timed t_(h, AMX_KC_CONV0);
        launch_conv0_groupnorm(prec, d_audio, (const int64_t*)d_len, (const float*)d_stats, N, L, (int)Ts[1], C, c.conv_kernel[0],
                               c.conv_stride[0], h->c0_w, h->conv_b[0], h->conv_g[0], h->conv_be[0], 1e-5f, c.do_normalize,
                               (double*)gn_partial, (float*)gn_scale, (float*)gn_shift, actA, pln(h, rows1 * C), ragged ? 1 : 0, s,
                               conv0_mfma_eligible(C, c.conv_kernel[0], c.conv_stride[0]) ? h->c0_wscale : 0.f);
    } else {
        Timed t_(h, AMX_KC_CONV0);
        launch_conv0(prec, d_audio, (const int64_t*)d_len, (const float*)d_stats, N, L, (int)Ts[1], C, c.conv_kernel[0],
                     c.conv_stride[0], h->c0_w, h->conv_b[0], h->conv_g[0], h->conv_be[0], 1e-5f, c.do_normalize, actA,
                     pln(h, rows1 * C), ragged ? 1 : 0, s, h->c0_stats, h->c0_wscale);
    }
    // ---- conv layers 1..n-1: implicit GEMM over overlapping channels-last windows, then LN + GELU rows ----
    void* cur = actA;
    int64_t cur_plane = pln(h, rows1 * C);
    void* other = actB;
    for (int i = 1; i < c.n_conv; ++i) {
        const int64_t rows_out = (int64_t)N * Ts[i + 1];
        GemmParams g{};
        g.A = cur; g.a_plane = cur_plane; g.lda = (int64_t)c.conv_stride[i] * C; g.rows_per_batch = Ts[i + 1];
        g.a_batch_stride = Ts[i] * C;
        g.W = h->conv_w[i]; g.w_plane = pln(h, (int64_t)C * C * c.conv_kernel[i]); g.ldw = (int64_t)C * c.conv_kernel[i];
        g.M = (int)rows_out; g.N = C; g.K = C * c.conv_kernel[i];
        g.scale = h->conv_r[i]; g.bias = h->conv_b[i];
        if (ragged) {  // honoured by the row-complete kernel only
            g.tile_list = d_tiles + tile_first[i];
            g.n_tiles = (int)(tile_first[i + 1] - tile_first[i]);
        }
        const bool last = i == c.n_conv - 1;
        const int64_t out_plane = pln(h, rows_out * C);
        if (h->gn) {
            // group-norm variant: conv + GELU, no norm (Wav2Vec2NoLayerNormConvLayer): the GELU sits in the product's epilogue
            g.act = 1;
            g.tile_list = nullptr; g.n_tiles = 0;
            if (!last) {
                g.out_p = other; g.out_plane = out_plane; g.ldp = C;
                { Timed t_(h, gemm_class(prec, g)); run_gemm(prec, g, s); }
            } else {
                // last layer: fp32 GELU output, then the feature-projection LayerNorm -> planes (packed_early: valid frames only)
                g.out_f32 = keep ? conv_dbg : (float*)preln; g.ldo = C;
                { Timed t_(h, AMX_KC_CONV_TAIL); run_gemm(prec, g, s); }
                Timed t_(h, AMX_KC_CONV_TAIL);
                if (packed_early)
                    launch_rownorm_to_packed(prec, g.out_f32, C, rows_out, C, nullptr, nullptr, 0, h->fp_g, h->fp_b, 0.f, c.eps, other,
                                             out_plane, C, (const int*)d_rowoff, (const int*)d_frames, T, s);
                else
                    launch_rownorm(prec, g.out_f32, C, rows_out, C, nullptr, nullptr, 0, h->fp_g, h->fp_b, 0.f, c.eps, other, out_plane,
                                   C, nullptr, 0, s);
            }
            std::swap(cur, other);
            cur_plane = out_plane;
            continue;
        }
        if (!last) {
            // LayerNorm + GELU fused into the GEMM epilogue when the row-complete kernel takes the shape (C == 512)
            GemmParams f = g;
            f.act = 1; f.ln_gamma = h->conv_g[i]; f.ln_beta = h->conv_be[i]; f.ln_eps = 1e-5f;
            f.out_p = other; f.out_plane = out_plane; f.ldp = C;
            // AMX_NO_FUSED_CONV_LN=1: developer A/B switch (separate fp32 GEMM output + row kernel)
            static const bool no_fuse = dev_switch("AMX_NO_FUSED_CONV_LN");
            if (!no_fuse && gemm_fuses_ln(prec, f)) {
                if (h->conv_w_tm[i] && gemm_ln_tap_minor_slice(prec, f) == h->conv_tm_slice) {
                    // tap-minor K order: the input row two output rows share is fetched in adjacent slices (an L2 hit)
                    f.W = h->conv_w_tm[i];
                    f.a_taps = c.conv_kernel[i];
                    f.a_tap_stride = C;
                }
                { Timed t_(h, AMX_KC_GEMM_LN); run_gemm(prec, f, s); }
                std::swap(cur, other);
                cur_plane = out_plane;
                continue;
            }
        }
        g.out_f32 = (float*)preln; g.ldo = C;
        { Timed t_(h, last ? AMX_KC_CONV_TAIL : gemm_class(prec, g)); run_gemm(prec, g, s); }
        if (!last) {
            { Timed t_(h, AMX_KC_ROWNORM); launch_rownorm(prec, (const float*)preln, C, rows_out, C, h->conv_g[i], h->conv_be[i], 1, nullptr, nullptr, 1e-5f,
                           0.f, other, out_plane, C, nullptr, 0, s); }
        } else if (!keep) {
            // last conv layer: LN + GELU, then the feature-projection LayerNorm in the same pass (packed_early: only the valid
            // frames, written back to back)
            Timed t_(h, AMX_KC_CONV_TAIL);
            if (packed_early)
                launch_rownorm_to_packed(prec, (const float*)preln, C, rows_out, C, h->conv_g[i], h->conv_be[i], 1, h->fp_g, h->fp_b, 1e-5f,
                                         c.eps, other, out_plane, C, (const int*)d_rowoff, (const int*)d_frames, T, s);
            else
                launch_rownorm(prec, (const float*)preln, C, rows_out, C, h->conv_g[i], h->conv_be[i], 1, h->fp_g, h->fp_b, 1e-5f,
                               c.eps, other, out_plane, C, nullptr, 0, s);
        } else {
            { Timed t_(h, AMX_KC_ROWNORM); launch_rownorm(prec, (const float*)preln, C, rows_out, C, h->conv_g[i], h->conv_be[i], 1, nullptr, nullptr, 1e-5f,
                           0.f, nullptr, 0, 0, conv_dbg, C, s); }
            { Timed t_(h, AMX_KC_ROWNORM); launch_rownorm(prec, conv_dbg, C, rows_out, C, h->fp_g, h->fp_b, 0, nullptr, nullptr, c.eps, 0.f, other, out_plane,
                           C, nullptr, 0, s); }
        }
        std::swap(cur, other);
        cur_plane = out_plane;
    }
    // ---- feature projection (+ zero padded frames) ----
    {
        GemmParams g{};
        const int64_t Mfp = packed_early ? Mp : M;  // packed_early: the valid frames only, no row mask needed
        g.A = cur; g.a_plane = cur_plane; g.lda = C; g.rows_per_batch = Mfp; g.a_batch_stride = 0;
        g.W = h->fp_w; g.w_plane = pln(h, (int64_t)D * C); g.ldw = C;
        g.M = (int)Mfp; g.N = D; g.K = C;
        g.scale = h->fp_r; g.bias = h->fp_bias;
        if (!packed_early && masked) { g.row_len = (const int*)d_frames; g.rows_T = T; }  // hidden_states[~mask] = 0
        g.out_f32 = (float*)(packed_early ? hpk : hbuf); g.ldo = D;
        { Timed t_(h, gemm_class(prec, g)); run_gemm(prec, g, s); }
    }
    // ---- positional conv embedding: h += GELU(grouped conv(h)) ----
    {
        const int* pk_off = packed_early ? (const int*)d_rowoff : nullptr;
        const int* pk_len = packed_early ? (const int*)d_frames : nullptr;
        float* hcur = (float*)(packed_early ? hpk : hbuf);
        { Timed t_(h, AMX_KC_OTHER); launch_posconv_pack(prec, hcur, N, T, D, c.pos_groups, c.pos_kernel / 2, Tpad, hg,
                            (int64_t)N * Tpad * D, pk_off, pk_len, s); }
        // AMX_NO_POSCONV_WINDOW=1: developer A/B switch (grouped implicit GEMM on the tile kernels instead)
        if (window_ok) {
            Timed t_(h, AMX_KC_GEMM_TILE);
            launch_posconv_window(prec, hg, (int64_t)N * Tpad * D, h->pos_w, (int64_t)D * cg * c.pos_kernel,
                                  (int64_t)cg * c.pos_kernel, h->pos_b, h->pos_r, hcur, N, T, Tpad, D, c.pos_groups, c.pos_kernel,
                                  pk_off, pk_len, s);
        } else {
        GemmParams g{};
        g.A = hg; g.a_plane = (int64_t)N * Tpad * D; g.lda = cg; g.rows_per_batch = T; g.a_batch_stride = (int64_t)Tpad * cg;
        g.za = (int64_t)N * Tpad * cg;
        g.W = h->pos_w; g.w_plane = (int64_t)D * cg * c.pos_kernel; g.ldw = (int64_t)cg * c.pos_kernel;
        g.zw = (int64_t)cg * cg * c.pos_kernel;
        g.M = (int)M; g.N = cg; g.K = cg * c.pos_kernel;
        g.scale = h->pos_r; g.bias = h->pos_b; g.zbias = cg; g.act = 1;
        g.residual = (const float*)hbuf; g.ldr = D; g.out_f32 = (float*)hbuf; g.ldo = D; g.zout = cg;
        { Timed t_(h, AMX_KC_GEMM_TILE); launch_gemm_grouped(prec, g, c.pos_groups, s); }
        }
    }
    // ---- transformer encoder (pre-LN) ----
    void* const hpad = hbuf;     // the padded residual stream [N * T, D]
    if (packed) {
        if (!packed_early) { Timed t_(h, AMX_KC_OTHER); launch_pack_rows((const float*)hpad, (float*)hpk, (const int*)d_rowoff, (const int*)d_frames, N, T, D, false, s); }
        hbuf = hpk;
    }
    // A residual product that launch_gemm cuts into K chunks (short batches) leaves its fix-up -- slab sum + bias + residual
    // -> h -- to the LayerNorm that follows it: one kernel instead of the fix-up and a LayerNorm pass that re-reads h
    // (AMX_NO_FUSED_FIXUP=1: developer A/B switch).
    static const bool no_fused_fixup = dev_switch("AMX_NO_FUSED_FIXUP");
    struct { bool on = false; GemmParams g; int splits = 0; } pending;
    auto residual_gemm = [&](GemmParams g, bool may_defer) {
        const int splits = may_defer && !no_fused_fixup ? gemm_planned_splits(prec, g) : 1;
        if (splits > 1 && fixup_rownorm_eligible(g)) {
            g.defer_fixup = 1;
            pending.on = true;
            pending.g = g;
            pending.splits = splits;
        }
        { Timed t_(h, gemm_class(prec, g)); launch_gemm(prec, g, s); }
    };
    // LayerNorm of the residual stream (rows of hbuf) -> planes (and the fp32 rows, for the final one)
    auto stream_norm = [&](const float* gamma, const float* beta, int64_t rows, int64_t plane, float* out_ln) {
        Timed t_(h, AMX_KC_ROWNORM);
        if (pending.on) {
            launch_fixup_rownorm(prec, pending.g, pending.splits, gamma, beta, c.eps, xp, plane, D, out_ln, D, s);
            pending.on = false;
        } else {
            launch_rownorm(prec, (const float*)hbuf, D, rows, D, gamma, beta, 0, nullptr, nullptr, c.eps, 0.f, xp, plane, D, out_ln, D, s);
        }
    };
    // post-LN encoder (Wav2Vec2Encoder): h = LayerNorm(h + pos_conv(h)) first; every layer is h = LN1(h + attention(h)),
    // h = LN2(h + FFN(h)); hidden_states[layers] is the last layer's output as it is.  The LayerNorm passes write the fp32
    // rows back in place (the residual stream IS the normalised tensor) next to the planes the products read.
    float* const ln_inplace = stable ? nullptr : (float*)hbuf;
    if (!stable) stream_norm(h->fln_g, h->fln_b, Mrows, xp_plane, ln_inplace);
    // (pre-LN layers: gamma / beta of both norms live in the QKV / FFN1 weights -- fold_layer_norm at amx_create -- so a row pass
    // normalises with (1, 0))
    auto ln_finalize = [&]() {
        Timed t_(h, AMX_KC_ROWNORM);
        launch_ln_finalize((const float2*)ln_partial, D / 64, Mrows, c.eps, (float4*)ln_rowps, (float2*)ln_coef, s);
    };
    // developer timing switch (WRONG results: every layer runs on layer 0's weights): what weights that are already on the chip are worth
    static const bool share_weights = dev_switch("AMX_DEV_SHARE_LAYER_WEIGHTS");
    for (int l = 0; l < c.layers; ++l) {
        const Layer& ly = h->layers[share_weights ? 0 : l];
        if (fold) {
            // the first norm of the stack from the stream itself; later ones: the previous FFN2 left planes and statistics
            if (l == 0) {
                Timed t_(h, AMX_KC_ROWNORM);
                launch_ln_rowprep(prec, (const float*)hbuf, D, Mrows, D, c.eps, xp, xp_plane, D, (float4*)ln_rowps, (float2*)ln_coef, s);
            }
        } else if (stable) {
            stream_norm(h->unit_g, h->zero_b, Mrows, xp_plane, nullptr);  // (completes the previous layer's FFN2 when that was deferred)
        }
        if (saved[l]) {
            // a classifier reads hidden state l (OUTPUT_l, acoustic_model.py:478-483): padded layout; with packed rows the padded
            // frames take their pre-encoder rows (finite, as meaningless as any padded frame) and the valid ones are scattered in
            // (packed_early: the classifiers read packed rows, the copy is the packed stream as it is)
            if (packed_early) {
                launch_copy(saved[l], hbuf, (size_t)Mp * D * 4, s);  // (a kernel, not a memcpy node: see launch_zero)
            } else {
                launch_copy(saved[l], packed ? hpad : hbuf, (size_t)M * D * 4, s);
                if (packed) launch_pack_rows(saved[l], (float*)hbuf, (const int*)d_rowoff, (const int*)d_frames, N, T, D, true, s);
            }
        }
        {
            GemmParams g = P.qkv_params(ly);
            if (fold) g = P.as_consumer(g, ly.c_qkv);
            { Timed t_(h, gemm_class(prec, g)); launch_gemm(prec, g, s); }
        }
        {
            AttnParams a{};
            a.q = qb; a.k = kb; a.v = vtb;
            a.qk_plane = qk_plane;
            a.out = ao; a.out_plane = xp_plane;
            a.frame_len = d_frames_enc;
            a.N = N; a.H = H; a.T = T; a.Tp = packed ? TpTot : Tp; a.dh = dh; a.dhp = dhp;
            a.row_off = packed ? (const int*)d_rowoff : nullptr;
            a.order = packed ? (const int*)d_rowoff + N + 1 : nullptr;
            { Timed t_(h, AMX_KC_ATTENTION); launch_attention(prec, a, s); }
        }
        if (fold) {
            residual_gemm(P.as_producer(P.oproj_params(ly), false), false);  // (nothing reads the stream between the two halves of a layer)
            ln_finalize();
        } else {
            residual_gemm(P.oproj_params(ly), true);
            // pre-LN: the LayerNorm in front of the FFN (`final_layer_norm`); post-LN: `layer_norm` behind the attention residual
            if (stable) stream_norm(h->unit_g, h->zero_b, Mrows, xp_plane, nullptr);
            else stream_norm(ly.ln1_g, ly.ln1_b, Mrows, xp_plane, ln_inplace);
        }
        {
            GemmParams g = P.ffn1_params(ly);
            if (fold) g = P.as_consumer(g, ly.c_1);
            { Timed t_(h, gemm_class(prec, g)); launch_gemm(prec, g, s); }
        }
        if (fold && (l + 1 < c.layers || stream_in_planes)) {
            // fp32 rows where the next reader needs them: hidden state l + 1 is published, or this is the last layer (final LayerNorm,
            // unpacking); the last layer's planes and statistics have no reader (it is a producer for the residual's sake)
            const bool last = l + 1 == c.layers;
            residual_gemm(P.as_producer(P.ffn2_params(ly), last || saved[l + 1] != nullptr), false);
            if (!last) ln_finalize();
        } else {
            // (the last layer of a packed batch is followed by the unpacking, not by a LayerNorm of these rows)
            residual_gemm(P.ffn2_params(ly), !fold && !(packed && !packed_early && l == c.layers - 1));
        }
        if (!stable) stream_norm(ly.ln2_g, ly.ln2_b, Mrows, xp_plane, ln_inplace);  // `final_layer_norm` closes the post-LN layer
    }
    if (packed && !packed_early) {
        // back to the padded layout for the final LayerNorm and the projection (padded frames keep their pre-encoder rows)
        { Timed t_(h, AMX_KC_OTHER); launch_pack_rows((const float*)hpad, (float*)hpk, (const int*)d_rowoff, (const int*)d_frames, N, T, D, true, s); }
        hbuf = hpad;
    }
    if (stable) {
        stream_norm(h->fln_g, h->fln_b, Mh, pln(h, Mh * D), (float*)hfin);
    } else if (keep) {
        // post-LN: hidden_states[layers] = the stream as the last layer left it (hfin_rows); its planes are in xp already
        launch_copy(hfin, hbuf, (size_t)Mh * D * 4, s);
    }

    // ---- hierarchical projection ----
    for (auto& st : h->steps) {
        const void* A = xp;
        int64_t a_plane = pln(h, Mh * D), lda = D;
        if (!st.direct_output) {
            { Timed t_(h, AMX_KC_OTHER); launch_concat(prec, st.parts_dev, (int)st.parts.size(), (const float*)logits, h->ld_logits, Mh, cat,
                          pln(h, Mh * st.Kpad), st.Kpad, st.Kpad, s); }
            A = cat; a_plane = pln(h, Mh * st.Kpad); lda = st.Kpad;
        }
        GemmParams g{};
        g.A = A; g.a_plane = a_plane; g.lda = lda; g.rows_per_batch = Mh;
        g.W = st.W; g.w_plane = pln(h, (int64_t)st.rows * st.Kpad); g.ldw = st.Kpad;
        g.M = (int)Mh; g.N = st.rows; g.K = st.Kpad;
        g.scale = st.r_w; g.bias = st.bias;
        if (st.time_heads > 0) {
            // ProjectingMultiheadAttention.forward (acoustic_model.py:255-268)
            const int Co = st.rows, Cp = st.Cpad;
            g.out_f32 = (float*)tl_x; g.ldo = Co;
            { Timed t_(h, gemm_class(prec, g)); run_gemm(prec, g, s); }
            { Timed t_(h, AMX_KC_OTHER); launch_time_ln_pe(prec, (const float*)tl_x, Mh, Co, T, st.tl_g, st.tl_b, 1e-5f, st.tl_pe, tl_p,
                              pln(h, Mh * Cp), Cp, s); }
            GemmParams gi{};
            gi.A = tl_p; gi.a_plane = pln(h, Mh * Cp); gi.lda = Cp; gi.rows_per_batch = Mh;
            gi.W = st.tl_win; gi.w_plane = pln(h, (int64_t)3 * Co * Cp); gi.ldw = Cp;
            gi.M = (int)Mh; gi.N = 3 * Co; gi.K = Cp;
            gi.scale = st.r_tin; gi.bias = st.tl_bin;
            gi.out_f32 = (float*)tl_qkv; gi.ldo = 3 * Co;
            { Timed t_(h, gemm_class(prec, gi)); run_gemm(prec, gi, s); }
            { Timed t_(h, AMX_KC_OTHER); launch_time_attention(prec, (const float*)tl_qkv, (const int*)d_frames, N, T, Co, st.time_heads,
                                  tl_p, pln(h, Mh * Cp), Cp, s); }
            // out_proj lands where the plain linear classifier would have written
            g = GemmParams{};
            g.A = tl_p; g.a_plane = pln(h, Mh * Cp); g.lda = Cp; g.rows_per_batch = Mh;
            g.W = st.tl_wout; g.w_plane = pln(h, (int64_t)Co * Cp); g.ldw = Cp;
            g.M = (int)Mh; g.N = Co; g.K = Cp;
            g.scale = st.r_tout; g.bias = st.tl_bout;
        }
        if (st.composed) {
            g.out_p = ebuf; g.out_plane = pln(h, Mh * Eld); g.ldp = Eld;
            { Timed t_(h, gemm_class(prec, g)); run_gemm(prec, g, s); }
            // logits = (e @ composed) / sqrt(E)   (acoustic_model.py:234)
            GemmParams g2{};
            g2.A = ebuf; g2.a_plane = pln(h, Mh * Eld); g2.lda = Eld; g2.rows_per_batch = Mh;
            g2.W = h->composed_w; g2.w_plane = pln(h, (int64_t)h->P1 * Eld); g2.ldw = Eld;
            g2.M = (int)Mh; g2.N = h->P1; g2.K = E;
            g2.scale = h->r_composed / sqrtf((float)E);
            g2.out_f32 = (float*)logits + h->col[st.classes[0]]; g2.ldo = h->ld_logits;
            { Timed t_(h, gemm_class(prec, g2)); run_gemm(prec, g2, s); }
        } else {
            g.out_f32 = (float*)logits + h->col[st.classes[0]]; g.ldo = h->ld_logits;
            { Timed t_(h, gemm_class(prec, g)); run_gemm(prec, g, s); }
        }
    }
    { Timed t_(h, AMX_KC_OTHER); launch_logsoftmax_out(h->out_unique_dev, (int)h->out_unique.size(), (const float*)logits, h->ld_logits, N, T,
                          (const int*)d_frames, (flags & AMX_FLAG_RAW_LOGITS) ? 0 : 1, d_out, h->nonfinite,
                          packed_early ? (const int*)d_rowoff : nullptr, s); }
    HIPCHK(h, hipGetLastError());
    return AMX_OK;
}

extern "C" int amx_forward(amx_handle h, const float* audio, const int64_t* lengths, int N, int64_t L, float* out,
                           int64_t* out_lengths, uint32_t flags, void* stream_) {
    if (!h) return AMX_EINVAL;
    if (!audio || !lengths || !out) return fail(h, AMX_EINVAL, "null buffer");
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream_;
    int rc;
    // Safe by default: the reference computes in fp32 and cannot overflow (estimator.py:1035-1046); the fp16 planes can.  A
    // pass that produced non-finite logits on valid frames is reported -- AMX_ERANGE -- by the first amx_forward /
    // amx_synchronize issued after it has completed; nothing of THIS call has been enqueued when that happens.  No host
    // synchronisation: only passes whose trailing event has completed are read.
    if (!(flags & AMX_FLAG_NO_RANGE_CHECK) && (rc = range_poll(h, false))) return rc;
    rc = compute_layout(h, N, L);
    if (rc) return rc;
    PassPlan P;
    if ((rc = plan_pass(h, audio, lengths, N, L, out, out_lengths, flags, s, P))) return rc;
    const float* const d_audio = P.d_audio;
    float* const d_out = P.d_out;
    const bool needs_qkv_zero = P.needs_qkv_zero, packed = P.packed, ragged = P.ragged, packed_early = P.packed_early, keep = P.keep;
    const int64_t total = P.total;
    const int T = P.T;
    auto enqueue = [&](hipStream_t stream) -> int { return enqueue_pass(h, P, stream); };

    // ---- run the pass: replay its graph, record one, or enqueue it eagerly ----
    const bool graph_ok = !h->graph_broken && !(flags & (AMX_FLAG_NO_GRAPH | AMX_FLAG_TIMING | AMX_FLAG_KEEP_HIDDEN));
    bool done = false;
    if (graph_ok) {
        std::vector<int64_t> key;
        key.reserve(10 + (size_t)N);
        key.insert(key.end(), {(int64_t)(intptr_t)d_audio, (int64_t)(intptr_t)d_out, (int64_t)N, L, (int64_t)flags, (int64_t)needs_qkv_zero,
                               (int64_t)h->inv, (int64_t)h->inventories[h->inv].generation, (int64_t)h->ws_gen});
        // (the stream is part of the key: an executable graph is launched on one stream at a time -- a handle is meant for one
        // stream, include/allophant_amx.h, but a caller that moves to another one gets a recording of its own, not a shared one)
        key.push_back((int64_t)(intptr_t)s);
        // A recording is keyed on GEOMETRY where it can be (ABI 6): in the padded layout without tile skipping every kernel reads
        // lengths / frame counts / masks from the device buffers the plan region has just refreshed (no length travels by value in a
        // kernel node), so batches of one (N, L) with different lengths -- the reference's loop feeds a new batch every iteration,
        // run.py:742-753 -- replay one recording.  Packed rows and skipped conv tiles size grids and tile lists by the lengths:
        // those passes keep them in the key.
        key.push_back(packed ? 2 : (ragged ? 1 : 0));
        if (packed || ragged) key.insert(key.end(), lengths, lengths + N);
        if (!h->graphs.empty() && h->graphs.front().key.size() >= 11 && h->graphs.front().key[8] != (int64_t)h->ws_gen) {
            // a workspace buffer moved since these were recorded (ws_get synchronised the device before freeing it)
            drop_graphs(h);
        }
        for (auto& g : h->graphs)
            if (g.key == key) {
                HIPCHK(h, hipGraphLaunch(g.exec, s));
                g.last_use = ++h->graph_clock;
                g.last_stream = s;
                ++h->graph_replays;
                h->last_graph = 2;
                done = true;
                break;
            }
        const bool seen_before = std::find(h->graph_seen.begin(), h->graph_seen.end(), key) != h->graph_seen.end();
        if (!done && seen_before) {
            // this key ran eagerly a moment ago: record it.  Capture runs on a private stream (the caller's may be the null
            // stream, which cannot be captured) in thread-local mode; nothing executes until the launch below.
            if (!h->capture_stream) HIPCHK(h, hipStreamCreateWithFlags(&h->capture_stream, hipStreamNonBlocking));
            hipGraph_t graph = nullptr;
            hipGraphExec_t exec = nullptr;
            if (hipStreamBeginCapture(h->capture_stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                const int erc = enqueue(h->capture_stream);
                const hipError_t end = hipStreamEndCapture(h->capture_stream, &graph);
                if (erc) {
                    if (graph) (void)hipGraphDestroy(graph);
                    return erc;
                }
                if (end == hipSuccess && graph && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess) {
                    if ((int)h->graphs.size() >= amx_handle_s::GRAPH_CAP) {
                        // evict the least recently used recording once nothing on the stream can still be running it
                        size_t lru = 0;
                        for (size_t i = 1; i < h->graphs.size(); ++i)
                            if (h->graphs[i].last_use < h->graphs[lru].last_use) lru = i;
                        HIPCHK(h, hipStreamSynchronize(h->graphs[lru].last_stream));  // (its last launch may be on another stream than `s`)
                        (void)hipGraphExecDestroy(h->graphs[lru].exec);
                        if (h->graphs[lru].graph) (void)hipGraphDestroy(h->graphs[lru].graph);
                        h->graphs.erase(h->graphs.begin() + (long)lru);
                    }
                    HIPCHK(h, hipGraphLaunch(exec, s));
                    amx_handle_s::GraphEntry entry;
                    entry.key = key;
                    entry.exec = exec;
                    entry.graph = graph;
                    // AMX_GRAPH_DROP_TEMPLATE=1 (developer switch, tools/r06_graph_fault.sh): the round-5 code before its fix --
                    // the template destroyed right after instantiation, only the executable kept
                    static const bool drop_template = dev_switch("AMX_GRAPH_DROP_TEMPLATE");
                    if (drop_template) {
                        (void)hipGraphDestroy(graph);
                        entry.graph = nullptr;
                    }
                    entry.last_use = ++h->graph_clock;
                    entry.last_stream = s;
                    h->graphs.push_back(std::move(entry));
                    ++h->graph_captures;
                    h->last_graph = 1;
                    done = true;
                } else {
                    if (graph) (void)hipGraphDestroy(graph);
                    (void)hipGetLastError();
                    h->graph_broken = true;  // this runtime does not record the pass: stay eager from here on
                }
            } else {
                (void)hipGetLastError();
                h->graph_broken = true;
            }
        }
        if (!done && !seen_before) {
            if ((int)h->graph_seen.size() >= amx_handle_s::GRAPH_SEEN) h->graph_seen.erase(h->graph_seen.begin());
            h->graph_seen.push_back(std::move(key));
        }
    }
    if (!done && (rc = enqueue(s))) return rc;
    // the pass's non-finite frame count travels to a pinned slot behind it; the next call reads it (see range_poll)
    if (!(flags & AMX_FLAG_NO_RANGE_CHECK) && (rc = range_record(h, (flags & AMX_FLAG_CONTINUE) != 0, s))) return rc;
    if (flags & AMX_FLAG_HOST_IO) {
        HIPCHK(h, hipMemcpyAsync(out, d_out, (size_t)total * 4, hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipStreamSynchronize(s));
    }
    bool report_now = false;
    if ((flags & AMX_FLAG_HOST_IO) && !(flags & AMX_FLAG_NO_RANGE_CHECK)) report_now = true;
    h->last_N = N; h->last_L = L; h->last_T = T; h->last_keep = keep;
    h->last_packed_rows = packed_early;
    if (packed_early) {
        h->last_rowoff = P.rowoff_host;
        h->last_frames = P.frames_host;
    }
    h->qkv_dirty = packed;  // a packed call leaves other rows in the Q / K / V planes: the next padded call re-zeroes them
    // a host-I/O call has synchronised and handed the outputs over already: its own range report does not wait for the next call
    if (report_now && (rc = range_poll(h, true))) return rc;
    return AMX_OK;
}

extern "C" int amx_pass_info(amx_handle h, int32_t* info, int n) {
    if (!h || !info || n < 0) return AMX_EINVAL;
    const int32_t values[AMX_PASS_INFO_COUNT] = {h->last_fold ? 1 : 0, h->last_packed, h->last_graph,
                                                 (int32_t)std::min<int64_t>(h->last_rows, INT32_MAX), (int32_t)(h->pass_counter & 0x7fffffff)};
    for (int i = 0; i < n && i < AMX_PASS_INFO_COUNT; ++i) info[i] = values[i];
    return AMX_OK;
}

extern "C" int amx_synchronize(amx_handle h, void* stream) {
    if (!h) return AMX_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize((hipStream_t)stream));
    HIPCHK(h, hipGetLastError());
    return range_poll(h, true);  // every pass issued has completed: AMX_ERANGE if one of them left the range of the planes
}

extern "C" int amx_check_finite(amx_handle h, void* stream, int64_t* frames) {
    if (!h) return AMX_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    int count = 0;
    // read and reset IN STREAM ORDER: callers pass non-blocking streams, which the null stream does not order against, so a
    // reset on the null stream could land after a later forward pass had started counting
    HIPCHK(h, hipMemcpyAsync(&count, h->nonfinite, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipMemsetAsync(h->nonfinite, 0, 4, s));  // a check closes its reporting period
    HIPCHK(h, hipStreamSynchronize(s));
    if (frames) *frames = count;
    // an explicit check also consumes the pending reports of earlier passes (the stream has drained: their slots are complete)
    const int earlier = range_poll(h, true);
    if (count == 0 && earlier) return earlier;
    if (count > 0)
        return fail(h, AMX_ERANGE, std::to_string(count) + " valid frame(s) of the last forward pass hold non-finite logits: an activation left the "
                                   "range of the 16-bit planes (fp16: |x| <= 65504) or the input was not finite; the bf16 planes "
                                   "(precision bf16x3) have the range of fp32");
    return AMX_OK;
}

extern "C" int amx_graph_info(amx_handle h, int64_t* captures, int64_t* replays) {
    if (!h) return AMX_EINVAL;
    if (captures) *captures = h->graph_captures;
    if (replays) *replays = h->graph_replays;
    return AMX_OK;
}

extern "C" int amx_greedy_ctc(amx_handle h, const float* out, const int64_t* frame_lengths, int N, int64_t L,
                              int64_t* tokens, int64_t* timesteps, int32_t* counts, float* scores, void* stream) {
    if (!h || !out || !frame_lengths || !tokens || !timesteps || !counts || !scores) return AMX_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t s = (hipStream_t)stream;
    int rc = compute_layout(h, N, L);  // output block geometry is a function of (N, L, inventory) only
    if (rc) return rc;
    const int T = (int)h->layout_T;
    std::vector<int> fl(N);
    for (int n = 0; n < N; ++n) {
        if (frame_lengths[n] < 0 || frame_lengths[n] > T) return fail(h, AMX_EINVAL, "frame length out of range");
        fl[n] = (int)frame_lengths[n];
    }
    void* d_fl;
    if ((rc = ws_get(h, "ctc_frames", (size_t)N * 4, &d_fl))) return rc;
    HIPCHK(h, hipMemcpyAsync(d_fl, fl.data(), (size_t)N * 4, hipMemcpyHostToDevice, s));
    HIPCHK(h, hipStreamSynchronize(s));  // fl is pageable host memory
    launch_greedy_ctc(h->out_all_dev, (int)h->out_all.size(), out, (const int*)d_fl, N, T, tokens, timesteps, counts, scores, s);
    HIPCHK(h, hipGetLastError());
    return AMX_OK;
}

extern "C" int amx_greedy_ctc_emissions(int device, const float* emissions, int64_t stride_n, int64_t stride_t,
                                        const int32_t* frame_lengths, int N, int64_t T, int C, int blank_index,
                                        int64_t* tokens, int64_t* timesteps, int32_t* counts, float* scores, void* stream) {
    if (!emissions || !frame_lengths || !tokens || !timesteps || !counts || !scores) return fail(nullptr, AMX_EINVAL, "null buffer");
    if (N < 1 || T < 1 || C < 1) return fail(nullptr, AMX_EINVAL, "empty emission tensor");
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, AMX_EHIP, "hipSetDevice failed");
    launch_greedy_ctc_emissions(emissions, stride_n, stride_t, frame_lengths, N, (int)T, C, blank_index, tokens, timesteps, counts,
                                scores, (hipStream_t)stream);
    if (hipGetLastError() != hipSuccess) return fail(nullptr, AMX_EHIP, "greedy CTC kernel launch failed");
    return AMX_OK;
}

extern "C" int amx_timing_fetch(amx_handle h, float* ms, int32_t* launches, int n_classes) {
    if (!h || !ms || !launches || n_classes < AMX_KC_COUNT) return AMX_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipStreamSynchronize(h->timing_stream));
    for (int i = 0; i < n_classes; ++i) { ms[i] = 0.f; launches[i] = 0; }
    for (auto& sp : h->spans) {
        float t = 0.f;
        HIPCHK(h, hipEventElapsedTime(&t, sp.a, sp.b));
        ms[sp.cls] += t;
        launches[sp.cls] += 1;
        h->event_pool.push_back(sp.a);
        h->event_pool.push_back(sp.b);
    }
    h->spans.clear();
    return AMX_OK;
}

extern "C" int amx_debug_fetch(amx_handle h, int what, int index, float* host_out, int64_t capacity, int64_t* ld_out) {
    if (!h || !host_out) return AMX_EINVAL;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipDeviceSynchronize());
    const int64_t M = (int64_t)h->last_N * h->last_T;
    const void* src = nullptr;
    int64_t n = 0;
    if (what == 0) {
        if (!h->last_keep) return fail(h, AMX_ESTATE, "conv output is only kept with AMX_FLAG_KEEP_HIDDEN");
        src = h->ws["conv_dbg"].p; n = M * h->cfg.conv_dim;
    } else if (what == 1) {
        if (index < 0 || index > h->cfg.layers) return fail(h, AMX_EINVAL, "hidden state index out of range");
        std::string nm = index == h->cfg.layers ? "hfin" : "hid" + std::to_string(index);
        if (index == h->cfg.layers && !h->stable && !h->last_keep)
            return fail(h, AMX_ESTATE, "the post-LN encoder keeps its last hidden state for the debug fetch only with AMX_FLAG_KEEP_HIDDEN");
        if (!h->ws.count(nm) || !h->ws[nm].p || (index < h->cfg.layers && !h->last_keep && !h->need_hidden[index]))
            return fail(h, AMX_ESTATE, "hidden state was not kept (AMX_FLAG_KEEP_HIDDEN)");
        src = h->ws[nm].p; n = M * h->cfg.hidden;
    } else if (what == 2) {
        src = h->ws["logits"].p; n = M * h->ld_logits;
        if (ld_out) *ld_out = h->ld_logits;
    } else if (what == 3) {
        // raw workspace bytes (developer diagnostics): index 0 actA, 1 actB, 2 preln
        const char* names[3] = {"actA", "actB", "preln"};
        if (index < 0 || index > 2) return fail(h, AMX_EINVAL, "bad raw buffer index");
        auto& w = h->ws[names[index]];
        src = w.p;
        n = std::min<int64_t>(capacity, (int64_t)(w.bytes / 4));
        if (ld_out) *ld_out = (int64_t)w.bytes;
    } else {
        return fail(h, AMX_EINVAL, "unknown debug item");
    }
    if (!src) return fail(h, AMX_ESTATE, "nothing to fetch before the first amx_forward");
    if (capacity < n) return fail(h, AMX_EINVAL, "debug buffer too small");
    if (h->last_packed_rows && (what == 1 || what == 2)) {
        // the buffer holds the packed rows of a ragged batch (utterance n at last_rowoff[n]): hand them out in the padded
        // [N, T] layout the caller expects, frames beyond an utterance as zeros
        const int64_t ld = n / M;
        int64_t Mp = 0;
        for (int f : h->last_frames) Mp += f;
        std::vector<float> rows((size_t)Mp * ld);
        HIPCHK(h, hipMemcpy(rows.data(), src, rows.size() * 4, hipMemcpyDeviceToHost));
        std::fill(host_out, host_out + n, 0.f);
        for (int u = 0; u < h->last_N; ++u)
            std::memcpy(host_out + (int64_t)u * h->last_T * ld, rows.data() + (int64_t)h->last_rowoff[u] * ld,
                        (size_t)h->last_frames[u] * ld * 4);
        return AMX_OK;
    }
    HIPCHK(h, hipMemcpy(host_out, src, (size_t)n * 4, hipMemcpyDeviceToHost));
    return AMX_OK;
}
