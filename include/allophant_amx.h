/* liballophant_amx -- C ABI of the MI355X-native Allophant acoustic-encoder forward path.
 *
 * The reference (kgnlp/allophant) has no FFI for its network: the only native boundary upstream is the pyo3 module
 * `allophant.phonemes` (src/lib.rs:9-18), which is not on this path.  The drop-in boundary is therefore the Python-level
 * call `Estimator.predict(batch, target_feature_indices, log_probabilities)` (allophant/estimator.py:1035-1046) on a
 * model restored by `Estimator.restore` (estimator.py:1085-1126).  Every entry point below names the reference interface
 * it replaces; the reference-side binding (ctypes) is shown in INTEGRATION.md and implemented in
 * allophant_amd/lib.py + allophant_amd/estimator.py.
 *
 * Conventions (following the reference's own FFI habits): every call returns 0 on success or a
 * negative AMX_E* code, with a message retrievable through amx_last_error(); handles are not re-entrant (one handle per
 * GPU / stream, like the single-threaded `torch.inference_mode` caller upstream); the library owns packed weights and
 * workspace, the caller owns input and output buffers.  No torch types appear in any signature.
 */
#ifndef ALLOPHANT_AMX_H
#define ALLOPHANT_AMX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AMX_ABI_VERSION 6

#define AMX_MAX_CONV 8
#define AMX_MAX_DEPS 64
#define AMX_NAME_LEN 48

/* error codes */
#define AMX_OK 0
#define AMX_EINVAL (-1)   /* bad argument / unsupported configuration (the reference raises ValueError) */
#define AMX_EHIP (-2)     /* HIP runtime failure */
#define AMX_ESTATE (-3)   /* call order violated (e.g. composition model without an inventory) */
#define AMX_ENOMEM (-4)
#define AMX_ERANGE (-5)   /* an activation left the range of the 16-bit planes (non-finite logits on valid frames): returned by
                           * the first amx_forward / amx_synchronize after the offending pass has completed (ABI 5: safe by
                           * default), and by amx_check_finite */

/* arithmetic modes of the GEMM-shaped products (activations between kernels are 16-bit planes, residual stream,
 * LayerNorm, softmax and all accumulation are fp32):
 *   BF16 / F16      one 16-bit plane per operand
 *   BF16X3 / F16X3  hi/lo split planes, 3 MFMAs per product.  F16X3 is the parity mode (max-abs log-prob error 4-9e-5
 *                   against the reference at XLS-R shape, all greedy alignments equal); BF16X3 has the range of fp32 but
 *                   16 mantissa bits per product: 4e-4, inside the 1e-3 gate, one of 76 golden alignments differs */
#define AMX_PREC_BF16 0
#define AMX_PREC_F16 1
#define AMX_PREC_BF16X3 2
#define AMX_PREC_F16X3 3

#define AMX_NORM_LAYER 0
#define AMX_NORM_GROUP 1

/* amx_forward flags */
#define AMX_FLAG_HOST_IO 1u      /* audio / out are host pointers; the library stages them over PCIe */
#define AMX_FLAG_RAW_LOGITS 2u   /* `log_probabilities=False` of Estimator.predict (estimator.py:1037,1040-1046) */
#define AMX_FLAG_KEEP_HIDDEN 4u  /* keep every encoder hidden state for amx_debug_fetch */
#define AMX_FLAG_TIMING 8u       /* bracket every kernel launch with HIP events on the launch stream (amx_timing_fetch) */
#define AMX_FLAG_PADDED 16u      /* L may exceed max(lengths): the call is one slice of a larger batch padded to L (the
                                    reference itself requires L == max(lengths), utils.py:62-63) */
#define AMX_FLAG_CONTINUE 64u    /* the call continues the range-check count of the previous call instead of restarting it: slices
                                  * 2.. of one over-long batch (amx_check_finite then reports on the whole batch) */
#define AMX_FLAG_NO_GRAPH 128u   /* enqueue the pass launch by launch even when a HIP graph of it exists or could be recorded */
#define AMX_FLAG_NO_RANGE_CHECK 256u /* do not report non-finite logits of this pass through a later call (amx_check_finite still does) */
#define AMX_FLAG_NO_PACK 32u     /* keep the padded [N, T] row layout through the encoder layers even for a ragged batch (the
                                  * default runs them on the valid frames only; results on valid frames are identical) */

/* kernel classes reported by amx_timing_fetch */
#define AMX_KC_GEMM_PP 0   /* gemm_pp_kernel<T, NT, MI>: ping-pong GEMM on 256x256 / 128x256 tiles -- feature projection, QKV/out/FFN, wide heads */
#define AMX_KC_GEMM_TILE 1 /* gemm_kernel<T, NT, 128, {128,64}>: grouped positional conv, narrow heads, shapes the ping-pong kernel rejects */
#define AMX_KC_ATTENTION 2
#define AMX_KC_ROWNORM 3
#define AMX_KC_CONV0 4
#define AMX_KC_OTHER 5
#define AMX_KC_GEMM_LN 6   /* gemm_ln_kernel<T, NT>: row-complete 128x512 GEMM with fused LayerNorm + GELU -- conv layers 1-5 */
#define AMX_KC_CONV_TAIL 7 /* last conv layer: its GEMM (gemm_pp_kernel<T, NT, 4>) and the LayerNorm + GELU + feature-projection LayerNorm rows */
#define AMX_KC_COUNT 8

/* dependency codes in amx_class_desc.deps */
#define AMX_DEP_OUTPUT (-1)                 /* "OUTPUT"   (allophant/config.py:636) */
#define AMX_DEP_OUTPUT_LAYER(i) (-2 - (i))  /* "OUTPUT_i" (allophant/config.py:637, acoustic_model.py:478-483) */

typedef struct amx_handle_s* amx_handle;

/* Shape of the wav2vec 2.0 encoder (`transformers.Wav2Vec2Config` read at acoustic_model.py:818-826) and of the
 * projection (`ProjectionConfig`, allophant/config.py:679-712). */
typedef struct amx_config {
    int32_t abi_version;                 /* AMX_ABI_VERSION */
    int32_t n_conv;                      /* feature-extractor conv layers (7) */
    int32_t conv_dim;                    /* 512 */
    int32_t conv_kernel[AMX_MAX_CONV];   /* 10,3,3,3,3,2,2 */
    int32_t conv_stride[AMX_MAX_CONV];   /* 5,2,2,2,2,2,2 */
    int32_t hidden;                      /* 1024 (a multiple of 8, at most 2048: XLS-R 1B / 2B have 1280 / 1920; ABI 6) */
    int32_t layers;                      /* 24 */
    int32_t heads;                       /* 16 (head_dim = hidden / heads: a multiple of 8, at most 128 -- 64 for every released
                                            checkpoint, 80 / 120 for XLS-R 1B / 2B shapes; ABI 6) */
    int32_t ffn;                         /* 4096 */
    int32_t pos_kernel;                  /* 128 */
    int32_t pos_groups;                  /* 16 (hidden / pos_groups: a multiple of 8, at most 128) */
    float eps;                           /* layer_norm_eps 1e-5 */
    int32_t do_normalize;                /* preprocessor do_normalize (acoustic_model.py:815,841-843) */
    int32_t dependency_blanks;           /* ProjectionConfig.dependency_blanks */
    int32_t embedding_size;              /* EmbeddingCompositionConfig.embedding_size, 0 = no composition layer */
    int32_t allophone_layer;             /* 1: predict mode also publishes "phone" (acoustic_model.py:161-167) */
    int32_t precision;                   /* AMX_PREC_* */
    /* ABI 4: the wav2vec 2.0 variant (`Wav2Vec2Config` fields the reference passes through untouched: it builds whatever
     * `model_id` names, acoustic_model.py:775-826).  XLS-R / every released Allophant checkpoint: 0, 1, 1, 1. */
    int32_t feat_extract_norm;           /* AMX_NORM_LAYER: LayerNorm over channels behind every conv layer; AMX_NORM_GROUP:
                                            GroupNorm(conv_dim groups) over time behind conv layer 0 only (wav2vec2-base/-large) */
    int32_t conv_bias;                   /* config.conv_bias: the conv layers carry a bias */
    int32_t stable_layer_norm;           /* config.do_stable_layer_norm: 1 = pre-LN layers + final LayerNorm
                                            (Wav2Vec2EncoderStableLayerNorm), 0 = LayerNorm behind the positional convolution and
                                            post-LN layers (Wav2Vec2Encoder) */
    int32_t use_attention_mask;          /* preprocessor return_attention_mask (acoustic_model.py:814,842-846): 0 = the model is
                                            called with attention_mask=None -- padded frames are neither zeroed nor masked as
                                            keys; `Predictions.lengths` are the downsampled lengths either way */
} amx_config;

/* One classifier of the hierarchical projection (`ProjectionEntryConfig` / `AttributeNode`,
 * allophant/config.py:624-644, allophant/attribute_graph.py:17-41), in configuration order. */
typedef struct amx_class_desc {
    char name[AMX_NAME_LEN];
    int32_t size;                   /* classes without the CTC blank */
    int32_t out_features;           /* rows of `_time_distributed_layer.weight` (size+1, or embedding_size) */
    int32_t n_deps;
    int32_t deps[AMX_MAX_DEPS];     /* >= 0: index of another class; AMX_DEP_OUTPUT; AMX_DEP_OUTPUT_LAYER(i) */
    /* `time_layer` = MultiheadAttentionConfig (allophant/config.py:596-610): 0 = plain nn.Linear; > 0 = the classifier is a
     * ProjectingMultiheadAttention (acoustic_model.py:237-268): Linear -> LayerNorm -> (+ sinusoidal positions) ->
     * nn.MultiheadAttention(out_features, time_heads) over time with the key-padding mask of the frame lengths */
    int32_t time_heads;
    int32_t time_positional;        /* add SinusoidalPositionEmbeddings (acoustic_model.py:34-69) */
} amx_class_desc;

/* One tensor of `Allophant.state_dict()` (= `Checkpoint.model_state`, estimator.py:216), host fp32, reference key
 * names.  The library packs them itself: conv weights to tap-major, weight-norm folded into
 * the positional conv, Q/K/V fused with the 1/sqrt(head_dim) scale, hi/lo 16-bit planes. */
typedef struct amx_tensor {
    const char* name;
    const float* data;
    int64_t numel;
} amx_tensor;

/* One entry of `Predictions.outputs` (acoustic_model.py:908-926): a [T, N, C] time-major fp32 block at `offset` floats
 * into the output buffer.  "phone" and "phoneme" share one block, as they share one tensor upstream. */
typedef struct amx_output_desc {
    char name[AMX_NAME_LEN];
    int32_t classes;   /* C (incl. blank) */
    int64_t offset;
} amx_output_desc;

/* Replaces `Estimator.restore` + `Allophant.from_config` + `load_state_dict` (estimator.py:1085-1126,
 * acoustic_model.py:988-1025): builds the device-resident model on `device`. */
int amx_create(amx_handle* out, int device, const amx_config* config, const amx_class_desc* classes, int n_classes,
               const amx_tensor* tensors, int n_tensors);
int amx_destroy(amx_handle h);
const char* amx_last_error(amx_handle h); /* h may be NULL for errors of amx_create */

/* Replaces the `target_feature_indices` argument of `Estimator.predict` / `EmbeddingCompositionLayer.forward`
 * (acoustic_model.py:219-234): `tfi` is the int64 [P, F] `composition_feature_matrix`
 * (phonetic_features.py:808-818), `category_offsets` the int64 [F] buffer `_category_offsets`
 * (acoustic_model.py:196-207, 214-217).  Host pointers.  Stays in effect until replaced.  The composed phoneme matrix of a
 * new inventory is built on `stream` (the stream of the following amx_forward calls) into buffers of its own; the library
 * keeps the matrices of the last 16 distinct inventories, so the per-language loop of the reference (run.py:742-753: one
 * `feature_matrix` per batch) neither recomputes them nor races a forward pass still in flight. */
int amx_set_inventory(amx_handle h, const int64_t* tfi, int phones, int features, const int64_t* category_offsets,
                      void* stream);

/* Output geometry for a batch of N utterances padded to L samples: T = frames of the padded length,
 * `total` = floats the caller must provide to amx_forward.  `descs` may be NULL to query `n_outputs` only. */
int amx_output_layout(amx_handle h, int N, int64_t L, amx_output_desc* descs, int* n_outputs, int64_t* T,
                      int64_t* total);

/* Largest N amx_forward accepts for utterances padded to L samples: the kernels address an activation plane with 32-bit
 * byte offsets, so in the two-plane modes every plane must stay below 4 GiB (e.g. 22 x 60 s of conv-0 output).  Larger
 * batches are refused with AMX_EINVAL; the caller runs them as slices of at most this many utterances with
 * AMX_FLAG_PADDED (allophant_amd/estimator.py does). */
int64_t amx_max_utterances(amx_handle h, int64_t L);

/* Replaces `Estimator.predict(batch, tfi, log_probabilities)` (estimator.py:1035-1046):
 *   audio        fp32 [N, L], zero right-padded to L == max(lengths) (batching.py:174, utils.py:62-63); device pointer
 *                unless AMX_FLAG_HOST_IO
 *   lengths      int64 [N] valid samples per utterance, HOST pointer
 *   out          fp32 `total` floats laid out per amx_output_layout; device pointer unless AMX_FLAG_HOST_IO
 *   out_lengths  int64 [N] frames per utterance (`Predictions.lengths`), HOST pointer
 *   stream       hipStream_t (NULL = default stream).  Work is enqueued asynchronously; call amx_synchronize or
 *                synchronize the stream before reading `out` (with AMX_FLAG_HOST_IO the call returns synchronised).
 *
 * Launch collapse (ABI 5): a pass whose buffers, geometry, lengths, flags and inventory equal those of one of the last few passes is
 * recorded into a HIP graph and replayed from then on -- one hipGraphLaunch instead of ~185 kernel launches (the host side of a
 * step drops from ~12 to ~2 us per kernel); results are bitwise those of the eager pass.  AMX_FLAG_NO_GRAPH opts out;
 * AMX_FLAG_TIMING / AMX_FLAG_KEEP_HIDDEN passes are never recorded.
 *
 * Range report (ABI 5, safe by default): the reference computes in fp32 and cannot overflow; the fp16 planes can (|x| <=
 * 65504).  A pass that produced non-finite logits on valid frames makes the FIRST amx_forward / amx_synchronize issued after
 * it has completed return AMX_ERANGE (that amx_forward enqueues nothing; call it again after handling the report).  No host
 * synchronisation is added to the hot path: a call only reads the pinned counters of passes that have already finished. */
int amx_forward(amx_handle h, const float* audio, const int64_t* lengths, int N, int64_t L, float* out,
                int64_t* out_lengths, uint32_t flags, void* stream);
int amx_synchronize(amx_handle h, void* stream);
/* number of forward passes recorded into HIP graphs / replayed from one so far (measurement and test hook) */
int amx_graph_info(amx_handle h, int64_t* captures, int64_t* replays);

/* Which of the plan's optional forms the last amx_forward took (ABI 6; measurement and test hook -- results do not depend on them
 * beyond rounding): fills info[0 .. min(n, AMX_PASS_INFO_COUNT)). */
#define AMX_PASS_INFO_LN_FOLD 0   /* 1: pre-LN encoder layers with their LayerNorm passes folded into the products around them */
#define AMX_PASS_INFO_PACKED 1    /* 0: padded rows; 1: encoder layers on the valid frames only; 2: packed from the feature projection on */
#define AMX_PASS_INFO_GRAPH 2     /* 0: enqueued launch by launch; 1: recorded into a HIP graph by this call; 2: replayed from one */
#define AMX_PASS_INFO_ROWS 3      /* rows (frames) the encoder layers worked on */
#define AMX_PASS_INFO_ID 4        /* number of the pass among the amx_forward calls of this handle (from 1; low 31 bits): an AMX_ERANGE
                                     report names the offending pass by this number */
#define AMX_PASS_INFO_COUNT 5
int amx_pass_info(amx_handle h, int32_t* info, int n);

/* Range check of the last amx_forward on `stream` (no upstream counterpart: the reference computes in fp32).  The 16-bit
 * planes of the fp16 modes hold |x| <= 65504; weights are packed under a per-tensor power-of-two scale, so only an
 * ACTIVATION (or a non-finite input sample) can leave that range, and it then reaches the logits as an infinity or a NaN.
 * Waits for `stream`, stores the number of valid frames with non-finite logits in *frames (may be NULL) and returns AMX_OK
 * when there are none, AMX_ERANGE otherwise (amx_last_error names the remedy: precision bf16x3 has the range of fp32).
 * The count covers the last amx_forward; calls with AMX_FLAG_CONTINUE (the later slices of one batch) add to it instead of
 * restarting it, and a check resets it (in stream order). */
int amx_check_finite(amx_handle h, void* stream, int64_t* frames);

/* Replaces `GreedyCTCDecoder.__call__` (predictions.py:194-207), applied to every output of a prediction as the
 * reference's decode loop does (run.py:767-774): `out` is the device output buffer amx_forward filled for a batch of
 * geometry (N, L) under the current inventory, `frame_lengths` the int64 [N] HOST `Predictions.lengths`.
 * tokens/timesteps are int64 [n_outputs, N, T] (first counts[o,n] entries valid, timesteps 1-based), counts int32
 * [n_outputs, N], scores fp32 [n_outputs, N] (sum of the per-frame maxima); all four are DEVICE pointers. */
int amx_greedy_ctc(amx_handle h, const float* out, const int64_t* frame_lengths, int N, int64_t L, int64_t* tokens,
                   int64_t* timesteps, int32_t* counts, float* scores, void* stream);

/* `GreedyCTCDecoder.__call__(log_emissions, lengths)` with the reference's own signature (predictions.py:194-207): one
 * fp32 emission tensor [N, T, C] on `device` with element strides (stride_n, stride_t, 1) -- the contiguous transpose the
 * reference's loop builds (run.py:770-773) or a strided view of a [T, N, C] output -- and int32 [N] DEVICE frame lengths.
 * tokens / timesteps: int64 [N, T] (first counts[n] entries valid, timesteps 1-based), counts int32 [N], scores fp32 [N]
 * (sum of the per-frame maxima over the valid frames); all DEVICE pointers.  `blank_index` is the constructor argument of
 * the reference class (0 everywhere upstream, config.py:555).  Needs no handle, like the reference class. */
int amx_greedy_ctc_emissions(int device, const float* emissions, int64_t stride_n, int64_t stride_t,
                             const int32_t* frame_lengths, int N, int64_t T, int C, int blank_index, int64_t* tokens,
                             int64_t* timesteps, int32_t* counts, float* scores, void* stream);

/* Test hook: copies an intermediate of the last amx_forward to host fp32.
 *   what = 0: conv feature extractor output [N, T, conv_dim] (after the last GELU)
 *   what = 1: hidden_states[index] [N, T, hidden]  (needs AMX_FLAG_KEEP_HIDDEN; index == layers is the final LayerNorm)
 *   what = 2: raw logits buffer [N*T, ld] (index ignored); returns ld through *ld_out */
int amx_debug_fetch(amx_handle h, int what, int index, float* host_out, int64_t capacity, int64_t* ld_out);

/* Measurement hook: sums the HIP-event durations (ms) and launch counts per kernel class (AMX_KC_*) of every
 * amx_forward issued with AMX_FLAG_TIMING since the previous fetch; synchronises the stream. */
int amx_timing_fetch(amx_handle h, float* ms, int32_t* launches, int n_classes);

/* number of bytes of device memory held by the handle (weights + workspace) */
int64_t amx_device_bytes(amx_handle h);

/* The exchange step of utterance-level data parallelism (no upstream counterpart: the reference is single-device,
 * run.py:576-580) for hosts that do not use allophant_amd/parallel.py: one process per GPU, every rank
 * runs amx_forward on its contiguous block of utterances and calls this with the SAME `count` (floats of its output block,
 * amx_output_layout of the shard geometry; equal shards) and `n_local` (utterances of a shard).  The root rank receives the
 * blocks in rank order into `recv` (world x count floats) and the frame lengths into `recv_lengths` (world x n_local int64);
 * block r holds, per output, the [T, n_local, C] tensor of rank r's utterances at the offsets amx_output_layout reports.
 * `send`, `recv`, `send_lengths`, `recv_lengths` are DEVICE pointers (copy `out_lengths` of amx_forward to the device first);
 * `recv` / `recv_lengths` may be NULL on the other ranks.  `nccl_comm` is an ncclComm_t the caller created (ncclCommInitRank, one
 * communicator per process); the work is enqueued on `stream` behind the forward pass, RCCL moves it over xGMI.  The library does
 * not link RCCL: it uses the ncclSend / ncclRecv of the RCCL already loaded in the process -- the one the communicator came from
 * (AMX_ESTATE if there is none, or if the library AMX_RCCL_LIBRARY names does not load).  A non-zero return on ANY rank must
 * abort the job: the arguments only one rank can check (the root's receive buffers) fail there alone, and the sends the other
 * ranks have already enqueued then never match.  Padding-sensitive models (feat_extract_norm = group or use_attention_mask = 0):
 * every rank must run amx_forward with the padded length L of the GLOBAL batch and AMX_FLAG_PADDED, else the shards are not
 * the function the single-device call computes.  Errors: amx_dist_last_error(). */
int amx_gather_outputs(void* nccl_comm, int rank, int world, int root, const float* send, int64_t count, float* recv,
                       const int64_t* send_lengths, int n_local, int64_t* recv_lengths, void* stream);
const char* amx_dist_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* ALLOPHANT_AMX_H */
