/*
 * savad.h -- C ABI of libsavad.so: the MI355X (gfx950) self-attentive-VAD forward pass.
 *
 * This is the drop-in boundary for ONE path of voithru/voice-activity-detection: the call
 *     model_outputs = self.model(features=batch_inputs["feature"])        vad/predictor.py:224
 * i.e. SelfAttentiveVAD.forward (vad/models/self_attention.py:23-28, blocks in
 * vad/modeling/transformer.py), plus the two host loops either side of it in
 * VADFromScratchPredictor.predict_probabilities: the window gather (vad/predictor.py:180-220)
 * and the boosted-prediction scatter/softmax/mean (vad/predictor.py:238-258 and :95).
 *
 * The reference has no FFI (it is pure Python + torch.nn); the reference-side binding a
 * maintainer would add is the ctypes stub shown in INTEGRATION.md -- in this repo it is
 * voice_activity_detection_amd/_lib.py + model.py (an nn.Module with the reference's
 * constructor signature and state_dict keys whose forward() calls savad_forward).
 *
 * Conventions: plain pointers and sizes only; all tensor pointers are DEVICE pointers unless
 * stated; row-major contiguous; fp32.  Every call returns 0 on success or a negative
 * SAVAD_E_* code, with a thread-local message in savad_last_error().  Handles share nothing: use one handle per stream that has a forward in
 * flight (two or three batches in flight on as many streams fill the CU slots a single forward leaves idle: +25 % batches per second at
 * [32,800,80] fp32 -- voice_activity_detection_amd/pipeline.py does exactly that), never one handle from two streams at once.  Kernels are enqueued
 * on the caller's HIP stream (`stream` is a hipStream_t passed as void*; NULL = default
 * stream).  savad_forward allocates no device memory and never synchronises (the caller supplies the
 * workspace) for every T covered by an earlier savad_reserve(h, T_max); a longer T first grows the
 * library's positional-encoding table on that call (one hipMalloc + one stream synchronisation, the
 * counterpart of the reference's cache growth: vad/modeling/transformer.py:392-397), and the first
 * forward after savad_set_param enqueues the weight re-packing kernels on `stream`.
 */
#ifndef SAVAD_H
#define SAVAD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAVAD_OK 0
#define SAVAD_E_INVALID (-1)     /* bad argument / shape */
#define SAVAD_E_UNSUPPORTED (-2) /* configuration outside what the gfx950 kernels implement */
#define SAVAD_E_HIP (-3)         /* HIP runtime error (message carries hipGetErrorString) */
#define SAVAD_E_STATE (-4)       /* e.g. forward before every parameter was set */
#define SAVAD_E_NOKEY (-5)       /* unknown state_dict key */

typedef struct savad_model* savad_handle;

/* Mirrors SelfAttentiveVAD.__init__(feature_size, num_layers, d_model, dropout)
 * (vad/models/self_attention.py:7-21; d_ff = 4*d_model :10, n_heads = 1 :18).  dropout is an
 * inference no-op and is not part of the ABI.  Any even d_model in [2, 4096] (the reference's positional encoding pairs
 * sin / cos columns) and any feature_size (rows are zero-padded to a multiple of 16 internally when needed).
 * d_model = 128 -- the only value the reference ships (tests/configs/vad/train_config.yaml:7-10) -- runs the tuned MFMA
 * kernels in fp32 or bf16; every other width runs plain fp32 kernels (csrc/savad_generic.h: same results to the fp32
 * tolerance, several times slower per FLOP, savad_set_precision(h, 1) is refused with SAVAD_E_UNSUPPORTED, and
 * savad_set_attention_splits(h, n > 1) there means "n query tiles per sequence"). */
typedef struct savad_config {
    int32_t feature_size;
    int32_t num_layers;
    int32_t d_model;
} savad_config;

/* Replaces SelfAttentiveVAD(...) construction + .to(device) (vad/predictor.py:277-279).
 * Allocates the library-owned packed weights on the CURRENT HIP device. */
int savad_create(const savad_config* cfg, savad_handle* out);
void savad_destroy(savad_handle h);

/* Replaces load_state_dict(checkpoint["state_dict"]) (vad/predictor.py:278), one tensor per call.
 * `key` is the reference state_dict key (e.g. "encoder.layers.0.self_attention.query_projection.weight");
 * `data` may be a host or a device pointer (hipMemcpyDefault); `numel` must match the key's shape.
 * The copy is enqueued on `stream`; the library keeps its own packed copy (LayerNorm affine
 * parameters are folded into the following Linear at the next forward). */
int savad_set_param(savad_handle h, const char* key, const float* data, size_t numel, void* stream);
/* state_dict introspection: number of keys, i-th key, its element count. */
int savad_num_params(savad_handle h);
const char* savad_param_key(savad_handle h, int i);
size_t savad_param_numel(savad_handle h, int i);

/* Pre-sizes the positional-encoding table (vad/modeling/transformer.py:392-397: the reference grows its cache
 * when T exceeds it) for sequences of up to T_max frames, so that savad_forward with T <= T_max neither
 * allocates nor synchronises.  May allocate and synchronise `stream` itself.  Once every parameter has been set
 * (savad_set_param) it also folds / packs the weights for the precision selected by savad_set_precision, so that the
 * FIRST forward after it enqueues nothing but its own kernels (HIP-graph capturable). */
int savad_reserve(savad_handle h, int T_max, void* stream);

/* Bytes of scratch savad_forward needs for a [B,T,F] batch (activations + attention partials). */
int savad_workspace_bytes(savad_handle h, int B, int T, size_t* bytes);

/* Replaces self.model(features=x) (vad/predictor.py:224): x [B,T,F] fp32 -> out [B,T,2] fp32
 * log-probabilities (LogSoftmax(dim=2), vad/models/self_attention.py:26-27).  Asynchronous on
 * `stream`.  B == 0 or T == 0 is a no-op.  Masks do not exist on this path (always None in the
 * reference: vad/modeling/transformer.py:24). */
int savad_forward(savad_handle h, const float* x, int B, int T, float* out, void* workspace, size_t workspace_bytes,
                  void* stream);

/* Arithmetic of the forward pass: 0 = fp32 operands on the exact-fp32 MFMA (default; log-probs within
 * 1e-4 of the reference), 1 = bf16 operands (weights, Q/K/V, probabilities, FFN activations) with fp32
 * accumulation, fp32 softmax / LayerNorm statistics, residual stream fp32 in registers / fp16 in memory (BASELINE.json
 * configs[2..3]; judged on AUC, not on 1e-4), 2 = "fp32s": the fp32 arithmetic of vad/modeling/transformer.py:281-284,
 * 347,351-363,370-375 on the bf16 matrix pipe -- every GEMM operand kept as three bf16 pieces (hi + mid + lo = the fp32 value,
 * exactly), six bf16 MFMA products per K-step into one fp32 accumulator, everything else fp32 (softmax, LayerNorm, residual
 * stream): the same 1e-4 bar as precision 0, at 6/16 of its matrix time (gfx950 has no TF32; csrc/savad_kernels_f32s.h). */
int savad_set_precision(savad_handle h, int precision);
/* bf16 precision only: the residual stream lives in HBM as fp16 between kernels and saturates at +-65504 where the
 * reference's fp32 stream (vad/modeling/transformer.py:234-238) would not.  Saturation is counted, not silent: this
 * returns the number of residual elements clamped (or non-finite) since the previous call and clears the counter;
 * it synchronises `stream`.  0 on every parity workload; a model that reports > 0 needs precision 0 (fp32). */
int savad_residual_saturations(savad_handle h, unsigned long long* count, void* stream);
/* savad_forward with an explicit feature dtype: x_dtype 0 = fp32 [B,T,F], 1 = bf16 [B,T,F] (bf16
 * precision only).  Output is always fp32 log-probabilities. */
int savad_forward_ex(savad_handle h, const void* x, int x_dtype, int B, int T, float* out, void* workspace,
                     size_t workspace_bytes, void* stream);
/* The same with sequences that OVERLAP in memory: sequence b starts x_batch_stride elements behind sequence b-1
 * (x_batch_stride = T * feature_size is savad_forward_ex).  The streaming mode reads its windows [hop w, hop w + T) in place out of
 * the feature matrix with x_batch_stride = hop * feature_size -- no window copies (T > 32, feature_size a multiple of 16). */
int savad_forward_strided(savad_handle h, const void* x, int x_dtype, int B, int T, long x_batch_stride, float* out,
                          void* workspace, size_t workspace_bytes, void* stream);

/* Tuning knob: number of key-range splits of the attention stage (0 = automatic). */
int savad_set_attention_splits(savad_handle h, int splits);
/* Tuning knob: the launch schedule (results differ only in summation order; default 0 = automatic).
 *   fp32 operands
 *     1  row-wise stages on 32-row tiles, output features split over the workgroup's 4 waves; attention is its own
 *        launch (what automatic picks for small T > 32 batches)
 *     4  T <= 32: the whole forward in ONE launch, a workgroup per packed tile of floor(32/T) sequences (what
 *        automatic picks up to 1024 tiles unless the dense 32-row tiles of mode 1 fit fewer rounds of the CUs);
 *        T > 32: as 0
 *     2  row-wise stages on 128-row tiles, weight stream shared through LDS; attention and row stages are
 *        separate launches
 *     3  as 2, with attention and row chain of a query-block group fused into one launch per layer whenever
 *        T > 32 and the key range is not split (what automatic picks for large batches)
 *   bf16 operands
 *     1  separate attention / row launches, 4-wave workgroups      2  the same with 8-wave workgroups
 *     3  fused launches (T > 32)                                    0  fused up to ~4 workgroups per CU
 *     5  separate launches with the attention stage as ONE persistent launch (4 waves x 64 query rows per CU walking
 *        (sequence, 8 query blocks) items; csrc/savad_attn_pw_bf16.h).  Same bits as 1 except for the frames of a sequence's
 *        tail group of one or two query blocks (ceil(T / 32) mod 8 in {1, 2}), whose keys are summed as four partial softmaxes
 *        (key-split item): those agree with 1 to the bf16 rounding of the context.  Automatic picks it where a cost model of
 *        both attention kernels (savad.hip, pw_pays) has it ahead: large batches of long sequences ([160+,800], [256,1000] ...)
 *     0, 4 and 5 run the input stage, from one 32-row block per CU up, as ONE persistent launch with its weights resident in LDS
 *        (input_qkv_kernel_bf16_p; the same bits as the ring kernel 1 - 3 keep)
 *   bf16 operands, T <= 32 (the reference pipeline's 7-frame windows): 0 and 4 run the WHOLE forward in one launch (a wave per
 *        packed block for all layers, csrc/savad_packed_bf16.h; same bits as the per-layer launches of 1 - 3) in the variant
 *        the number of blocks suggests; 5 / 6 / 7 / 8 pin a variant (8-wave workgroups / 4 waves + 4 that move the weight stream
 *        through a 4-slot ring / 4 waves with a 2-slot ring / ONE block per workgroup, its four waves splitting every GEMM's
 *        output features: the latency variant, picked up to one block per CU)
 *   fp32s (split-bf16 operands), T <= 32: 0 and 4 run the WHOLE forward in one launch (csrc/savad_kernels_f32s.h) -- the latency
 *        variant (ONE packed block per workgroup, its four waves splitting every GEMM's output features, weights straight from L2
 *        into registers) up to two blocks per CU, a wave per block with the weight stream shared through the LDS ring beyond;
 *        8 pins the latency variant, 5 - 7 the wave-per-block one, 1 - 3 keep the per-layer launches.  The two variants agree to
 *        fp32 rounding (one fp32 sum is taken in another order), each is deterministic and independent of the batch. */
int savad_set_row_mode(savad_handle h, int mode);
/* bf16 operands: results that do not depend on the batch a sequence is evaluated in (0 = off, the default; 1 = on).  Every bf16 launch
 * schedule computes a frame with the same arithmetic, bit for bit -- with one exception: the persistent attention kernel that automatic
 * picks for large batches of long sequences sums the keys of a sequence's tail group of one or two query blocks as four partial
 * softmaxes (row_mode 5 above), so the same window can differ in the bf16 rounding of the context (<= 3e-3 in the log-probs) between a
 * 256-sequence batch and a 100-sequence remainder.  On: that kernel runs its tail groups as ordinary items (a second generated
 * instruction stream, csrc/savad_attn_pw_bf16_nosplit.inc) and every schedule -- hence every batching, chunk size and shard size --
 * gives the same bits, at about 9 % of that launch (3 - 4 % of a [256,800,80] forward).  fp32 operands: no effect (there a sequence's
 * result depends on its batch only through the automatic key-split count, <= 2e-6; savad_set_attention_splits(h, 1) pins that).
 * The reference itself (BLAS blocking) is not batch-invariant at the bit level. */
int savad_set_batch_invariant(savad_handle h, int on);
/* Per-kernel timing (bench.py's roofline block).  savad_set_profiling(h, capacity): the next `capacity` calls of
 * savad_forward bracket every launch with hipEvents on `stream` (capacity 0 switches profiling off and frees the
 * events); savad_profiling_skip(h, n): the next n forwards run un-recorded first (event creation idles the GPU for
 * a moment and the first ~15 ms of kernels afterwards run at lower clocks: 230 -> 207 us measured);
 * savad_last_kernel_times: after the caller has synchronised the stream, the launch names and their average duration
 * (ms) over the recorded forwards; returns the number of kernels written (<= max) and starts a new recording. */
int savad_set_profiling(savad_handle h, int capacity);
int savad_profiling_skip(savad_handle h, int forwards);
int savad_last_kernel_times(savad_handle h, const char** names, float* ms, int max);

/* Window geometry of the predictor: W = 2*(half-1)/jump + 3 (vad/predictor.py:57-59); offsets =
 * arange(-half,0,jump) ++ [0] ++ arange(1,half+1,jump) (vad/predictor.py:186-212). Returns W;
 * offsets (host pointer, >= W ints) may be NULL. */
int savad_window_offsets(int half, int jump, int32_t* offsets);

/* Replaces the per-window python gather loop (vad/predictor.py:182-220): for item i in
 * [first, first+count): windows[i-first][w][:] = feature[half+i+off[w]][:], positions[i-first][w] =
 * half+i+off[w].  feature [N,F] fp32, windows [count,W,F] fp32, positions [count,W] int64 (may be NULL). */
int savad_gather_windows(const float* feature, int N, int F, int half, int jump, int first, int count, float* windows,
                         int64_t* positions, void* stream);

/* The whole of VADFromScratchPredictor.predict_probabilities (vad/predictor.py:159-262) in one call: feature [N,F] fp32
 * (device) -> probs [N,W] (the boosted positive-class probabilities, unfilled slots exactly 0.5) and mean [N] (may be
 * NULL) = probs.mean(axis=1) (vad/predictor.py:95).  Windows are cut as savad_gather_windows does, `chunk` windows per
 * forward (the reference's chunk_size, :180); results are those of savad_gather_windows + savad_forward + savad_boost
 * (bit for bit when the forwards run the same launch schedule).  With fp32 arithmetic, W <= 32 and at most 1024 packed
 * tiles (4096 windows of 7 frames) the forward reads its windows straight out of the feature matrix (no window copies,
 * no positions, no scatter): two launches for the whole clip.  Asynchronous on `stream`, no allocation;
 * workspace from savad_predict_workspace_bytes (same N, half, jump, chunk and the handle's current precision). */
int savad_predict_workspace_bytes(savad_handle h, int N, int half, int jump, int chunk, size_t* bytes);
int savad_predict_probabilities(savad_handle h, const float* feature, int N, int half, int jump, int chunk, float* probs,
                                float* mean, void* workspace, size_t workspace_bytes, void* stream);

/* Replaces the boosted prediction (vad/predictor.py:238-258 and :95):
 *   boosted[N][W][2] = 0; boosted[positions[b][w]][w] = logp[b][w]  (scatter)
 *   probs[n][w] = softmax(boosted[n][w])[1]  (unfilled slots = exactly 0.5, and averaged in)
 *   mean[n] = mean_w probs[n][w]
 * logp [count,W,2] fp32, positions [count,W] int64, boosted_ws [N,W,2] fp32 scratch,
 * probs [N,W] fp32, mean [N] fp32 (may be NULL). */
int savad_boost(const float* logp, const int64_t* positions, int count, int N, int W, float* boosted_ws, float* probs,
                float* mean, void* stream);

/* Streaming long-form mode (BASELINE.json configs[4]: sliding windows T=800, hop=400; NOT a mode of
 * the reference, whose predictor only cuts 7-frame windows -- the rule below is this build's):
 * window w covers frames [hop*w, hop*w+T) of feature[N,F], zero-padded past N; the number of windows
 * is savad_stream_window_count = N <= T ? 1 : ceil((N-T)/hop) + 1.  After savad_forward on the
 * windows, savad_overlap_merge gives probs[n] = mean over the windows covering frame n of
 * softmax(logp[w][n-hop*w])[1] (mirrors the averaging of vad/predictor.py:95). */
int savad_stream_window_count(int N, int T, int hop);
int savad_gather_strided(const float* feature, int N, int F, int T, int hop, int first, int count, float* windows,
                         void* stream);
int savad_overlap_merge(const float* logp, int W, int N, int T, int hop, float* probs, void* stream);

/* Log-mel front-end (next-row 1 of the scope table): replaces, for the reference's only transform
 * configuration (n_fft 512, hop 160, window 400, 80 mels, 16 kHz: tests/configs/vad/train_config.yaml:
 * 18-26), FeatureExtractor.extract_with_postprocessing = librosa.feature.melspectrogram(...) ->
 * log(x + 1e-6) -> transpose (vad/acoustics/transforms/log_mel_spectrogram.py:19-32,
 * vad/acoustics/feature_extractor.py:71-80).  audio: n_samples fp32 mono 16 kHz (device);
 * features: [savad_logmel_frames(n_samples) = 1 + n_samples/160][80] fp32 (device), ready to be the
 * model's / predictor's feature matrix; workspace: savad_logmel_workspace_bytes(n_samples) bytes. */
int savad_logmel_frames(int n_samples);
size_t savad_logmel_workspace_bytes(int n_samples);
int savad_logmel(const float* audio, int n_samples, float* workspace, float* features, void* stream);
/* The same for a SPAN of the frames -- what one rank of a sharded run computes (each rank only the frames its
 * windows cover).  audio points at sample audio_first of a signal of n_samples samples and holds audio_count of them;
 * frames [frame_first, frame_first + frame_count) are written to features[frame_count][80].  The slice must hold the
 * samples savad_logmel_span_samples names for these frames (160 f - 208 .. 160 f + 207 of every frame, plus the
 * mirrored stretch when the span touches an end of the signal: centre=True, reflect padding); *first is a multiple of
 * 4, so that a slice cut there keeps the 16-byte alignment of the direct reads (any other slice is copied first).
 * workspace: savad_logmel_span_workspace_bytes(frame_count) bytes, 16-byte aligned. */
int savad_logmel_span_samples(long n_samples, int frame_first, int frame_count, long* first, long* count);
/* 16-bit PCM (device) -> float32 samples in [-1, 1): sample / 32768, exactly what the reference's AudioData.load obtains from
 * soundfile for a PCM16 source (vad/data_models/audio_data.py:21-24,32).  For uploads from the host: PCM16 is half the bytes
 * of the float signal the log-mel entry points take (StreamingPredictor.predict_audio_host).  pcm: 2-byte aligned (8-byte
 * aligned slices move four samples per load); audio: 16-byte aligned. */
int savad_pcm16_to_f32(const short* pcm, long n_samples, float* audio, void* stream);
size_t savad_logmel_span_workspace_bytes(int frame_count);
int savad_logmel_span(const float* audio, long audio_first, long audio_count, long n_samples, int frame_first,
                      int frame_count, float* workspace, float* features, void* stream);
/* Experiments and tests: algorithm 0 (default) = the STFT as a factored DFT (512 = 32 x 16, two small GEMMs with the
 * twiddles folded into the second), 1 = the DFT as one GEMM (rounds 1-4; whole-signal calls only).  Process-wide.
 * savad_logmel_tables_host copies the factored kernel's three A-operand tables (savad_logmel_table_floats(0..2)
 * floats each) to HOST memory: the CPU suite replays the kernel's data flow on them against the oracle. */
int savad_logmel_set_algorithm(int algorithm);
int savad_logmel_table_floats(int which);
int savad_logmel_tables_host(float* t1, float* t3, float* tm);

/* Post-processing of the predict path (next-row 3 of the scope table); HOST pointers.
 * savad_trim_voice_activity  : vad/postprocessing/trim.py:4-66 (valley fill, hill flatten, hang before/over; the
 *                              hang pass only runs when hang_before > 0, as in the reference)
 * savad_frames_to_samples    : vad/postprocessing/convert.py:6-24; returns int((n-1)*hop + window) samples
 *                              (out == NULL: size query)
 * savad_samples_to_segments  : vad/postprocessing/convert.py:27-61; returns the segment count, writes up to `cap`
 *                              (start, end) SAMPLE indices (the reference's times are index / sample_rate)
 * savad_optimal_split        : vad/postprocessing/split.py:26-109 */
int savad_trim_voice_activity(const uint8_t* pred, int n, int min_vally, int min_hill, int hang_before, int hang_over,
                              uint8_t* out);
long savad_frames_to_samples(const double* frames, int n, int sample_rate, double hop_ms, double window_ms, double* out);
int savad_samples_to_segments(const double* samples, long n, long* starts, long* ends, int cap);
int savad_optimal_split(const double* pred, const double* probs, long n, long max_samples, double* out);

const char* savad_last_error(void);
const char* savad_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SAVAD_H */
