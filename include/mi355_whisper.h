/*
 * mi355_whisper.h — entry points of libmi355_whisper.so beyond the whisper.h-compatible set.
 *
 * These have no counterpart in whisper.h; they exist for (a) choosing the arithmetic mode at load
 * time, (b) the stage-level parity tests (each stage of the path that whisper_full runs as one call at
 * /root/reference/src/whisper-utils/whisper-processing.cpp:218-219 can be run and read back on its own)
 * and (c) measurement.  All pointers are plain host pointers unless a name says `_device`.
 * Every function returns 0 on success and a negative code on failure; none throws.
 */
#ifndef MI355_WHISPER_H
#define MI355_WHISPER_H

#include "whisper.h"

#ifdef __cplusplus
extern "C" {
#endif

enum mw_precision {
    MW_PRECISION_BF16 = 0, /* bf16 weights / MFMA operands / KV caches, fp32 accumulate + residual stream */
    MW_PRECISION_FP32 = 1, /* exact fp32 everywhere (f32 MFMA): the bit-exact-token mode for tiny/base */
    MW_PRECISION_FP8 = 2,  /* MW_PRECISION_BF16 with the decoder's projection matrices and the logits matrix quantised to OCP
                              e4m3 at load (one power-of-two scale per output row) and streamed as bytes by the decode-step
                              kernels (bf16 activations, fp32 accumulate); the encoder keeps bf16 weights */
    MW_PRECISION_FP8_MX = 3 /* MW_PRECISION_FP8 and, in the encoder, the Q|K|V, attention-output, fc1 and fc2 products of every block and the cross K / V projections in MX-fp8 on
                              the fp8 matrix cores: weights AND activations as e4m3 with one power-of-two scale per block of
                              32 along k (OCP microscaling; scale = the smallest power of two that brings the block under
                              448, round to nearest even).  Needs n_audio_state % 256 == 0 (base, small, medium, large) */
};

/* Same as whisper_init_from_file_with_params (whisper-processing.cpp:112) with the arithmetic mode made
   explicit.  whisper_init_* use MW_PRECISION_BF16 unless the environment variable
   MI355_WHISPER_PRECISION is "fp32", "fp8" or "fp8mx". */
struct whisper_context *mw_init_from_file(const char *path_model, int gpu_device, int precision);
struct whisper_context *mw_init_from_buffer(const void *buffer, size_t size, int gpu_device, int precision);
/* A further context (one more audio stream / OBS source) on src's GPU that SHARES src's HBM-resident weights: nothing is
   parsed or uploaded again; KV caches, workspace, HIP stream and whisper_full state are its own, so the two contexts run
   concurrently from different threads.  The weights are freed with the last context that uses them (either may be freed
   first).  whisper_init_from_file_with_params (whisper-processing.cpp:112) does this by itself when the same file
   (path, size, mtime) is already resident on the same device in the same precision — one upload per GPU however many
   filter instances load the model — unless MI355_WHISPER_NO_SHARED_WEIGHTS is set.  whisper.cpp's counterpart is
   whisper_init_state, which LocalVocal does not call (one whisper_context per filter, whisper-utils.cpp:27-51). */
struct whisper_context *mw_init_shared(struct whisper_context *src);

/* model facts: n_vocab,n_audio_ctx,n_audio_state,n_audio_head,n_audio_layer,n_text_ctx,n_text_state,
   n_text_head,n_text_layer,n_mels,ftype */
int mw_hparams(struct whisper_context *ctx, int32_t out[11]);

/* ---- stage-level entry points (same stages whisper_full chains) ---- */
/* log-mel of host PCM; result stays in HBM */
int mw_pcm_to_mel(struct whisper_context *ctx, const float *samples, int n_samples);
/* same, PCM already resident in HBM (the timed region of bench.py starts here) */
int mw_pcm_to_mel_device(struct whisper_context *ctx, const float *samples_device, int n_samples);
/* The same log-mel for callers that cannot allocate (tight serving loops, harnesses that forbid hipMalloc, code that
   must not disturb an allocator while another thread captures a HIP graph).  The call itself cannot be captured: it
   launches on the context's own stream and synchronises that stream before it returns.  No counterpart in the reference.  Calls no hipMalloc / hipFree / allocating routine: the
   intermediate rows and the global-maximum word live in a static __device__ array of the library (1.54 MB per GPU, held
   until the process ends).  Input and output are the caller's device buffers: samples_device[n_samples] ->
   mel_out_device[n_len][n_mel] fp32, frame-major (the engine's HBM layout, the transpose of mw_get_mel's),
   n_len = (n_samples + 480000) / 160 written to *n_len_out.  Limits: n_samples <= 480640 (one 30 s window plus
   LocalVocal's margins), mel_out_floats >= n_len * n_mel; beyond them the call fails (-100, mw_last_error says why).
   The context's own mel state is left untouched.  Every call on one GPU shares that GPU's array: calls that overlap on
   one GPU race (the call synchronises the context's stream before it returns). */
int mw_pcm_to_mel_noalloc(struct whisper_context *ctx, const float *samples_device, int n_samples, float *mel_out_device,
                          size_t mel_out_floats, int *n_len_out);
int mw_mel_n_len(struct whisper_context *ctx);
/* MX-fp8 stages of MW_PRECISION_FP8_MX on host arrays (parity entry points; the encoder chains the same kernels inside
   whisper_full, whisper-processing.cpp:218-219).  epilogue: 0 = bf16 rows (+ bias), 1 = fp32 rows + bias + residual R,
   2 = GELU(+ bias) re-quantised to MX-fp8 (returned decoded).  A [M][K] and W [N][K] are quantised by the library's kernels
   (W from its bf16 rounding, as the loader does) and handed back decoded when the pointers are given; C_out is [M][N]
   fp32.  K % 128 == 0, N % 256 == 0. */
int mw_mx_gemm(int device, const float *A, int M, int K, const float *W, int N, const float *bias, const float *R, int epilogue,
               float *C_out, float *A_decoded, float *W_decoded);
/* LayerNorm (ggml_norm + weight, bias) whose rows leave as MX-fp8 (returned decoded) and, optionally, as bf16; d % 256 == 0 */
int mw_mx_layernorm(int device, const float *x, const float *w, const float *b, int rows, int d, float *out_decoded, float *out_bf16);
/* whisper_full (whisper-processing.cpp:218-219) with the PCM already resident in HBM on the context's device */
int mw_full_device(struct whisper_context *ctx, struct whisper_full_params params, const float *samples_device,
                   int n_samples);
/* out[n_mel][n_len] (whisper.cpp layout) */
int mw_get_mel(struct whisper_context *ctx, float *out);
/* encoder + cross-K/V for the 30 s window starting at mel frame `seek` */
int mw_encode(struct whisper_context *ctx, int seek, int audio_ctx);
/* out[n_ctx][n_audio_state] fp32 */
int mw_get_encoder_out(struct whisper_context *ctx, float *out);
/* out[n_ctx][n_text_state] fp32: cross-attention keys (pre-scaled by dh^-1/4) or values of one layer */
int mw_get_cross_kv(struct whisper_context *ctx, int layer, int is_v, float *out);
/* decode n tokens of sequence `seq` at positions n_past..n_past+n-1; keeps logits of the last
   min(n, 32) tokens */
int mw_decode(struct whisper_context *ctx, const whisper_token *tokens, int n, int n_past, int seq);
/* ONE decode step over n rows of different sequences (the shape of a beam step: n <= 32 rows, row i = token tokens[i] of
   sequence seq[i] at position pos[i]); keeps the logits of every row.  For teacher-forced parity tests. */
int mw_decode_rows(struct whisper_context *ctx, const whisper_token *tokens, const int *pos, const int *seq, int n);
/* out[rows][n_vocab]; rows = min(n, 32) of the last mw_decode / mw_decode_rows */
int mw_get_logits(struct whisper_context *ctx, float *out);
/* forget every cached key/value (start of a new prompt) */
int mw_kv_clear(struct whisper_context *ctx);
/* logits rules + sampler on one caller-supplied logits row (whisper_process_logits + samplers).
   history: tokens sampled so far.  n_draws = 0: greedy arg-max, else inverse-CDF draws with u[].
   Any of the three [n_vocab] outputs may be null.  tok_out gets max(1, n_draws) records. */
int mw_process_logits(struct whisper_context *ctx, const struct whisper_full_params *params, float temperature,
                      const float *logits_in, const whisper_token *history, int n_history, int has_ts, int seek_delta,
                      int n_draws, const double *u, float *logits_out, float *logprobs_out, float *probs_out,
                      whisper_token_data *tok_out);

/* ---- measurement ---- */
struct mw_timings {
    double mel_ms, encode_ms, decode_ms, sample_ms, total_ms; /* host wall clock, cumulative */
    int n_encode, n_decode, n_sample;
};
int mw_get_timings(struct whisper_context *ctx, struct mw_timings *out);
int mw_reset_timings(struct whisper_context *ctx);
/* the HIP stream every kernel of this context is launched on (for HIP-event timing by the caller) */
void *mw_stream(struct whisper_context *ctx);
int mw_synchronize(struct whisper_context *ctx);
/* Benchmark-only: with n > 0, EOT is suppressed and no decoder may complete before n sampled tokens
   (synthetic random weights otherwise end a window after a handful of tokens).  0 restores normal
   behaviour.  See DESIGN.md "forced decode length". */
int mw_set_force_decode_steps(struct whisper_context *ctx, int n);
/* Counters of the batcher that runs the decode steps of every context sharing ctx's weights on its GPU:
   out = {batched steps run, activation rows in them, decode requests served, encoder passes run, 30 s windows in them}.
   rows / steps > rows per request means rows of different streams really shared weight passes; windows / passes > 1 the
   same for the encoder.  Returns 1 (and zeros) when ctx does not use the batcher
   (fp32 mode, widths without tile kernels).  No counterpart in the reference. */
int mw_hub_stats(struct whisper_context *ctx, unsigned long long out[5]);
/* Measurement-only: every launch of kernel class `step_class` (MW_STEP_*, -1: off) inside the replayed decode step is
   issued twice (the second writes a scratch output where the kernel updates in place), so that
   (time with - time without) / launches is the class's cost INSIDE graph replay; results are unchanged.  No
   counterpart in the reference. */
int mw_set_duplicate_class(struct whisper_context *ctx, int step_class);
/* micro-benchmarks of single kernels on device-resident synthetic operands (bench.py roofline legs):
   run `iters` launches of the decode step's weight-streaming projections for `rows` activation rows and
   return the average milliseconds per launch measured with HIP events on the context's stream. */
int mw_bench_decode_step(struct whisper_context *ctx, int rows, int n_past, int iters, float *ms_per_step);
int mw_bench_mel(struct whisper_context *ctx, int n_samples, int iters, float *ms_per_call);
int mw_bench_encode(struct whisper_context *ctx, int iters, float *ms_per_call);

/* Live timing of ONE kernel class of the decode step (the projections of dec.hip / mega.hip and the two
   attention kernels): between begin and end every launch of that class on the context's stream is bracketed
   by HIP events (the step then runs launch by launch, not as a captured graph); end returns the launch count
   and the summed elapsed time.  Classes: */
enum mw_gemv_class {
    MW_GEMV_QKV = 0, MW_GEMV_ATTN_OUT = 1, MW_GEMV_CROSS_Q = 2, MW_GEMV_FC1 = 3, MW_GEMV_FC2 = 4, MW_GEMV_LOGITS = 5,
    MW_STEP_SELF_ATTN = 6, MW_STEP_CROSS_ATTN = 7
};
int mw_profile_begin(struct whisper_context *ctx, int gemv_class, int max_launches);
int mw_profile_end(struct whisper_context *ctx, int *n_launches, float *total_ms);

const char *mw_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
