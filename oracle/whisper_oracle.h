/*
 * whisper_oracle.h — C interface of the CPU ORACLE.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and there only
 * as the checker.  The product (obs-localvocal_amd/csrc) never links, includes or calls it.
 *
 * What it restates: the arithmetic LocalVocal reaches through whisper.cpp (third-party, pinned to
 * git tag v1.6.2 by /root/reference/cmake/BuildWhispercpp.cmake:118, NOT vendored in the reference
 * tree and absent from this container).  Call sites the behaviour is anchored on:
 *   /root/reference/src/whisper-utils/whisper-processing.cpp:59,112,218-219,236-237,250-258
 *   /root/reference/src/whisper-utils/whisper-params.cpp:56,104-143
 *   /root/reference/src/whisper-utils/whisper-utils.cpp:15
 * Algorithm source: whisper.cpp v1.6.2 as published (log_mel_spectrogram, whisper_encode_internal,
 * whisper_decode_internal, whisper_process_logits, whisper_sample_token[_topk], whisper_full_with_state),
 * restated from SURVEY.md Appendix A; pure fp32 (no f16 activation rounding, no f16 GELU table).
 *
 * PARITY UNPINNED at the whisper.cpp boundary: the reference holds no golden vectors, transcripts,
 * token or mel dumps for this path.  What IS pinned: the model arithmetic
 * (mel, encoder, decoder logits, timestamp rules, greedy loop) against HuggingFace transformers'
 * independent implementation of the same architecture, through tests/golden/ fixtures generated
 * by tools/gen_golden.py in the build container.
 */
#ifndef WHISPER_ORACLE_H
#define WHISPER_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ow_context ow_context;

/* flat parameter block (field meaning = whisper_full_params of the same name) */
typedef struct ow_params {
    int32_t strategy;         /* 0 greedy, 1 beam search */
    int32_t n_threads;
    int32_t n_max_text_ctx;
    int32_t offset_ms;
    int32_t duration_ms;
    int32_t translate;
    int32_t no_context;
    int32_t no_timestamps;
    int32_t single_segment;
    int32_t print_special;
    int32_t max_tokens;
    int32_t audio_ctx;
    int32_t tdrz_enable;
    int32_t detect_language;
    int32_t suppress_blank;
    int32_t suppress_non_speech_tokens;
    float temperature;
    float max_initial_ts;
    float length_penalty;
    float temperature_inc;
    float entropy_thold;
    float logprob_thold;
    float no_speech_thold;
    int32_t best_of;
    int32_t beam_size;
    float patience;
    /* bench-only extension (not in whisper.h): when > 0, EOT is suppressed and a decoder is not
       allowed to complete before this many sampled tokens; see DESIGN.md "forced decode length" */
    int32_t force_decode_steps;
    const char *suppress_regex;
    const char *initial_prompt;
    const char *language;
} ow_params;

typedef struct ow_token_data {
    int32_t id;
    int32_t tid;
    float p;
    float plog;
    float pt;
    float ptsum;
    int64_t t0;
    int64_t t1;
    int64_t t_dtw;
    float vlen;
} ow_token_data;

ow_context *ow_init_from_file(const char *path);
void ow_free(ow_context *ctx);
void ow_set_threads(int n);

/* hparams: n_vocab,n_audio_ctx,n_audio_state,n_audio_head,n_audio_layer,n_text_ctx,n_text_state,
   n_text_head,n_text_layer,n_mels,ftype */
void ow_hparams(ow_context *ctx, int32_t out[11]);
/* eot,sot,translate,transcribe,solm,prev,nosp,not,beg */
void ow_special_tokens(ow_context *ctx, int32_t out[9]);
const char *ow_token_to_str(ow_context *ctx, int id);
int ow_tokenize(ow_context *ctx, const char *text, int32_t *out, int n_max);

ow_params ow_default_params(int strategy);

/* stage entry points (state lives in ctx) */
int ow_pcm_to_mel(ow_context *ctx, const float *pcm, int n);
int ow_mel_n_len(ow_context *ctx);
int ow_mel_n_len_org(ow_context *ctx);
const float *ow_mel_data(ow_context *ctx); /* [n_mel][n_len] */
int ow_encode(ow_context *ctx, int seek, int audio_ctx);
/* MX-fp8 twin of the engine's MW_PRECISION_FP8_MX encoder: the operands of the Q, K, V, attention-output, fc1 and fc2 products of every encoder
   block and of the decoder layers' cross K / V projections go through ow_mx_qdq first (blocks of 32 along k: power-of-two scale bringing the block under 448, e4m3 elements,
   round to nearest even) */
void ow_set_mx_encoder(ow_context *ctx, int on);
void ow_mx_qdq(float *a, int rows, int K); /* in place, K % 32 == 0 */
const float *ow_encoder_out(ow_context *ctx);           /* [n_ctx][d] */
const float *ow_cross_kv(ow_context *ctx, int layer, int is_v); /* [n_ctx][d] */
/* decode n tokens of sequence `seq` at positions n_past.. ; logits for every token are kept */
int ow_decode(ow_context *ctx, const int32_t *tokens, int n, int n_past, int seq);
const float *ow_logits(ow_context *ctx); /* [n][n_vocab] of the last ow_decode */
/* the logits rules on a caller-supplied logits row: history = sampled tokens so far */
int ow_process_logits(ow_context *ctx, const ow_params *p, float temperature, const float *logits_in,
                      const int32_t *history, int n_history, int has_ts, int seek_delta,
                      float *logits_out, float *logprobs_out, float *probs_out);

int ow_full(ow_context *ctx, const ow_params *p, const float *pcm, int n);
int ow_full_n_segments(ow_context *ctx);
int64_t ow_full_get_segment_t0(ow_context *ctx, int i);
int64_t ow_full_get_segment_t1(ow_context *ctx, int i);
const char *ow_full_get_segment_text(ow_context *ctx, int i);
int ow_full_n_tokens(ow_context *ctx, int i);
ow_token_data ow_full_get_token_data(ow_context *ctx, int i, int j);
void ow_full_get_token_data_p(ow_context *ctx, int i, int j, ow_token_data *out);
int ow_full_lang_id(ow_context *ctx);
int ow_lang_auto_detect(ow_context *ctx, int offset_ms, float *probs);
const char *ow_lang_str(int id);
int ow_lang_id(const char *lang);
/* counters for the cpu_baseline leg */
int ow_n_decode_calls(ow_context *ctx);
int ow_n_encode_calls(ow_context *ctx);

#ifdef __cplusplus
}
#endif
#endif
