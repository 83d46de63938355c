/*
 * whisper.h — the drop-in boundary.
 *
 * LocalVocal reaches its Whisper engine only through `#include <whisper.h>` and the C symbols below
 * (grep of `whisper_[a-z_]+\(` over /root/reference/src).  Today that header comes from
 * whisper.cpp v1.6.2, fetched and built by /root/reference/cmake/BuildWhispercpp.cmake:118-149.  This
 * header + libmi355_whisper.so replace that dependency: same names, same argument meaning, same
 * struct field order, so the sources under /root/reference/src/whisper-utils compile and link against it
 * unchanged (INTEGRATION.md shows the cmake edit).  The implementation behind it is a from-scratch
 * HIP engine for gfx950 (obs-localvocal_amd/csrc); there is no ggml and no CPU fallback: every entry
 * point that computes fails (nullptr / negative code) when no MI355X is usable.
 *
 * Struct layouts are restated from the published whisper.cpp v1.6.2 API (the library is not in the
 * reference tree; see SURVEY.md "[W-mem]").  They are part of the compile-time contract because
 * /root/reference/src/transcription-filter-data.h:126-127 embeds `whisper_context *` and a
 * `whisper_full_params` BY VALUE in the filter state.
 *
 * Each declaration cites the reference call site it serves.
 */
#ifndef MI355_WHISPER_COMPAT_WHISPER_H
#define MI355_WHISPER_COMPAT_WHISPER_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WHISPER_SAMPLE_RATE 16000 /* used ~25x, e.g. whisper-processing.cpp:151,160,162,165,183,332 */
#define WHISPER_N_FFT 400
#define WHISPER_HOP_LENGTH 160
#define WHISPER_CHUNK_SIZE 30

struct whisper_context; /* opaque; transcription-filter-data.h:126 */
struct whisper_state;

typedef int32_t whisper_pos;
typedef int32_t whisper_token;
typedef int32_t whisper_seq_id;

/* whisper-processing.cpp:47 (the lambda's first parameter) */
enum ggml_log_level {
    GGML_LOG_LEVEL_ERROR = 2,
    GGML_LOG_LEVEL_WARN = 3,
    GGML_LOG_LEVEL_INFO = 4,
    GGML_LOG_LEVEL_DEBUG = 5
};
typedef void (*ggml_log_callback)(enum ggml_log_level level, const char *text, void *user_data);

/* whisper-processing.cpp:77,81 */
enum whisper_alignment_heads_preset {
    WHISPER_AHEADS_NONE,
    WHISPER_AHEADS_N_TOP_MOST,
    WHISPER_AHEADS_CUSTOM,
    WHISPER_AHEADS_TINY_EN,
    WHISPER_AHEADS_TINY,
    WHISPER_AHEADS_BASE_EN,
    WHISPER_AHEADS_BASE,
    WHISPER_AHEADS_SMALL_EN,
    WHISPER_AHEADS_SMALL,
    WHISPER_AHEADS_MEDIUM_EN,
    WHISPER_AHEADS_MEDIUM,
    WHISPER_AHEADS_LARGE_V1,
    WHISPER_AHEADS_LARGE_V2,
    WHISPER_AHEADS_LARGE_V3
};

typedef struct whisper_ahead {
    int n_text_layer;
    int n_head;
} whisper_ahead;

typedef struct whisper_aheads {
    size_t n_heads;
    const whisper_ahead *heads;
} whisper_aheads;

/* whisper-processing.cpp:59-82: LocalVocal writes use_gpu, dtw_token_timestamps, dtw_aheads_preset and
   reads gpu_device.  `use_gpu` is accepted and ignored: this engine only runs on the GPU. */
struct whisper_context_params {
    bool use_gpu;
    bool flash_attn;
    int gpu_device; /* HIP device ordinal the context lives on */

    bool dtw_token_timestamps; /* accepted; DTW token timestamps are not implemented (t_dtw stays -1) */
    enum whisper_alignment_heads_preset dtw_aheads_preset;
    int dtw_n_top;
    struct whisper_aheads dtw_aheads;
    size_t dtw_mem_size;
};

/* whisper-processing.cpp:255-256,293; transcription-filter-data.h (vector<whisper_token_data>) */
typedef struct whisper_token_data {
    whisper_token id;  /* token id */
    whisper_token tid; /* forced timestamp token id */

    float p;     /* probability of the token */
    float plog;  /* log probability of the token */
    float pt;    /* probability of the timestamp token */
    float ptsum; /* sum of probabilities of all timestamp tokens */

    int64_t t0; /* token-level timestamps: not computed (-1) */
    int64_t t1;
    int64_t t_dtw; /* -1 */

    float vlen; /* voice length of the token */
} whisper_token_data;

/* whisper-params.cpp:56-60,104-105 */
enum whisper_sampling_strategy {
    WHISPER_SAMPLING_GREEDY,      /* similar to OpenAI's GreedyDecoder */
    WHISPER_SAMPLING_BEAM_SEARCH  /* similar to OpenAI's BeamSearchDecoder */
};

typedef void (*whisper_new_segment_callback)(struct whisper_context *ctx, struct whisper_state *state, int n_new,
                                             void *user_data);
typedef void (*whisper_progress_callback)(struct whisper_context *ctx, struct whisper_state *state, int progress,
                                          void *user_data);
typedef bool (*whisper_encoder_begin_callback)(struct whisper_context *ctx, struct whisper_state *state,
                                               void *user_data);
typedef bool (*ggml_abort_callback)(void *data);
typedef void (*whisper_logits_filter_callback)(struct whisper_context *ctx, struct whisper_state *state,
                                               const whisper_token_data *tokens, int n_tokens, float *logits,
                                               void *user_data);

typedef struct whisper_grammar_element {
    int type;
    uint32_t value;
} whisper_grammar_element;

/* Every field read or written at whisper-params.cpp:104-143, whisper-processing.cpp:197,206,233-235 and
   transcription-filter.cpp:415-450 is honoured unless marked "accepted, no effect". */
struct whisper_full_params {
    enum whisper_sampling_strategy strategy;

    int n_threads;      /* accepted, no effect (no CPU compute) */
    int n_max_text_ctx; /* max tokens to use from past text as prompt for the decoder */
    int offset_ms;      /* start offset in ms */
    int duration_ms;    /* audio duration to process in ms (whisper-processing.cpp:206) */

    bool translate;
    bool no_context;       /* do not use past transcription (if any) as initial prompt for the decoder */
    bool no_timestamps;    /* do not generate timestamps */
    bool single_segment;   /* force single segment output */
    bool print_special;    /* include special tokens in segment text */
    bool print_progress;   /* accepted, no effect */
    bool print_realtime;   /* accepted, no effect */
    bool print_timestamps; /* accepted, no effect */

    bool token_timestamps; /* accepted, no effect (experimental heuristic not implemented) */
    float thold_pt;
    float thold_ptsum;
    int max_len;
    bool split_on_word;
    int max_tokens; /* max tokens per segment (0 = no limit) */

    bool speed_up;   /* accepted, no effect (removed upstream) */
    bool debug_mode; /* accepted, no effect */
    int audio_ctx;   /* overwrite the audio context size (0 = use default) */

    bool tdrz_enable; /* keep the speaker-turn token unsuppressed */

    const char *suppress_regex; /* tokens whose text fully matches are suppressed (borrowed pointer) */

    const char *initial_prompt; /* borrowed pointer; whisper-processing.cpp:197 */
    const whisper_token *prompt_tokens;
    int prompt_n_tokens;

    const char *language; /* nullptr, "" or "auto" => auto-detect; whisper-processing.cpp:233-235 */
    bool detect_language;

    bool suppress_blank;
    bool suppress_non_speech_tokens;

    float temperature;
    float max_initial_ts;
    float length_penalty;

    float temperature_inc;
    float entropy_thold;
    float logprob_thold;
    float no_speech_thold; /* not applied (as in v1.6.2) */

    struct {
        int best_of;
    } greedy;

    struct {
        int beam_size;
        float patience; /* not implemented upstream either */
    } beam_search;

    /* callbacks: LocalVocal leaves all of them null.  new_segment, progress,
       encoder_begin and abort are called when set; logits_filter is rejected (the logits rules run
       on the device) — whisper_full returns -10 if it is non-null. */
    whisper_new_segment_callback new_segment_callback;
    void *new_segment_callback_user_data;
    whisper_progress_callback progress_callback;
    void *progress_callback_user_data;
    whisper_encoder_begin_callback encoder_begin_callback;
    void *encoder_begin_callback_user_data;
    ggml_abort_callback abort_callback;
    void *abort_callback_user_data;
    whisper_logits_filter_callback logits_filter_callback;
    void *logits_filter_callback_user_data;

    const whisper_grammar_element **grammar_rules; /* must be null: grammars are not implemented (-10) */
    size_t n_grammar_rules;
    size_t i_start_rule;
    float grammar_penalty;
};

/* whisper-processing.cpp:46-57 — global; cb may fire from any thread that calls into the engine */
void whisper_log_set(ggml_log_callback log_callback, void *user_data);

/* whisper-processing.cpp:59 */
struct whisper_context_params whisper_context_default_params(void);
struct whisper_context_params *whisper_context_default_params_by_ref(void);
void whisper_free_context_params(struct whisper_context_params *params);

/* whisper-processing.cpp:112 (non-Windows) / :109 (Windows).  nullptr on any failure, never throws.
   The buffer variant copies what it needs before returning (the caller frees its buffer, :104-110).
   GGML weights are parsed once and moved into HBM here. */
struct whisper_context *whisper_init_from_file_with_params(const char *path_model,
                                                           struct whisper_context_params params);
struct whisper_context *whisper_init_from_buffer_with_params(void *buffer, size_t buffer_size,
                                                             struct whisper_context_params params);

/* whisper-processing.cpp:123 */
const char *whisper_print_system_info(void);

/* whisper-utils.cpp:15, whisper-processing.cpp:222 — never concurrent with whisper_full on the same ctx */
void whisper_free(struct whisper_context *ctx);

/* whisper-params.cpp:56,104; src/tests/localvocal-offline-test.cpp:176 */
struct whisper_full_params whisper_full_default_params(enum whisper_sampling_strategy strategy);
struct whisper_full_params *whisper_full_default_params_by_ref(enum whisper_sampling_strategy strategy);
void whisper_free_params(struct whisper_full_params *params);

/* whisper-processing.cpp:218-219 — blocking; 0 on success; results live in ctx until the next call */
int whisper_full(struct whisper_context *ctx, struct whisper_full_params params, const float *samples,
                 int n_samples);

/* whisper-processing.cpp:250-258 */
int whisper_full_n_segments(struct whisper_context *ctx);
int whisper_full_n_tokens(struct whisper_context *ctx, int i_segment);
whisper_token_data whisper_full_get_token_data(struct whisper_context *ctx, int i_segment, int i_token);
const char *whisper_token_to_str(struct whisper_context *ctx, whisper_token token);

/* whisper-processing.cpp:236-237.  lang_probs may be null.  Negative on error; whisper_lang_str
   clamps a negative / out-of-range id to "en" instead of returning an invalid pointer (LocalVocal does
   not check the id). */
int whisper_lang_auto_detect(struct whisper_context *ctx, int offset_ms, int n_threads, float *lang_probs);
const char *whisper_lang_str(int id);

/* remaining read-only accessors of the same API family (used by the offline test harness and by
   callers that print segments; src/tests/localvocal-offline-test.cpp) */
int64_t whisper_full_get_segment_t0(struct whisper_context *ctx, int i_segment);
int64_t whisper_full_get_segment_t1(struct whisper_context *ctx, int i_segment);
const char *whisper_full_get_segment_text(struct whisper_context *ctx, int i_segment);
whisper_token whisper_full_get_token_id(struct whisper_context *ctx, int i_segment, int i_token);
float whisper_full_get_token_p(struct whisper_context *ctx, int i_segment, int i_token);
int whisper_full_lang_id(struct whisper_context *ctx);
int whisper_lang_id(const char *lang);
int whisper_lang_max_id(void);
int whisper_is_multilingual(struct whisper_context *ctx);
int whisper_n_vocab(struct whisper_context *ctx);
int whisper_n_text_ctx(struct whisper_context *ctx);
int whisper_n_audio_ctx(struct whisper_context *ctx);
int whisper_n_len(struct whisper_context *ctx);
whisper_token whisper_token_eot(struct whisper_context *ctx);
whisper_token whisper_token_sot(struct whisper_context *ctx);
whisper_token whisper_token_solm(struct whisper_context *ctx);
whisper_token whisper_token_prev(struct whisper_context *ctx);
whisper_token whisper_token_nosp(struct whisper_context *ctx);
whisper_token whisper_token_not(struct whisper_context *ctx);
whisper_token whisper_token_beg(struct whisper_context *ctx);
whisper_token whisper_token_lang(struct whisper_context *ctx, int lang_id);
whisper_token whisper_token_translate(struct whisper_context *ctx);
whisper_token whisper_token_transcribe(struct whisper_context *ctx);
int whisper_tokenize(struct whisper_context *ctx, const char *text, whisper_token *tokens, int n_max_tokens);

#ifdef __cplusplus
}
#endif
#endif
