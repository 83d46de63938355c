/*
 * localvocal_hotpath.h — C ABI over the LocalVocal-side half of the hot path (the code that sits
 * directly above whisper.h in the reference).  The C++ mirror with the reference's own names and
 * signatures is obs-localvocal_amd/csrc/localvocal.hpp; this header flattens it for FFI callers
 * (tests call it through ctypes).  No torch types, no C++ types.
 */
#ifndef LOCALVOCAL_HOTPATH_H
#define LOCALVOCAL_HOTPATH_H

#include "whisper.h"

#ifdef __cplusplus
extern "C" {
#endif

/* enum DetectionResult, /root/reference/src/whisper-utils/whisper-processing.h:14-21 */
enum lv_detection_result {
    LV_DETECTION_RESULT_UNKNOWN = 0,
    LV_DETECTION_RESULT_SILENCE = 1,
    LV_DETECTION_RESULT_SPEECH = 2,
    LV_DETECTION_RESULT_SUPPRESSED = 3,
    LV_DETECTION_RESULT_NO_INFERENCE = 4,
    LV_DETECTION_RESULT_PARTIAL = 5
};
/* enum VadState, /root/reference/src/whisper-utils/vad-processing.h:21 */
enum lv_vad_state { LV_VAD_STATE_WAS_ON = 0, LV_VAD_STATE_WAS_OFF, LV_VAD_STATE_IS_OFF, LV_VAD_STATE_PARTIAL };

typedef struct lv_filter lv_filter; /* the slice of transcription_filter_data the path reads */

/* struct DetectionResultWithText, whisper-processing.h:23-30, flattened.  Owned by the lv_filter and
   valid until the next lv_run_whisper_inference on it. */
typedef struct lv_detection {
    int result;
    const char *text;
    uint64_t start_timestamp_ms;
    uint64_t end_timestamp_ms;
    const whisper_token_data *tokens;
    int n_tokens;
    const char *language;
} lv_detection;

/* transcription_filter_data construction + defaults (transcription-filter-properties.cpp:680-750,
   whisper-params.cpp:54-100: BEAM_SEARCH strategy, suppress_blank=false, language "en", ...) */
lv_filter *lv_filter_create(void);
void lv_filter_destroy(lv_filter *f);
/* the whisper_full_params embedded in the filter (transcription-filter-data.h:127); edit in place */
struct whisper_full_params *lv_filter_params(lv_filter *f);
/* language string is copied (OBS settings own theirs) */
void lv_filter_set_language(lv_filter *f, const char *language);
void lv_filter_set_thresholds(lv_filter *f, float sentence_psum_accept_thresh, float duration_filter_threshold);
void lv_filter_set_context_sentences(lv_filter *f, int n_context_sentences);
void lv_filter_push_sentence(lv_filter *f, const char *sentence);
void lv_filter_set_gpu_device(lv_filter *f, int device, int precision);

/* start_whisper_thread_with_path minus the thread (whisper-utils.cpp:27-51) -> init_whisper_context
   (whisper-processing.cpp:26-125): accepts a file or a directory holding a *.bin.  0 on success. */
int lv_filter_load_model(lv_filter *f, const char *model_path);
/* shutdown_whisper_thread's context half (whisper-utils.cpp:9-18) */
void lv_filter_unload_model(lv_filter *f);
struct whisper_context *lv_filter_context(lv_filter *f);

/* run_whisper_inference (whisper-processing.cpp:127-324) */
int lv_run_whisper_inference(lv_filter *f, const float *pcm32f_data, size_t pcm32f_num_samples, uint64_t t0, uint64_t t1,
                             int vad_state, lv_detection *out);

/* to_timestamp (whisper-utils.cpp:150-161); writes at most n bytes incl. NUL */
void lv_to_timestamp(uint64_t t_ms_offset, char *buf, size_t n);

#ifdef __cplusplus
}
#endif
#endif
