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

/* ---- the callers of the path: settings, audio tap, segmentation, worker thread ---- */

/* enum VadMode, /root/reference/src/whisper-utils/vad-processing.h:31 */
enum lv_vad_mode { LV_VAD_MODE_ACTIVE = 0, LV_VAD_MODE_HYBRID, LV_VAD_MODE_DISABLED };

/* obs_data_t stand-in holding the keys of whisper-params.cpp:54-100 ("strategy", "n_threads", ...,
   "beam_search.patience").  Defaults are installed at lv_filter_create
   (apply_whisper_params_defaults_on_settings). Setters return 0, or -1 for a null argument. */
int lv_settings_set_int(lv_filter *f, const char *key, long long v);
int lv_settings_set_double(lv_filter *f, const char *key, double v);
int lv_settings_set_bool(lv_filter *f, const char *key, int v);
int lv_settings_set_string(lv_filter *f, const char *key, const char *v);
long long lv_settings_get_int(lv_filter *f, const char *key);
double lv_settings_get_double(lv_filter *f, const char *key);
int lv_settings_get_bool(lv_filter *f, const char *key);
/* apply_whisper_params_from_settings (whisper-params.cpp:102-144) followed by the language override of
   transcription_filter_update (transcription-filter.cpp:415-450): key "whisper_language_select"
   ("auto"/"" -> language "auto"... kept as given).  Rewrites the filter's whisper_full_params. */
int lv_filter_apply_settings(lv_filter *f);

/* stream shape + segmentation knobs (transcription-filter.cpp:476-541, transcription-filter-data.h:132-160) */
int lv_filter_setup_audio(lv_filter *f, uint32_t sample_rate, uint32_t channels);
void lv_filter_set_vad_mode(lv_filter *f, int vad_mode);
void lv_filter_set_partial(lv_filter *f, int partial_transcription, int partial_latency_ms);
void lv_filter_set_segment_duration(lv_filter *f, int segment_duration_ms);
void lv_filter_set_max_sub_duration(lv_filter *f, int max_sub_duration_ms);
void lv_filter_enable_audio_chunks_callback(lv_filter *f, int enable);
/* parameters of the energy gate that stands in for Silero (initialize_vad, vad-processing.cpp:456-475) */
void lv_filter_set_vad_gate(lv_filter *f, float gate_db, float slope_db, float threshold);

/* callbacks; `user` is passed back.  Strings and arrays are valid only during the call.
   on_text     <- set_text_callback (transcription-filter-callbacks.cpp:272-433): result after the English trim
   on_chunk    <- audio_chunk_callback (transcription-filter-data.h:266-269)
   on_request  <- fired at the top of run_inference_and_callbacks (whisper-processing.cpp:326-348) with the
                  padded PCM about to be transcribed: how a harness observes the segmentation itself */
typedef void (*lv_text_cb)(void *user, const lv_detection *result);
typedef void (*lv_chunk_cb)(void *user, const float *pcm32f, size_t frames, int vad_state, const lv_detection *result);
typedef void (*lv_request_cb)(void *user, const float *pcm32f, size_t n, uint64_t t0, uint64_t t1, int vad_state);
void lv_filter_set_callbacks(lv_filter *f, lv_text_cb on_text, lv_chunk_cb on_chunk, lv_request_cb on_request, void *user);

/* transcription_filter_filter_audio (transcription-filter.cpp:60-119): planar channel pointers, stamped
   with the wall clock; dropped while inactive or without a model.  Returns 1 if buffered, 0 if dropped. */
int lv_filter_filter_audio(lv_filter *f, const float *const *data, uint32_t frames);
/* the offline harness's push (localvocal-offline-test.cpp:527-560): explicit timestamp, no model needed;
   data == NULL pushes silence */
int lv_filter_push_packet(lv_filter *f, const float *const *data, uint32_t frames, uint64_t timestamp_offset_ns);
void lv_filter_clear_buffers(lv_filter *f); /* sets gf->clear_buffers (transcription-filter-callbacks.cpp:690-710) */
size_t lv_filter_input_bytes(lv_filter *f);   /* input_buffers[0].size */
size_t lv_filter_whisper_bytes(lv_filter *f); /* whisper_buffer.size */

/* One pass of the segmentation the worker would run for the current vad_mode
   (vad_based_segmentation / hybrid_vad_segmentation / vad_disabled_segmentation, vad-processing.cpp:125-454)
   on the calling thread, with the vad_state carried inside the filter.  Works without a model (every
   inference then returns DETECTION_RESULT_UNKNOWN).  state_out, if not NULL, receives
   {vad_on, start_ts_offest_ms, end_ts_offset_ms, last_partial_segment_end_ts}. */
int lv_filter_segmentation_step(lv_filter *f, uint64_t state_out[4]);
void lv_filter_reset_vad_state(lv_filter *f);

/* start_whisper_thread_with_path / shutdown_whisper_thread (whisper-utils.cpp:9-51): model + worker
   thread running whisper_loop (whisper-processing.cpp:361-431).  0 on success. */
int lv_filter_start(lv_filter *f, const char *model_path);
void lv_filter_stop(lv_filter *f);
/* update_whisper_model (whisper-model-utils.cpp:12-142): reads "whisper_model_path", "whisper_model_path_external" and
   "dtw_token_timestamps" from the filter's settings (lv_settings_set_*) and switches the model + worker exactly as the
   reference does: same name -> nothing; another name -> shutdown_whisper_thread (context freed under the context mutex,
   parked worker woken and joined) then start with the catalogue's file; "!!!external!!!" -> the external file, unless that
   file is the one loaded; a changed DTW flag -> the reference's restart with the path it has just cleared (fails; the
   next call reloads).  Returns 0 when a context is loaded afterwards, 1 when the filter is left stopped, < 0 on errors. */
int lv_filter_update_model(lv_filter *f);
/* the catalogue update_whisper_model looks names up in (models_info() + find_model_bin_file, model-downloader.h):
   name -> file on disk; file NULL or "" = listed but not downloaded (the download dialog is outside this library) */
int lv_filter_register_model(lv_filter *f, const char *name, const char *file);
/* The catalogue itself: load_models_info + find_model_bin_file (model-infos.cpp:114-260,
   model-downloader.cpp:9-89) over a models_directory.json on disk (the reference tries GitHub first; no network here):
   every MODEL_TYPE_TRANSCRIPTION entry is registered under its friendly name with the first *.bin found in
   <data_models_dir>/<local_folder_name> or, failing that, <config_models_dir>/<local_folder_name> ("" = not on disk).
   Returns the number of transcription models, < 0 on errors (unreadable file, malformed JSON). */
int lv_filter_load_catalogue(lv_filter *f, const char *json_path, const char *data_models_dir, const char *config_models_dir);
/* The check the catalogue's "sha256" field is there for; the reference parses the field and never compares it (no
   counterpart).  0: the registered file's SHA-256 equals the catalogue's; 1: it differs; 2: model not in the catalogue,
   not on disk, or listed without a digest; < 0: errors. */
int lv_filter_verify_model(lv_filter *f, const char *friendly_name);
int lv_model_file_sha256(const char *path, char *out_hex, size_t n); /* lower-case hex, n >= 65 */
/* gf->whisper_model_path, gf->whisper_model_file_currently_loaded, gf->whisper_model_loaded_new, whisper_thread.joinable(),
   gf->enable_token_ts_dtw — for lifecycle tests and harnesses; any pointer may be NULL */
int lv_filter_model_state(lv_filter *f, char *model_path, size_t n_path, char *file_loaded, size_t n_file, int *loaded_new,
                          int *worker_running, int *dtw);

/* VadIterator::process + get_speech_timestamps (silero-vad-onnx.cpp:262-309) on a standalone iterator
   with the filter's gate: writes up to cap (start,end) sample pairs, returns the count. */
int lv_vad_speech_timestamps(lv_filter *f, const float *pcm, size_t n, int reset_state, int *pairs, int cap);
/* The Silero VAD network behind VadIterator::predict_one (silero-vad-onnx.cpp:111-140: one Ort::Session::Run per 32 ms
   window on models/silero-vad/silero_vad.onnx) as HIP kernels that score every window of a process() call in one pass.
   lv_filter_set_silero_model: what obs_module_file("models/silero-vad/silero_vad.onnx") resolves to
   (whisper-model-utils.cpp:49-56); 0 = the filter's VadIterator now runs the network on the filter's GPU, 1 = no such
   file (the energy gate stands in, logged), < 0 = the file is not a Silero v5 graph (mw_last_error()). */
int lv_filter_set_silero_model(lv_filter *f, const char *path);
int lv_filter_vad_is_silero(lv_filter *f);
/* speech probabilities of the windows of the last VadIterator::process call; returns their count, writes up to cap */
int lv_filter_vad_probabilities(lv_filter *f, float *out, int cap);
/* the network alone (VadIterator::predict_one over consecutive 512-sample windows, state carried between calls unless
   reset: silero-vad-onnx.cpp:95-109,262-284); wav and probs are host buffers */
void *lv_silero_create(const char *onnx_path, int device);
int lv_silero_process(void *h, const float *wav, int n_windows, int reset, float *probs);
int lv_silero_state(void *h, float *hc256); /* the model's `state` tensor: h[128], c[128] */
/* the windows of several streams (up to 256 handles created from the same file on the same device, each at most once)
   in one pass: one front-end launch over all windows, one recurrent launch with a workgroup per stream; per stream the
   same probabilities and state as lv_silero_process.  No counterpart in the reference (one ORT session call per window
   and source); for hosts that own many sources. */
int lv_silero_process_batch(void *const *handles, const float *const *wavs, const int *n_windows, const int *resets,
                            float *const *probs, int n_streams);
void lv_silero_destroy(void *h);
/* host only: position-weighted sums of the 15 tensors the ONNX reader takes from the file, in the order basis,
   encoder.{0..3}.{weight,bias}, rnn.weight_ih, weight_hh, bias_ih, bias_hh, decoder.2.weight, bias; returns 15 */
int lv_silero_inspect(const char *onnx_path, double *sums, int cap);
/* remove_leading_trailing_nonalpha (transcription-utils.cpp:91-129) */
void lv_remove_leading_trailing_nonalpha(const char *in, char *out, size_t n);
/* stream_resampler: feeds one packet, returns the number of 16 kHz samples written (<= cap) */
int lv_resample(lv_filter *f, const float *const *data, uint32_t frames, float *out, int cap);

#ifdef __cplusplus
}
#endif
#endif
