// localvocal.hpp — host-side mirror of LocalVocal's whisper boundary (reference names and signatures).
//
// A maintainer swapping engines keeps src/whisper-utils/whisper-processing.cpp as it is and links
// libmi355_whisper.so instead of whisper.cpp.  This mirror exists so the path can be
// exercised and tested HERE without libobs: it is a fresh implementation of the same contract
//   init_whisper_context   /root/reference/src/whisper-utils/whisper-processing.cpp:26-125
//   run_whisper_inference  /root/reference/src/whisper-utils/whisper-processing.cpp:127-324
//   shutdown (ctx half)    /root/reference/src/whisper-utils/whisper-utils.cpp:9-18
//   to_timestamp           /root/reference/src/whisper-utils/whisper-utils.cpp:150-161
// over a transcription_filter_data reduced to the members those functions read
// (/root/reference/src/transcription-filter-data.h:93-253).
#pragma once

#include "../../include/whisper.h"

#include <cstdint>
#include <deque>
#include <mutex>
#include <string>
#include <vector>

enum DetectionResult {
    DETECTION_RESULT_UNKNOWN = 0,
    DETECTION_RESULT_SILENCE = 1,
    DETECTION_RESULT_SPEECH = 2,
    DETECTION_RESULT_SUPPRESSED = 3,
    DETECTION_RESULT_NO_INFERENCE = 4,
    DETECTION_RESULT_PARTIAL = 5,
};

enum VadState { VAD_STATE_WAS_ON = 0, VAD_STATE_WAS_OFF, VAD_STATE_IS_OFF, VAD_STATE_PARTIAL };

struct DetectionResultWithText {
    DetectionResult result;
    std::string text;
    uint64_t start_timestamp_ms;
    uint64_t end_timestamp_ms;
    std::vector<whisper_token_data> tokens;
    std::string language;
};

struct transcription_filter_data {
    std::string whisper_model_path;
    struct whisper_context *whisper_context = nullptr;
    whisper_full_params whisper_params;
    float sentence_psum_accept_thresh = 0.4f;
    float duration_filter_threshold = 2.25f;
    int log_level = 5;
    bool log_words = false;
    bool enable_token_ts_dtw = false;
    int n_context_sentences = 0;
    std::deque<std::string> last_transcription_sentence;
    std::string whisper_model_file_currently_loaded;
    std::mutex whisper_ctx_mutex;
    // engine placement (the reference only logs cparams.gpu_device, whisper-processing.cpp:62)
    int gpu_device = 0;
    int precision = -1;  // -1: library default / MI355_WHISPER_PRECISION
    // storage behind the borrowed const char* members of whisper_params
    std::string language_storage;

    transcription_filter_data();
};

struct whisper_context *init_whisper_context(const std::string &model_path, struct transcription_filter_data *gf);
struct DetectionResultWithText run_whisper_inference(struct transcription_filter_data *gf, const float *pcm32f_data,
                                                     size_t pcm32f_num_samples, uint64_t t0 = 0, uint64_t t1 = 0,
                                                     int vad_state = VAD_STATE_WAS_OFF);
void shutdown_whisper_context(struct transcription_filter_data *gf);
std::string to_timestamp(uint64_t t_ms_offset);
