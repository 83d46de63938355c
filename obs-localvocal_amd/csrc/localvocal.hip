// localvocal.hip — see localvocal.hpp.  Written against the reference's observable behaviour; the
// control flow below is this project's own arrangement of the same contract.
#include "localvocal.hpp"

#include "../../include/localvocal_hotpath.h"
#include "../../include/mi355_whisper.h"
#include "common.hpp"

#include <cmath>
#include <cstring>
#include <filesystem>
#include <memory>

namespace {

constexpr int kLogInfo = mw::LOG_INFO, kLogWarn = mw::LOG_WARN, kLogError = mw::LOG_ERROR;

DetectionResultWithText no_result(DetectionResult r, uint64_t t0, uint64_t t1, const std::string &lang = "") {
    return {r, "", t0, t1, {}, lang};
}

// the first *.bin a directory iteration yields (model-find-utils.cpp:36-50)
std::string first_bin_in(const std::string &dir) {
    std::error_code ec;
    for (const auto &entry : std::filesystem::directory_iterator(dir, ec))
        if (entry.path().extension() == ".bin") return entry.path().string();
    return "";
}

bool wrapped_in_brackets(const std::string &s) { return !s.empty() && s.front() == '[' && s.back() == ']'; }

}  // namespace

transcription_filter_data::transcription_filter_data() {
    // whisper-params.cpp:54-100 (plugin defaults over whisper_full_default_params(BEAM_SEARCH)) and
    // transcription-filter-properties.cpp:690-710 / transcription-filter.cpp:415-450 (language "en")
    whisper_params = whisper_full_default_params(WHISPER_SAMPLING_BEAM_SEARCH);
    whisper_params.print_special = false;
    whisper_params.print_progress = false;
    whisper_params.print_realtime = false;
    whisper_params.print_timestamps = false;
    whisper_params.suppress_blank = false;
    whisper_params.suppress_non_speech_tokens = false;
    language_storage = "en";
    whisper_params.language = language_storage.c_str();
}

struct whisper_context *init_whisper_context(const std::string &model_path_in, struct transcription_filter_data *gf) {
    std::string model_path = model_path_in;
    mw::log_msg(kLogInfo, "Loading whisper model from %s\n", model_path.c_str());
    std::error_code ec;
    if (std::filesystem::is_directory(model_path, ec)) {
        const std::string bin = first_bin_in(model_path);
        if (bin.empty()) {
            mw::log_msg(kLogError, "Model bin file not found in folder: %s\n", model_path.c_str());
            return nullptr;
        }
        model_path = bin;
    }
    whisper_context_params cparams = whisper_context_default_params();
    cparams.use_gpu = true;  // the only mode this engine has
    cparams.gpu_device = gf ? gf->gpu_device : 0;
    cparams.dtw_token_timestamps = gf && gf->enable_token_ts_dtw;
    cparams.dtw_aheads_preset = cparams.dtw_token_timestamps ? WHISPER_AHEADS_TINY_EN : WHISPER_AHEADS_NONE;
    whisper_context *ctx = (gf && gf->precision >= 0)
                               ? mw_init_from_file(model_path.c_str(), cparams.gpu_device, gf->precision)
                               : whisper_init_from_file_with_params(model_path.c_str(), cparams);
    if (ctx == nullptr) {
        mw::log_msg(kLogError, "Failed to load whisper model\n");
        return nullptr;
    }
    mw::log_msg(kLogInfo, "Whisper model loaded: %s\n", whisper_print_system_info());
    return ctx;
}

void shutdown_whisper_context(struct transcription_filter_data *gf) {
    if (!gf) return;
    if (gf->whisper_context != nullptr) {
        std::lock_guard<std::mutex> lock(gf->whisper_ctx_mutex);
        whisper_free(gf->whisper_context);
        gf->whisper_context = nullptr;
    }
    gf->whisper_model_path.clear();
}

struct DetectionResultWithText run_whisper_inference(struct transcription_filter_data *gf, const float *pcm_in,
                                                     size_t n_in, uint64_t t0, uint64_t t1, int vad_state) {
    if (gf == nullptr) {
        mw::log_msg(kLogError, "run_whisper_inference: gf is null\n");
        return no_result(DETECTION_RESULT_UNKNOWN, t0, t1);
    }
    if (pcm_in == nullptr || n_in == 0) {
        mw::log_msg(kLogError, "run_whisper_inference: pcm32f_data is null or size is 0\n");
        return no_result(DETECTION_RESULT_UNKNOWN, t0, t1);
    }
    if (t1 - t0 < 50) return no_result(DETECTION_RESULT_UNKNOWN, t0, t1);  // under 50 ms: skip

    const uint64_t incoming_duration_ms = (uint64_t)(n_in * 1000 / WHISPER_SAMPLE_RATE);
    // Segments under one second are centred in 1.01 s of +-0.01 uniform noise (whisper_full refuses
    // inputs under 1 s).  The noise comes from rand(), as in the reference.
    std::vector<float> padded;
    const float *pcm = pcm_in;
    size_t n = n_in;
    if (n_in < WHISPER_SAMPLE_RATE) {
        const size_t new_size = (size_t)(1.01f * (float)(WHISPER_SAMPLE_RATE));
        padded.resize(new_size);
        const float noise_level = 0.01f;
        for (size_t i = 0; i < new_size; ++i) padded[i] = noise_level * ((float)rand() / (float)RAND_MAX * 2.0f - 1.0f);
        memcpy(padded.data() + (new_size - n_in) / 2, pcm_in, n_in * sizeof(float));
        pcm = padded.data();
        n = new_size;
    }
    const uint64_t whisper_duration_ms = (uint64_t)(n * 1000 / WHISPER_SAMPLE_RATE);

    std::lock_guard<std::mutex> lock(gf->whisper_ctx_mutex);
    if (gf->whisper_context == nullptr) {
        mw::log_msg(kLogWarn, "whisper context is null\n");
        return no_result(DETECTION_RESULT_UNKNOWN, t0, t1);
    }
    std::string initial_prompt;  // must outlive whisper_full (borrowed by whisper_params)
    if (gf->n_context_sentences > 0 && !gf->last_transcription_sentence.empty()) {
        for (size_t i = 0; i < gf->last_transcription_sentence.size(); ++i)
            initial_prompt += (i ? " " : "") + gf->last_transcription_sentence[i];
        gf->whisper_params.initial_prompt = initial_prompt.c_str();
    }
    gf->whisper_params.duration_ms = (int)whisper_duration_ms;
    const int rc = whisper_full(gf->whisper_context, gf->whisper_params, pcm, (int)n);
    if (!initial_prompt.empty()) gf->whisper_params.initial_prompt = nullptr;  // do not leave a dangling pointer

    const char *lang_c = gf->whisper_params.language;
    std::string language = lang_c ? lang_c : "";
    if (lang_c == nullptr || lang_c[0] == 0 || strcmp(lang_c, "auto") == 0)
        language = whisper_lang_str(whisper_lang_auto_detect(gf->whisper_context, 0, 1, nullptr));
    if (rc != 0) {
        mw::log_msg(kLogWarn, "failed to process audio, error %d\n", rc);
        return no_result(DETECTION_RESULT_UNKNOWN, t0, t1);
    }

    // token walk: drop bracketed / special / time tokens, and a '.' in second-to-last position; a time
    // token far beyond the incoming duration marks the whole call as silence
    float sentence_p = 0.0f;
    std::string text;
    std::vector<whisper_token_data> kept;
    const int n_segments = whisper_full_n_segments(gf->whisper_context);
    for (int s = 0; s < n_segments; ++s) {
        const int n_tokens = whisper_full_n_tokens(gf->whisper_context, s);
        for (int j = 0; j < n_tokens; ++j) {
            const whisper_token_data token = whisper_full_get_token_data(gf->whisper_context, s, j);
            const std::string piece = whisper_token_to_str(gf->whisper_context, token.id);
            bool keep = !wrapped_in_brackets(piece) && token.id < 50256 && !(j == n_tokens - 2 && token.id == 13);
            if (token.id > 50365 && token.id <= 51865) {  // large-v3 id layout, literal as in the reference
                const float time = ((float)token.id - 50365.0f) * 0.02f;
                const float duration_s = (float)incoming_duration_ms / 1000.0f;
                if (time / duration_s > gf->duration_filter_threshold)
                    return no_result(DETECTION_RESULT_SILENCE, t0, t1, language);
                keep = false;
            }
            if (keep) {
                sentence_p += token.p;
                text += piece;
                kept.push_back(token);
            }
        }
    }
    sentence_p /= (float)kept.size();  // 0/0 = NaN when nothing was kept; NaN < thresh is false (kept as is)
    if (sentence_p < gf->sentence_psum_accept_thresh) return no_result(DETECTION_RESULT_SILENCE, t0, t1, language);
    if (gf->log_words)
        mw::log_msg(kLogInfo, "[%s --> %s]%s(%.3f) %s\n", to_timestamp(t0).c_str(), to_timestamp(t1).c_str(),
                    vad_state == VAD_STATE_PARTIAL ? "P" : " ", sentence_p, text.c_str());
    if (text.empty() || text == "." || text == " " || text == "\n")
        return no_result(DETECTION_RESULT_SILENCE, t0, t1, language);
    return {vad_state == VAD_STATE_PARTIAL ? DETECTION_RESULT_PARTIAL : DETECTION_RESULT_SPEECH, text, t0, t1, kept,
            language};
}

std::string to_timestamp(uint64_t t_ms_offset) {
    const uint64_t total_s = t_ms_offset / 1000, msec = t_ms_offset % 1000;
    char buf[32];
    snprintf(buf, sizeof(buf), "%02d:%02d.%03d", (int)(total_s / 60), (int)(total_s % 60), (int)msec);
    return buf;
}

// ---------------------------------------------------------------------------------------------
// flat C ABI
// ---------------------------------------------------------------------------------------------
struct lv_filter {
    transcription_filter_data gf;
    DetectionResultWithText last;
};

extern "C" {

lv_filter *lv_filter_create(void) {
    try {
        return new lv_filter();
    } catch (...) {
        return nullptr;
    }
}
void lv_filter_destroy(lv_filter *f) {
    if (!f) return;
    shutdown_whisper_context(&f->gf);
    delete f;
}
struct whisper_full_params *lv_filter_params(lv_filter *f) { return f ? &f->gf.whisper_params : nullptr; }
void lv_filter_set_language(lv_filter *f, const char *language) {
    if (!f) return;
    if (language == nullptr) {
        f->gf.whisper_params.language = nullptr;
        return;
    }
    f->gf.language_storage = language;
    f->gf.whisper_params.language = f->gf.language_storage.c_str();
}
void lv_filter_set_thresholds(lv_filter *f, float psum, float dur) {
    if (!f) return;
    f->gf.sentence_psum_accept_thresh = psum;
    f->gf.duration_filter_threshold = dur;
}
void lv_filter_set_context_sentences(lv_filter *f, int n) {
    if (f) f->gf.n_context_sentences = n;
}
void lv_filter_push_sentence(lv_filter *f, const char *sentence) {
    if (!f || !sentence) return;
    // transcription-filter-callbacks.cpp:423-431: keep the last n_context_sentences sentences
    f->gf.last_transcription_sentence.push_back(sentence);
    while ((int)f->gf.last_transcription_sentence.size() > std::max(f->gf.n_context_sentences, 0))
        f->gf.last_transcription_sentence.pop_front();
}
void lv_filter_set_gpu_device(lv_filter *f, int device, int precision) {
    if (!f) return;
    f->gf.gpu_device = device;
    f->gf.precision = precision;
}
int lv_filter_load_model(lv_filter *f, const char *model_path) {
    if (!f || !model_path) return -1;
    try {
        std::lock_guard<std::mutex> lock(f->gf.whisper_ctx_mutex);
        if (f->gf.whisper_context != nullptr) {
            mw::log_msg(kLogError, "cannot init whisper: whisper_context is not null\n");
            return -2;
        }
        f->gf.whisper_context = init_whisper_context(model_path, &f->gf);
        if (f->gf.whisper_context == nullptr) return -3;
        f->gf.whisper_model_file_currently_loaded = model_path;
        f->gf.whisper_model_path = model_path;
        return 0;
    } catch (...) {
        return -100;
    }
}
void lv_filter_unload_model(lv_filter *f) {
    if (f) shutdown_whisper_context(&f->gf);
}
struct whisper_context *lv_filter_context(lv_filter *f) { return f ? f->gf.whisper_context : nullptr; }

int lv_run_whisper_inference(lv_filter *f, const float *pcm, size_t n, uint64_t t0, uint64_t t1, int vad_state,
                             lv_detection *out) {
    if (!f || !out) return -1;
    try {
        f->last = run_whisper_inference(&f->gf, pcm, n, t0, t1, vad_state);
    } catch (const std::exception &e) {
        mw::log_msg(kLogError, "Whisper exception: %s. Filter restart is required\n", e.what());
        f->last = {DETECTION_RESULT_UNKNOWN, "", t0, t1, {}, ""};
    }
    out->result = (int)f->last.result;
    out->text = f->last.text.c_str();
    out->start_timestamp_ms = f->last.start_timestamp_ms;
    out->end_timestamp_ms = f->last.end_timestamp_ms;
    out->tokens = f->last.tokens.data();
    out->n_tokens = (int)f->last.tokens.size();
    out->language = f->last.language.c_str();
    return 0;
}

void lv_to_timestamp(uint64_t t_ms_offset, char *buf, size_t n) {
    if (!buf || n == 0) return;
    const std::string s = to_timestamp(t_ms_offset);
    snprintf(buf, n, "%s", s.c_str());
}

}  // extern "C"
