// localvocal.hip — see localvocal.hpp.  Written against the reference's observable behaviour; the
// control flow below is this project's own arrangement of the same contract.
#include "localvocal.hpp"
#include "silero_vad.hpp"

#include <fstream>
#include <iterator>

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
    input_cv.emplace();
    last_sub_render_time = now_ms();
    // 16 kHz mono until the host says otherwise (transcription_filter_setup_audio)
    frames = (size_t)(16000.0f / (1000.0f / 11000));
    copy_buffers[0].assign(frames, 0.0f);
    resampler_to_whisper.reset(new stream_resampler(16000, 1));
}

transcription_filter_data::~transcription_filter_data() {
    // a worker still running would touch freed members: stop it first (transcription-filter.cpp:143-170)
    try {
        shutdown_whisper_thread(this);
    } catch (...) {
    }
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
    lv_settings settings;
    vad_state vs = {false, 0, 0, 0};
    lv_text_cb on_text = nullptr;
    lv_chunk_cb on_chunk = nullptr;
    lv_request_cb on_request = nullptr;
    void *user = nullptr;
};

namespace {

lv_detection flatten(const DetectionResultWithText &r) {
    lv_detection d;
    d.result = (int)r.result;
    d.text = r.text.c_str();
    d.start_timestamp_ms = r.start_timestamp_ms;
    d.end_timestamp_ms = r.end_timestamp_ms;
    d.tokens = r.tokens.data();
    d.n_tokens = (int)r.tokens.size();
    d.language = r.language.c_str();
    return d;
}

// (Re)builds the filter's VAD from its configuration.  The worker thread uses gf.vad outside whisper_ctx_mutex
// (segmentation.hip), so the object is never replaced while a worker exists: callers on a running filter get an error and
// the new configuration takes effect at the next start (start_whisper_thread_with_path -> initialize_vad).
void make_vad(lv_filter *f) {
    if (f->gf.whisper_thread.joinable()) throw std::runtime_error("the VAD cannot be rebuilt while the worker thread is running");
    f->gf.vad.reset(build_vad(f->gf.silero_vad_model_file.c_str(), f->gf.gpu_device, f->gf.vad_gate_db, f->gf.vad_gate_slope_db,
                              f->gf.vad_threshold));
}

// every entry below returns an error code instead of letting an exception cross the C boundary
template <typename F> int lv_guard(F &&fn) {
    try {
        return fn();
    } catch (const std::exception &e) {
        mw::log_msg(kLogError, "localvocal: %s\n", e.what());
        mw::set_last_error(e.what());
        return -100;
    } catch (...) {
        mw::set_last_error("unknown error");
        return -101;
    }
}

}  // namespace

extern "C" {

lv_filter *lv_filter_create(void) {
    try {
        lv_filter *f = new lv_filter();
        apply_whisper_params_defaults_on_settings(&f->settings);
        // transcription-filter-properties.cpp:680-750 defaults for the keys read outside whisper-params.cpp
        f->settings.def_strings["whisper_language_select"] = "en";
        f->settings.def_ints["n_context_sentences"] = 0;
        f->settings.def_doubles["sentence_psum_accept_thresh"] = 0.4;
        f->settings.def_doubles["vad_threshold"] = 0.65;
        make_vad(f);
        return f;
    } catch (...) {
        return nullptr;
    }
}
void lv_filter_destroy(lv_filter *f) {
    if (!f) return;
    try {
        shutdown_whisper_thread(&f->gf);
    } catch (...) {
    }
    delete f;
}

int lv_settings_set_int(lv_filter *f, const char *key, long long v) {
    if (!f || !key) return -1;
    f->settings.ints[key] = v;
    return 0;
}
int lv_settings_set_double(lv_filter *f, const char *key, double v) {
    if (!f || !key) return -1;
    f->settings.doubles[key] = v;
    return 0;
}
int lv_settings_set_bool(lv_filter *f, const char *key, int v) {
    if (!f || !key) return -1;
    f->settings.bools[key] = v != 0;
    return 0;
}
int lv_settings_set_string(lv_filter *f, const char *key, const char *v) {
    if (!f || !key || !v) return -1;
    f->settings.strings[key] = v;
    return 0;
}
long long lv_settings_get_int(lv_filter *f, const char *key) { return f && key ? f->settings.get_int(key) : 0; }
double lv_settings_get_double(lv_filter *f, const char *key) { return f && key ? f->settings.get_double(key) : 0.0; }
int lv_settings_get_bool(lv_filter *f, const char *key) { return f && key ? (int)f->settings.get_bool(key) : 0; }

int lv_filter_apply_settings(lv_filter *f) {
    if (!f) return -1;
    return lv_guard([&] {
        transcription_filter_data &gf = f->gf;
        std::lock_guard<std::mutex> lock(gf.whisper_ctx_mutex);
        gf.n_context_sentences = (int)f->settings.get_int("n_context_sentences");
        gf.sentence_psum_accept_thresh = (float)f->settings.get_double("sentence_psum_accept_thresh");
        apply_whisper_params_from_settings(gf.whisper_params, &f->settings);
        const char *sel = f->settings.get_string("whisper_language_select");
        const bool language_selected = sel != nullptr && sel[0] != 0;
        gf.language_storage = language_selected ? sel : "auto";
        gf.whisper_params.language = gf.language_storage.c_str();
        gf.whisper_params.detect_language = !language_selected;
        gf.vad_threshold = (float)f->settings.get_double("vad_threshold");
        if (gf.vad) gf.vad->set_threshold(gf.vad_threshold);  // a float store, as transcription-filter.cpp:447-448 does on a live filter
        return 0;
    });
}

int lv_filter_setup_audio(lv_filter *f, uint32_t sample_rate, uint32_t channels) {
    if (!f) return -1;
    return lv_guard([&] {
        transcription_filter_setup_audio(&f->gf, sample_rate, channels);
        f->vs = {false, 0, 0, 0};
        return 0;
    });
}
void lv_filter_set_vad_mode(lv_filter *f, int vad_mode) {
    if (f) f->gf.vad_mode = vad_mode;
}
void lv_filter_set_partial(lv_filter *f, int partial_transcription, int partial_latency_ms) {
    if (!f) return;
    f->gf.partial_transcription = partial_transcription != 0;
    f->gf.partial_latency = partial_latency_ms;
}
void lv_filter_set_segment_duration(lv_filter *f, int ms) {
    if (f) f->gf.segment_duration = ms;
}
void lv_filter_set_max_sub_duration(lv_filter *f, int ms) {
    if (f) f->gf.max_sub_duration = (size_t)std::max(ms, 0);
}
void lv_filter_enable_audio_chunks_callback(lv_filter *f, int enable) {
    if (f) f->gf.enable_audio_chunks_callback = enable != 0;
}
void lv_filter_set_vad_gate(lv_filter *f, float gate_db, float slope_db, float threshold) {
    if (!f) return;
    f->gf.vad_gate_db = gate_db;
    f->gf.vad_gate_slope_db = slope_db > 0.0f ? slope_db : 3.0f;
    f->gf.vad_threshold = threshold;
    if (f->gf.whisper_thread.joinable()) {  // live worker: only the threshold changes now, the gate shape at the next start
        if (f->gf.vad) f->gf.vad->set_threshold(threshold);
        return;
    }
    try {
        make_vad(f);
    } catch (...) {
    }
}

void lv_filter_set_callbacks(lv_filter *f, lv_text_cb on_text, lv_chunk_cb on_chunk, lv_request_cb on_request, void *user) {
    if (!f) return;
    f->on_text = on_text;
    f->on_chunk = on_chunk;
    f->on_request = on_request;
    f->user = user;
    f->gf.setTextCallback = nullptr;
    f->gf.audioChunkCallback = nullptr;
    f->gf.inferenceRequestObserver = nullptr;
    if (on_text)
        f->gf.setTextCallback = [f](const DetectionResultWithText &r) {
            const lv_detection d = flatten(r);
            f->on_text(f->user, &d);
        };
    if (on_chunk)
        f->gf.audioChunkCallback = [f](const float *pcm, size_t frames, int vad_state, const DetectionResultWithText &r) {
            const lv_detection d = flatten(r);
            f->on_chunk(f->user, pcm, frames, vad_state, &d);
        };
    if (on_request)
        f->gf.inferenceRequestObserver = [f](const float *pcm, size_t n, uint64_t t0, uint64_t t1, int vad_state) {
            f->on_request(f->user, pcm, n, t0, t1, vad_state);
        };
}

int lv_filter_filter_audio(lv_filter *f, const float *const *data, uint32_t frames) {
    if (!f || !data || frames == 0) return 0;
    return lv_guard([&] {
        if (!f->gf.active || f->gf.whisper_context == nullptr) return 0;
        transcription_filter_filter_audio(&f->gf, data, frames);
        return 1;
    });
}
int lv_filter_push_packet(lv_filter *f, const float *const *data, uint32_t frames, uint64_t timestamp_offset_ns) {
    if (!f || frames == 0) return -1;
    return lv_guard([&] {
        transcription_filter_push_packet(&f->gf, data, frames, timestamp_offset_ns);
        return 0;
    });
}
void lv_filter_clear_buffers(lv_filter *f) {
    if (f) f->gf.clear_buffers = true;
}
size_t lv_filter_input_bytes(lv_filter *f) {
    if (!f) return 0;
    std::lock_guard<std::mutex> lock(f->gf.whisper_buf_mutex);
    return f->gf.input_buffers[0].size;
}
size_t lv_filter_whisper_bytes(lv_filter *f) { return f ? f->gf.whisper_buffer.size : 0; }

int lv_filter_segmentation_step(lv_filter *f, uint64_t state_out[4]) {
    if (!f) return -1;
    return lv_guard([&] {
        transcription_filter_data *gf = &f->gf;
        if (gf->clear_buffers) {
            gf->resampled_buffer.clear();
            gf->whisper_buffer.clear();
            f->vs = {false, now_ms(), 0, 0};
            gf->clear_buffers = false;
        }
        if (gf->vad_mode == VAD_MODE_HYBRID) f->vs = hybrid_vad_segmentation(gf, f->vs);
        else if (gf->vad_mode == VAD_MODE_ACTIVE) f->vs = vad_based_segmentation(gf, f->vs);
        else if (gf->vad_mode == VAD_MODE_DISABLED) f->vs = vad_disabled_segmentation(gf, f->vs);
        else return -2;
        if (state_out) {
            state_out[0] = f->vs.vad_on;
            state_out[1] = f->vs.start_ts_offest_ms;
            state_out[2] = f->vs.end_ts_offset_ms;
            state_out[3] = f->vs.last_partial_segment_end_ts;
        }
        return 0;
    });
}
void lv_filter_reset_vad_state(lv_filter *f) {
    if (f) f->vs = {false, 0, 0, 0};
}

int lv_filter_start(lv_filter *f, const char *model_path) {
    if (!f || !model_path) return -1;
    return lv_guard([&] {
        if (f->gf.whisper_context != nullptr) return -2;
        start_whisper_thread_with_path(&f->gf, model_path,
                                       f->gf.silero_vad_model_file.empty() ? "silero-vad/silero_vad.onnx" : f->gf.silero_vad_model_file.c_str());
        if (f->gf.whisper_context == nullptr) return -3;
        f->gf.whisper_model_path = model_path;
        return 0;  // initialize_vad built the VAD from gf's gate / threshold configuration before the worker started
    });
}
void lv_filter_stop(lv_filter *f) {
    if (!f) return;
    try {
        shutdown_whisper_thread(&f->gf);
    } catch (...) {
    }
}

int lv_filter_register_model(lv_filter *f, const char *name, const char *file) {
    if (!f || !name) return -1;
    return lv_guard([&] {
        f->gf.models_info[name] = file ? file : "";
        return 0;
    });
}
int lv_filter_load_catalogue(lv_filter *f, const char *json_path, const char *data_models_dir, const char *config_models_dir) {
    if (!f || !json_path) return -1;
    return lv_guard([&] {
        std::ifstream in(json_path, std::ios::binary);
        if (!in) throw std::runtime_error(std::string("cannot open ") + json_path);
        const std::string text((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        f->gf.catalogue = load_models_info(text);
        int n = 0;
        for (const auto &kv : f->gf.catalogue) {
            if (kv.second.type != MODEL_TYPE_TRANSCRIPTION) continue;
            f->gf.models_info[kv.first] = find_model_bin_file(kv.second, data_models_dir ? data_models_dir : "",
                                                              config_models_dir ? config_models_dir : "");
            ++n;
        }
        return n;
    });
}
int lv_filter_verify_model(lv_filter *f, const char *friendly_name) {
    if (!f || !friendly_name) return -1;
    return lv_guard([&] {
        const auto it = f->gf.catalogue.find(friendly_name);
        const auto fit = f->gf.models_info.find(friendly_name);
        if (it == f->gf.catalogue.end() || fit == f->gf.models_info.end() || fit->second.empty()) return 2;
        return verify_model_file(it->second, fit->second);
    });
}
int lv_model_file_sha256(const char *path, char *out_hex, size_t n) {
    if (!path || !out_hex || n < 65) return -1;
    return lv_guard([&] {
        snprintf(out_hex, n, "%s", sha256_file(path).c_str());
        return 0;
    });
}
int lv_filter_update_model(lv_filter *f) {
    if (!f) return -1;
    return lv_guard([&] {
        f->gf.context = &f->settings;
        update_whisper_model(&f->gf);  // a restart rebuilds the VAD itself (initialize_vad, before the new worker exists)
        return f->gf.whisper_context != nullptr ? 0 : 1;
    });
}
int lv_filter_model_state(lv_filter *f, char *model_path, size_t n_path, char *file_loaded, size_t n_file, int *loaded_new,
                          int *worker_running, int *dtw) {
    if (!f) return -1;
    return lv_guard([&] {
        if (model_path && n_path) snprintf(model_path, n_path, "%s", f->gf.whisper_model_path.c_str());
        if (file_loaded && n_file) snprintf(file_loaded, n_file, "%s", f->gf.whisper_model_file_currently_loaded.c_str());
        if (loaded_new) *loaded_new = f->gf.whisper_model_loaded_new ? 1 : 0;
        if (worker_running) *worker_running = f->gf.whisper_thread.joinable() ? 1 : 0;
        if (dtw) *dtw = f->gf.enable_token_ts_dtw ? 1 : 0;
        return 0;
    });
}

int lv_vad_speech_timestamps(lv_filter *f, const float *pcm, size_t n, int reset_state, int *pairs, int cap) {
    if (!f || (!pcm && n) || (!pairs && cap > 0)) return -1;
    return lv_guard([&] {
        if (f->gf.whisper_thread.joinable()) throw std::runtime_error("lv_vad_speech_timestamps: the worker thread owns the VAD");
        if (!f->gf.vad) make_vad(f);
        std::vector<float> wav(pcm, pcm + n);
        f->gf.vad->process(wav, reset_state != 0);
        const std::vector<timestamp_t> st = f->gf.vad->get_speech_timestamps();
        for (int i = 0; i < (int)st.size() && i < cap; ++i) {
            pairs[2 * i] = st[i].start;
            pairs[2 * i + 1] = st[i].end;
        }
        return (int)st.size();
    });
}
int lv_filter_set_silero_model(lv_filter *f, const char *path) {
    if (!f) return -1;
    return lv_guard([&] {
        if (f->gf.whisper_thread.joinable()) {
            mw::set_last_error("lv_filter_set_silero_model: stop the filter first (the worker thread is using its VAD)");
            return -2;
        }
        f->gf.silero_vad_model_file = path ? path : "";
        make_vad(f);  // throws (-> -100, mw_last_error) when the file is there but is not a Silero v5 graph
        return f->gf.vad->has_batch_source() ? 0 : 1;
    });
}
int lv_filter_vad_is_silero(lv_filter *f) { return f && f->gf.vad && f->gf.vad->has_batch_source() ? 1 : 0; }
int lv_filter_vad_probabilities(lv_filter *f, float *out, int cap) {
    if (!f || !f->gf.vad || (!out && cap > 0)) return -1;
    const std::vector<float> &p = f->gf.vad->last_probabilities();
    for (int i = 0; i < (int)p.size() && i < cap; ++i) out[i] = p[(size_t)i];
    return (int)p.size();
}
int lv_silero_inspect(const char *path, double *sums, int cap) {
    if (!path || (!sums && cap > 0)) return -1;
    return lv_guard([&] {
        const mw::SileroWeights w = mw::load_silero_onnx(path);
        const std::vector<float> out_b{w.out_b};
        const std::vector<float> *t[15] = {&w.basis, &w.c_w[0], &w.c_b[0], &w.c_w[1], &w.c_b[1], &w.c_w[2], &w.c_b[2], &w.c_w[3],
                                           &w.c_b[3], &w.w_ih, &w.w_hh, &w.b_ih, &w.b_hh, &w.out_w, &out_b};
        for (int k = 0; k < 15 && k < cap; ++k) {
            double s = 0.0;
            for (size_t i = 0; i < t[k]->size(); ++i) s += (double)((i % 7) + 1) * (double)(*t[k])[i];
            sums[k] = s;
        }
        return 15;
    });
}
void *lv_silero_create(const char *path, int device) {
    if (!path) return nullptr;
    mw::SileroVad *v = nullptr;
    lv_guard([&] {
        v = new mw::SileroVad(mw::load_silero_onnx(path), device);
        return 0;
    });
    return v;
}
int lv_silero_process(void *h, const float *wav, int n_windows, int reset, float *probs) {
    if (!h || n_windows < 0 || (n_windows > 0 && (!wav || !probs))) return -1;
    return lv_guard([&] {
        static_cast<mw::SileroVad *>(h)->process(wav, n_windows, reset != 0, probs);
        return 0;
    });
}
int lv_silero_process_batch(void *const *handles, const float *const *wavs, const int *n_windows, const int *resets,
                            float *const *probs, int n_streams) {
    if (!handles || !wavs || !n_windows || !resets || !probs || n_streams < 1) return -1;
    return lv_guard([&] {
        std::vector<mw::SileroBatchItem> items((size_t)n_streams);
        for (int i = 0; i < n_streams; ++i)
            items[(size_t)i] = mw::SileroBatchItem{static_cast<mw::SileroVad *>(handles[i]), wavs[i], n_windows[i], resets[i] != 0, probs[i]};
        mw::SileroVad::process_batch(items.data(), n_streams);
        return 0;
    });
}
int lv_silero_state(void *h, float *hc256) {
    if (!h || !hc256) return -1;
    return lv_guard([&] {
        static_cast<mw::SileroVad *>(h)->get_state(hc256);
        return 0;
    });
}
void lv_silero_destroy(void *h) { delete static_cast<mw::SileroVad *>(h); }

void lv_remove_leading_trailing_nonalpha(const char *in, char *out, size_t n) {
    if (!in || !out || n == 0) return;
    try {
        snprintf(out, n, "%s", remove_leading_trailing_nonalpha(in).c_str());
    } catch (...) {
        out[0] = 0;
    }
}
int lv_resample(lv_filter *f, const float *const *data, uint32_t frames, float *out, int cap) {
    if (!f || !data || !out) return -1;
    return lv_guard([&] {
        std::vector<float> o;
        f->gf.resampler_to_whisper->resample(data, frames, o);
        if ((int)o.size() > cap) return -2;
        memcpy(out, o.data(), o.size() * sizeof(float));
        return (int)o.size();
    });
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
    if (!f) return;
    try {
        shutdown_whisper_thread(&f->gf);  // context half + joins a worker if one was started
    } catch (...) {
    }
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
