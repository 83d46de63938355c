// segmentation.hip — the callers either side of run_whisper_inference, re-hosted on plain C++ (see
// localvocal.hpp for the reference file:line of each function).  Host code only; it is part of
// libmi355_whisper.so so the whole LocalVocal worker (audio tap -> segmentation -> whisper_full ->
// result callbacks) can run against the HIP engine without libobs.
#include "localvocal.hpp"
#include "silero_vad.hpp"

#include "common.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>

#define MAX_MS_WORK_BUFFER 11000  // whisper-processing.h:12

namespace {
constexpr int kLogInfo = mw::LOG_INFO, kLogWarn = mw::LOG_WARN, kLogError = mw::LOG_ERROR, kLogDebug = mw::LOG_DEBUG;
constexpr size_t kF = sizeof(float);
}  // namespace

uint64_t now_ns() {
    return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch())
        .count();
}
uint64_t now_ms() { return now_ns() / 1000000; }

// ---------------------------------------------------------------------------------------------
// ring_buffer
// ---------------------------------------------------------------------------------------------
void ring_buffer::reserve(size_t want) {
    if (want <= buf_.size()) return;
    size_t cap = std::max<size_t>(buf_.size() * 2, 4096);
    while (cap < want) cap *= 2;
    std::vector<uint8_t> nb(cap);
    if (size) {
        const size_t first = std::min(size, buf_.size() - head_);
        memcpy(nb.data(), buf_.data() + head_, first);
        memcpy(nb.data() + first, buf_.data(), size - first);
    }
    buf_.swap(nb);
    head_ = 0;
}

void ring_buffer::push_back(const void *data, size_t n) {
    if (n == 0) return;
    reserve(size + n);
    const size_t cap = buf_.size(), tail = (head_ + size) % cap, first = std::min(n, cap - tail);
    if (data) {
        memcpy(buf_.data() + tail, data, first);
        memcpy(buf_.data(), (const uint8_t *)data + first, n - first);
    } else {
        memset(buf_.data() + tail, 0, first);
        memset(buf_.data(), 0, n - first);
    }
    size += n;
}

void ring_buffer::push_front(const void *data, size_t n) {
    if (n == 0) return;
    reserve(size + n);
    const size_t cap = buf_.size();
    head_ = (head_ + cap - n) % cap;
    const size_t first = std::min(n, cap - head_);
    memcpy(buf_.data() + head_, data, first);
    memcpy(buf_.data(), (const uint8_t *)data + first, n - first);
    size += n;
}

void ring_buffer::peek_front(void *out, size_t n) const {
    MW_CHECK(n <= size, "ring_buffer: read beyond the stored bytes");
    if (n == 0 || !out) return;
    const size_t cap = buf_.size(), first = std::min(n, cap - head_);
    memcpy(out, buf_.data() + head_, first);
    memcpy((uint8_t *)out + first, buf_.data(), n - first);
}

void ring_buffer::pop_front(void *out, size_t n) {
    peek_front(out, n);
    if (n == 0) return;
    head_ = (head_ + n) % buf_.size();
    size -= n;
    if (size == 0) head_ = 0;
}

void ring_buffer::clear() {
    size = 0;
    head_ = 0;
}

// ---------------------------------------------------------------------------------------------
// stream_resampler
// ---------------------------------------------------------------------------------------------
namespace {
double bessel_i0(double x) {
    double sum = 1.0, term = 1.0;
    for (int k = 1; k < 32; ++k) {
        term *= (x / (2.0 * k)) * (x / (2.0 * k));
        sum += term;
        if (term < 1e-12 * sum) break;
    }
    return sum;
}
}  // namespace

stream_resampler::stream_resampler(uint32_t in_rate, size_t channels) : in_rate_(in_rate), channels_(channels) {
    MW_CHECK(in_rate >= 8000 && in_rate <= 384000, "resampler: unsupported input rate");
    MW_CHECK(channels >= 1 && channels <= MAX_PREPROC_CHANNELS, "resampler: unsupported channel count");
    cutoff_ = std::min(1.0, (double)WHISPER_SAMPLE_RATE / in_rate) * 0.94;  // fraction of input Nyquist
    half_taps_ = (int)std::ceil(16.0 / std::min(1.0, (double)WHISPER_SAMPLE_RATE / in_rate));
    hist_.assign(half_taps_, 0.0f);  // zero history so the first output sits at input sample 0
    pos_ = half_taps_;
}

void stream_resampler::resample(const float *const *in, uint32_t in_frames, std::vector<float> &out) {
    if (in_frames == 0) return;
    if (in_rate_ == WHISPER_SAMPLE_RATE && channels_ == 1) {  // identity: bit-exact pass-through
        out.insert(out.end(), in[0], in[0] + in_frames);
        return;
    }
    const size_t old = hist_.size();
    hist_.resize(old + in_frames);
    const float inv = 1.0f / (float)channels_;
    for (uint32_t i = 0; i < in_frames; ++i) {
        float s = 0.0f;
        for (size_t c = 0; c < channels_; ++c) s += in[c][i];
        hist_[old + i] = s * inv;
    }
    if (in_rate_ == WHISPER_SAMPLE_RATE) {  // downmix only
        out.insert(out.end(), hist_.begin() + half_taps_, hist_.end());
        hist_.assign(half_taps_, 0.0f);
        return;
    }
    const double step = (double)in_rate_ / WHISPER_SAMPLE_RATE, beta = 8.6, i0b = bessel_i0(beta);
    while (pos_ + half_taps_ < (double)hist_.size()) {
        const long centre = (long)std::floor(pos_);
        double acc = 0.0, wsum = 0.0;
        for (long k = centre - half_taps_ + 1; k <= centre + half_taps_; ++k) {
            const double t = (double)k - pos_, u = t / half_taps_;
            if (std::fabs(u) >= 1.0) continue;
            const double x = M_PI * t * cutoff_;
            const double sinc = std::fabs(x) < 1e-9 ? 1.0 : std::sin(x) / x;
            const double w = sinc * bessel_i0(beta * std::sqrt(1.0 - u * u)) / i0b;
            wsum += w;
            if (k >= 0) acc += w * hist_[k];
        }
        out.push_back((float)(acc / wsum));
        pos_ += step;
    }
    // drop input that no future output can reach
    const long keep_from = std::max(0L, (long)std::floor(pos_) - half_taps_);
    if (keep_from > 0) {
        hist_.erase(hist_.begin(), hist_.begin() + keep_from);
        pos_ -= (double)keep_from;
    }
}

// ---------------------------------------------------------------------------------------------
// VadIterator
// ---------------------------------------------------------------------------------------------
VadIterator::VadIterator(prob_fn fn, int Sample_rate, int windows_frame_size, float Threshold, int min_silence_duration_ms,
                         int speech_pad_ms, int min_speech_duration_ms, float max_speech_duration_s)
    : prob(std::move(fn)) {
    threshold = Threshold;
    sample_rate = Sample_rate;
    sr_per_ms = sample_rate / 1000;
    window_size_samples = (int64_t)windows_frame_size * sr_per_ms;
    min_speech_samples = sr_per_ms * min_speech_duration_ms;
    speech_pad_samples = sr_per_ms * speech_pad_ms;
    max_speech_samples = (float)sample_rate * max_speech_duration_s - (float)window_size_samples - 2.0f * (float)speech_pad_samples;
    min_silence_samples = sr_per_ms * min_silence_duration_ms;
    min_silence_samples_at_max_speech = sr_per_ms * 98;
}

VadIterator::prob_fn VadIterator::energy_gate(float gate_db, float slope_db) {
    return [gate_db, slope_db](const float *w, size_t n, bool) {
        double e = 0.0;
        for (size_t i = 0; i < n; ++i) e += (double)w[i] * w[i];
        const double db = 10.0 * std::log10(e / (double)std::max<size_t>(n, 1) + 1e-12);
        return (float)(1.0 / (1.0 + std::exp(-(db - gate_db) / slope_db)));
    };
}

void VadIterator::reset_states(bool reset_state) {
    if (reset_state) {
        state_was_reset = true;  // the recurrent state of the probability source starts over
        triggered = false;
    }
    temp_end = 0;
    current_sample = 0;
    prev_end = next_start = 0;
    speeches.clear();
    current_speech = timestamp_t();
}

// One window of the speech-timestamp automaton: start on p >= threshold, tentative end below
// threshold-0.15, confirmed end after min_silence, forced split after max_speech.
void VadIterator::predict(float p) {
    const unsigned int W = (unsigned int)window_size_samples;
    current_sample += W;

    if (p >= threshold) {
        if (temp_end != 0) {
            temp_end = 0;
            if (next_start < prev_end) next_start = (int)(current_sample - W);
        }
        if (!triggered) {
            triggered = true;
            current_speech.start = (int)(current_sample - W);
        }
        return;
    }
    if (triggered && (float)(current_sample - (unsigned int)current_speech.start) > max_speech_samples) {
        if (prev_end > 0) {
            current_speech.end = prev_end;
            speeches.push_back(current_speech);
            current_speech = timestamp_t();
            if (next_start < prev_end) triggered = false;
            else current_speech.start = next_start;
        } else {
            current_speech.end = (int)current_sample;
            speeches.push_back(current_speech);
            current_speech = timestamp_t();
            triggered = false;
        }
        prev_end = next_start = 0;
        temp_end = 0;
        return;
    }
    if (p >= threshold - 0.15) return;  // hysteresis band: keep the current state
    if (!triggered) return;
    if (temp_end == 0) temp_end = current_sample;
    if (current_sample - temp_end > (unsigned int)min_silence_samples_at_max_speech) prev_end = (int)temp_end;
    if (current_sample - temp_end < (unsigned int)min_silence_samples) return;
    current_speech.end = (int)temp_end;
    if (current_speech.end - current_speech.start > min_speech_samples) {
        speeches.push_back(current_speech);
        current_speech = timestamp_t();
        prev_end = next_start = 0;
        temp_end = 0;
        triggered = false;
    }
}

void VadIterator::process(const std::vector<float> &input_wav, bool reset_state) {
    reset_states(reset_state);
    audio_length_samples = (int)input_wav.size();
    const int n_windows = audio_length_samples / (int)window_size_samples;
    probs.assign((size_t)n_windows, 0.0f);
    if (batch) {  // every window of the call in one pass over the device network
        batch(input_wav.data(), n_windows, state_was_reset, probs.data());
        if (n_windows > 0) state_was_reset = false;
    } else {
        for (int j = 0; j < n_windows; ++j) {
            probs[(size_t)j] = prob(input_wav.data() + (size_t)j * window_size_samples, (size_t)window_size_samples, state_was_reset);
            state_was_reset = false;
        }
    }
    for (int j = 0; j < n_windows; ++j) predict(probs[(size_t)j]);
    if (current_speech.start >= 0) {  // speech still open at the end of the buffer
        current_speech.end = audio_length_samples;
        speeches.push_back(current_speech);
        current_speech = timestamp_t();
        prev_end = next_start = 0;
        temp_end = 0;
        triggered = false;
    }
}

void VadIterator::collect_chunks(const std::vector<float> &in, std::vector<float> &out) const {
    out.clear();
    for (const timestamp_t &s : speeches) out.insert(out.end(), in.begin() + s.start, in.begin() + s.end);
}

void VadIterator::drop_chunks(const std::vector<float> &in, std::vector<float> &out) const {
    out.clear();
    int cur = 0;
    for (const timestamp_t &s : speeches) {
        out.insert(out.end(), in.begin() + cur, in.begin() + s.start);
        cur = s.end;
    }
    out.insert(out.end(), in.begin() + cur, in.end());
}

VadIterator *build_vad(const char *silero_file, int device, float gate_db, float slope_db, float threshold) {
    std::unique_ptr<VadIterator> vad(new VadIterator(VadIterator::energy_gate(gate_db, slope_db), WHISPER_SAMPLE_RATE, 32, threshold, 100, 100, 100));
    bool readable = false;
    if (silero_file && silero_file[0]) {
        if (FILE *fp = fopen(silero_file, "rb")) {
            readable = true;
            fclose(fp);
        }
    }
    if (readable) {
        // the reference's Ort::Session on this file (silero-vad-onnx.cpp:47-93), as kernels on the GPU
        mw::log_msg(kLogDebug, "Create silero VAD: %s\n", silero_file);
        auto net = std::make_shared<mw::SileroVad>(mw::load_silero_onnx(silero_file), device);
        MW_CHECK(vad->get_window_size_samples() == mw::SileroVad::WINDOW, "silero: 32 ms windows at 16 kHz");
        vad->set_batch_source([net](const float *wav, int n_windows, bool reset, float *probs) { net->process(wav, n_windows, reset, probs); });
    } else if (silero_file && silero_file[0]) {
        mw::log_msg(kLogWarn, "Silero VAD model file not found (%s): the energy gate stands in\n", silero_file);
    } else {
        mw::log_msg(kLogDebug, "no Silero VAD model file configured: energy gate\n");
    }
    return vad.release();
}

void initialize_vad(transcription_filter_data *gf, const char *silero_vad_model_file) {
    // vad-processing.cpp:456-478: VadIterator(model, 16 kHz, 32 ms windows, 0.5, 100 / 100 / 100 ms)
    // Only ever called with no worker running (start_whisper_thread_with_path, before the thread exists): the worker uses
    // gf->vad outside whisper_ctx_mutex, so the object must not be replaced under it.
    gf->vad.reset(build_vad(silero_vad_model_file, gf->gpu_device, gf->vad_gate_db, gf->vad_gate_slope_db, gf->vad_threshold));
}

// ---------------------------------------------------------------------------------------------
// settings -> whisper_full_params
// ---------------------------------------------------------------------------------------------
long long lv_settings::get_int(const std::string &k) const {
    auto it = ints.find(k);
    if (it != ints.end()) return it->second;
    auto d = def_ints.find(k);
    return d != def_ints.end() ? d->second : 0;
}
double lv_settings::get_double(const std::string &k) const {
    auto it = doubles.find(k);
    if (it != doubles.end()) return it->second;
    auto d = def_doubles.find(k);
    return d != def_doubles.end() ? d->second : 0.0;
}
bool lv_settings::get_bool(const std::string &k) const {
    auto it = bools.find(k);
    if (it != bools.end()) return it->second;
    auto d = def_bools.find(k);
    return d != def_bools.end() ? d->second : false;
}
const char *lv_settings::get_string(const std::string &k) const {
    auto it = strings.find(k);
    if (it != strings.end()) return it->second.c_str();
    auto d = def_strings.find(k);
    return d != def_strings.end() ? d->second.c_str() : "";
}

void apply_whisper_params_defaults_on_settings(lv_settings *s) {
    const whisper_full_params d = whisper_full_default_params(WHISPER_SAMPLING_BEAM_SEARCH);
    s->def_ints = {{"strategy", WHISPER_SAMPLING_BEAM_SEARCH}, {"n_threads", d.n_threads},
                   {"n_max_text_ctx", d.n_max_text_ctx},       {"offset_ms", d.offset_ms},
                   {"duration_ms", d.duration_ms},             {"max_len", d.max_len},
                   {"max_tokens", d.max_tokens},               {"audio_ctx", d.audio_ctx},
                   {"greedy.best_of", d.greedy.best_of},       {"beam_search.beam_size", d.beam_search.beam_size}};
    s->def_bools = {{"whisper_translate", d.translate},
                    {"no_context", d.no_context},
                    {"no_timestamps", d.no_timestamps},
                    {"single_segment", d.single_segment},
                    {"print_special", false},
                    {"print_progress", false},
                    {"print_realtime", false},
                    {"print_timestamps", false},
                    {"token_timestamps", d.token_timestamps},
                    {"split_on_word", d.split_on_word},
                    {"debug_mode", d.debug_mode},
                    {"tdrz_enable", d.tdrz_enable},
                    {"detect_language", d.detect_language},
                    {"suppress_blank", false},
                    {"suppress_non_speech_tokens", false}};
    s->def_doubles = {{"thold_pt", d.thold_pt},
                      {"thold_ptsum", d.thold_ptsum},
                      {"temperature", d.temperature},
                      {"max_initial_ts", d.max_initial_ts},
                      {"length_penalty", d.length_penalty},
                      {"temperature_inc", d.temperature_inc},
                      {"entropy_thold", d.entropy_thold},
                      {"logprob_thold", d.logprob_thold},
                      {"no_speech_thold", d.no_speech_thold},
                      {"beam_search.patience", d.beam_search.patience}};
    s->def_strings = {{"suppress_regex", d.suppress_regex ? d.suppress_regex : ""},
                      {"initial_prompt", d.initial_prompt ? d.initial_prompt : ""}};
}

// The string members point into `settings`, which must outlive the params (OBS settings do, too).
void apply_whisper_params_from_settings(whisper_full_params &p, lv_settings *s) {
    p = whisper_full_default_params((whisper_sampling_strategy)s->get_int("strategy"));
    p.n_threads = (int)s->get_int("n_threads");
    p.n_max_text_ctx = (int)s->get_int("n_max_text_ctx");
    p.offset_ms = (int)s->get_int("offset_ms");
    p.duration_ms = (int)s->get_int("duration_ms");
    p.translate = s->get_bool("whisper_translate");
    p.no_context = s->get_bool("no_context");
    p.no_timestamps = s->get_bool("no_timestamps");
    p.single_segment = s->get_bool("single_segment");
    p.print_special = s->get_bool("print_special");
    p.print_progress = s->get_bool("print_progress");
    p.print_realtime = s->get_bool("print_realtime");
    p.print_timestamps = s->get_bool("print_timestamps");
    p.token_timestamps = s->get_bool("token_timestamps");
    p.thold_pt = (float)s->get_double("thold_pt");
    p.thold_ptsum = (float)s->get_double("thold_ptsum");
    p.max_len = (int)s->get_int("max_len");
    p.split_on_word = s->get_bool("split_on_word");
    p.max_tokens = (int)s->get_int("max_tokens");
    p.debug_mode = s->get_bool("debug_mode");
    p.audio_ctx = (int)s->get_int("audio_ctx");
    p.tdrz_enable = s->get_bool("tdrz_enable");
    p.suppress_regex = s->get_string("suppress_regex");
    p.initial_prompt = s->get_string("initial_prompt");
    p.detect_language = s->get_bool("detect_language");
    p.suppress_blank = s->get_bool("suppress_blank");
    p.suppress_non_speech_tokens = s->get_bool("suppress_non_speech_tokens");
    p.temperature = (float)s->get_double("temperature");
    p.max_initial_ts = (float)s->get_double("max_initial_ts");
    p.length_penalty = (float)s->get_double("length_penalty");
    p.temperature_inc = (float)s->get_double("temperature_inc");
    p.entropy_thold = (float)s->get_double("entropy_thold");
    p.logprob_thold = (float)s->get_double("logprob_thold");
    p.no_speech_thold = (float)s->get_double("no_speech_thold");
    p.greedy.best_of = (int)s->get_int("greedy.best_of");
    p.beam_search.beam_size = (int)s->get_int("beam_search.beam_size");
    p.beam_search.patience = (float)s->get_double("beam_search.patience");
}

// ---------------------------------------------------------------------------------------------
// audio tap and buffers
// ---------------------------------------------------------------------------------------------
void transcription_filter_setup_audio(transcription_filter_data *gf, uint32_t sample_rate, size_t channels) {
    MW_CHECK(channels >= 1 && channels <= MAX_PREPROC_CHANNELS, "setup_audio: channel count");
    std::lock_guard<std::mutex> lock(gf->whisper_buf_mutex);
    gf->channels = channels;
    gf->sample_rate = sample_rate;
    gf->frames = (size_t)((float)sample_rate / (1000.0f / MAX_MS_WORK_BUFFER));
    gf->last_num_frames = 0;
    for (size_t c = 0; c < MAX_PREPROC_CHANNELS; ++c) {
        gf->input_buffers[c].clear();
        gf->copy_buffers[c].assign(c < channels ? gf->frames : 0, 0.0f);
    }
    gf->info_buffer.clear();
    gf->whisper_buffer.clear();
    gf->resampled_buffer.clear();
    gf->resampler_to_whisper.reset(new stream_resampler(sample_rate, channels));
    gf->last_sub_render_time = now_ms();
    gf->start_timestamp_ms = now_ms();
}

// what both the OBS audio tap and the offline harness do once they decide to feed the worker
// (transcription-filter.cpp:102-116, localvocal-offline-test.cpp:527-560)
void transcription_filter_push_packet(transcription_filter_data *gf, const float *const *data, uint32_t frames,
                                      uint64_t timestamp_offset_ns) {
    std::lock_guard<std::mutex> lock(gf->whisper_buf_mutex);
    for (size_t c = 0; c < gf->channels; ++c) gf->input_buffers[c].push_back(data ? data[c] : nullptr, frames * kF);
    transcription_filter_audio_info info = {};
    info.frames = frames;
    info.timestamp_offset_ns = timestamp_offset_ns;
    gf->info_buffer.push_back(&info, sizeof(info));
    gf->wshiper_thread_cv.notify_one();
}

void transcription_filter_filter_audio(transcription_filter_data *gf, const float *const *data, uint32_t frames) {
    if (gf == nullptr || data == nullptr || frames == 0) return;
    if (!gf->active) return;
    if (gf->whisper_context == nullptr) return;  // no model loaded: audio passes through untouched
    transcription_filter_push_packet(gf, data, frames, now_ns() - gf->start_timestamp_ms * 1000000ull);
}

int get_data_from_buf_and_resample(transcription_filter_data *gf, uint64_t &start_timestamp_offset_ns,
                                   uint64_t &end_timestamp_offset_ns) {
    uint32_t num_frames_from_infos = 0;
    {
        std::lock_guard<std::mutex> lock(gf->whisper_buf_mutex);
        if (gf->input_buffers[0].size == 0) return 1;
        const size_t max_num_frames = (size_t)gf->sample_rate * 10;  // at most 10 s per pass
        transcription_filter_audio_info info = {};
        while (gf->info_buffer.size >= sizeof(info)) {
            gf->info_buffer.pop_front(&info, sizeof(info));
            num_frames_from_infos += info.frames;
            if (start_timestamp_offset_ns == 0) start_timestamp_offset_ns = info.timestamp_offset_ns;
            if (num_frames_from_infos > max_num_frames) {  // over the cap: this packet waits for the next pass
                num_frames_from_infos -= info.frames;
                gf->info_buffer.push_front(&info, sizeof(info));
                break;
            }
        }
        // (the end timestamp is taken from the last packet looked at, pushed back or not, as in the reference)
        end_timestamp_offset_ns = info.timestamp_offset_ns + (uint64_t)info.frames * 1000000000ull / gf->sample_rate;
        if (start_timestamp_offset_ns > end_timestamp_offset_ns)  // timestamp reset in the incoming media
            start_timestamp_offset_ns = end_timestamp_offset_ns - (uint64_t)num_frames_from_infos * 1000000000ull / gf->sample_rate;
        MW_CHECK(num_frames_from_infos <= gf->frames, "segmentation: packet run exceeds the work buffer");
        for (size_t c = 0; c < gf->channels; ++c) {
            std::fill(gf->copy_buffers[c].begin(), gf->copy_buffers[c].end(), 0.0f);
            gf->input_buffers[c].pop_front(gf->copy_buffers[c].data(), num_frames_from_infos * kF);
        }
    }
    gf->last_num_frames = num_frames_from_infos;
    const float *chans[MAX_PREPROC_CHANNELS];
    for (size_t c = 0; c < gf->channels; ++c) chans[c] = gf->copy_buffers[c].data();
    std::vector<float> resampled;
    gf->resampler_to_whisper->resample(chans, num_frames_from_infos, resampled);
    gf->resampled_buffer.push_back(resampled.data(), resampled.size() * kF);
    return 0;
}

// ---------------------------------------------------------------------------------------------
// result plumbing
// ---------------------------------------------------------------------------------------------
std::string remove_leading_trailing_nonalpha(const std::string &str) {
    // Observable behaviour of the reference function (transcription-utils.cpp:91-129): strings of up
    // to two bytes keep their alphabetic bytes; longer strings lose exactly their first byte (its two
    // find_if predicates `!isspace || !ispunct` are true for every byte, so nothing is trimmed at the
    // end and the erase at the front removes one byte — normally the leading space of the sentence).
    if (str.size() <= 2) {
        std::string out;
        for (unsigned char ch : str)
            if (std::isalpha(ch)) out.push_back((char)ch);
        return out;
    }
    return str.substr(1);
}

void set_text_callback(uint64_t /*possible_end_ts*/, transcription_filter_data *gf, const DetectionResultWithText &in) {
    DetectionResultWithText result = in;
    // English output is trimmed; other languages pass fix_utf8, which is the identity off Windows
    if (!(gf->whisper_params.language != nullptr && strcmp(gf->whisper_params.language, "en") != 0))
        result.text = remove_leading_trailing_nonalpha(result.text);
    if (gf->setTextCallback) gf->setTextCallback(result);
    if (!in.text.empty() && (in.result == DETECTION_RESULT_SPEECH || in.result == DETECTION_RESULT_PARTIAL)) {
        gf->last_sub_render_time = now_ms();
        gf->cleared_last_sub = false;
        if (in.result == DETECTION_RESULT_SPEECH) {  // full sentences feed the next call's initial prompt
            gf->last_transcription_sentence.push_back(in.text);
            while (gf->last_transcription_sentence.size() > (size_t)std::max(gf->n_context_sentences, 0))
                gf->last_transcription_sentence.pop_front();
        }
    }
}

void clear_current_caption(transcription_filter_data *gf) {
    if (gf->setTextCallback) gf->setTextCallback({DETECTION_RESULT_UNKNOWN, "", 0, 0, {}, ""});
    gf->last_transcription_sentence.clear();
    gf->cleared_last_sub = true;
}

void audio_chunk_callback(transcription_filter_data *gf, const float *pcm, size_t frames, int vad_state,
                          const DetectionResultWithText &result) {
    if (gf->audioChunkCallback) gf->audioChunkCallback(pcm, frames, vad_state, result);
}

void run_inference_and_callbacks(transcription_filter_data *gf, uint64_t start_offset_ms, uint64_t end_offset_ms,
                                 int vad_state) {
    // the whole whisper buffer between two 10 ms margins of silence; a partial run leaves it in place
    const size_t n = gf->whisper_buffer.size / kF, margin = WHISPER_SAMPLE_RATE / 100;
    std::vector<float> pcm(n + 2 * margin, 0.0f);
    if (vad_state == VAD_STATE_PARTIAL) gf->whisper_buffer.peek_front(pcm.data() + margin, n * kF);
    else gf->whisper_buffer.pop_front(pcm.data() + margin, n * kF);
    if (gf->inferenceRequestObserver) gf->inferenceRequestObserver(pcm.data(), pcm.size(), start_offset_ms, end_offset_ms, vad_state);
    const uint64_t inference_start_ts = now_ms();
    const DetectionResultWithText result = run_whisper_inference(gf, pcm.data(), pcm.size(), start_offset_ms, end_offset_ms, vad_state);
    set_text_callback(inference_start_ts, gf, result);
    if (gf->enable_audio_chunks_callback && vad_state != VAD_STATE_PARTIAL)
        audio_chunk_callback(gf, pcm.data(), pcm.size(), vad_state, result);
}

// ---------------------------------------------------------------------------------------------
// segmentation
// ---------------------------------------------------------------------------------------------
vad_state vad_disabled_segmentation(transcription_filter_data *gf, vad_state last) {
    uint64_t start_ns = 0, end_ns = 0;
    if (get_data_from_buf_and_resample(gf, start_ns, end_ns) != 0) {
        // nothing new: what is left in the whisper buffer goes out as a final segment
        if (gf->whisper_buffer.size > 0)
            run_inference_and_callbacks(gf, last.start_ts_offest_ms, last.end_ts_offset_ms, VAD_STATE_WAS_OFF);
        return last;
    }
    {  // everything resampled moves to the whisper buffer
        std::vector<uint8_t> tmp(gf->resampled_buffer.size);
        gf->resampled_buffer.pop_front(tmp.data(), tmp.size());
        gf->whisper_buffer.push_back(tmp.data(), tmp.size());
    }
    const uint64_t samples = gf->whisper_buffer.size / kF;
    const bool partial = samples < (uint64_t)(gf->segment_duration * WHISPER_SAMPLE_RATE / 1000);
    const uint64_t end_ms = end_ns / 1000000;
    if (!partial) {
        run_inference_and_callbacks(gf, last.start_ts_offest_ms, end_ms, VAD_STATE_WAS_OFF);
        return {false, end_ms, end_ms, end_ms};
    }
    const uint64_t unprocessed_ms = end_ms - last.last_partial_segment_end_ts;
    if (unprocessed_ms > (uint64_t)gf->partial_latency) {
        if (gf->partial_transcription) run_inference_and_callbacks(gf, last.start_ts_offest_ms, end_ms, VAD_STATE_PARTIAL);
        last.last_partial_segment_end_ts = end_ms;
    }
    return {false, last.start_ts_offest_ms, end_ms, last.last_partial_segment_end_ts};
}

vad_state vad_based_segmentation(transcription_filter_data *gf, vad_state last) {
    uint64_t start_ns = 0, end_ns = 0;
    if (get_data_from_buf_and_resample(gf, start_ns, end_ns) != 0) return last;

    const size_t W = (size_t)gf->vad->get_window_size_samples(), window_bytes = W * kF;
    if (gf->resampled_buffer.size < window_bytes * 8) return last;  // wait for 8 windows
    const size_t n_windows = gf->resampled_buffer.size / window_bytes;
    std::vector<float> vad_input(n_windows * W);
    gf->resampled_buffer.pop_front(vad_input.data(), vad_input.size() * kF);
    gf->vad->process(vad_input, !last.vad_on);

    const uint64_t start_ms = start_ns / 1000000, end_ms = end_ns / 1000000;
    vad_state cur = {false, start_ms, end_ms, last.last_partial_segment_end_ts};
    const std::vector<timestamp_t> stamps = gf->vad->get_speech_timestamps();
    if (stamps.empty()) {
        if (last.vad_on) {  // speech ended with the previous pass
            run_inference_and_callbacks(gf, last.start_ts_offest_ms, last.end_ts_offset_ms, VAD_STATE_WAS_ON);
            cur.last_partial_segment_end_ts = 0;
        }
        if (gf->enable_audio_chunks_callback)
            audio_chunk_callback(gf, vad_input.data(), vad_input.size(), VAD_STATE_IS_OFF,
                                 {DETECTION_RESULT_SILENCE, "[silence]", cur.start_ts_offest_ms, cur.end_ts_offset_ms, {}, ""});
        return cur;
    }
    for (size_t i = 0; i < stamps.size(); ++i) {
        // from the previous segment's end, or 100 ms ahead of the first one
        const int start_frame = i > 0 ? stamps[i - 1].end : std::max(0, stamps[i].start - WHISPER_SAMPLE_RATE / 10);
        const int end_frame = stamps[i].end;
        gf->whisper_buffer.push_back(vad_input.data() + start_frame, (size_t)(end_frame - start_frame) * kF);

        if (stamps[i].end < (int)vad_input.size()) {  // the segment closes inside this buffer
            const uint64_t segment_end_ts = start_ms + (uint64_t)(end_frame * 1000 / WHISPER_SAMPLE_RATE);
            run_inference_and_callbacks(gf, last.start_ts_offest_ms, segment_end_ts,
                                        last.vad_on ? VAD_STATE_WAS_ON : VAD_STATE_WAS_OFF);
            cur.vad_on = false;
            cur.start_ts_offest_ms = cur.end_ts_offset_ms;
            cur.end_ts_offset_ms = 0;
            cur.last_partial_segment_end_ts = 0;
            last = cur;
            continue;
        }
        // speech runs on past the end of this buffer
        cur.vad_on = true;
        cur.start_ts_offest_ms =
            last.vad_on ? last.start_ts_offest_ms : start_ms + (uint64_t)(start_frame * 1000 / WHISPER_SAMPLE_RATE);
        cur.end_ts_offset_ms = start_ms + (uint64_t)(end_frame * 1000 / WHISPER_SAMPLE_RATE);
        last = cur;
        if (!gf->partial_transcription) continue;
        const uint64_t from = cur.last_partial_segment_end_ts > 0 ? cur.last_partial_segment_end_ts : cur.start_ts_offest_ms;
        const uint64_t to = cur.end_ts_offset_ms > 0 ? cur.end_ts_offset_ms : cur.start_ts_offest_ms;
        if (to - from > (uint64_t)gf->partial_latency) {
            cur.last_partial_segment_end_ts = cur.end_ts_offset_ms;
            run_inference_and_callbacks(gf, cur.start_ts_offest_ms, cur.end_ts_offset_ms, VAD_STATE_PARTIAL);
        }
    }
    return cur;
}

vad_state hybrid_vad_segmentation(transcription_filter_data *gf, vad_state last) {
    uint64_t start_ns = 0, end_ns = 0;
    if (get_data_from_buf_and_resample(gf, start_ns, end_ns) != 0) return last;
    last.end_ts_offset_ms = end_ns / 1000000;
    {
        std::vector<uint8_t> tmp(gf->resampled_buffer.size);
        gf->resampled_buffer.pop_front(tmp.data(), tmp.size());
        gf->whisper_buffer.push_back(tmp.data(), tmp.size());
    }
    if (last.end_ts_offset_ms - last.start_ts_offest_ms >= (uint64_t)gf->segment_duration) {
        run_inference_and_callbacks(gf, last.start_ts_offest_ms, last.end_ts_offset_ms, VAD_STATE_WAS_ON);
        last.start_ts_offest_ms = end_ns / 1000000;
        last.last_partial_segment_end_ts = 0;
        return last;
    }
    if (!gf->partial_transcription) return last;
    const uint64_t from = last.last_partial_segment_end_ts > 0 ? last.last_partial_segment_end_ts : last.start_ts_offest_ms;
    const uint64_t to = last.end_ts_offset_ms > 0 ? last.end_ts_offset_ms : last.start_ts_offest_ms;
    if (to - from > (uint64_t)gf->partial_latency) {
        last.last_partial_segment_end_ts = last.end_ts_offset_ms;
        std::vector<float> vad_input(gf->whisper_buffer.size / kF);
        gf->whisper_buffer.peek_front(vad_input.data(), vad_input.size() * kF);
        gf->vad->process(vad_input, true);
        if (!gf->vad->get_speech_timestamps().empty()) {
            run_inference_and_callbacks(gf, last.start_ts_offest_ms, last.end_ts_offset_ms, VAD_STATE_PARTIAL);
        } else {
            // silence so far: drop it, keeping a quarter second as lead-in for the next segment
            const size_t keep = (WHISPER_SAMPLE_RATE / 4) * kF;
            // (the reference subtracts unconditionally; a buffer shorter than `keep` is left alone here
            // instead of underflowing)
            if (gf->whisper_buffer.size > keep) gf->whisper_buffer.pop_front(nullptr, gf->whisper_buffer.size - keep);
        }
    }
    return last;
}

// ---------------------------------------------------------------------------------------------
// worker thread
// ---------------------------------------------------------------------------------------------
bool whisper_loop_iteration(transcription_filter_data *gf, vad_state &cur) {
    {
        std::lock_guard<std::mutex> lock(gf->whisper_ctx_mutex);
        if (gf->whisper_context == nullptr) return false;
    }
    if (gf->clear_buffers) {
        gf->resampled_buffer.clear();
        gf->whisper_buffer.clear();
        cur = {false, now_ms(), 0, 0};
        gf->clear_buffers = false;
    }
    if (gf->vad_mode == VAD_MODE_HYBRID) cur = hybrid_vad_segmentation(gf, cur);
    else if (gf->vad_mode == VAD_MODE_ACTIVE) cur = vad_based_segmentation(gf, cur);
    else if (gf->vad_mode == VAD_MODE_DISABLED) cur = vad_disabled_segmentation(gf, cur);

    if (!gf->cleared_last_sub && (now_ms() - gf->last_sub_render_time) > gf->max_sub_duration) clear_current_caption(gf);
    if (gf->input_cv.has_value()) gf->input_cv->notify_one();
    return true;
}

void whisper_loop(void *data) {
    if (data == nullptr) {
        mw::log_msg(kLogError, "whisper_loop: data is null\n");
        return;
    }
    transcription_filter_data *gf = static_cast<transcription_filter_data *>(data);
    mw::log_msg(kLogDebug, "Starting whisper thread\n");
    vad_state cur = {false, 0, 0, 0};
    while (true) {
        bool alive = false;
        try {
            alive = whisper_loop_iteration(gf, cur);
        } catch (const std::exception &e) {
            mw::log_msg(kLogError, "whisper_loop: %s\n", e.what());
            alive = true;  // drop this pass, keep serving (the reference's worker has no exit on errors either)
        }
        if (!alive) {
            mw::log_msg(kLogWarn, "Whisper context is null, exiting thread\n");
            break;
        }
        // sleep until the audio tap signals new data, the context goes away, or 250 ms pass
        std::unique_lock<std::mutex> lock(gf->whisper_ctx_mutex);
        if (gf->input_buffers[0].size == 0) gf->wshiper_thread_cv.wait_for(lock, std::chrono::milliseconds(250));
    }
    mw::log_msg(kLogDebug, "Exiting whisper thread\n");
}

void start_whisper_thread_with_path(transcription_filter_data *gf, const std::string &whisper_model_path,
                                    const char *silero_vad_model_file) {
    std::lock_guard<std::mutex> lock(gf->whisper_ctx_mutex);
    if (gf->whisper_context != nullptr) {
        mw::log_msg(kLogError, "cannot init whisper: whisper_context is not null\n");
        return;
    }
    initialize_vad(gf, silero_vad_model_file);
    gf->whisper_context = init_whisper_context(whisper_model_path, gf);
    if (gf->whisper_context == nullptr) {
        mw::log_msg(kLogError, "Failed to initialize whisper context\n");
        return;
    }
    gf->whisper_model_file_currently_loaded = whisper_model_path;  // (gf->whisper_model_path is the CALLER's to set, as in the reference)
    std::thread new_whisper_thread(whisper_loop, gf);
    gf->whisper_thread.swap(new_whisper_thread);
}

void shutdown_whisper_thread(transcription_filter_data *gf) {
    if (gf->whisper_context != nullptr) {
        std::lock_guard<std::mutex> lock(gf->whisper_ctx_mutex);  // never free under a running whisper_full
        whisper_free(gf->whisper_context);
        gf->whisper_context = nullptr;
        gf->wshiper_thread_cv.notify_all();
    }
    if (gf->whisper_thread.joinable()) gf->whisper_thread.join();
    gf->whisper_model_path.clear();
}

// update_whisper_model (/root/reference/src/whisper-utils/whisper-model-utils.cpp:12-142), decision for decision:
//   * the settings' "whisper_model_path" is a catalogue name, or contains "!!!external!!!" and the file comes from
//     "whisper_model_path_external";
//   * same name as loaded and not external: nothing happens (the worker keeps running);
//   * another name: shutdown_whisper_thread FIRST (the context is freed under whisper_ctx_mutex, a worker parked in its
//     250 ms wait is woken and joined), then the catalogue lookup — an unknown name or a file that is not on disk leaves
//     the filter stopped (the reference opens its download dialog there; the downloader is out of scope);
//   * external: an identical file that is already loaded is left alone, anything else is a restart;
//   * a changed "dtw_token_timestamps" restarts the worker with gf->whisper_model_path — which shutdown_whisper_thread
//     has just cleared, so, exactly like the reference, that restart fails ("Failed to initialize whisper context") and
//     the filter stays stopped until the next call finds path "" != settings and loads the model with the new flag.
void update_whisper_model(transcription_filter_data *gf) {
    if (gf->context == nullptr) {
        mw::log_msg(kLogError, "obs_source_t context is null\n");
        return;
    }
    lv_settings *s = gf->context;
    const std::string new_model_path = s->get_string("whisper_model_path");
    const std::string external_model_file_path = s->get_string("whisper_model_path_external");
    const bool new_dtw_timestamps = s->get_bool("dtw_token_timestamps");
    const bool is_external_model = new_model_path.find("!!!external!!!") != std::string::npos;
    if (!is_external_model && new_model_path.empty()) {
        mw::log_msg(kLogWarn, "Whisper model path is empty\n");
        return;
    }
    if (is_external_model && external_model_file_path.empty()) {
        mw::log_msg(kLogWarn, "External model file path is empty\n");
        return;
    }
    const char *silero = gf->silero_vad_model_file.empty() ? "models/silero-vad/silero_vad.onnx" : gf->silero_vad_model_file.c_str();
    if (gf->whisper_model_path.empty() || gf->whisper_model_path != new_model_path || is_external_model) {
        if (gf->whisper_model_path != new_model_path) {
            mw::log_msg(kLogDebug, "model path changed from %s to %s\n", gf->whisper_model_path.c_str(), new_model_path.c_str());
            gf->whisper_model_loaded_new = !gf->whisper_model_path.empty();  // a switch, not the initial load
        }
        if (!is_external_model) {
            shutdown_whisper_thread(gf);
            const auto it = gf->models_info.find(new_model_path);
            if (it == gf->models_info.end()) {
                mw::log_msg(kLogWarn, "Model '%s' does not exist\n", new_model_path.c_str());
                return;
            }
            if (it->second.empty()) {
                mw::log_msg(kLogWarn, "Whisper model does not exist (download is outside this library)\n");
                return;
            }
            gf->whisper_model_path = new_model_path;
            start_whisper_thread_with_path(gf, it->second, silero);
        } else if (gf->whisper_model_file_currently_loaded == external_model_file_path) {
            mw::log_msg(kLogInfo, "External model file is already loaded\n");
            return;
        } else {
            shutdown_whisper_thread(gf);
            gf->whisper_model_path = new_model_path;
            start_whisper_thread_with_path(gf, external_model_file_path, silero);
        }
    } else {
        mw::log_msg(kLogDebug, "Model path did not change: %s == %s\n", gf->whisper_model_path.c_str(), new_model_path.c_str());
    }
    if (new_dtw_timestamps != gf->enable_token_ts_dtw) {
        mw::log_msg(kLogDebug, "dtw_token_timestamps changed from %d to %d\n", (int)gf->enable_token_ts_dtw, (int)new_dtw_timestamps);
        gf->enable_token_ts_dtw = new_dtw_timestamps;
        shutdown_whisper_thread(gf);
        start_whisper_thread_with_path(gf, gf->whisper_model_path, silero);
    }
}
