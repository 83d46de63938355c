// localvocal.hpp — host-side mirror of LocalVocal's whisper boundary (reference names and signatures).
//
// A maintainer swapping engines keeps src/whisper-utils/*.cpp as they are and links
// libmi355_whisper.so instead of whisper.cpp.  This mirror exists so the path and its
// immediate callers can be exercised and tested HERE without libobs: it is a fresh implementation of
// the same contracts
//   init_whisper_context              /root/reference/src/whisper-utils/whisper-processing.cpp:26-125
//   run_whisper_inference             /root/reference/src/whisper-utils/whisper-processing.cpp:127-324
//   run_inference_and_callbacks       /root/reference/src/whisper-utils/whisper-processing.cpp:326-359
//   whisper_loop                      /root/reference/src/whisper-utils/whisper-processing.cpp:361-431
//   start_whisper_thread_with_path /
//   shutdown_whisper_thread           /root/reference/src/whisper-utils/whisper-utils.cpp:9-51
//   to_timestamp                      /root/reference/src/whisper-utils/whisper-utils.cpp:150-161
//   get_data_from_buf_and_resample,
//   vad_disabled_segmentation,
//   vad_based_segmentation,
//   hybrid_vad_segmentation,
//   initialize_vad                    /root/reference/src/whisper-utils/vad-processing.cpp:24-475
//   VadIterator (state machine only)  /root/reference/src/whisper-utils/silero-vad-onnx.cpp:95-109,142-284
//   apply_whisper_params_*            /root/reference/src/whisper-utils/whisper-params.cpp:54-144
//   the audio tap                     /root/reference/src/transcription-filter.cpp:60-119
//   set_text_callback (result half)   /root/reference/src/transcription-filter-callbacks.cpp:272-287,419-432
//   remove_leading_trailing_nonalpha  /root/reference/src/transcription-utils.cpp:91-129
// over a transcription_filter_data reduced to the members those functions read
// (/root/reference/src/transcription-filter-data.h:93-253).  libobs pieces are replaced by plain C++:
// circlebuf -> ring_buffer, audio_resampler -> stream_resampler, obs_data_t -> lv_settings,
// Silero/ONNX Runtime (absent from this image) -> a pluggable speech-probability function with a
// deterministic energy gate as the default.
#pragma once

#include "../../include/whisper.h"
#include "catalogue.hpp"

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <functional>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <optional>
#include <string>
#include <thread>
#include <vector>

#define MAX_PREPROC_CHANNELS 10

enum DetectionResult {
    DETECTION_RESULT_UNKNOWN = 0,
    DETECTION_RESULT_SILENCE = 1,
    DETECTION_RESULT_SPEECH = 2,
    DETECTION_RESULT_SUPPRESSED = 3,
    DETECTION_RESULT_NO_INFERENCE = 4,
    DETECTION_RESULT_PARTIAL = 5,
};

enum VadState { VAD_STATE_WAS_ON = 0, VAD_STATE_WAS_OFF, VAD_STATE_IS_OFF, VAD_STATE_PARTIAL };
enum VadMode { VAD_MODE_ACTIVE = 0, VAD_MODE_HYBRID, VAD_MODE_DISABLED };

struct vad_state {
    bool vad_on;
    uint64_t start_ts_offest_ms;  // (sic) the reference's spelling, vad-processing.h:48
    uint64_t end_ts_offset_ms;
    uint64_t last_partial_segment_end_ts;
};

struct DetectionResultWithText {
    DetectionResult result;
    std::string text;
    uint64_t start_timestamp_ms;
    uint64_t end_timestamp_ms;
    std::vector<whisper_token_data> tokens;
    std::string language;
};

// ---------------------------------------------------------------------------------------------
// ring_buffer — byte FIFO with the four libobs circlebuf operations the path uses
// (push_back, push_front, pop_front with a null sink, peek_front) and a public `size` in bytes.
// ---------------------------------------------------------------------------------------------
struct ring_buffer {
    size_t size = 0;

    void push_back(const void *data, size_t n);
    void push_front(const void *data, size_t n);
    void pop_front(void *out, size_t n);  // out may be null: discard
    void peek_front(void *out, size_t n) const;
    void clear();

private:
    std::vector<uint8_t> buf_;
    size_t head_ = 0;
    void reserve(size_t want);
};

// ---------------------------------------------------------------------------------------------
// stream_resampler — planar float, `channels` at `in_rate`  ->  mono float at 16 kHz.
// Equal rates and one channel pass through bit-exactly; otherwise channels are averaged and a
// Kaiser-windowed sinc (streaming, phase-continuous across calls) converts the rate.  libobs uses
// libswresample here; bit parity with it is not claimed (no swresample in this image).
// ---------------------------------------------------------------------------------------------
struct stream_resampler {
    stream_resampler(uint32_t in_rate, size_t channels);
    // appends the produced 16 kHz samples to `out`
    void resample(const float *const *in, uint32_t in_frames, std::vector<float> &out);

private:
    uint32_t in_rate_;
    size_t channels_;
    int half_taps_ = 16;
    double cutoff_ = 1.0;
    std::vector<float> hist_;  // mono input not yet fully consumed
    double pos_ = 0.0;         // position of the next output sample, in input samples, relative to hist_[0]
};

// ---------------------------------------------------------------------------------------------
// VadIterator — the speech-timestamp state machine of silero-vad-onnx.cpp driven by a probability
// function instead of the ONNX session.
// ---------------------------------------------------------------------------------------------
struct timestamp_t {
    int start;
    int end;
    timestamp_t(int start_ = -1, int end_ = -1) : start(start_), end(end_) {}
    bool operator==(const timestamp_t &a) const { return start == a.start && end == a.end; }
};

class VadIterator {
public:
    // p(window) in [0,1]; `reset` is true on the first window after the recurrent state was cleared
    using prob_fn = std::function<float(const float *window, size_t n, bool reset)>;
    // all windows of one process() call at once (the Silero network on the GPU scores them in one pass): probs[j] for the
    // n_windows consecutive windows of `wav`; `reset` as above.  When set, it replaces prob_fn.
    using batch_fn = std::function<void(const float *wav, int n_windows, bool reset, float *probs)>;

    VadIterator(prob_fn fn, int Sample_rate = 16000, int windows_frame_size = 32, float Threshold = 0.5f,
                int min_silence_duration_ms = 0, int speech_pad_ms = 32, int min_speech_duration_ms = 32,
                float max_speech_duration_s = std::numeric_limits<float>::infinity());

    void process(const std::vector<float> &input_wav, bool reset_state = true);
    void collect_chunks(const std::vector<float> &input_wav, std::vector<float> &output_wav) const;
    void drop_chunks(const std::vector<float> &input_wav, std::vector<float> &output_wav) const;
    const std::vector<timestamp_t> get_speech_timestamps() const { return speeches; }
    void set_threshold(float threshold_) { threshold = threshold_; }
    void set_batch_source(batch_fn fn) { batch = std::move(fn); }
    bool has_batch_source() const { return (bool)batch; }
    const std::vector<float> &last_probabilities() const { return probs; }  // of the last process() call (tests)
    int64_t get_window_size_samples() const { return window_size_samples; }

    // default probability source: short-time energy gate (logistic in dB around `gate_db`)
    static prob_fn energy_gate(float gate_db = -40.0f, float slope_db = 3.0f);

private:
    void reset_states(bool reset_state);
    void predict(float speech_prob);

    prob_fn prob;
    batch_fn batch;
    std::vector<float> probs;
    bool state_was_reset = true;
    int64_t window_size_samples;
    int sample_rate;
    int sr_per_ms;
    float threshold;
    int min_silence_samples;
    int min_silence_samples_at_max_speech;
    int min_speech_samples;
    float max_speech_samples;
    int speech_pad_samples;
    int audio_length_samples = 0;

    bool triggered = false;
    unsigned int temp_end = 0;
    unsigned int current_sample = 0;
    int prev_end = 0;
    int next_start = 0;
    std::vector<timestamp_t> speeches;
    timestamp_t current_speech;
};

// obs_data_t stand-in: flat string-keyed settings with the reference's key names
struct lv_settings {
    std::map<std::string, long long> ints;
    std::map<std::string, double> doubles;
    std::map<std::string, bool> bools;
    std::map<std::string, std::string> strings;
    std::map<std::string, long long> def_ints;
    std::map<std::string, double> def_doubles;
    std::map<std::string, bool> def_bools;
    std::map<std::string, std::string> def_strings;

    long long get_int(const std::string &k) const;
    double get_double(const std::string &k) const;
    bool get_bool(const std::string &k) const;
    const char *get_string(const std::string &k) const;  // "" when unset (obs_data_get_string)
};

// Audio packet info (transcription-filter-data.h:256-259)
struct transcription_filter_audio_info {
    uint32_t frames;
    uint64_t timestamp_offset_ns;
};

struct transcription_filter_data {
    size_t channels = 1;
    uint32_t sample_rate = 16000;
    size_t frames = 0;           // capacity of copy_buffers, in input frames
    size_t last_num_frames = 0;
    uint64_t start_timestamp_ms = 0;
    size_t max_sub_duration = 3000;
    uint64_t last_sub_render_time = 0;
    bool cleared_last_sub = true;

    /* PCM buffers */
    std::vector<float> copy_buffers[MAX_PREPROC_CHANNELS];
    ring_buffer info_buffer;
    ring_buffer input_buffers[MAX_PREPROC_CHANNELS];
    std::atomic<bool> clear_buffers{false};
    ring_buffer whisper_buffer;

    /* Resampler */
    std::unique_ptr<stream_resampler> resampler_to_whisper;
    ring_buffer resampled_buffer;

    /* whisper */
    std::string whisper_model_path;
    struct whisper_context *whisper_context = nullptr;
    whisper_full_params whisper_params;

    /* VAD */
    std::unique_ptr<VadIterator> vad;
    // what initialize_vad builds it with (the reference passes 0.5 and the UI sets the threshold afterwards,
    // vad-processing.cpp:473-474); the gate shape only matters while the energy gate stands in for the Silero file
    float vad_gate_db = -40.0f, vad_gate_slope_db = 3.0f, vad_threshold = 0.5f;

    float sentence_psum_accept_thresh = 0.4f;
    int vad_mode = VAD_MODE_ACTIVE;
    int log_level = 5;
    bool log_words = false;
    bool active = true;
    bool enable_token_ts_dtw = false;
    bool enable_audio_chunks_callback = false;
    bool partial_transcription = false;
    int partial_latency = 1100;  // transcription-filter-properties.cpp:710
    float duration_filter_threshold = 2.25f;
    int segment_duration = 7000;

    int n_context_sentences = 0;
    std::deque<std::string> last_transcription_sentence;

    // downstream consumers (the reference routes these to OBS text sources / files / captions)
    std::function<void(const DetectionResultWithText &result)> setTextCallback;
    std::function<void(const float *pcm32f_data, size_t frames, int vad_state, const DetectionResultWithText &result)>
        audioChunkCallback;
    // observer fired at the top of run_inference_and_callbacks with what is about to be transcribed
    std::function<void(const float *pcm32f_data, size_t n, uint64_t t0, uint64_t t1, int vad_state)> inferenceRequestObserver;

    std::string whisper_model_file_currently_loaded;
    bool whisper_model_loaded_new = false;
    // what update_whisper_model reads through gf->context in the reference: the source's settings (obs_data_t stand-in) and
    // the model catalogue (models_info() + find_model_bin_file: catalogue name -> local file, "" = listed but not on disk;
    // the downloader behind it is out of scope)
    lv_settings *context = nullptr;
    std::map<std::string, std::string> models_info;
    std::map<std::string, ModelInfo> catalogue;  // what lv_filter_load_catalogue parsed (digests for lv_filter_verify_model)

    std::thread whisper_thread;
    std::mutex whisper_buf_mutex;
    std::mutex whisper_ctx_mutex;
    std::condition_variable wshiper_thread_cv;  // (sic)
    std::optional<std::condition_variable> input_cv;

    // engine placement (the reference only logs cparams.gpu_device, whisper-processing.cpp:62)
    int gpu_device = 0;
    // what obs_module_file("models/silero-vad/silero_vad.onnx") resolves to on this host (whisper-model-utils.cpp:49-56);
    // empty or not a file: the energy gate stands in (logged)
    std::string silero_vad_model_file;
    int precision = -1;  // -1: library default / MI355_WHISPER_PRECISION
    // storage behind the borrowed const char* members of whisper_params
    std::string language_storage, suppress_regex_storage, initial_prompt_storage;

    transcription_filter_data();
    ~transcription_filter_data();
};

struct whisper_context *init_whisper_context(const std::string &model_path, struct transcription_filter_data *gf);
struct DetectionResultWithText run_whisper_inference(struct transcription_filter_data *gf, const float *pcm32f_data,
                                                     size_t pcm32f_num_samples, uint64_t t0 = 0, uint64_t t1 = 0,
                                                     int vad_state = VAD_STATE_WAS_OFF);
void run_inference_and_callbacks(transcription_filter_data *gf, uint64_t start_offset_ms, uint64_t end_offset_ms,
                                 int vad_state);
void whisper_loop(void *data);
// one pass of the loop body (clear_buffers, segmentation dispatch, caption timeout); false = context gone
bool whisper_loop_iteration(transcription_filter_data *gf, vad_state &current_vad_state);
void start_whisper_thread_with_path(struct transcription_filter_data *gf, const std::string &whisper_model_path,
                                    const char *silero_vad_model_file);
void shutdown_whisper_thread(struct transcription_filter_data *gf);
// whisper-model-utils.cpp:12-142: reload the model when the settings name another one (or an external file, or the DTW flag)
void update_whisper_model(struct transcription_filter_data *gf);
void shutdown_whisper_context(struct transcription_filter_data *gf);
std::string to_timestamp(uint64_t t_ms_offset);

int get_data_from_buf_and_resample(transcription_filter_data *gf, uint64_t &start_timestamp_offset_ns,
                                   uint64_t &end_timestamp_offset_ns);
vad_state vad_disabled_segmentation(transcription_filter_data *gf, vad_state last_vad_state);
vad_state vad_based_segmentation(transcription_filter_data *gf, vad_state last_vad_state);
vad_state hybrid_vad_segmentation(transcription_filter_data *gf, vad_state last_vad_state);
void initialize_vad(transcription_filter_data *gf, const char *silero_vad_model_file);
// A VadIterator with the reference's parameters (16 kHz, 32 ms windows, 100 / 100 / 100 ms): the Silero network on `device`
// when `silero_file` names a readable file (throws if that file is not a Silero v5 graph), else the energy gate.
VadIterator *build_vad(const char *silero_file, int device, float gate_db, float slope_db, float threshold);

void apply_whisper_params_defaults_on_settings(lv_settings *s);
void apply_whisper_params_from_settings(whisper_full_params &params, lv_settings *settings);

// the audio tap: planar float packets at gf->sample_rate, stamped now - start of stream; dropped while
// the filter is inactive or no model is loaded
void transcription_filter_filter_audio(transcription_filter_data *gf, const float *const *data, uint32_t frames);
// the push itself, with an explicit timestamp (what the offline harness does; data == null pushes silence)
void transcription_filter_push_packet(transcription_filter_data *gf, const float *const *data, uint32_t frames,
                                      uint64_t timestamp_offset_ns);
// (re)allocates buffers + resampler for (sample_rate, channels); transcription-filter.cpp:560-600
void transcription_filter_setup_audio(transcription_filter_data *gf, uint32_t sample_rate, size_t channels);

void set_text_callback(uint64_t possible_end_ts, struct transcription_filter_data *gf, const DetectionResultWithText &str);
void clear_current_caption(transcription_filter_data *gf);
void audio_chunk_callback(struct transcription_filter_data *gf, const float *pcm32f_data, size_t frames, int vad_state,
                          const DetectionResultWithText &result);
std::string remove_leading_trailing_nonalpha(const std::string &str);
uint64_t now_ms();
uint64_t now_ns();
