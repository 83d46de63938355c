// api.hip — the C ABI of libmi355_whisper.so (include/whisper.h + include/mi355_whisper.h).
// Nothing crosses this boundary but plain pointers, sizes and PODs; no exception escapes.
#include "engine.hpp"

#include "../../include/mi355_whisper.h"

#include <fstream>
#include <sys/stat.h>
#include <mutex>

namespace mw {

namespace {
ggml_log_callback g_log_cb = nullptr;
void *g_log_user = nullptr;
std::mutex g_log_mutex;
thread_local std::string g_last_error;
}  // namespace

void set_last_error(const std::string &msg) { g_last_error = msg; }  // what mw_last_error() returns on this thread

void log_msg(int level, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    ggml_log_callback cb;
    void *user;
    {
        std::lock_guard<std::mutex> lk(g_log_mutex);
        cb = g_log_cb;
        user = g_log_user;
    }
    if (cb) cb((ggml_log_level)level, buf, user);
    else if (level <= LOG_WARN) fputs(buf, stderr);
}

namespace {

template <typename F> int guarded(const char *what, F &&f) {
    try {
        return f();
    } catch (const std::exception &e) {
        g_last_error = std::string(what) + ": " + e.what();
        log_msg(LOG_ERROR, "%s\n", g_last_error.c_str());
        return -100;
    } catch (...) {
        g_last_error = std::string(what) + ": unknown error";
        log_msg(LOG_ERROR, "%s\n", g_last_error.c_str());
        return -101;
    }
}

// what a context computes in: element type of the matrix weights / activations, and whether the decoder's matrices are
// additionally kept (and streamed) as e4m3
struct Precision {
    DType wt;
    int fp8;  // 0, 1 = e4m3 decoder weights, 2 = that + the MX-fp8 encoder products
    bool operator==(const Precision &o) const { return wt == o.wt && fp8 == o.fp8; }
};
Precision precision_of(int mw_precision) {
    return Precision{mw_precision == MW_PRECISION_FP32 ? DType::F32 : DType::BF16,
                     mw_precision == MW_PRECISION_FP8_MX ? 2 : mw_precision == MW_PRECISION_FP8 ? 1 : 0};
}
Precision default_precision() {
    const char *e = getenv("MI355_WHISPER_PRECISION");
    if (e && (strcmp(e, "fp32") == 0 || strcmp(e, "f32") == 0)) return precision_of(MW_PRECISION_FP32);
    if (e && (strcmp(e, "fp8") == 0 || strcmp(e, "f8") == 0)) return precision_of(MW_PRECISION_FP8);
    if (e && (strcmp(e, "fp8mx") == 0 || strcmp(e, "mxfp8") == 0)) return precision_of(MW_PRECISION_FP8_MX);
    return precision_of(MW_PRECISION_BF16);
}

whisper_context *init_from_buffer(const void *buffer, size_t size, int device, Precision wt) {
    whisper_context *ctx = nullptr;
    const int rc = guarded("whisper_init", [&]() {
        if (!buffer || size < 64) throw std::runtime_error("empty model buffer");
        std::unique_ptr<whisper_context> w(new whisper_context());
        w->c = create_context((const unsigned char *)buffer, size, device, wt.wt, wt.fp8);
        ctx = w.release();
        return 0;
    });
    return rc == 0 ? ctx : nullptr;
}

// Filter instances (OBS sources) that load the same model file on the same GPU share one HBM copy of the weights: the
// registry keeps a view without ownership plus weak references; the arena is freed with its last context.
struct SharedModelEntry {
    std::string path;
    int device;
    Precision wt;
    long long size, mtime_ns;
    Model view;  // owners reset
    std::weak_ptr<void> arena, mel_tab;
    std::weak_ptr<Hub> hub;
};
std::mutex g_models_mutex;
std::vector<SharedModelEntry> g_models;

bool file_identity(const char *path, long long &size, long long &mtime_ns) {
    struct stat sb;
    if (stat(path, &sb) != 0) return false;
    size = (long long)sb.st_size;
    mtime_ns = (long long)sb.st_mtim.tv_sec * 1000000000ll + sb.st_mtim.tv_nsec;
    return true;
}

whisper_context *init_shared_from(const Context &src) {
    whisper_context *ctx = nullptr;
    const int rc = guarded("whisper_init", [&]() {
        std::unique_ptr<whisper_context> w(new whisper_context());
        w->c = create_shared_context(src);
        ctx = w.release();
        return 0;
    });
    return rc == 0 ? ctx : nullptr;
}

whisper_context *init_from_file(const char *path, int device, Precision wt) {
    if (!path) return nullptr;
    long long fsize = 0, fmtime = 0;
    const bool have_id = getenv("MI355_WHISPER_NO_SHARED_WEIGHTS") == nullptr && file_identity(path, fsize, fmtime);
    if (have_id) {
        // a context whose Model is a view of the registered weights (kept alive by the locked owners while it is built)
        std::unique_ptr<Context> donor;
        {
            std::lock_guard<std::mutex> lk(g_models_mutex);
            for (size_t i = 0; i < g_models.size();) {
                SharedModelEntry &e = g_models[i];
                std::shared_ptr<void> a = e.arena.lock(), t = e.mel_tab.lock();
                if (!a || !t) {
                    g_models.erase(g_models.begin() + i);
                    continue;
                }
                if (!donor && e.path == path && e.device == device && e.wt == wt && e.size == fsize && e.mtime_ns == fmtime) {
                    donor.reset(new Context());
                    donor->device = device;
                    donor->model = e.view;
                    donor->model.arena_owner = a;
                    donor->model.mel_tab_owner = t;
                    donor->model.hub = e.hub.lock();
                }
                ++i;
            }
        }
        if (donor) {
            log_msg(LOG_INFO, "whisper_init_from_file: '%s' is already resident on device %d, sharing its weights\n", path, device);
            return init_shared_from(*donor);
        }
    }
    std::vector<unsigned char> buf;
    const int rc = guarded("whisper_init", [&]() {
        std::ifstream f(path, std::ios::binary | std::ios::ate);
        if (!f) throw std::runtime_error(std::string("failed to open ") + path);
        const std::streamsize n = f.tellg();
        f.seekg(0);
        buf.resize((size_t)n);
        if (!f.read((char *)buf.data(), n)) throw std::runtime_error(std::string("failed to read ") + path);
        return 0;
    });
    if (rc != 0) return nullptr;
    log_msg(LOG_INFO, "whisper_init_from_file: loading model from '%s'\n", path);
    whisper_context *ctx = init_from_buffer(buf.data(), buf.size(), device, wt);
    if (ctx && have_id) {
        std::lock_guard<std::mutex> lk(g_models_mutex);
        SharedModelEntry e{path, device, wt, fsize, fmtime, ctx->c->model, ctx->c->model.arena_owner, ctx->c->model.mel_tab_owner,
                           ctx->c->model.hub};
        e.view.arena_owner.reset();
        e.view.mel_tab_owner.reset();
        e.view.hub.reset();
        g_models.push_back(std::move(e));
    }
    return ctx;
}

struct DeviceGuard {
    explicit DeviceGuard(const Context &c) { MW_HIP(hipSetDevice(c.device)); }
};

}  // namespace
}  // namespace mw

using namespace mw;

namespace {
template <typename F> float time_launches(Context &c, int iters, F &&f) {
    hipEvent_t e0, e1;
    MW_HIP(hipEventCreate(&e0));
    MW_HIP(hipEventCreate(&e1));
    f();  // warm-up
    MW_HIP(hipStreamSynchronize(c.st));
    MW_HIP(hipEventRecord(e0, c.st));
    for (int i = 0; i < iters; ++i) f();
    MW_HIP(hipEventRecord(e1, c.st));
    MW_HIP(hipEventSynchronize(e1));
    float ms = 0;
    MW_HIP(hipEventElapsedTime(&ms, e0, e1));
    MW_HIP(hipEventDestroy(e0));
    MW_HIP(hipEventDestroy(e1));
    return ms / iters;
}
}  // namespace

extern "C" {

void whisper_log_set(ggml_log_callback log_callback, void *user_data) {
    std::lock_guard<std::mutex> lk(g_log_mutex);
    g_log_cb = log_callback;
    g_log_user = user_data;
}

struct whisper_context_params whisper_context_default_params(void) {
    whisper_context_params p;
    memset(&p, 0, sizeof(p));
    p.use_gpu = true;
    p.flash_attn = false;
    p.gpu_device = 0;
    p.dtw_token_timestamps = false;
    p.dtw_aheads_preset = WHISPER_AHEADS_NONE;
    p.dtw_n_top = -1;
    p.dtw_mem_size = 1024 * 1024 * 128;
    return p;
}
struct whisper_context_params *whisper_context_default_params_by_ref(void) {
    whisper_context_params *p = (whisper_context_params *)malloc(sizeof(whisper_context_params));
    if (p) *p = whisper_context_default_params();
    return p;
}
void whisper_free_context_params(struct whisper_context_params *params) { free(params); }

struct whisper_context *whisper_init_from_file_with_params(const char *path_model, struct whisper_context_params params) {
    return init_from_file(path_model, params.gpu_device, default_precision());
}
struct whisper_context *whisper_init_from_buffer_with_params(void *buffer, size_t buffer_size,
                                                             struct whisper_context_params params) {
    return init_from_buffer(buffer, buffer_size, params.gpu_device, default_precision());
}
struct whisper_context *mw_init_from_file(const char *path_model, int gpu_device, int precision) {
    return init_from_file(path_model, gpu_device, precision_of(precision));
}
struct whisper_context *mw_init_from_buffer(const void *buffer, size_t size, int gpu_device, int precision) {
    return init_from_buffer(buffer, size, gpu_device, precision_of(precision));
}
struct whisper_context *mw_init_shared(struct whisper_context *src) {
    if (!src || !src->c) return nullptr;
    return init_shared_from(*src->c);
}

const char *whisper_print_system_info(void) {
    return "MI355X = 1 | gfx950 = 1 | HIP = 1 | MFMA_BF16 = 1 | MFMA_F32 = 1 | CPU_FALLBACK = 0 | ";
}

void whisper_free(struct whisper_context *ctx) {
    if (!ctx) return;
    (void)guarded("whisper_free", [&]() {
        delete ctx;
        return 0;
    });
}

struct whisper_full_params whisper_full_default_params(enum whisper_sampling_strategy strategy) {
    whisper_full_params p;
    memset(&p, 0, sizeof(p));
    p.strategy = strategy;
    p.n_threads = 4;
    p.n_max_text_ctx = 16384;
    p.offset_ms = 0;
    p.duration_ms = 0;
    p.translate = false;
    p.no_context = true;
    p.no_timestamps = false;
    p.single_segment = false;
    p.print_special = false;
    p.print_progress = true;
    p.print_realtime = false;
    p.print_timestamps = true;
    p.token_timestamps = false;
    p.thold_pt = 0.01f;
    p.thold_ptsum = 0.01f;
    p.max_len = 0;
    p.split_on_word = false;
    p.max_tokens = 0;
    p.speed_up = false;
    p.debug_mode = false;
    p.audio_ctx = 0;
    p.tdrz_enable = false;
    p.suppress_regex = nullptr;
    p.initial_prompt = nullptr;
    p.prompt_tokens = nullptr;
    p.prompt_n_tokens = 0;
    p.language = "en";
    p.detect_language = false;
    p.suppress_blank = true;
    p.suppress_non_speech_tokens = false;
    p.temperature = 0.0f;
    p.max_initial_ts = 1.0f;
    p.length_penalty = -1.0f;
    p.temperature_inc = 0.2f;
    p.entropy_thold = 2.4f;
    p.logprob_thold = -1.0f;
    p.no_speech_thold = 0.6f;
    p.greedy.best_of = -1;
    p.beam_search.beam_size = -1;
    p.beam_search.patience = -1.0f;
    p.grammar_penalty = 100.0f;
    if (strategy == WHISPER_SAMPLING_GREEDY) p.greedy.best_of = 5;
    else p.beam_search.beam_size = 5;
    return p;
}
struct whisper_full_params *whisper_full_default_params_by_ref(enum whisper_sampling_strategy strategy) {
    whisper_full_params *p = (whisper_full_params *)malloc(sizeof(whisper_full_params));
    if (p) *p = whisper_full_default_params(strategy);
    return p;
}
void whisper_free_params(struct whisper_full_params *params) { free(params); }

int whisper_full(struct whisper_context *ctx, struct whisper_full_params params, const float *samples, int n_samples) {
    if (!ctx || !ctx->c) return -1;
    return guarded("whisper_full", [&]() {
        DeviceGuard g(*ctx->c);
        return full(*ctx->c, params, samples, n_samples);
    });
}

int mw_full_device(struct whisper_context *ctx, struct whisper_full_params params, const float *samples_device, int n_samples) {
    if (!ctx || !ctx->c || !samples_device || n_samples <= 0) return -1;
    return guarded("mw_full_device", [&]() {
        DeviceGuard g(*ctx->c);
        return full(*ctx->c, params, samples_device, n_samples, true);
    });
}

int whisper_full_n_segments(struct whisper_context *ctx) { return ctx && ctx->c ? (int)ctx->c->result_all.size() : 0; }
int whisper_full_n_tokens(struct whisper_context *ctx, int i) {
    if (!ctx || !ctx->c || i < 0 || i >= (int)ctx->c->result_all.size()) return 0;
    return (int)ctx->c->result_all[i].tokens.size();
}
whisper_token_data whisper_full_get_token_data(struct whisper_context *ctx, int i, int j) {
    whisper_token_data z = {0, 0, 0, 0, 0, 0, -1, -1, -1, 0};
    if (!ctx || !ctx->c || i < 0 || i >= (int)ctx->c->result_all.size()) return z;
    const auto &t = ctx->c->result_all[i].tokens;
    if (j < 0 || j >= (int)t.size()) return z;
    return t[j];
}
whisper_token whisper_full_get_token_id(struct whisper_context *ctx, int i, int j) {
    return whisper_full_get_token_data(ctx, i, j).id;
}
float whisper_full_get_token_p(struct whisper_context *ctx, int i, int j) {
    return whisper_full_get_token_data(ctx, i, j).p;
}
const char *whisper_token_to_str(struct whisper_context *ctx, whisper_token token) {
    if (!ctx || !ctx->c || token < 0 || token >= ctx->c->model.vocab.n_vocab) return "";
    return ctx->c->model.vocab.id_to_token[token].c_str();
}
int64_t whisper_full_get_segment_t0(struct whisper_context *ctx, int i) {
    return (ctx && ctx->c && i >= 0 && i < (int)ctx->c->result_all.size()) ? ctx->c->result_all[i].t0 : 0;
}
int64_t whisper_full_get_segment_t1(struct whisper_context *ctx, int i) {
    return (ctx && ctx->c && i >= 0 && i < (int)ctx->c->result_all.size()) ? ctx->c->result_all[i].t1 : 0;
}
const char *whisper_full_get_segment_text(struct whisper_context *ctx, int i) {
    return (ctx && ctx->c && i >= 0 && i < (int)ctx->c->result_all.size()) ? ctx->c->result_all[i].text.c_str() : "";
}
int whisper_full_lang_id(struct whisper_context *ctx) { return ctx && ctx->c ? ctx->c->lang_id : -1; }

int whisper_lang_auto_detect(struct whisper_context *ctx, int offset_ms, int n_threads, float *lang_probs) {
    (void)n_threads;
    if (!ctx || !ctx->c) return -1;
    return guarded("whisper_lang_auto_detect", [&]() {
        DeviceGuard g(*ctx->c);
        if (ctx->c->d_mel == nullptr) return -2;
        return lang_auto_detect(*ctx->c, offset_ms, lang_probs);
    });
}
const char *whisper_lang_str(int id) {
    const char *s = lang_str(id);
    return s ? s : "en";  // LocalVocal does not check the id (whisper-processing.cpp:236-237): never return null
}
int whisper_lang_id(const char *lang) { return lang_id(lang); }
int whisper_lang_max_id(void) { return N_LANGS - 1; }
int whisper_is_multilingual(struct whisper_context *ctx) {
    return ctx && ctx->c && ctx->c->model.vocab.is_multilingual() ? 1 : 0;
}
int whisper_n_vocab(struct whisper_context *ctx) { return ctx && ctx->c ? ctx->c->model.hp.n_vocab : 0; }
int whisper_n_text_ctx(struct whisper_context *ctx) { return ctx && ctx->c ? ctx->c->model.hp.n_text_ctx : 0; }
int whisper_n_audio_ctx(struct whisper_context *ctx) { return ctx && ctx->c ? ctx->c->model.hp.n_audio_ctx : 0; }
int whisper_n_len(struct whisper_context *ctx) { return ctx && ctx->c ? ctx->c->n_len : 0; }
#define MW_TOKEN_GETTER(name, field) \
    whisper_token name(struct whisper_context *ctx) { return ctx && ctx->c ? ctx->c->model.vocab.field : -1; }
MW_TOKEN_GETTER(whisper_token_eot, eot)
MW_TOKEN_GETTER(whisper_token_sot, sot)
MW_TOKEN_GETTER(whisper_token_solm, solm)
MW_TOKEN_GETTER(whisper_token_prev, prev)
MW_TOKEN_GETTER(whisper_token_nosp, nosp)
MW_TOKEN_GETTER(whisper_token_not, not_)
MW_TOKEN_GETTER(whisper_token_beg, beg)
MW_TOKEN_GETTER(whisper_token_translate, translate)
MW_TOKEN_GETTER(whisper_token_transcribe, transcribe)
#undef MW_TOKEN_GETTER
whisper_token whisper_token_lang(struct whisper_context *ctx, int id) {
    return ctx && ctx->c ? ctx->c->model.vocab.lang_token(id) : -1;
}
int whisper_tokenize(struct whisper_context *ctx, const char *text, whisper_token *tokens, int n_max_tokens) {
    if (!ctx || !ctx->c || !text) return 0;
    int n = 0;
    const int rc = guarded("whisper_tokenize", [&]() {
        const auto t = ctx->c->model.vocab.tokenize(text);
        if ((int)t.size() > n_max_tokens) {
            n = -(int)t.size();
            return 0;
        }
        for (size_t i = 0; i < t.size(); ++i) tokens[i] = t[i];
        n = (int)t.size();
        return 0;
    });
    return rc == 0 ? n : 0;
}

// ---------------------------------------------------------------------------------------------
// extension entry points (mi355_whisper.h)
// ---------------------------------------------------------------------------------------------
const char *mw_last_error(void) { return g_last_error.c_str(); }

int mw_hparams(struct whisper_context *ctx, int32_t out[11]) {
    if (!ctx || !ctx->c || !out) return -1;
    const HParams &h = ctx->c->model.hp;
    const int v[11] = {h.n_vocab, h.n_audio_ctx, h.n_audio_state, h.n_audio_head, h.n_audio_layer, h.n_text_ctx,
                       h.n_text_state, h.n_text_head, h.n_text_layer, h.n_mels, h.ftype};
    for (int i = 0; i < 11; ++i) out[i] = v[i];
    return 0;
}

int mw_pcm_to_mel(struct whisper_context *ctx, const float *samples, int n_samples) {
    if (!ctx || !ctx->c || !samples || n_samples <= 0) return -1;
    return guarded("mw_pcm_to_mel", [&]() {
        DeviceGuard g(*ctx->c);
        pcm_to_mel(*ctx->c, samples, n_samples, false);
        MW_HIP(hipStreamSynchronize(ctx->c->st));
        return 0;
    });
}
int mw_pcm_to_mel_device(struct whisper_context *ctx, const float *samples_device, int n_samples) {
    if (!ctx || !ctx->c || !samples_device || n_samples <= 0) return -1;
    return guarded("mw_pcm_to_mel_device", [&]() {
        DeviceGuard g(*ctx->c);
        pcm_to_mel(*ctx->c, samples_device, n_samples, true);
        return 0;
    });
}
int mw_pcm_to_mel_noalloc(struct whisper_context *ctx, const float *samples_device, int n_samples, float *mel_out_device,
                          size_t mel_out_floats, int *n_len_out) {
    if (!ctx || !ctx->c || !samples_device || !mel_out_device || n_samples <= 0) return -1;
    return guarded("mw_pcm_to_mel_noalloc", [&]() {
        Context &c = *ctx->c;
        DeviceGuard g(c);
        mel_noalloc(samples_device, n_samples, c.model.d_mel_tab, c.model.hp.n_mels, mel_out_device, mel_out_floats, n_len_out,
                    c.device, c.st);
        MW_HIP(hipStreamSynchronize(c.st));
        return 0;
    });
}
int mw_mel_n_len(struct whisper_context *ctx) { return ctx && ctx->c ? ctx->c->n_len : -1; }
int mw_get_mel(struct whisper_context *ctx, float *out) {
    if (!ctx || !ctx->c || !out || !ctx->c->d_mel) return -1;
    return guarded("mw_get_mel", [&]() {
        Context &c = *ctx->c;
        DeviceGuard g(c);
        const int n_mel = c.model.hp.n_mels;
        float *tmp = nullptr;
        MW_HIP(hipMalloc((void **)&tmp, (size_t)c.n_len * n_mel * 4));
        launch_transpose_f32(c.d_mel, tmp, c.n_len, n_mel, c.st);
        MW_HIP(hipMemcpyAsync(out, tmp, (size_t)c.n_len * n_mel * 4, hipMemcpyDeviceToHost, c.st));
        MW_HIP(hipStreamSynchronize(c.st));
        MW_HIP(hipFree(tmp));
        return 0;
    });
}
int mw_encode(struct whisper_context *ctx, int seek, int audio_ctx) {
    if (!ctx || !ctx->c || seek < 0) return -1;
    return guarded("mw_encode", [&]() {
        Context &c = *ctx->c;
        DeviceGuard g(c);
        if (audio_ctx < 0 || audio_ctx > c.model.hp.n_audio_ctx) return -5;
        c.exp_n_audio_ctx = audio_ctx;
        encode(c, seek);
        MW_HIP(hipStreamSynchronize(c.st));
        return 0;
    });
}

namespace {
// copy `n` elements of the context's compute type to host fp32
void fetch_as_f32(Context &c, const void *src, size_t n, float *out) {
    if (c.model.wt == DType::F32) {
        MW_HIP(hipMemcpyAsync(out, src, n * 4, hipMemcpyDeviceToHost, c.st));
        MW_HIP(hipStreamSynchronize(c.st));
        return;
    }
    std::vector<bf16_t> tmp(n);
    MW_HIP(hipMemcpyAsync(tmp.data(), src, n * 2, hipMemcpyDeviceToHost, c.st));
    MW_HIP(hipStreamSynchronize(c.st));
    for (size_t i = 0; i < n; ++i) out[i] = bf16_to_f32_host(tmp[i]);
}
}  // namespace

int mw_get_encoder_out(struct whisper_context *ctx, float *out) {
    if (!ctx || !ctx->c || !out) return -1;
    return guarded("mw_get_encoder_out", [&]() {
        Context &c = *ctx->c;
        DeviceGuard g(c);
        fetch_as_f32(c, c.d_enc_out, (size_t)c.n_ctx_cur * c.model.hp.n_audio_state, out);
        return 0;
    });
}
int mw_get_cross_kv(struct whisper_context *ctx, int layer, int is_v, float *out) {
    if (!ctx || !ctx->c || !out) return -1;
    return guarded("mw_get_cross_kv", [&]() {
        Context &c = *ctx->c;
        DeviceGuard g(c);
        const HParams &hp = c.model.hp;
        if (layer < 0 || layer >= hp.n_text_layer) return -2;
        const size_t ws = dtype_size(c.model.wt), per = (size_t)hp.n_audio_ctx * hp.n_text_state;
        const char *p = (const char *)c.d_cross + ((size_t)layer * 2 + (is_v ? 1 : 0)) * per * ws;
        fetch_as_f32(c, p, (size_t)c.n_ctx_cur * hp.n_text_state, out);
        return 0;
    });
}
int mw_kv_clear(struct whisper_context *ctx) {
    if (!ctx || !ctx->c) return -1;
    return guarded("mw_kv_clear", [&]() {
        DeviceGuard g(*ctx->c);
        kv_clear(*ctx->c);
        return 0;
    });
}
int mw_decode(struct whisper_context *ctx, const whisper_token *tokens, int n, int n_past, int seq) {
    if (!ctx || !ctx->c || !tokens || n <= 0 || n_past < 0 || seq < 0 || seq >= MAX_DECODERS) return -1;
    return guarded("mw_decode", [&]() {
        Context &c = *ctx->c;
        DeviceGuard g(c);
        if (n_past + n > c.model.hp.n_text_ctx) return -2;
        c.ps_armed = c.ps_fused = false;  // stage call: logits only, no sampler rows ride along
        std::vector<int> pos(n), slot(n, seq), rows;
        for (int i = 0; i < n; ++i) pos[i] = n_past + i;
        for (int i = std::max(0, n - LOGIT_ROWS); i < n; ++i) rows.push_back(i);
        // stage API: sequence `seq` keeps its own keys/values (identity index table)
        std::vector<int> idx((size_t)KV_SLOTS * c.model.hp.n_text_ctx);
        for (int s = 0; s < KV_SLOTS; ++s)
            for (int p = 0; p < c.model.hp.n_text_ctx; ++p) idx[(size_t)s * c.model.hp.n_text_ctx + p] = s;
        MW_HIP(hipMemcpyAsync(c.d_kv_idx[c.kv_cur], idx.data(), idx.size() * sizeof(int), hipMemcpyHostToDevice, c.st));
        MW_HIP(hipStreamSynchronize(c.st));
        decode(c, tokens, pos.data(), slot.data(), n, rows.data(), (int)rows.size());
        MW_HIP(hipStreamSynchronize(c.st));
        return 0;
    });
}
int mw_decode_rows(struct whisper_context *ctx, const whisper_token *tokens, const int *pos, const int *seq, int n) {
    if (!ctx || !ctx->c || !tokens || !pos || !seq || n <= 0 || n > LOGIT_ROWS) return -1;
    return guarded("mw_decode_rows", [&]() {
        Context &c = *ctx->c;
        DeviceGuard g(c);
        const int TC = c.model.hp.n_text_ctx;
        for (int i = 0; i < n; ++i)
            if (pos[i] < 0 || pos[i] >= TC || seq[i] < 0 || seq[i] >= MAX_DECODERS) return -2;
        c.ps_armed = c.ps_fused = false;  // stage call: logits only
        std::vector<int> rows(n);
        for (int i = 0; i < n; ++i) rows[i] = i;
        std::vector<int> idx((size_t)KV_SLOTS * TC);  // every sequence keeps its own keys/values (identity index table)
        for (int s = 0; s < KV_SLOTS; ++s)
            for (int p = 0; p < TC; ++p) idx[(size_t)s * TC + p] = s;
        MW_HIP(hipMemcpyAsync(c.d_kv_idx[c.kv_cur], idx.data(), idx.size() * sizeof(int), hipMemcpyHostToDevice, c.st));
        MW_HIP(hipStreamSynchronize(c.st));
        decode(c, tokens, pos, seq, n, rows.data(), n);
        MW_HIP(hipStreamSynchronize(c.st));
        return 0;
    });
}
int mw_get_logits(struct whisper_context *ctx, float *out) {
    if (!ctx || !ctx->c || !out) return -1;
    return guarded("mw_get_logits", [&]() {
        Context &c = *ctx->c;
        DeviceGuard g(c);
        MW_HIP(hipMemcpyAsync(out, c.d_logits, (size_t)c.n_logit_rows * c.model.hp.n_vocab * 4, hipMemcpyDeviceToHost, c.st));
        MW_HIP(hipStreamSynchronize(c.st));
        return 0;
    });
}
int mw_process_logits(struct whisper_context *ctx, const struct whisper_full_params *params, float temperature,
                      const float *logits_in, const whisper_token *history, int n_history, int has_ts, int seek_delta,
                      int n_draws, const double *u, float *logits_out, float *logprobs_out, float *probs_out,
                      whisper_token_data *tok_out) {
    if (!ctx || !ctx->c || !params || !logits_in || n_history < 0 || n_draws < 0 || n_draws > MAX_CAND) return -1;
    return guarded("mw_process_logits", [&]() {
        Context &c = *ctx->c;
        DeviceGuard g(c);
        const Vocab &v = c.model.vocab;
        const size_t nv = v.n_vocab;
        upload_suppress_mask(c, *params);
        MW_HIP(hipMemcpyAsync(c.d_logits, logits_in, nv * 4, hipMemcpyHostToDevice, c.st));
        LogitsRowState &s = c.h_row_state[0];
        s.logits_row = 0;
        s.n_sampled = n_history;
        s.last_is_ts = n_history > 0 && history[n_history - 1] >= v.beg;
        s.penult_is_ts = n_history < 2 || history[n_history - 2] >= v.beg;
        s.has_ts = has_ts;
        s.seek_delta = seek_delta;
        for (int k = 0; k < n_draws; ++k) s.u[k] = u ? u[k] : 0.5;
        MW_HIP(hipMemcpyAsync(c.d_row_state, &s, sizeof(s), hipMemcpyHostToDevice, c.st));
        float *dump = nullptr;
        const bool want_dump = logits_out || logprobs_out || probs_out;
        if (want_dump) MW_HIP(hipMalloc((void **)&dump, 3 * nv * 4));
        LogitsParams lp = make_logits_params(c, *params, temperature, n_draws);
        launch_process_logits(c.d_logits, (long)nv, c.d_suppress, lp, c.d_row_state, c.d_row_res, 1, c.d_work, dump,
                              dump ? dump + nv : nullptr, dump ? dump + 2 * nv : nullptr, c.st);
        MW_HIP(hipMemcpyAsync(c.h_row_res, c.d_row_res, sizeof(LogitsRowResult), hipMemcpyDeviceToHost, c.st));
        if (logits_out) MW_HIP(hipMemcpyAsync(logits_out, dump, nv * 4, hipMemcpyDeviceToHost, c.st));
        if (logprobs_out) MW_HIP(hipMemcpyAsync(logprobs_out, dump + nv, nv * 4, hipMemcpyDeviceToHost, c.st));
        if (probs_out) MW_HIP(hipMemcpyAsync(probs_out, dump + 2 * nv, nv * 4, hipMemcpyDeviceToHost, c.st));
        MW_HIP(hipStreamSynchronize(c.st));
        if (dump) MW_HIP(hipFree(dump));
        if (tok_out) {
            const LogitsRowResult &r = c.h_row_res[0];
            const int n_out = std::max(1, n_draws);
            for (int k = 0; k < n_out; ++k) {
                whisper_token_data t = {r.id[k], r.tid >= 0 ? r.tid : (n_draws > 1 ? v.beg : 0), r.p[k], r.plog[k], r.pt,
                                        r.ptsum, -1, -1, -1, 0.0f};
                if (t.id >= v.beg) {
                    t.tid = t.id;
                    t.pt = t.p;
                }
                tok_out[k] = t;
            }
        }
        return 0;
    });
}

int mw_get_timings(struct whisper_context *ctx, struct mw_timings *out) {
    if (!ctx || !ctx->c || !out) return -1;
    const Timings &t = ctx->c->tm;
    *out = {t.mel_ms, t.encode_ms, t.decode_ms, t.sample_ms, t.total_ms, t.n_encode, t.n_decode, t.n_sample};
    return 0;
}
int mw_reset_timings(struct whisper_context *ctx) {
    if (!ctx || !ctx->c) return -1;
    ctx->c->tm = Timings();
    return 0;
}
void *mw_stream(struct whisper_context *ctx) { return ctx && ctx->c ? (void *)ctx->c->st : nullptr; }
int mw_synchronize(struct whisper_context *ctx) {
    if (!ctx || !ctx->c) return -1;
    return guarded("mw_synchronize", [&]() {
        DeviceGuard g(*ctx->c);
        MW_HIP(hipStreamSynchronize(ctx->c->st));
        return 0;
    });
}
int mw_set_force_decode_steps(struct whisper_context *ctx, int n) {
    if (!ctx || !ctx->c || n < 0) return -1;
    ctx->c->force_decode_steps = n;
    return 0;
}

int mw_hub_stats(struct whisper_context *ctx, unsigned long long out[5]) {
    if (!ctx || !ctx->c || !out) return -1;
    for (int i = 0; i < 5; ++i) out[i] = 0;
    Hub *h = ctx->c->model.hub.get();
    if (!h || !ctx->c->use_hub) return 1;  // this context runs its own steps (fp32 mode or a width without tile kernels)
    std::lock_guard<std::mutex> lk(h->mu);
    out[0] = h->n_steps;
    out[1] = h->n_rows;
    out[2] = h->n_requests;
    std::lock_guard<std::mutex> lke(h->mu_enc);
    out[3] = h->n_enc_passes;
    out[4] = h->n_enc_windows;
    return 0;
}

int mw_set_duplicate_class(struct whisper_context *ctx, int step_class) {
    if (!ctx || !ctx->c || step_class < -1 || step_class >= GEMV_N_TAGS + 2) return -1;
    ctx->c->dup_tag = step_class;
    return 0;
}

int mw_profile_begin(struct whisper_context *ctx, int gemv_class, int max_launches) {
    if (!ctx || !ctx->c || gemv_class < 0 || gemv_class >= GEMV_N_TAGS + 2 || max_launches <= 0 || max_launches > (1 << 20)) return -1;
    return guarded("mw_profile_begin", [&]() {
        Context &c = *ctx->c;
        DeviceGuard g(c);
        while ((int)c.prof_ev.size() < 2 * max_launches) {
            hipEvent_t e;
            MW_HIP(hipEventCreate(&e));
            c.prof_ev.push_back(e);
        }
        c.prof_n = 0;
        c.prof_tag = gemv_class;
        return 0;
    });
}
int mw_profile_end(struct whisper_context *ctx, int *n_launches, float *total_ms) {
    if (!ctx || !ctx->c || !n_launches || !total_ms) return -1;
    return guarded("mw_profile_end", [&]() {
        Context &c = *ctx->c;
        DeviceGuard g(c);
        MW_HIP(hipStreamSynchronize(c.st));
        float total = 0.0f;
        for (int i = 0; i + 1 < c.prof_n; i += 2) {
            float ms = 0.0f;
            MW_HIP(hipEventElapsedTime(&ms, c.prof_ev[i], c.prof_ev[i + 1]));
            total += ms;
        }
        *n_launches = c.prof_n / 2;
        *total_ms = total;
        c.prof_tag = -1;
        c.prof_n = 0;
        return 0;
    });
}
int mw_bench_decode_step(struct whisper_context *ctx, int rows, int n_past, int iters, float *ms_per_step) {
    if (!ctx || !ctx->c || rows <= 0 || rows > MAX_DECODERS || iters <= 0 || !ms_per_step) return -1;
    return guarded("mw_bench_decode_step", [&]() {
        Context &c = *ctx->c;
        DeviceGuard g(c);
        if (n_past < 0 || n_past >= c.model.hp.n_text_ctx) return -2;
        std::vector<int> tok(rows, 100), pos(rows, n_past), slot(rows), lr(rows);
        for (int i = 0; i < rows; ++i) slot[i] = lr[i] = i;
        *ms_per_step = time_launches(c, iters, [&]() { decode(c, tok.data(), pos.data(), slot.data(), rows, lr.data(), rows); });
        return 0;
    });
}
// ---- MX-fp8 stage entries (include/mi355_whisper.h): the quantisers, the LayerNorm producer and the product of gemm_mx.hip
// on host arrays, for the parity tests; operands come back decoded (e4m3 byte x 2^(scale - 127)) ----
namespace {
float e4m3_decode(unsigned char v) {
    const int s = v >> 7, e = (v >> 3) & 15, mnt = v & 7;
    float x = e == 0 ? ldexpf((float)mnt, -9) : (e == 15 && mnt == 7) ? NAN : ldexpf(1.0f + mnt / 8.0f, e - 7);
    return s ? -x : x;
}
struct DevBuf {
    void *p = nullptr;
    explicit DevBuf(size_t n) { MW_HIP(hipMalloc(&p, std::max<size_t>(n, 16))); }
    ~DevBuf() { (void)hipFree(p); }
    DevBuf(const DevBuf &) = delete;
    float *f() { return (float *)p; }
    unsigned char *u8() { return (unsigned char *)p; }
};
void decode_rows(const std::vector<unsigned char> &q, const std::vector<unsigned char> &s, int rows, int K, bool act, float *out) {
    const int RB = act ? mx_row_blocks_a(rows) : rows / 64;
    for (int r = 0; r < rows; ++r)
        for (int k = 0; k < K; ++k) {
            const int sb = s[act ? mx_sa_off(r, k >> 5, RB) : mx_sw_off(r, k >> 5, RB)];
            out[(size_t)r * K + k] = e4m3_decode(q[(size_t)r * K + k]) * ldexpf(1.0f, sb - 127);
        }
}
}  // namespace
int mw_mx_gemm(int device, const float *A, int M, int K, const float *W, int N, const float *bias, const float *R, int epilogue, float *C_out,
               float *A_decoded, float *W_decoded) {
    if (!A || !W || !C_out || M <= 0 || N <= 0 || K <= 0 || K % 128 || N % 256) return -1;
    return guarded("mw_mx_gemm", [&]() {
        MW_HIP(hipSetDevice(device));
        hipStream_t st = nullptr;  // the null stream: a test entry, nothing to overlap with
        const size_t na = (size_t)M * K, nw = (size_t)N * K, nc = (size_t)M * N;
        std::vector<bf16_t> wb(nw);
        for (size_t i = 0; i < nw; ++i) wb[i] = f32_to_bf16_host(W[i]);
        DevBuf dA(na * 4), dWb(nw * 2), dA8(na), dW8(nw), dSA(mx_scale_bytes_a(M, K)), dSW(mx_scale_bytes_w(N, K)), dC(nc * 4),
            dSC(mx_scale_bytes_a(M, N)), dB((size_t)N * 4), dR(nc * 4);
        MW_HIP(hipMemcpy(dA.p, A, na * 4, hipMemcpyHostToDevice));
        MW_HIP(hipMemcpy(dWb.p, wb.data(), nw * 2, hipMemcpyHostToDevice));
        if (bias) MW_HIP(hipMemcpy(dB.p, bias, (size_t)N * 4, hipMemcpyHostToDevice));
        if (R) MW_HIP(hipMemcpy(dR.p, R, nc * 4, hipMemcpyHostToDevice));
        launch_quantize_mx_rows(dA.f(), K, M, K, dA8.u8(), K, dSA.u8(), st);
        launch_quantize_mx_weights(dWb.p, N, K, dW8.u8(), dSW.u8(), st);
        GemmMxArgs g;
        g.A = dA8.u8(); g.SA = dSA.u8(); g.lda = K; g.W = dW8.u8(); g.SW = dSW.u8();
        g.ldw = K; g.C = dC.p; g.ldc = N; g.bias = bias ? dB.f() : nullptr; g.M = M; g.N = N; g.K = K;
        if (epilogue == MXE_F32_RES) {
            if (!R) return -1;
            g.R = dR.f(); g.ldr = N;
        }
        if (epilogue == MXE_MX_GELU) g.SC = dSC.u8();
        launch_gemm_mx(g, epilogue, st);
        MW_HIP(hipDeviceSynchronize());
        if (epilogue == MXE_F32_RES) {
            MW_HIP(hipMemcpy(C_out, dC.p, nc * 4, hipMemcpyDeviceToHost));
        } else if (epilogue == MXE_BF16) {
            std::vector<bf16_t> h(nc);
            MW_HIP(hipMemcpy(h.data(), dC.p, nc * 2, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < nc; ++i) C_out[i] = bf16_to_f32_host(h[i]);
        } else {
            std::vector<unsigned char> q(nc), s(mx_scale_bytes_a(M, N));
            MW_HIP(hipMemcpy(q.data(), dC.p, nc, hipMemcpyDeviceToHost));
            MW_HIP(hipMemcpy(s.data(), dSC.p, s.size(), hipMemcpyDeviceToHost));
            decode_rows(q, s, M, N, true, C_out);
        }
        if (A_decoded) {
            std::vector<unsigned char> q(na), s(mx_scale_bytes_a(M, K));
            MW_HIP(hipMemcpy(q.data(), dA8.p, na, hipMemcpyDeviceToHost));
            MW_HIP(hipMemcpy(s.data(), dSA.p, s.size(), hipMemcpyDeviceToHost));
            decode_rows(q, s, M, K, true, A_decoded);
        }
        if (W_decoded) {
            std::vector<unsigned char> q(nw), s(mx_scale_bytes_w(N, K));
            MW_HIP(hipMemcpy(q.data(), dW8.p, nw, hipMemcpyDeviceToHost));
            MW_HIP(hipMemcpy(s.data(), dSW.p, s.size(), hipMemcpyDeviceToHost));
            decode_rows(q, s, N, K, false, W_decoded);
        }
        return 0;
    });
}
int mw_mx_layernorm(int device, const float *x, const float *w, const float *b, int rows, int d, float *out_decoded, float *out_bf16) {
    if (!x || !w || !b || !out_decoded || rows <= 0 || d <= 0 || d % 256 || d > 2048) return -1;
    return guarded("mw_mx_layernorm", [&]() {
        MW_HIP(hipSetDevice(device));
        hipStream_t st = nullptr;
        const size_t n = (size_t)rows * d;
        DevBuf dx(n * 4), dw((size_t)d * 4), db((size_t)d * 4), dq(n), ds(mx_scale_bytes_a(rows, d)), dh(n * 2);
        MW_HIP(hipMemcpy(dx.p, x, n * 4, hipMemcpyHostToDevice));
        MW_HIP(hipMemcpy(dw.p, w, (size_t)d * 4, hipMemcpyHostToDevice));
        MW_HIP(hipMemcpy(db.p, b, (size_t)d * 4, hipMemcpyHostToDevice));
        launch_layernorm_mx(dx.f(), d, dw.f(), db.f(), out_bf16 ? dh.p : nullptr, d, dq.u8(), d,
                            ds.u8(), rows, d, st);
        MW_HIP(hipDeviceSynchronize());
        std::vector<unsigned char> q(n), s(mx_scale_bytes_a(rows, d));
        MW_HIP(hipMemcpy(q.data(), dq.p, n, hipMemcpyDeviceToHost));
        MW_HIP(hipMemcpy(s.data(), ds.p, s.size(), hipMemcpyDeviceToHost));
        decode_rows(q, s, rows, d, true, out_decoded);
        if (out_bf16) {
            std::vector<bf16_t> h(n);
            MW_HIP(hipMemcpy(h.data(), dh.p, n * 2, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < n; ++i) out_bf16[i] = bf16_to_f32_host(h[i]);
        }
        return 0;
    });
}

int mw_bench_mel(struct whisper_context *ctx, int n_samples, int iters, float *ms_per_call) {
    if (!ctx || !ctx->c || n_samples <= 0 || iters <= 0 || !ms_per_call) return -1;
    return guarded("mw_bench_mel", [&]() {
        Context &c = *ctx->c;
        DeviceGuard g(c);
        std::vector<float> pcm(n_samples);
        for (int i = 0; i < n_samples; ++i) pcm[i] = 0.1f * sinf(0.01f * i) + 0.01f * sinf(0.37f * i);
        pcm_to_mel(c, pcm.data(), n_samples, false);
        MW_HIP(hipStreamSynchronize(c.st));
        const int n_mel = c.model.hp.n_mels;
        const int n_active = std::min((n_samples + 200) / MEL_HOP + 1, c.n_len);
        *ms_per_call = time_launches(c, iters, [&]() {
            launch_mel_frames(c.d_pcm, 0, n_samples, 1, c.model.d_mel_tab, c.d_mel_raw, (long)c.mel_cap_frames * n_mel,
                              n_active, n_mel, c.d_gmax, c.st);
            launch_mel_normalize(c.d_mel_raw, (long)c.mel_cap_frames * n_mel, c.d_mel, (long)c.mel_cap_frames * n_mel, c.n_len,
                                 n_active, n_mel, 1, c.d_gmax, c.st);
        });
        return 0;
    });
}
int mw_bench_encode(struct whisper_context *ctx, int iters, float *ms_per_call) {
    if (!ctx || !ctx->c || iters <= 0 || !ms_per_call) return -1;
    return guarded("mw_bench_encode", [&]() {
        Context &c = *ctx->c;
        DeviceGuard g(c);
        if (!c.d_mel) return -2;
        c.exp_n_audio_ctx = 0;
        *ms_per_call = time_launches(c, iters, [&]() { encode(c, 0); });
        return 0;
    });
}

}  // extern "C"
