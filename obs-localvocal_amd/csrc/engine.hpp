// engine.hpp — the whisper_context of the MI355X engine: model + per-context state + driver.
#pragma once

#include "model.hpp"

#include "../../include/whisper.h"

#include <memory>
#include <random>
#include <string>
#include <vector>

namespace mw {

constexpr int MAX_DECODERS = KV_SLOTS;  // WHISPER_MAX_DECODERS
constexpr int LOGIT_ROWS = 32;          // logits rows kept per decode call

struct Sequence {
    std::vector<whisper_token_data> tokens;
    int result_len = 0;
    double sum_logprobs_all = 0, sum_logprobs = 0, avg_logprobs = 0, entropy = 0, score = 0;
};

struct Decoder {
    Sequence sequence;
    int seek_delta = 0;
    bool failed = false, completed = false, has_ts = false;
    int i_batch = 0;
    std::mt19937 rng;
    // what the device sampler returned for this decoder's next token(s)
    LogitsRowResult sampled;
};

struct Segment {
    int64_t t0, t1;
    std::string text;
    std::vector<whisper_token_data> tokens;
};

struct Timings {
    double mel_ms = 0, encode_ms = 0, decode_ms = 0, sample_ms = 0, total_ms = 0;
    int n_encode = 0, n_decode = 0, n_sample = 0;
};

struct Context {
    int device = 0;
    hipStream_t st = nullptr;
    Model model;
    // ---- mel state ----
    int n_len = 0, n_len_org = 0;
    size_t pcm_cap = 0, mel_cap_frames = 0;
    float *d_pcm = nullptr, *d_mel_raw = nullptr, *d_mel = nullptr;
    unsigned *d_gmax = nullptr;
    // ---- encoder workspace ----
    int n_ctx_cur = 0;  // audio context of the last encode
    int tp = 0;         // padded key count of the unfused attention path
    void *d_conv_in = nullptr, *d_x1 = nullptr, *d_h = nullptr, *d_qk = nullptr, *d_vt = nullptr, *d_p = nullptr,
         *d_att = nullptr, *d_ff = nullptr, *d_enc_out = nullptr;
    float *d_x = nullptr, *d_s = nullptr;
    void *d_cross = nullptr;  // [L][2][n_audio_ctx][d]
    // ---- decoder workspace ----
    void *d_kself = nullptr, *d_vself = nullptr;  // [L][KV_SLOTS][n_text_ctx][d]
    int *d_kv_idx[2] = {nullptr, nullptr};
    int kv_cur = 0;
    int *d_tokens = nullptr, *d_pos = nullptr, *d_slot = nullptr, *d_rows = nullptr;
    float *d_xd = nullptr, *d_q = nullptr, *d_part = nullptr, *d_logits = nullptr, *d_work = nullptr;
    void *d_attd = nullptr, *d_ffd = nullptr;  // attention output / MLP hidden rows (compute type)
    unsigned char *d_suppress = nullptr;
    LogitsRowState *d_row_state = nullptr;
    LogitsRowResult *d_row_res = nullptr;
    // pinned host staging
    int *h_ints = nullptr;  // tokens | pos | slot | rows + index-table update (src[KV_SLOTS], n_past, flag): one copy per step
    int *h_ints_dev = nullptr;  // device view of h_ints: the step's first kernel reads the block from there (env MI355_WHISPER_NO_STEP_BEGIN: copy node)
    hipEvent_t ev_staging = nullptr;  // recorded behind each decode launch: the staging block is free once it has passed
    bool staging_pending = false;
    // index-table update of the next decode call (kv_reorder defers it into that call's launch sequence)
    bool reorder_pending = false;
    int reorder_src[KV_SLOTS] = {0}, reorder_n_past = 0;
    LogitsRowState *h_row_state = nullptr;
    LogitsRowResult *h_row_res = nullptr;
    LogitsRowResult *d_row_res_host = nullptr;  // device view of h_row_res: the fused sampler stores its results straight into it
    int n_logit_rows = 0;
    // rules + samplers of a step ride in the decode call that produces their logits: prepare_sample() fills the row
    // states (they travel in the step block), decode() enqueues the four kernels and the result copy behind the logits
    // projection inside the replayed graph, process_and_sample() then only waits and reads (one copy up, one graph
    // launch, one copy down per beam step instead of three copies and five more launches)
    bool fuse_sample = true;  // env MI355_WHISPER_NO_FUSED_SAMPLE
    bool ps_armed = false, ps_fused = false;
    int ps_n = 0;
    LogitsParams ps_lp{};
    // ---- whisper_full state ----
    Decoder decoders[MAX_DECODERS];
    std::vector<Segment> result_all;
    std::vector<int> prompt_past;
    int lang_id = 0;
    int exp_n_audio_ctx = 0;
    int force_decode_steps = 0;  // bench-only extension (mw_set_force_decode_steps)
    std::string suppress_key;    // cache key of the uploaded suppression mask
    Timings tm;
    // live per-kernel timing of one projection class (bench.py roofline leg)
    int prof_tag = -1, prof_n = 0;
    int dup_tag = -1;  // measurement: launches of this class are issued twice inside the replayed step (mw_set_duplicate_class)
    std::vector<hipEvent_t> prof_ev;
    // captured decode steps, keyed by (rows, logits rows, audio ctx, index-table parity)
    // switches read ONCE at context creation from the environment (each has a parity test)
    bool use_graphs = true;   // MI355_WHISPER_NO_GRAPH: eager launches instead of the replayed step
    bool flash_attn = false;  // fused encoder attention (bf16 only; MI355_WHISPER_NO_FLASH_ATTN: QK^T, softmax, PV as three launches)
    bool gemm_glds = false;   // MI355_WHISPER_GEMM_GLDS=1: encoder GEMM operands staged by global_load_lds
    std::unordered_map<uint64_t, hipGraphExec_t> graphs;
    std::vector<uint64_t> graph_lru;  // keys, least recently used first
    // matrix-core tile kernels of the beam step (dec_tile.hip): bf16 weights, <= 5 rows, 64-wide heads, d % 256 == 0
    bool tile_step = false;
    unsigned *d_head_cnt = nullptr;  // per-head arrival counters of the fused cross-attention combine

    ~Context();
};

// throws on failure
std::unique_ptr<Context> create_context(const unsigned char *data, size_t size, int device, DType wt);
// another stream context on src's GPU that shares src's uploaded weights (throws on failure)
std::unique_ptr<Context> create_shared_context(const Context &src);

// stage entry points (used by whisper_full and by the parity API)
void pcm_to_mel(Context &c, const float *pcm, int n_samples, bool pcm_on_device);
void encode(Context &c, int seek);
// decode n tokens; logits are produced for the rows listed in logit_rows (indices into the batch)
void decode(Context &c, const int *tokens, const int *pos, const int *slot, int n, const int *logit_rows, int n_logit_rows);
void kv_clear(Context &c);
void kv_reorder(Context &c, const int *src /*[MAX_DECODERS]*/, int n_past);
int lang_auto_detect(Context &c, int offset_ms, float *probs);
int full(Context &c, whisper_full_params params, const float *samples, int n_samples, bool samples_on_device = false);
// builds + uploads the static suppression mask for these params (cached)
void upload_suppress_mask(Context &c, const whisper_full_params &params);
LogitsParams make_logits_params(const Context &c, const whisper_full_params &params, float temperature, int n_samples);

}  // namespace mw

struct whisper_context {
    std::unique_ptr<mw::Context> c;
};
