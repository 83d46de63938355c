// engine.hpp — the whisper_context of the MI355X engine: model + per-context state + driver.
#pragma once

#include "model.hpp"

#include "../../include/whisper.h"

#include <condition_variable>
#include <deque>
#include <unordered_map>
#include <exception>
#include <memory>
#include <mutex>
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

struct Context;

// One decode call of one context, as the hub sees it (lives on the calling thread's stack until it is done).
struct StepRequest {
    Context *c = nullptr;
    const int *tokens = nullptr, *pos = nullptr, *slot = nullptr;
    int n = 0;
    const int *logit_rows = nullptr;
    int n_logit_rows = 0;
    bool sample = false;       // the context's prepared rules + sampler rows ride in the step (results -> c->h_row_res)
    bool keep_logits = false;  // copy the request's logits rows into c->d_logits (stage calls, language detection, unfused sampler)
    int logits_dst_row = 0;    // ... from this row of c->d_logits on (a long call travels as several requests)
    bool solo = false;         // measurement hooks armed on the context: never batched with other requests
    std::string rules_key;     // requests share a step only when their rules + sampler parameters are identical
    std::condition_variable cv;  // the owner sleeps here until its step is done or it is its turn to lead
    bool taken = false;        // part of a step that is running on some lane
    bool done = false;
    std::exception_ptr err;
};

// One encoder pass (a 30 s window starting at mel frame `seek`) of one context, as the hub sees it.
struct EncRequest {
    Context *c = nullptr;
    int seek = 0, n_ctx = 0;
    int peers = 1;  // contexts attached to the same hub when the request was made (1: nobody to wait for)
    bool taken = false;  // part of a pass that is running on some encoder lane
    bool done = false;
    std::exception_ptr err;
};

// The batcher of decode steps for the contexts (streams, OBS sources) that share one upload of the weights on one GPU.
// LocalVocal runs one worker thread per filter, each calling whisper_full on its own context
// (/root/reference/src/whisper-utils/whisper-processing.cpp:361-431, whisper-utils.cpp:49-50); their decode calls meet
// here: the first thread to find the hub idle becomes the leader, takes the requests queued so far (up to STEP_MAX_ROWS
// rows), runs ONE step over all their rows — one pass over the weights — and hands every request its rows' results.
// Requests that arrive while a step runs form the next step (no timed waiting: one stream alone is never delayed).
struct Hub {
    // One lane = a stream with the workspace and the captured graphs of one batched step.  Steps are latency-bound (eight
    // dependent kernels per layer, each far from filling the chip), so several lanes run their steps side by side: a
    // request that arrives while a lane is busy leads the next step on an idle lane instead of waiting.  One stream alone
    // only ever uses lane 0.  Two lanes: measured on 64 turbo streams 4018 audio-s/s with one, 4465 with two, 4266 with four
    // (the box gives a process 4 hardware queues; lanes + encoder stream + the contexts' own streams share them).
    static constexpr int N_LANES = 2;
    struct Lane {
        hipStream_t st = nullptr;
        // step block: pinned host copy (and its device view) + the device copy the kernels read
        StepBlock *h_blk = nullptr, *h_blk_dev = nullptr, *d_blk = nullptr;
        float *d_x = nullptr, *d_q = nullptr, *d_part = nullptr, *d_logits = nullptr, *d_work = nullptr;
        void *d_attd = nullptr, *d_ffd = nullptr;
        unsigned *d_head_cnt = nullptr;
        int *d_phys = nullptr;  // [STEP_MAX_ROWS][TILE_TC_MAX] physical KV slot per (row, position) of the step (step_begin_batch_kernel)
        LogitsRowResult *h_res = nullptr, *h_res_dev = nullptr;  // pinned: the sampler stores its results straight into it
        unsigned char *d_suppress = nullptr;  // the suppression mask of the lane's last rules (copied from the leader when the key changes)
        std::string suppress_key;
        std::unordered_map<uint64_t, hipGraphExec_t> graphs;
        std::vector<uint64_t> graph_lru;
        bool busy = false;
        // measurement only (MI355_WHISPER_CROSS_PREFETCH): a second stream and fork / join events for a parallel graph branch
        hipStream_t st2 = nullptr;
        std::vector<hipEvent_t> ev_branch;
    };
    bool cross_prefetch = false;
    int device = 0;
    Model m;  // non-owning view of the weights (owners reset: the hub is owned BY the models that point to it)
    Lane lanes[N_LANES];
    int n_lanes = N_LANES;  // lanes in use (MI355_WHISPER_LANES, read when the hub is created)
    bool use_graphs = true;
    std::mutex mu;
    std::deque<StepRequest *> queue;
    // Gathering: a leader that finds fewer requests queued than streams are decoding right now gives the others a moment to
    // arrive (their host-side bookkeeping between two steps takes tens of microseconds) before it closes the step — 8 greedy
    // streams otherwise drifted apart and rode 2.4 rows per step on two lanes.  `hot`: when a context last asked for a step;
    // `n_running`: requests inside steps that are running now.  A stream alone (nobody else hot) never waits.
    std::unordered_map<const Context *, double> hot;
    int n_running = 0, n_gathering = 0;
    std::condition_variable cv_gather;
    int gather_us = 100;  // MI355_WHISPER_GATHER_US (0: take what is queued)
    // statistics (mw_hub_stats)
    unsigned long long n_steps = 0, n_rows = 0, n_requests = 0, n_enc_passes = 0, n_enc_windows = 0;
    // encoder passes of the same contexts, batched the same way on a stream of their own: the windows of up to
    // ENC_MAX_WINDOWS streams run as ONE pass with M = windows x n_ctx rows in every projection (more tiles per launch,
    // one pass over the encoder weights); each window's cross K / V goes to its own context
    static constexpr int ENC_MAX_WINDOWS = 16;  // measured on 64 turbo streams: 8 -> 6 225, 16 -> 6 460, 32 -> 6 260 audio-s/s
    // ONE encoder lane (stream + workspace).  Two were measured: encoder passes alone 3.88 -> 3.75 ms per window (the second pass
    // fills the LayerNorm / V-projection / tail gaps of the first), but 64 streams with their decode steps beside them
    // 5 985 -> 5 650 audio-s/s: two chip-filling GEMMs at a time leave the latency-bound step kernels even fewer CUs.
    static constexpr int N_ENC_LANES = 1;
    struct EncLane {
        hipStream_t st = nullptr;
        void *e_conv_in = nullptr, *e_x1 = nullptr, *e_h = nullptr, *e_qk = nullptr, *e_vt = nullptr, *e_att = nullptr, *e_ff = nullptr;
        float *e_x = nullptr;
        unsigned char *e_mx = nullptr;  // MX-fp8 mode: LayerNorm rows as e4m3 | their block scales | the block scales of fc1's output
        bool busy = false;
    };
    EncLane enc_lanes[N_ENC_LANES];
    std::mutex mu_enc;
    std::condition_variable cv_enc;
    std::deque<EncRequest *> queue_enc;
    bool collecting = false;  // a leader is gathering the windows of the next pass

    ~Hub();
    void submit(StepRequest &r);  // blocks until the request's step has run; rethrows its error
    void submit_encode(EncRequest &r);  // blocks until the window is encoded and its cross K / V are in the context
  private:
    void run(Lane &ln, const std::vector<StepRequest *> &batch);
    void run_encode(EncLane &ln, const std::vector<EncRequest *> &batch);
};
std::shared_ptr<Hub> create_hub(const Model &m, int device, bool use_graphs);

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
    unsigned char *d_mx = nullptr;  // MX-fp8 mode (Model::enc_mx): as Hub::EncLane::e_mx, for one window
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
    // batched path: work enqueued on `st` that a hub step must wait for (encoder, cache clears, a long prompt's prefill)
    bool st_dirty = false;
    hipEvent_t ev_ready = nullptr;
    bool use_hub = true;  // MI355_WHISPER_NO_BATCH: every context runs its own steps (the per-context graph path)

    ~Context();
};

// throws on failure
std::unique_ptr<Context> create_context(const unsigned char *data, size_t size, int device, DType wt, int fp8 = 0);
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
