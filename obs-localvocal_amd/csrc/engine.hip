// engine.hip — context creation, the stage pipelines (mel / encoder / decoder) and the whisper_full driver.
//
// The driver restates whisper.cpp v1.6.2's whisper_full_with_state — windows of 30 s,
// temperature fallback, greedy / beam bookkeeping, timestamp-driven seeking, segment assembly — as
// LocalVocal uses it (/root/reference/src/whisper-utils/whisper-processing.cpp:218-258), but every
// arithmetic step runs on the GPU: the host only keeps token sequences and launches kernels.
#include "engine.hpp"

#include <chrono>
#include <cmath>
#include <map>
#include <regex>

namespace mw {

namespace {

double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

template <typename T> void dev_alloc(T *&p, size_t n) {
    MW_HIP(hipMalloc((void **)&p, std::max<size_t>(n, 1) * sizeof(T)));
}
void dev_alloc_bytes(void *&p, size_t bytes) { MW_HIP(hipMalloc(&p, std::max<size_t>(bytes, 16))); }

const char *const k_non_speech[] = {
    "\"", "#", "(", ")", "*", "+", "/", ":", ";", "<", "=", ">", "@", "[", "\\", "]", "^", "_", "`", "{", "|", "}",
    "~", "「", "」", "『", "』", "<<", ">>", "<<<", ">>>", "--", "---", "-(", "-[", "('", "(\"", "((", "))", "(((",
    ")))", "[[", "]]", "{{", "}}", "♪♪", "♪♪♪", "♩", "♪", "♫", "♬", "♭", "♮", "♯"};

}  // namespace

// MX-fp8 encoder workspace for M rows of width d: the LayerNorm rows as e4m3 [M][d] | their block scales | the block scales
// of fc1's output (K = 4 d); fc1's e4m3 rows themselves reuse the bf16 fc1 buffer.  Each part 256-byte aligned.
struct MxWorkspace {
    unsigned char *h8, *hs, *ffs;
};
static size_t mx_align(size_t n) { return (n + 255) & ~(size_t)255; }
static size_t mx_workspace_bytes(int M, int d) {
    return mx_align((size_t)M * d) + mx_align(mx_scale_bytes_a(M, d)) + mx_align(mx_scale_bytes_a(M, 4 * d));
}
static MxWorkspace mx_workspace(unsigned char *base, int M_alloc, int d) {
    MxWorkspace w;
    w.h8 = base;
    w.hs = w.h8 + mx_align((size_t)M_alloc * d);
    w.ffs = w.hs + mx_align(mx_scale_bytes_a(M_alloc, d));
    return w;
}
// One encoder block's MX-fp8 products (Model::enc_mx) over M = windows x n_ctx rows: x -> LayerNorm (e4m3) -> Q|K|V as ONE
// product with rows of 3 d (the fused attention reads Q | K with that stride), V's columns transposed per window into the
// layout the attention reads; and LayerNorm -> fc1 + GELU (e4m3) -> fc2 + residual.  The bf16 path's attention and
// attention-output launches sit between the two (the callers).
static void mx_ln_qkv(const EncLayer &e, const float *x, void *qkv, void *vt, const MxWorkspace &w, int windows, int n_ctx, int tp, int d,
                      hipStream_t s) {
    const int M = windows * n_ctx;
    launch_layernorm_mx(x, d, e.ln1_w, e.ln1_b, nullptr, d, w.h8, d, w.hs, M, d, s);
    GemmMxArgs g;
    g.A = w.h8; g.SA = w.hs; g.lda = d; g.W = e.qk_w8; g.SW = e.qk_s8; g.ldw = d; g.C = qkv; g.ldc = 3 * d; g.bias = e.qk_b;
    g.M = M; g.N = 3 * d; g.K = d;
    launch_gemm_mx(g, MXE_BF16, s);
    launch_transpose_v((const bf16_t *)qkv + 2 * d, 3L * d, vt, n_ctx, tp, d, windows, s);
}
// ln_post (bf16 rows for the getter + e4m3 rows) and the cross K (scaled by dh^-1/4, no bias) / V (+ bias) of every decoder
// layer as two batched MX-fp8 launches, into one context's buffers
static void mx_ln_post_cross(const Model &m, const float *x, void *enc_out_bf16, void *cross, const MxWorkspace &w, int n_ctx, hipStream_t s) {
    const HParams &hp = m.hp;
    const int d = hp.n_audio_state, L = hp.n_text_layer;
    launch_layernorm_mx(x, d, m.ln_post_w, m.ln_post_b, enc_out_bf16, d, w.h8, d, w.hs, n_ctx, d, s);
    const long sb = (long)mx_scale_bytes_w(d, d);
    GemmMxArgs g;
    g.A = w.h8; g.SA = w.hs; g.lda = d; g.W = m.ckv_w8; g.SW = m.ckv_s8; g.ldw = d; g.C = cross; g.ldc = d;
    g.M = n_ctx; g.N = d; g.K = d; g.nz = L; g.sW = 2L * d * d; g.sSW = 2 * sb; g.sC = 2L * hp.n_audio_ctx * d;
    g.alpha = powf(64.0f, -0.25f);
    launch_gemm_mx(g, MXE_BF16, s);
    GemmMxArgs v = g;
    v.W = m.ckv_w8 + (size_t)d * d; v.SW = m.ckv_s8 + sb; v.C = (char *)cross + (size_t)hp.n_audio_ctx * d * 2;
    v.bias = m.ckv_b + d; v.sBias = 2L * d; v.alpha = 1.0f;
    launch_gemm_mx(v, MXE_BF16, s);
}
// x += att Wo^T + bo with the attention's rows as MX-fp8 in the workspace's row buffer (free between the Q|K|V product and
// the second LayerNorm): written there by the fused attention itself, or quantised from its bf16 rows (unfused attention)
static void mx_attn_out(const EncLayer &e, float *x, const MxWorkspace &w, int M, int d, hipStream_t s) {
    GemmMxArgs g;
    g.A = w.h8; g.SA = w.hs; g.lda = d; g.W = e.out_w8; g.SW = e.out_s8; g.ldw = d; g.C = x; g.ldc = d; g.bias = e.out_b;
    g.R = x; g.ldr = d; g.M = M; g.N = d; g.K = d;
    launch_gemm_mx(g, MXE_F32_RES, s);
}
static void mx_mlp(const EncLayer &e, float *x, void *ff, const MxWorkspace &w, int M, int d, hipStream_t s) {
    launch_layernorm_mx(x, d, e.ln2_w, e.ln2_b, nullptr, d, w.h8, d, w.hs, M, d, s);
    GemmMxArgs g;
    g.A = w.h8; g.SA = w.hs; g.lda = d; g.W = e.fc1_w8; g.SW = e.fc1_s8; g.ldw = d; g.C = ff; g.ldc = 4 * d; g.SC = w.ffs; g.bias = e.fc1_b;
    g.M = M; g.N = 4 * d; g.K = d;
    launch_gemm_mx(g, MXE_MX_GELU, s);
    GemmMxArgs f;
    f.A = (const unsigned char *)ff; f.SA = w.ffs; f.lda = 4 * d; f.W = e.fc2_w8; f.SW = e.fc2_s8; f.ldw = 4 * d; f.C = x; f.ldc = d;
    f.bias = e.fc2_b; f.R = x; f.ldr = d; f.M = M; f.N = d; f.K = 4 * d;
    launch_gemm_mx(f, MXE_F32_RES, s);
}

Context::~Context() {
    (void)hipSetDevice(device);
    void *ptrs[] = {d_pcm, d_mel_raw, d_mel, d_gmax, d_conv_in, d_x1, d_h, d_qk, d_vt, d_p, d_att, d_ff, d_enc_out, d_x,
                    d_s, d_mx, d_cross, d_kself, d_vself, d_kv_idx[0], d_kv_idx[1], d_tokens, d_xd, d_q,
                    d_attd, d_ffd, d_part, d_logits, d_work, d_suppress, d_row_state, d_row_res, d_head_cnt};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (h_ints) (void)hipHostFree(h_ints);
    if (ev_staging) (void)hipEventDestroy(ev_staging);
    if (ev_ready) (void)hipEventDestroy(ev_ready);
    if (h_row_state) (void)hipHostFree(h_row_state);
    if (h_row_res) (void)hipHostFree(h_row_res);
    for (hipEvent_t e : prof_ev) (void)hipEventDestroy(e);
    for (auto &kv : graphs) (void)hipGraphExecDestroy(kv.second);
    if (st) (void)hipStreamDestroy(st);
}

namespace {
std::unique_ptr<Context> create_context_impl(const unsigned char *data, size_t size, int device, DType wt, const Context *share,
                                             int fp8 = 0);
}

std::unique_ptr<Context> create_context(const unsigned char *data, size_t size, int device, DType wt, int fp8) {
    return create_context_impl(data, size, device, wt, nullptr, fp8);
}

// A further stream context on the same GPU: its Model is a view of `src`'s HBM-resident weights (reference-counted,
// nothing is uploaded again); workspace, KV caches, stream, graphs and whisper_full state are its own.
std::unique_ptr<Context> create_shared_context(const Context &src) {
    return create_context_impl(nullptr, 0, src.device, src.model.wt, &src);
}

namespace {
std::unique_ptr<Context> create_context_impl(const unsigned char *data, size_t size, int device, DType wt, const Context *share,
                                             int fp8) {
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        throw std::runtime_error("no HIP device available: this engine has no CPU path");
    if (device < 0 || device >= n_dev) throw std::runtime_error("gpu_device out of range");
    MW_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    MW_HIP(hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
        throw std::runtime_error(std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
    std::unique_ptr<Context> c(new Context());
    c->device = device;
    MW_HIP(hipStreamCreateWithFlags(&c->st, hipStreamNonBlocking));
    (void)mel_noalloc_init(device);  // device address of the allocation-free mel entry's scratch array, once per device
    if (share) c->model = share->model;
    else load_model(c->model, data, size, wt, c->st, fp8);
    const HParams &hp = c->model.hp;
    const size_t d = hp.n_audio_state, T = hp.n_audio_ctx, ws = dtype_size(wt), H = hp.n_audio_head;
    const size_t L = hp.n_text_layer, TC = hp.n_text_ctx, nv = hp.n_vocab;
    c->tp = (int)((T + 63) / 64 * 64);
    // encoder workspace (sized for the full audio context)
    dev_alloc_bytes(c->d_conv_in, (2 * T + 2) * hp.n_mels * ws);
    dev_alloc_bytes(c->d_x1, (2 * T + 1) * d * ws);
    dev_alloc(c->d_x, T * d);
    dev_alloc_bytes(c->d_h, T * d * ws);
    dev_alloc_bytes(c->d_qk, T * (c->model.enc_mx ? 3 : 2) * d * ws);  // (MX-fp8 mode: Q | K | V rows)
    dev_alloc_bytes(c->d_vt, d * c->tp * ws);
    dev_alloc(c->d_s, H * T * c->tp);
    dev_alloc_bytes(c->d_p, H * T * c->tp * ws);
    dev_alloc_bytes(c->d_att, T * d * ws);
    dev_alloc_bytes(c->d_ff, T * 4 * d * ws);
    dev_alloc_bytes(c->d_enc_out, T * d * ws);
    if (c->model.enc_mx) dev_alloc(c->d_mx, mx_workspace_bytes((int)T, (int)d));
    dev_alloc_bytes(c->d_cross, L * 2 * T * d * ws);
    MW_HIP(hipMemsetAsync(c->d_x1, 0, (2 * T + 1) * d * ws, c->st));  // row 0 is the conv2 left zero pad
    MW_HIP(hipMemsetAsync(c->d_vt, 0, d * c->tp * ws, c->st));        // pad columns stay zero
    MW_HIP(hipMemsetAsync(c->d_p, 0, H * T * c->tp * ws, c->st));
    // decoder workspace
    dev_alloc_bytes(c->d_kself, L * KV_SLOTS * TC * d * ws);
    dev_alloc_bytes(c->d_vself, L * KV_SLOTS * TC * d * ws);
    for (int i = 0; i < 2; ++i) {
        dev_alloc(c->d_kv_idx[i], KV_SLOTS * TC);
        MW_HIP(hipMemsetAsync(c->d_kv_idx[i], 0, KV_SLOTS * TC * sizeof(int), c->st));
    }
    dev_alloc(c->d_tokens, 4 * TC);  // one block, laid out like the pinned staging block: tokens | pos | slot | rows + update args
    c->d_pos = c->d_tokens + TC;
    c->d_slot = c->d_tokens + 2 * TC;
    c->d_rows = c->d_tokens + 3 * TC;
    dev_alloc(c->d_xd, TC * d);
    dev_alloc(c->d_q, TC * d);
    dev_alloc_bytes(c->d_attd, TC * d * 4);
    dev_alloc_bytes(c->d_ffd, TC * 4 * d * 4);
    dev_alloc(c->d_part, TC * H * CROSS_SPLITS_MAX * CROSS_PART_STRIDE);
    dev_alloc(c->d_logits, (size_t)LOGIT_ROWS * nv);
    dev_alloc(c->d_work, process_logits_work_floats(LOGIT_ROWS, nv));
    dev_alloc(c->d_suppress, nv);
    dev_alloc(c->d_row_state, LOGIT_ROWS);
    dev_alloc(c->d_row_res, LOGIT_ROWS);
    MW_CHECK(TC >= LOGIT_ROWS + KV_SLOTS + 2, "n_text_ctx too small for the step block");
    MW_HIP(hipHostMalloc((void **)&c->h_ints, 4 * TC * sizeof(int)));
    if (!getenv("MI355_WHISPER_NO_STEP_BEGIN")) MW_HIP(hipHostGetDevicePointer((void **)&c->h_ints_dev, c->h_ints, 0));
    MW_HIP(hipEventCreateWithFlags(&c->ev_staging, hipEventDisableTiming));
    MW_HIP(hipHostMalloc((void **)&c->h_row_state, LOGIT_ROWS * sizeof(LogitsRowState)));
    MW_HIP(hipHostMalloc((void **)&c->h_row_res, LOGIT_ROWS * sizeof(LogitsRowResult)));
    MW_HIP(hipHostGetDevicePointer((void **)&c->d_row_res_host, c->h_row_res, 0));
    c->decoders[0].rng = std::mt19937(0);
    c->n_ctx_cur = (int)T;
    // the environment is read here, once per context, never on a launch path
    if (getenv("MI355_WHISPER_NO_GRAPH")) c->use_graphs = false;  // debugging aid: eager launches
    if (getenv("MI355_WHISPER_NO_FUSED_SAMPLE")) c->fuse_sample = false;  // rules + samplers as launches of their own
    c->flash_attn = wt == DType::BF16 && getenv("MI355_WHISPER_NO_FLASH_ATTN") == nullptr;
    {
        const char *e = getenv("MI355_WHISPER_GEMM_GLDS");
        c->gemm_glds = wt == DType::BF16 && e && e[0] == '1';
    }
    // the beam step on matrix-core tile kernels (dec_tile.hip); other shapes and fp32 use the VALU kernels of dec.hip
    if (wt == DType::BF16 && TC <= (size_t)TILE_TC_MAX && hp.n_text_state == hp.n_text_head * 64 && hp.n_text_state % 256 == 0 &&
        hp.n_text_state >= 512 && hp.n_text_state <= 1280) {
        dev_alloc(c->d_head_cnt, 64);
        MW_HIP(hipMemsetAsync(c->d_head_cnt, 0, 64 * sizeof(unsigned), c->st));
        c->tile_step = true;
        // decode steps of the contexts that share these weights are batched by one hub per upload (engine.hpp)
        c->use_hub = getenv("MI355_WHISPER_NO_BATCH") == nullptr;
        if (c->use_hub && !c->model.hub) c->model.hub = create_hub(c->model, device, c->use_graphs);
    }
    MW_HIP(hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming));
    c->st_dirty = true;
    MW_HIP(hipStreamSynchronize(c->st));
    log_msg(LOG_INFO, "mi355_whisper: %s (%zu params, %s weights, %.1f MB arena) on device %d (%s)\n",
            share ? "stream context on shared weights" : "model loaded", c->model.n_params, wt == DType::F32 ? "fp32" : "bf16",
            c->model.arena_used / 1048576.0, device, prop.gcnArchName);
    return c;
}
}  // namespace


std::mutex g_capture_mu;  // see the capture sites

// ------------------------------------------------------------------------------------------
// the hub: one batched decode step over the rows of several contexts (engine.hpp)
// ------------------------------------------------------------------------------------------
Hub::~Hub() {
    (void)hipSetDevice(device);
    for (Lane &ln : lanes)
        if (ln.st) (void)hipStreamSynchronize(ln.st);
    for (EncLane &el : enc_lanes)
        if (el.st) (void)hipStreamSynchronize(el.st);
    for (Lane &ln : lanes) {
        void *ptrs[] = {ln.d_blk, ln.d_x, ln.d_q, ln.d_part, ln.d_logits, ln.d_work, ln.d_attd, ln.d_ffd, ln.d_head_cnt, ln.d_suppress, ln.d_phys};
        for (void *p : ptrs)
            if (p) (void)hipFree(p);
        if (ln.h_blk) (void)hipHostFree(ln.h_blk);
        if (ln.h_res) (void)hipHostFree(ln.h_res);
        for (auto &kv : ln.graphs) (void)hipGraphExecDestroy(kv.second);
        if (ln.st) (void)hipStreamDestroy(ln.st);
        if (ln.st2) (void)hipStreamDestroy(ln.st2);
        for (hipEvent_t e : ln.ev_branch) (void)hipEventDestroy(e);
    }
    for (EncLane &el : enc_lanes) {
        void *ptrs[] = {el.e_conv_in, el.e_x1, el.e_h, el.e_qk, el.e_vt, el.e_att, el.e_ff, el.e_x, el.e_mx};
        for (void *p : ptrs)
            if (p) (void)hipFree(p);
        if (el.st) (void)hipStreamDestroy(el.st);
    }
}

std::shared_ptr<Hub> create_hub(const Model &m, int device, bool use_graphs) {
    std::shared_ptr<Hub> h(new Hub());
    h->device = device;
    h->m = m;
    h->m.arena_owner.reset();  // the hub is owned by the models that point to it, never the other way round
    h->m.mel_tab_owner.reset();
    h->m.hub.reset();
    h->use_graphs = use_graphs;
    h->cross_prefetch = getenv("MI355_WHISPER_CROSS_PREFETCH") != nullptr;  // measurement only
    const HParams &hp = m.hp;
    const size_t d = hp.n_text_state, H = hp.n_text_head, nv = hp.n_vocab, R = STEP_MAX_ROWS;
    // Plain streams on purpose.  High-priority lane streams (hipStreamCreateWithPriority) changed nothing with one or two
    // lanes and collapsed throughput 30x with three or more (64 turbo streams: 4266 audio-s/s -> 151): HIP keeps a separate,
    // smaller pool of hardware queues per priority class.
    if (const char *e = getenv("MI355_WHISPER_LANES")) h->n_lanes = std::max(1, std::min(Hub::N_LANES, atoi(e)));
    if (const char *e = getenv("MI355_WHISPER_GATHER_US")) h->gather_us = std::max(0, std::min(2000, atoi(e)));
    for (int li = 0; li < h->n_lanes; ++li) {
        Hub::Lane &ln = h->lanes[li];
        MW_HIP(hipStreamCreateWithFlags(&ln.st, hipStreamNonBlocking));
        if (h->cross_prefetch) {
            MW_HIP(hipStreamCreateWithFlags(&ln.st2, hipStreamNonBlocking));
            ln.ev_branch.resize(2 * (size_t)m.hp.n_text_layer);
            for (hipEvent_t &e : ln.ev_branch) MW_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        MW_HIP(hipHostMalloc((void **)&ln.h_blk, sizeof(StepBlock)));
        memset(ln.h_blk, 0, sizeof(StepBlock));
        MW_HIP(hipHostGetDevicePointer((void **)&ln.h_blk_dev, ln.h_blk, 0));
        MW_HIP(hipHostMalloc((void **)&ln.h_res, R * sizeof(LogitsRowResult)));
        MW_HIP(hipHostGetDevicePointer((void **)&ln.h_res_dev, ln.h_res, 0));
        dev_alloc(ln.d_blk, 1);
        dev_alloc(ln.d_x, R * d);
        dev_alloc(ln.d_q, R * d);
        dev_alloc_bytes(ln.d_attd, R * d * 2);
        dev_alloc_bytes(ln.d_ffd, R * 4 * d * 2);
        dev_alloc(ln.d_part, R * H * CROSS_SPLITS_MAX * CROSS_PART_STRIDE);
        dev_alloc(ln.d_logits, R * nv);
        dev_alloc(ln.d_work, process_logits_work_floats((int)R, (int)nv));
        dev_alloc(ln.d_suppress, nv);
        dev_alloc(ln.d_phys, (size_t)STEP_MAX_ROWS * TILE_TC_MAX);
        dev_alloc(ln.d_head_cnt, (size_t)STEP_MAX_GROUPS * 64);
        MW_HIP(hipMemsetAsync(ln.d_head_cnt, 0, (size_t)STEP_MAX_GROUPS * 64 * sizeof(unsigned), ln.st));
    }
    for (Hub::EncLane &el : h->enc_lanes) {  // encoder workspace for ENC_MAX_WINDOWS windows per lane (slabs back to back)
        const size_t W = Hub::ENC_MAX_WINDOWS, da = hp.n_audio_state, T = hp.n_audio_ctx, tp = (T + 63) / 64 * 64;
        MW_HIP(hipStreamCreateWithFlags(&el.st, hipStreamNonBlocking));
        dev_alloc_bytes(el.e_conv_in, W * (2 * T + 2) * hp.n_mels * 2);
        dev_alloc_bytes(el.e_x1, W * (2 * T + 1) * da * 2);
        dev_alloc(el.e_x, W * T * da);
        dev_alloc_bytes(el.e_h, W * T * da * 2);
        dev_alloc_bytes(el.e_qk, W * T * 3 * da * 2);  // Q | K | V rows (batched passes; a single bf16 window uses two thirds)
        dev_alloc_bytes(el.e_vt, W * da * tp * 2);
        dev_alloc_bytes(el.e_att, W * T * da * 2);
        dev_alloc_bytes(el.e_ff, W * T * 4 * da * 2);
        if (m.enc_mx) dev_alloc(el.e_mx, mx_workspace_bytes((int)(W * T), (int)da));
        MW_HIP(hipMemsetAsync(el.e_x1, 0, W * (2 * T + 1) * da * 2, el.st));  // row 0 of every window is conv2's left zero pad
        MW_HIP(hipMemsetAsync(el.e_vt, 0, W * da * tp * 2, el.st));           // pad columns stay zero
        MW_HIP(hipStreamSynchronize(el.st));
    }
    for (int li = 0; li < h->n_lanes; ++li) MW_HIP(hipStreamSynchronize(h->lanes[li].st));
    return h;
}

void Hub::submit_encode(EncRequest &r) {
    std::unique_lock<std::mutex> lk(mu_enc);
    queue_enc.push_back(&r);
    if (collecting) cv_enc.notify_all();  // a leader is gathering windows (below)
    auto idle_lane = [&]() -> EncLane * {
        for (EncLane &el : enc_lanes)
            if (!el.busy) return &el;
        return nullptr;
    };
    while (!r.done) {
        EncLane *lane = (r.taken || collecting) ? nullptr : idle_lane();
        if (!lane) {
            cv_enc.wait(lk);
            continue;
        }
        // a lane is idle and nobody is gathering: this thread leads the next pass over the oldest request and every later one
        // with the same audio context
        collecting = true;
        if (r.peers > 1) {
            // Other streams share these weights: their windows tend to arrive within microseconds of each other (workers woken
            // by the same audio clock, threads released by the same finished pass), and a pass costs milliseconds — so the
            // leader gives them 300 us to join before it starts (measured: 4 streams encoding in step formed passes of 2
            // without this).  A context alone on its weights never waits.
            const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(300);
            const size_t want = (size_t)std::min(ENC_MAX_WINDOWS, r.peers);
            while (queue_enc.size() < want && cv_enc.wait_until(lk, deadline) != std::cv_status::timeout) {
            }
        }
        std::vector<EncRequest *> batch;
        EncRequest *first = queue_enc.front();
        for (auto it = queue_enc.begin(); it != queue_enc.end();) {
            if ((int)batch.size() < ENC_MAX_WINDOWS && (*it)->n_ctx == first->n_ctx) {
                (*it)->taken = true;
                batch.push_back(*it);
                it = queue_enc.erase(it);
            } else {
                ++it;
            }
        }
        lane->busy = true;
        collecting = false;
        if (!queue_enc.empty()) cv_enc.notify_all();  // more windows queued: somebody leads them on the other lane
        lk.unlock();
        std::exception_ptr err;
        try {
            run_encode(*lane, batch);
        } catch (...) {
            err = std::current_exception();
            (void)hipStreamSynchronize(lane->st);
        }
        lk.lock();
        for (EncRequest *q : batch) {
            q->err = err;
            q->done = true;
        }
        n_enc_passes += 1;
        n_enc_windows += batch.size();
        lane->busy = false;
        cv_enc.notify_all();
    }
    lk.unlock();
    if (r.err) std::rethrow_exception(r.err);
}

// Encoder + cross-K/V (A.3) of W windows in one pass: the conv stem and the attention are batched over the windows, every
// projection and LayerNorm of the transformer blocks runs over all W x n_ctx rows at once.
void Hub::run_encode(EncLane &ln, const std::vector<EncRequest *> &batch) {
    MW_HIP(hipSetDevice(device));
    void *const e_conv_in = ln.e_conv_in, *const e_x1 = ln.e_x1, *const e_h = ln.e_h, *const e_qk = ln.e_qk, *const e_vt = ln.e_vt,
                *const e_att = ln.e_att, *const e_ff = ln.e_ff;
    float *const e_x = ln.e_x;
    const HParams &hp = m.hp;
    const DType wt = DType::BF16;
    const size_t ws = 2;
    const int W = (int)batch.size(), n_ctx = batch[0]->n_ctx;
    const int d = hp.n_audio_state, H = hp.n_audio_head, n_mel = hp.n_mels, T2 = 2 * n_ctx;
    const int tp = (n_ctx + 63) / 64 * 64;
    hipStream_t s = ln.st;
    const long cin_w = (long)(T2 + 2) * n_mel, x1_w = (long)(T2 + 1) * d;  // per-window slab sizes (elements)
    for (int w = 0; w < W; ++w) {
        Context &c = *batch[w]->c;
        MW_CHECK(c.model.arena == m.arena && c.d_mel != nullptr, "hub: encode request of a foreign context / no mel");
        if (c.st_dirty) {  // the window's mel is still being produced on the context's own stream
            MW_HIP(hipStreamWaitEvent(s, c.ev_ready, 0));
            c.st_dirty = false;
        }
        launch_mel_window(c.d_mel, 0, c.n_len, batch[w]->seek, T2, n_mel, 1, (char *)e_conv_in + (size_t)w * cin_w * ws, cin_w, wt, s);
    }
    {   // conv1 over the overlapping row view of every window: output row t <- padded rows t, t+1, t+2
        GemmArgs g;
        g.A = e_conv_in; g.lda = n_mel; g.W = m.conv1_w; g.ldw = 3 * n_mel;
        g.C = (char *)e_x1 + (size_t)d * ws; g.ldc = d;  // row 0 of every window stays zero (conv2's left pad)
        g.bias = m.conv1_b; g.M = T2; g.N = d; g.K = 3 * n_mel; g.act = ACT_GELU;
        g.nz0 = W; g.sA0 = cin_w; g.sC0 = x1_w;
        launch_gemm(g, wt, wt, s);
    }
    {   // conv2, stride 2: output row t <- padded x1 rows 2t, 2t+1, 2t+2; + positional embedding
        GemmArgs g;
        g.A = e_x1; g.lda = 2 * d; g.W = m.conv2_w; g.ldw = 3 * d;
        g.C = e_x; g.ldc = d; g.bias = m.conv2_b; g.M = n_ctx; g.N = d; g.K = 3 * d; g.act = ACT_GELU;
        g.R = m.enc_pe; g.ldr = d;
        g.nz0 = W; g.sA0 = x1_w; g.sC0 = (long)n_ctx * d; g.sR0 = 0;
        launch_gemm(g, wt, DType::F32, s);
    }
    const int M = W * n_ctx;
    const MxWorkspace mxw = mx_workspace(ln.e_mx, ENC_MAX_WINDOWS * hp.n_audio_ctx, d);
    // bf16: one Q | K | V product once it runs on the 256 x 256 kernel (>= 200 tiles of 3 d columns; launch_gemm's rule)
    const bool qkv_fused = !m.enc_mx && d % 256 == 0 && (long)(3 * d / 256) * ceil_div(M, 256) >= 200;
    for (int l = 0; l < hp.n_audio_layer; ++l) {
        const EncLayer &e = m.enc[l];
        if (m.enc_mx) {
            mx_ln_qkv(e, e_x, e_qk, e_vt, mxw, W, n_ctx, tp, d, s);
        } else if (qkv_fused) {
            // Q | K | V of all windows as one product with rows of 3 d (the model keeps the three matrices back to back), V's
            // columns transposed per window afterwards: at 16 windows 205 + 24 us against 140 (Q | K) + 130 (V on the
            // transposed-store kernel); the same chain of MFMAs per element either way
            launch_layernorm(e_x, d, e.ln1_w, e.ln1_b, e_h, d, wt, M, d, s);
            GemmArgs g;
            g.A = e_h; g.lda = d; g.W = e.qk_w; g.ldw = d; g.C = e_qk; g.ldc = 3 * d; g.bias = e.qk_b;
            g.M = M; g.N = 3 * d; g.K = d;
            launch_gemm(g, wt, wt, s);
            launch_transpose_v((const bf16_t *)e_qk + 2 * d, 3L * d, e_vt, n_ctx, tp, d, W, s);
        } else {
            launch_layernorm(e_x, d, e.ln1_w, e.ln1_b, e_h, d, wt, M, d, s);
            GemmArgs g;  // Q | K of all windows
            g.A = e_h; g.lda = d; g.W = e.qk_w; g.ldw = d; g.C = e_qk; g.ldc = 2 * d; g.bias = e.qk_b;
            g.M = M; g.N = 2 * d; g.K = d;
            launch_gemm(g, wt, wt, s);
        }
        if (!m.enc_mx && !qkv_fused) {
            GemmArgs g;  // V, stored transposed [d][tp] per window
            g.A = e_h; g.lda = d; g.W = e.v_w; g.ldw = d; g.C = e_vt; g.ldc = tp; g.bias = e.v_b;
            g.M = n_ctx; g.N = d; g.K = d; g.trans_c = 1;
            g.nz0 = W; g.sA0 = (long)n_ctx * d; g.sC0 = (long)d * tp;
            launch_gemm(g, wt, wt, s);
        }
        if (m.enc_mx) {
            launch_enc_flash_attn(e_qk, e_vt, nullptr, n_ctx, tp, d, H, s, W, 3 * d, mxw.h8, mxw.hs);
            mx_attn_out(e, e_x, mxw, M, d, s);
            mx_mlp(e, e_x, e_ff, mxw, M, d, s);
            continue;
        }
        launch_enc_flash_attn(e_qk, e_vt, e_att, n_ctx, tp, d, H, s, W, qkv_fused ? 3 * d : 2 * d);
        {
            GemmArgs g;  // x += att Wo^T + bo
            g.A = e_att; g.lda = d; g.W = e.out_w; g.ldw = d; g.C = e_x; g.ldc = d; g.bias = e.out_b;
            g.R = e_x; g.ldr = d; g.M = M; g.N = d; g.K = d;
            launch_gemm(g, wt, DType::F32, s);
        }
        launch_layernorm(e_x, d, e.ln2_w, e.ln2_b, e_h, d, wt, M, d, s);
        {
            GemmArgs g;
            g.A = e_h; g.lda = d; g.W = e.fc1_w; g.ldw = d; g.C = e_ff; g.ldc = 4 * d; g.bias = e.fc1_b;
            g.M = M; g.N = 4 * d; g.K = d; g.act = ACT_GELU;
            launch_gemm(g, wt, wt, s);
        }
        {
            GemmArgs g;
            g.A = e_ff; g.lda = 4 * d; g.W = e.fc2_w; g.ldw = 4 * d; g.C = e_x; g.ldc = d; g.bias = e.fc2_b;
            g.R = e_x; g.ldr = d; g.M = M; g.N = d; g.K = 4 * d;
            launch_gemm(g, wt, DType::F32, s);
        }
    }
    const int L = hp.n_text_layer;
    const long per_layer = 2L * hp.n_audio_ctx * d;
    for (int w = 0; w < W; ++w) {  // ln_post and the cross K / V of every decoder layer, into the window's own context
        Context &c = *batch[w]->c;
        if (m.enc_mx) {
            mx_ln_post_cross(m, e_x + (size_t)w * n_ctx * d, c.d_enc_out, c.d_cross, mxw, n_ctx, s);
            c.n_ctx_cur = n_ctx;
            continue;
        }
        launch_layernorm(e_x + (size_t)w * n_ctx * d, d, m.ln_post_w, m.ln_post_b, c.d_enc_out, d, wt, n_ctx, d, s);
        GemmArgs g;
        g.A = c.d_enc_out; g.lda = d; g.W = m.ckv_w; g.ldw = d; g.C = c.d_cross; g.ldc = d;
        g.M = n_ctx; g.N = d; g.K = d; g.nz1 = L; g.sW1 = 2L * d * d; g.sC1 = per_layer;
        g.alpha = powf(64.0f, -0.25f);
        launch_gemm(g, wt, wt, s);
        GemmArgs v = g;
        v.W = (char *)m.ckv_w + (size_t)d * d * ws;
        v.C = (char *)c.d_cross + (size_t)hp.n_audio_ctx * d * ws;
        v.bias = m.ckv_b + d; v.sB1 = 2L * d; v.alpha = 1.0f;
        launch_gemm(v, wt, wt, s);
        c.n_ctx_cur = n_ctx;
    }
    MW_HIP(hipStreamSynchronize(s));
}


void Hub::submit(StepRequest &r) {
    std::unique_lock<std::mutex> lk(mu);
    queue.push_back(&r);
    hot[r.c] = now_ms();
    if (n_gathering > 0) cv_gather.notify_all();
    auto idle_lane = [&]() -> Lane * {
        for (int li = 0; li < n_lanes; ++li)
            if (!lanes[li].busy) return &lanes[li];
        return nullptr;
    };
    while (!r.done) {
        Lane *lane = (r.taken || n_gathering > 0) ? nullptr : idle_lane();  // (one leader gathers at a time: it will take or wake us)
        if (!lane) {
            // every lane is busy, or this request is already part of a running step: sleep on the REQUEST's condition variable.
            // Wake-ups are targeted (the requests of a finished step; the owner of the oldest queued request when a lane is
            // free): with one shared condition variable every step woke all ~60 waiting threads of a 64-stream load, and the
            // hand-over of the mutex among them delayed the next leader by more than the step itself took.
            r.cv.wait(lk);
            continue;
        }
        // a lane is idle: this thread leads the next step on it.  It takes the oldest request and every later one that can
        // share its step (same rules, rows / groups / logits rows / sampler rows within the step's limits), in order.
        lane->busy = true;  // (this thread leads on it from here on)
        if (gather_us > 0 && !r.solo) {
            // streams that asked for a step within the last 2 ms and are neither queued nor inside a running step are about to
            // ask again: wait for them, at most gather_us
            auto expected = [&]() {
                const double t = now_ms();
                int n = 0;
                for (auto it = hot.begin(); it != hot.end();) {
                    if (t - it->second > 2.0) it = hot.erase(it);
                    else ++n, ++it;
                }
                return n - n_running;
            };
            const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(gather_us);
            n_gathering += 1;
            while ((int)queue.size() < std::min(expected(), STEP_MAX_GROUPS) && cv_gather.wait_until(lk, deadline) != std::cv_status::timeout) {
            }
            n_gathering -= 1;
        }
        std::vector<StepRequest *> batch;
        int rows = 0, lrows = 0, srows = 0;
        StepRequest *first = queue.front();
        for (auto it = queue.begin(); it != queue.end();) {
            StepRequest *q = *it;
            const int sr = q->sample ? q->c->ps_n : 0;
            const bool fits = batch.empty() || (!first->solo && !q->solo && q->rules_key == first->rules_key &&
                                                (int)batch.size() < STEP_MAX_GROUPS && rows + q->n <= STEP_MAX_ROWS &&
                                                lrows + q->n_logit_rows <= STEP_MAX_ROWS && srows + sr <= STEP_MAX_ROWS);
            if (fits) {
                batch.push_back(q);
                q->taken = true;
                rows += q->n;
                lrows += q->n_logit_rows;
                srows += sr;
                it = queue.erase(it);
            } else {
                ++it;
            }
        }
        n_running += (int)batch.size();
        if (!queue.empty() && idle_lane()) queue.front()->cv.notify_one();  // more work and another lane free: next leader
        lk.unlock();
        std::exception_ptr err;
        try {
            run(*lane, batch);
        } catch (...) {
            err = std::current_exception();
            (void)hipStreamSynchronize(lane->st);  // nothing of the failed step may still be running when its requests return
        }
        lk.lock();
        for (StepRequest *q : batch) {
            q->err = err;
            q->done = true;
            if (q != &r) q->cv.notify_one();
        }
        n_steps += 1;
        n_rows += (unsigned long long)rows;
        n_requests += batch.size();
        n_running -= (int)batch.size();
        lane->busy = false;
        // the lane is free again: unless this thread leads the next step itself (its own request is still queued), hand the
        // lead to the owner of the oldest queued request
        if (!queue.empty() && (r.done || r.taken)) queue.front()->cv.notify_one();
    }
    lk.unlock();
    if (r.err) std::rethrow_exception(r.err);
}

void Hub::run(Lane &ln, const std::vector<StepRequest *> &batch) {
    MW_HIP(hipSetDevice(device));
    hipStream_t st = ln.st;
    StepBlock *const h_blk = ln.h_blk, *const h_blk_dev = ln.h_blk_dev, *const d_blk = ln.d_blk;
    float *const d_x = ln.d_x, *const d_q = ln.d_q, *const d_part = ln.d_part, *const d_logits = ln.d_logits, *const d_work = ln.d_work;
    void *const d_attd = ln.d_attd, *const d_ffd = ln.d_ffd;
    unsigned *const d_head_cnt = ln.d_head_cnt;
    LogitsRowResult *const h_res = ln.h_res, *const h_res_dev = ln.h_res_dev;
    unsigned char *const d_suppress = ln.d_suppress;
    std::string &suppress_key = ln.suppress_key;
    auto &graphs = ln.graphs;
    auto &graph_lru = ln.graph_lru;
    const HParams &hp = m.hp;
    const int d = hp.n_text_state, H = hp.n_text_head, TC = hp.n_text_ctx, T = hp.n_audio_ctx, nv = hp.n_vocab;
    StepBlock &B = *h_blk;
    int R = 0, G = 0, LR = 0, SR = 0, n_ctx_max = 1, g_rows_max = 1;  // g_rows_max: most rows of one request (cross-attention variant)
    Context *lead = batch[0]->c;
    const bool sample = batch[0]->sample;
    for (StepRequest *q : batch) {
        Context &c = *q->c;
        MW_CHECK(c.model.wt == DType::BF16 && c.model.arena == m.arena, "hub: context does not share the hub's weights");
        if (c.st_dirty) {  // the step reads what the context's own stream is still producing (encoder output, cleared tables)
            MW_HIP(hipStreamWaitEvent(st, c.ev_ready, 0));
            c.st_dirty = false;
        }
        StepGroup &g = B.groups[G];
        g.cross = (const bf16_t *)c.d_cross;
        g.idx_old = c.d_kv_idx[c.kv_cur ^ 1];
        g.idx_new = c.d_kv_idx[c.kv_cur];
        g.row0 = R;
        g.nrows = q->n;
        g.n_ctx = c.n_ctx_cur;
        g.reorder_flag = c.reorder_pending ? 1 : 0;
        g.reorder_n_past = c.reorder_n_past;
        for (int j = 0; j < KV_SLOTS; ++j) g.src[j] = c.reorder_pending ? c.reorder_src[j] : -1;
        g.pad = 0;
        c.reorder_pending = false;
        n_ctx_max = std::max(n_ctx_max, c.n_ctx_cur);
        g_rows_max = std::max(g_rows_max, q->n);
        for (int i = 0; i < q->n; ++i) {
            StepRow &r = B.rows[R + i];
            r.kself = (bf16_t *)c.d_kself;
            r.vself = (bf16_t *)c.d_vself;
            r.kv_idx = c.d_kv_idx[c.kv_cur];
            r.slot = q->slot[i];
            r.pos = q->pos[i];
            r.token = q->tokens[i];
            r.group = G;
        }
        for (int i = 0; i < q->n_logit_rows; ++i) B.logit_rows[LR + i] = R + q->logit_rows[i];
        if (q->sample) {
            for (int i = 0; i < c.ps_n; ++i) {
                B.rs[SR + i] = c.h_row_state[i];
                B.rs[SR + i].logits_row = LR + c.h_row_state[i].logits_row;
            }
            SR += c.ps_n;
        }
        LR += q->n_logit_rows;
        R += q->n;
        G += 1;
    }
    B.n_rows = R;
    B.n_groups = G;
    const int splits = std::min(CROSS_SPLITS_MAX, ceil_div(n_ctx_max, 128));
    MW_CHECK(ceil_div(n_ctx_max, splits) <= 128, "hub: audio context too long for the cross-attention tiles");
    LogitsParams lp{};
    if (sample) {
        lp = lead->ps_lp;
        if (suppress_key != lead->suppress_key || suppress_key.empty()) {
            MW_HIP(hipMemcpyAsync(d_suppress, lead->d_suppress, (size_t)nv, hipMemcpyDeviceToDevice, st));
            suppress_key = lead->suppress_key;
        }
    }
    // measurement hooks of a solo request's context (mw_profile_begin / mw_set_duplicate_class)
    Context *pc = batch[0]->solo ? lead : nullptr;
    const int prof_tag = pc ? pc->prof_tag : -1, dup_tag = pc ? pc->dup_tag : -1;
    const size_t kv_layer = (size_t)KV_SLOTS * TC * d;
    const float qk_scale = powf(64.0f, -0.25f);
    auto enqueue = [&]() {
        launch_step_begin_batch(h_blk_dev, d_blk, R, G, TC, m.tok_emb, m.dec_pe, d_x, d, ln.d_phys, st);
        auto bracket = [&](int tag, auto &&launch) {
            const bool prof = pc && tag == prof_tag && pc->prof_n + 2 <= (int)pc->prof_ev.size();
            if (prof) MW_HIP(hipEventRecord(pc->prof_ev[pc->prof_n++], st));
            launch();
            if (prof) MW_HIP(hipEventRecord(pc->prof_ev[pc->prof_n++], st));
        };
        auto proj = [&](GemvArgs g, long kv_off) {
            bracket(g.tag, [&]() { launch_proj_batch(g, d_blk, kv_off, st); });
            if (g.tag == dup_tag && prof_tag < 0) {  // measurement: the class's launches are issued twice
                if (g.R && g.out == g.R) g.out = d_part;
                launch_proj_batch(g, d_blk, kv_off, st);
            }
        };
        for (int l = 0; l < hp.n_text_layer; ++l) {
            const DecLayer &e = m.dec[l];
            const long kv_off = (long)(l * kv_layer);
            if (cross_prefetch) {  // measurement: this layer's cross K / V touched on a parallel branch while the chain runs up to cross-q
                const long ko = (long)l * 2 * T * d;
                MW_HIP(hipEventRecord(ln.ev_branch[2 * l], st));
                MW_HIP(hipStreamWaitEvent(ln.st2, ln.ev_branch[2 * l], 0));
                launch_cross_touch_batch(d_blk, ko, ko + (long)T * d, G, H, d, splits, (int *)nullptr, ln.st2);
                MW_HIP(hipEventRecord(ln.ev_branch[2 * l + 1], ln.st2));
            }
            {
                GemvArgs g;  // LN + Q|K|V; K and V go straight into each row's cache
                g.x = d_x; g.ldx = d; g.n = R; g.W = e.qkv_w; if (m.dec_fp8) { g.W8 = e.qkv_w8; g.w_scale = e.qkv_s; } g.bias = e.qkv_b; g.N = 3 * d; g.K = d;
                g.ln_w = e.ln1_w; g.ln_b = e.ln1_b; g.out = d_q; g.ldo = d; g.qkv = 1; g.qk_scale = qk_scale;
                g.kv_t = DType::BF16; g.n_text_ctx = TC; g.tag = GEMV_QKV;
                proj(g, kv_off);
            }
            bracket(GEMV_N_TAGS, [&]() { launch_self_attn_batch(d_q, d_blk, ln.d_phys, kv_off, d_attd, R, H, d, TC, st); });
            if (dup_tag == GEMV_N_TAGS && prof_tag < 0) launch_self_attn_batch(d_q, d_blk, ln.d_phys, kv_off, d_attd, R, H, d, TC, st);
            {
                GemvArgs g;
                g.x = (const float *)d_attd; g.x_bf16 = 1; g.ldx = d; g.n = R; g.W = e.out_w; if (m.dec_fp8) { g.W8 = e.out_w8; g.w_scale = e.out_s; } g.bias = e.out_b; g.N = d; g.K = d;
                g.out = d_x; g.ldo = d; g.R = d_x; g.tag = GEMV_ATTN_OUT;
                proj(g, 0);
            }
            {
                GemvArgs g;  // LN + cross query (scaled)
                g.x = d_x; g.ldx = d; g.n = R; g.W = e.cq_w; if (m.dec_fp8) { g.W8 = e.cq_w8; g.w_scale = e.cq_s; } g.bias = e.cq_b; g.N = d; g.K = d;
                g.ln_w = e.lnc_w; g.ln_b = e.lnc_b; g.out = d_q; g.ldo = d; g.out_scale = qk_scale; g.tag = GEMV_CROSS_Q;
                proj(g, 0);
            }
            const long k_off = (long)l * 2 * T * d, v_off = k_off + (long)T * d;
            if (cross_prefetch) MW_HIP(hipStreamWaitEvent(st, ln.ev_branch[2 * l + 1], 0));
            bracket(GEMV_N_TAGS + 1, [&]() {
                launch_cross_attn_batch(d_q, d_blk, k_off, v_off, d_part, d_attd, d_head_cnt, R, g_rows_max, G, H, d, n_ctx_max, splits, st);
            });
            if (dup_tag == GEMV_N_TAGS + 1 && prof_tag < 0)
                launch_cross_attn_batch(d_q, d_blk, k_off, v_off, d_part, d_attd, d_head_cnt, R, g_rows_max, G, H, d, n_ctx_max, splits, st);
            {
                GemvArgs g;
                g.x = (const float *)d_attd; g.x_bf16 = 1; g.ldx = d; g.n = R; g.W = e.co_w; if (m.dec_fp8) { g.W8 = e.co_w8; g.w_scale = e.co_s; } g.bias = e.co_b; g.N = d; g.K = d;
                g.out = d_x; g.ldo = d; g.R = d_x; g.tag = GEMV_ATTN_OUT;
                proj(g, 0);
            }
            {
                GemvArgs g;
                g.x = d_x; g.ldx = d; g.n = R; g.W = e.fc1_w; if (m.dec_fp8) { g.W8 = e.fc1_w8; g.w_scale = e.fc1_s; } g.bias = e.fc1_b; g.N = 4 * d; g.K = d;
                g.ln_w = e.ln2_w; g.ln_b = e.ln2_b; g.act = ACT_GELU; g.out = (float *)d_ffd; g.out_bf16 = 1; g.ldo = 4 * d; g.tag = GEMV_FC1;
                proj(g, 0);
            }
            {
                GemvArgs g;
                g.x = (const float *)d_ffd; g.x_bf16 = 1; g.ldx = 4 * d; g.n = R; g.W = e.fc2_w; if (m.dec_fp8) { g.W8 = e.fc2_w8; g.w_scale = e.fc2_s; } g.bias = e.fc2_b; g.N = d; g.K = 4 * d;
                g.out = d_x; g.ldo = d; g.R = d_x; g.tag = GEMV_FC2;
                proj(g, 0);
            }
        }
        if (LR > 0) {
            GemvArgs g;  // final LN + tied-embedding logits, only for the rows that are sampled from
            g.x = d_x; g.ldx = d; g.n = LR; g.W = m.tok_emb; if (m.dec_fp8) { g.W8 = m.tok_emb8; g.w_scale = m.tok_emb_s; } g.N = nv; g.K = d;
            g.ln_w = m.dec_ln_w; g.ln_b = m.dec_ln_b; g.out = d_logits; g.ldo = nv; g.d_rows = d_blk->logit_rows;
            g.tag = GEMV_LOGITS;
            MW_CHECK(logits_tile_supported(g), "hub: logits projection shape");
            bracket(GEMV_LOGITS, [&]() { launch_logits_tile(g, st); });
            if (dup_tag == GEMV_LOGITS && prof_tag < 0) launch_logits_tile(g, st);
        }
        if (sample && SR > 0)  // results (100 bytes per row) are stored straight into pinned host memory
            launch_process_logits(d_logits, nv, d_suppress, lp, d_blk->rs, h_res_dev, SR, d_work, nullptr, nullptr, nullptr, st);
    };
    if (!use_graphs || prof_tag >= 0 || (pc && !pc->use_graphs)) {
        enqueue();
    } else {
        uint64_t key = (uint64_t)R | ((uint64_t)G << 8) | ((uint64_t)LR << 16) | ((uint64_t)SR << 24) | ((uint64_t)splits << 32) |
                       ((uint64_t)(dup_tag + 1) << 40) | ((uint64_t)(sample ? 1 : 0) << 48) | ((uint64_t)(g_rows_max > 5 ? 1 : 0) << 49);
        if (sample) {  // the rules' parameters are kernel arguments of the captured nodes: part of the key (FNV-1a over their bytes)
            uint64_t h = 1469598103934665603ull;
            for (size_t i = 0; i < sizeof(lp); ++i) h = (h ^ ((const unsigned char *)&lp)[i]) * 1099511628211ull;
            key ^= (h | 1ull) * 0x9e3779b97f4a7c15ull;
        }
        auto it = graphs.find(key);
        if (it != graphs.end()) {
            auto pos = std::find(graph_lru.begin(), graph_lru.end(), key);
            if (pos != graph_lru.end() && pos + 1 != graph_lru.end()) std::rotate(pos, pos + 1, graph_lru.end());
        } else {
            if (graphs.size() >= 96) {  // bounded cache: evict the least recently used graph
                const uint64_t old = graph_lru.front();
                graph_lru.erase(graph_lru.begin());
                auto ot = graphs.find(old);
                if (ot != graphs.end()) {
                    (void)hipGraphExecDestroy(ot->second);
                    graphs.erase(ot);
                }
            }
            hipGraph_t graph = nullptr;
            // one capture (and instantiation) at a time in the process: a precaution, lanes and contexts capture from several
            // threads and a capture is rare (once per step shape); replays of finished graphs run concurrently
            std::lock_guard<std::mutex> capture_lock(g_capture_mu);
            MW_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            try {
                enqueue();
            } catch (...) {
                (void)hipStreamEndCapture(st, &graph);
                if (graph) (void)hipGraphDestroy(graph);
                throw;
            }
            MW_HIP(hipStreamEndCapture(st, &graph));
            hipGraphExec_t exec = nullptr;
            MW_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            MW_HIP(hipGraphDestroy(graph));
            it = graphs.emplace(key, exec).first;
            graph_lru.push_back(key);
        }
        MW_HIP(hipGraphLaunch(it->second, st));
    }
    {   // logits rows of the requests that read them back (stage calls, language detection, unfused sampler)
        int lr = 0;
        for (StepRequest *q : batch) {
            if (q->keep_logits && q->n_logit_rows > 0)
                MW_HIP(hipMemcpyAsync(q->c->d_logits + (size_t)q->logits_dst_row * nv, d_logits + (size_t)lr * nv,
                                      (size_t)q->n_logit_rows * nv * sizeof(float), hipMemcpyDeviceToDevice, st));
            lr += q->n_logit_rows;
        }
    }
    MW_HIP(hipStreamSynchronize(st));
    int sr = 0;
    for (StepRequest *q : batch) {
        if (!q->sample) continue;
        Context &c = *q->c;
        memcpy(c.h_row_res, h_res + sr, (size_t)c.ps_n * sizeof(LogitsRowResult));
        sr += c.ps_n;
    }
}

// ------------------------------------------------------------------------------------------
// mel (A.2)
// ------------------------------------------------------------------------------------------
void pcm_to_mel(Context &c, const float *pcm, int n_samples, bool pcm_on_device) {
    MW_CHECK(pcm && n_samples > 0, "pcm_to_mel: empty input");
    const int n_mel = c.model.hp.n_mels;
    const double t0 = now_ms();
    if ((size_t)n_samples > c.pcm_cap) {
        if (c.d_pcm) MW_HIP(hipFree(c.d_pcm));
        c.d_pcm = nullptr;
        c.pcm_cap = std::max<size_t>(n_samples, 30 * WHISPER_SAMPLE_RATE + 640);
        dev_alloc(c.d_pcm, c.pcm_cap);
    }
    MW_HIP(hipMemcpyAsync(c.d_pcm, pcm, (size_t)n_samples * 4, pcm_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice,
                          c.st));
    // reference frame counts: 30 s of zero padding at the end, 200-sample reflect pad at the start
    const int n_len = (n_samples + 30 * WHISPER_SAMPLE_RATE) / MEL_HOP;
    const int n_real = n_samples + MEL_N_FFT / 2;
    const int n_active = std::min(n_real / MEL_HOP + 1, n_len);
    if ((size_t)n_len > c.mel_cap_frames) {
        if (c.d_mel_raw) MW_HIP(hipFree(c.d_mel_raw));
        if (c.d_mel) MW_HIP(hipFree(c.d_mel));
        c.d_mel_raw = c.d_mel = nullptr;
        c.mel_cap_frames = std::max<size_t>(n_len, 6000);
        dev_alloc(c.d_mel_raw, c.mel_cap_frames * n_mel);
        dev_alloc(c.d_mel, c.mel_cap_frames * n_mel);
        if (!c.d_gmax) dev_alloc(c.d_gmax, 16);
    }
    launch_mel_frames(c.d_pcm, 0, n_samples, 1, c.model.d_mel_tab, c.d_mel_raw, (long)c.mel_cap_frames * n_mel, n_active,
                      n_mel, c.d_gmax, c.st);
    launch_mel_normalize(c.d_mel_raw, (long)c.mel_cap_frames * n_mel, c.d_mel, (long)c.mel_cap_frames * n_mel, n_len,
                         n_active, n_mel, 1, c.d_gmax, c.st);
    c.n_len = n_len;
    c.n_len_org = 1 + (n_samples + MEL_N_FFT / 2 - MEL_N_FFT) / MEL_HOP;
    // the hub's encoder lane reads d_mel on ITS stream: it must wait for these kernels (Hub::run_encode waits on ev_ready
    // only when the flag is set; after a context's first call the last hub step had cleared it)
    c.st_dirty = true;
    c.tm.mel_ms += now_ms() - t0;
}

// ------------------------------------------------------------------------------------------
// encoder + cross-KV (A.3)
// ------------------------------------------------------------------------------------------
void encode(Context &c, int seek) {
    const Model &m = c.model;
    const HParams &hp = m.hp;
    const DType wt = m.wt;
    const size_t ws = dtype_size(wt);
    const int n_ctx = c.exp_n_audio_ctx > 0 ? c.exp_n_audio_ctx : hp.n_audio_ctx;
    MW_CHECK(n_ctx <= hp.n_audio_ctx, "audio_ctx larger than the model's");
    const int d = hp.n_audio_state, H = hp.n_audio_head, n_mel = hp.n_mels, T2 = 2 * n_ctx;
    const int tp = (n_ctx + 63) / 64 * 64;
    hipStream_t st = c.st;
    const double t0 = now_ms();
    MW_CHECK(c.d_mel != nullptr, "encode called before any mel was computed");
    if (wt == DType::BF16 && c.use_hub && c.model.hub && c.flash_attn && !c.gemm_glds) {
        // batched path: the window joins the encoder pass of the hub (with the windows other contexts have queued)
        EncRequest r;
        r.c = &c;
        r.seek = seek;
        r.n_ctx = n_ctx;
        r.peers = (int)c.model.hub.use_count();
        if (c.st_dirty) MW_HIP(hipEventRecord(c.ev_ready, c.st));
        c.model.hub->submit_encode(r);
        c.tm.n_encode++;
        c.tm.encode_ms += now_ms() - t0;
        return;
    }
    c.n_ctx_cur = n_ctx;
    auto gemm = [&](GemmArgs &g, DType in, DType out, hipStream_t s) {
        g.glds = c.gemm_glds ? 1 : 0;
        launch_gemm(g, in, out, s);
    };
    launch_mel_window(c.d_mel, 0, c.n_len, seek, T2, n_mel, 1, c.d_conv_in, (long)(T2 + 2) * n_mel, wt, st);
    {   // conv1 as a GEMM over the overlapping row view: output row t <- padded rows t, t+1, t+2
        GemmArgs g;
        g.A = c.d_conv_in; g.lda = n_mel; g.W = m.conv1_w; g.ldw = 3 * n_mel;
        g.C = (char *)c.d_x1 + (size_t)d * ws; g.ldc = d;  // x1 row 0 stays zero (conv2's left pad)
        g.bias = m.conv1_b; g.M = T2; g.N = d; g.K = 3 * n_mel; g.act = ACT_GELU;
        gemm(g, wt, wt, st);
    }
    {   // conv2, stride 2: output row t <- padded x1 rows 2t, 2t+1, 2t+2; + positional embedding
        GemmArgs g;
        g.A = c.d_x1; g.lda = 2 * d; g.W = m.conv2_w; g.ldw = 3 * d;
        g.C = c.d_x; g.ldc = d; g.bias = m.conv2_b; g.M = n_ctx; g.N = d; g.K = 3 * d; g.act = ACT_GELU;
        g.R = m.enc_pe; g.ldr = d;
        gemm(g, wt, DType::F32, st);
    }
    const MxWorkspace mxw = mx_workspace(c.d_mx, hp.n_audio_ctx, d);
    for (int l = 0; l < hp.n_audio_layer; ++l) {
        const EncLayer &e = m.enc[l];
        if (m.enc_mx) {
            mx_ln_qkv(e, c.d_x, c.d_qk, c.d_vt, mxw, 1, n_ctx, tp, d, st);
        } else {
            launch_layernorm(c.d_x, d, e.ln1_w, e.ln1_b, c.d_h, d, wt, n_ctx, d, st);
            GemmArgs g;  // Q | K
            g.A = c.d_h; g.lda = d; g.W = e.qk_w; g.ldw = d; g.C = c.d_qk; g.ldc = 2 * d; g.bias = e.qk_b;
            g.M = n_ctx; g.N = 2 * d; g.K = d;
            gemm(g, wt, wt, st);
        }
        if (!m.enc_mx) {
            GemmArgs g;  // V, stored transposed [d][tp]
            g.A = c.d_h; g.lda = d; g.W = e.v_w; g.ldw = d; g.C = c.d_vt; g.ldc = tp; g.bias = e.v_b;
            g.M = n_ctx; g.N = d; g.K = d; g.trans_c = 1;
            gemm(g, wt, wt, st);
        }
        if (c.flash_attn && m.enc_mx) {
            launch_enc_flash_attn(c.d_qk, c.d_vt, nullptr, n_ctx, tp, d, H, st, 1, 3 * d, mxw.h8, mxw.hs);
        } else if (c.flash_attn) {
            launch_enc_flash_attn(c.d_qk, c.d_vt, c.d_att, n_ctx, tp, d, H, st, 1, 2 * d);  // fused: scores never leave the chip
        } else {
            {
                GemmArgs g;  // S_h = Q_h K_h^T, batched over heads
                const int qk_ld = m.enc_mx ? 3 * d : 2 * d;  // (MX-fp8 mode: rows of Q | K | V)
                g.A = c.d_qk; g.lda = qk_ld; g.W = (char *)c.d_qk + (size_t)d * ws; g.ldw = qk_ld;
                g.C = c.d_s; g.ldc = tp; g.M = n_ctx; g.N = n_ctx; g.K = 64;
                g.nz1 = H; g.sA1 = 64; g.sW1 = 64; g.sC1 = (long)n_ctx * tp;
                gemm(g, wt, DType::F32, st);
            }
            launch_softmax_rows(c.d_s, tp, c.d_p, tp, wt, (long)H * n_ctx, n_ctx, tp, 1.0f / sqrtf(64.0f), st);
            {
                GemmArgs g;  // O_h = P_h V_h
                g.A = c.d_p; g.lda = tp; g.W = c.d_vt; g.ldw = tp; g.C = c.d_att; g.ldc = d;
                g.M = n_ctx; g.N = 64; g.K = tp;
                g.nz1 = H; g.sA1 = (long)n_ctx * tp; g.sW1 = (long)64 * tp; g.sC1 = 64;
                gemm(g, wt, wt, st);
            }
        }
        if (m.enc_mx) {
            if (!c.flash_attn) launch_quantize_mx_rows_bf16(c.d_att, d, n_ctx, d, mxw.h8, d, mxw.hs, st);
            mx_attn_out(e, c.d_x, mxw, n_ctx, d, st);
            mx_mlp(e, c.d_x, c.d_ff, mxw, n_ctx, d, st);
            continue;
        }
        {
            GemmArgs g;  // x += att Wo^T + bo
            g.A = c.d_att; g.lda = d; g.W = e.out_w; g.ldw = d; g.C = c.d_x; g.ldc = d; g.bias = e.out_b;
            g.R = c.d_x; g.ldr = d; g.M = n_ctx; g.N = d; g.K = d;
            gemm(g, wt, DType::F32, st);
        }
        launch_layernorm(c.d_x, d, e.ln2_w, e.ln2_b, c.d_h, d, wt, n_ctx, d, st);
        {
            GemmArgs g;
            g.A = c.d_h; g.lda = d; g.W = e.fc1_w; g.ldw = d; g.C = c.d_ff; g.ldc = 4 * d; g.bias = e.fc1_b;
            g.M = n_ctx; g.N = 4 * d; g.K = d; g.act = ACT_GELU;
            gemm(g, wt, wt, st);
        }
        {
            GemmArgs g;
            g.A = c.d_ff; g.lda = 4 * d; g.W = e.fc2_w; g.ldw = 4 * d; g.C = c.d_x; g.ldc = d; g.bias = e.fc2_b;
            g.R = c.d_x; g.ldr = d; g.M = n_ctx; g.N = d; g.K = 4 * d;
            gemm(g, wt, DType::F32, st);
        }
    }
    if (m.enc_mx) {
        mx_ln_post_cross(m, c.d_x, c.d_enc_out, c.d_cross, mxw, n_ctx, st);
    } else {
        launch_layernorm(c.d_x, d, m.ln_post_w, m.ln_post_b, c.d_enc_out, d, wt, n_ctx, d, st);
        // cross K (scaled by dh^-1/4, no bias) and V (+bias) of every decoder layer, batched over layers
        const int L = hp.n_text_layer;
        const long per_layer = 2L * hp.n_audio_ctx * d;
        GemmArgs g;
        g.A = c.d_enc_out; g.lda = d; g.W = m.ckv_w; g.ldw = d; g.C = c.d_cross; g.ldc = d;
        g.M = n_ctx; g.N = d; g.K = d; g.nz1 = L; g.sW1 = 2L * d * d; g.sC1 = per_layer;
        g.alpha = powf(64.0f, -0.25f);
        gemm(g, wt, wt, st);
        GemmArgs v = g;
        v.W = (char *)m.ckv_w + (size_t)d * d * ws;
        v.C = (char *)c.d_cross + (size_t)hp.n_audio_ctx * d * ws;
        v.bias = m.ckv_b + d; v.sB1 = 2L * d; v.alpha = 1.0f;
        gemm(v, wt, wt, st);
    }
    c.st_dirty = true;
    c.tm.n_encode++;
    c.tm.encode_ms += now_ms() - t0;
}

// ------------------------------------------------------------------------------------------
// decoder (A.4)
// ------------------------------------------------------------------------------------------
void kv_clear(Context &c) {
    c.reorder_pending = false;  // superseded
    const size_t bytes = (size_t)KV_SLOTS * c.model.hp.n_text_ctx * sizeof(int);
    MW_HIP(hipMemsetAsync(c.d_kv_idx[0], 0, bytes, c.st));
    MW_HIP(hipMemsetAsync(c.d_kv_idx[1], 0, bytes, c.st));
    c.kv_cur = 0;
    c.st_dirty = true;
}

namespace {
void flush_reorder(Context &c) {  // a deferred index-table update that no decode call picked up
    if (!c.reorder_pending) return;
    BeamSrc b;
    for (int j = 0; j < KV_SLOTS; ++j) b.src[j] = c.reorder_src[j];
    launch_kv_reorder(c.d_kv_idx[c.kv_cur ^ 1], c.d_kv_idx[c.kv_cur], b, c.reorder_n_past, c.model.hp.n_text_ctx, c.st);
    c.reorder_pending = false;
    c.st_dirty = true;
}
}  // namespace

// The update runs as the first kernel of the next decode call's launch sequence (arguments travel in that call's
// step block), so a beam step is one graph launch; the table parity flips here.
void kv_reorder(Context &c, const int *src, int n_past) {
    MW_CHECK(n_past >= 0 && n_past < c.model.hp.n_text_ctx, "kv reorder: position out of range");
    flush_reorder(c);
    for (int j = 0; j < KV_SLOTS; ++j) c.reorder_src[j] = src[j];
    c.reorder_n_past = n_past;
    c.reorder_pending = true;
    c.kv_cur ^= 1;
}

void decode(Context &c, const int *tokens, const int *pos, const int *slot, int n, const int *logit_rows,
            int n_logit_rows) {
    const Model &m = c.model;
    const HParams &hp = m.hp;
    const DType wt = m.wt;
    const size_t ws = dtype_size(wt);
    const int d = hp.n_text_state, H = hp.n_text_head, TC = hp.n_text_ctx, n_ctx = c.n_ctx_cur;
    MW_CHECK(n > 0 && n <= TC, "decode: batch size out of range");
    MW_CHECK(n_logit_rows >= 0 && n_logit_rows <= LOGIT_ROWS, "decode: too many logits rows");
    for (int i = 0; i < n; ++i) {
        MW_CHECK(tokens[i] >= 0 && tokens[i] < hp.n_vocab, "decode: token id out of range");
        MW_CHECK(pos[i] >= 0 && pos[i] < TC, "decode: position beyond n_text_ctx");
        MW_CHECK(slot[i] >= 0 && slot[i] < KV_SLOTS, "decode: bad sequence slot");
    }
    for (int i = 0; i < n_logit_rows; ++i) MW_CHECK(logit_rows[i] >= 0 && logit_rows[i] < n, "decode: bad logits row");
    if (c.tile_step && c.use_hub && c.model.hub && (!c.ps_armed || c.ps_n <= STEP_MAX_REQ_ROWS)) {
        // batched path: the step runs on the hub of the contexts that share these weights, together with whatever other
        // contexts have queued meanwhile (one pass over the weights for all their rows).  A call of more than STEP_MAX_REQ_ROWS
        // rows (a long prompt: initial_prompt / prompt_past, whisper-processing.cpp:191-199) goes through in chunks of
        // STEP_MAX_REQ_ROWS consecutive rows, one step each — one pass over the weights per 16 tokens on the matrix-core tile
        // kernels; a chunk's rows see the K / V of the chunks before it through the cache (a kernel boundary apart).
        const bool want_sample = c.ps_armed && c.fuse_sample && n_logit_rows > 0 && c.prof_tag < 0;
        const int n_chunks = ceil_div(n, STEP_MAX_REQ_ROWS), last0 = (n_chunks - 1) * STEP_MAX_REQ_ROWS;
        bool chunkable = n_chunks == 1 || (c.prof_tag < 0 && c.dup_tag < 0 && c.use_graphs);
        if (n_chunks > 1) {
            // logits rows in call order (the copies below land them in that order); a fused sampler wants them all in the
            // last chunk (its rows index the step's logits rows)
            for (int i = 1; i < n_logit_rows; ++i) chunkable = chunkable && logit_rows[i] > logit_rows[i - 1];
            if (want_sample)
                for (int i = 0; i < n_logit_rows; ++i) chunkable = chunkable && logit_rows[i] >= last0;
        }
        if (chunkable) {
            const double t0h = now_ms();
            int lr_done = 0;
            bool fused = false;
            for (int ch = 0; ch < n_chunks; ++ch) {
                const int r0 = ch * STEP_MAX_REQ_ROWS, nr = std::min(STEP_MAX_REQ_ROWS, n - r0);
                int lrows[STEP_MAX_REQ_ROWS], nl = 0;
                for (int i = 0; i < n_logit_rows; ++i)
                    if (logit_rows[i] >= r0 && logit_rows[i] < r0 + nr) lrows[nl++] = logit_rows[i] - r0;
                StepRequest r;
                r.c = &c;
                r.tokens = tokens + r0; r.pos = pos + r0; r.slot = slot + r0; r.n = nr;
                r.logit_rows = lrows; r.n_logit_rows = nl;
                r.sample = want_sample && nl > 0 && ch == n_chunks - 1;
                r.keep_logits = !r.sample && nl > 0;
                r.logits_dst_row = lr_done;
                r.solo = c.prof_tag >= 0 || c.dup_tag >= 0 || !c.use_graphs;
                if (r.sample) r.rules_key = std::string((const char *)&c.ps_lp, sizeof(c.ps_lp)) + "|" + c.suppress_key;
                if (c.st_dirty) MW_HIP(hipEventRecord(c.ev_ready, c.st));
                c.model.hub->submit(r);
                fused = fused || r.sample;
                lr_done += nl;
                c.tm.n_decode++;
            }
            c.ps_fused = fused;  // the results are in h_row_res already
            c.n_logit_rows = n_logit_rows;
            c.tm.decode_ms += now_ms() - t0h;
            return;
        }
    }
    hipStream_t st = c.st;
    const bool tile = c.tile_step && n <= 5;  // bf16 rows between attention / MLP kernels and the residual projections
    const bool mf = tile;
    const DType act_t = mf ? DType::BF16 : DType::F32;
    const double t0 = now_ms();
    // the pinned staging block may still be read by the previous call's copy: wait for THAT call only (an event
    // recorded behind it), not for whatever else is in flight on the stream
    if (c.staging_pending) MW_HIP(hipEventSynchronize(c.ev_staging));
    c.staging_pending = false;
    int *hi = c.h_ints;
    memcpy(hi, tokens, n * sizeof(int));
    memcpy(hi + TC, pos, n * sizeof(int));
    memcpy(hi + 2 * TC, slot, n * sizeof(int));
    memcpy(hi + 3 * TC, logit_rows, n_logit_rows * sizeof(int));
    int *hu = hi + 3 * TC + LOGIT_ROWS;  // index-table update: src[KV_SLOTS], n_past, flag
    for (int j = 0; j < KV_SLOTS; ++j) hu[j] = c.reorder_pending ? c.reorder_src[j] : -1;
    hu[KV_SLOTS] = c.reorder_n_past;
    hu[KV_SLOTS + 1] = c.reorder_pending ? 1 : 0;
    c.reorder_pending = false;
    size_t step_bytes = (size_t)(3 * TC + LOGIT_ROWS + KV_SLOTS + 2) * sizeof(int);
    // prepared rules + sampler rows (prepare_sample): appended to the step block, 8-byte aligned
    const size_t rs_off = (step_bytes + 7) / 8 * 8;
    const bool fuse = c.ps_armed && c.fuse_sample && c.use_graphs && c.prof_tag < 0 && n_logit_rows > 0 &&
                      rs_off + (size_t)c.ps_n * sizeof(LogitsRowState) <= (size_t)4 * TC * sizeof(int);
    if (fuse) {
        memcpy((char *)hi + rs_off, c.h_row_state, (size_t)c.ps_n * sizeof(LogitsRowState));
        step_bytes = rs_off + (size_t)c.ps_n * sizeof(LogitsRowState);
    }
    const LogitsRowState *d_rs_block = reinterpret_cast<const LogitsRowState *>((const char *)c.d_tokens + rs_off);
    // Everything below is one fixed launch sequence for a given (rows, logits rows, index table, audio
    // context): it is captured into a hipGraph once and replayed, because the step is launch-bound
    // (~290 launches of 3-10 us kernels).  Per-step values (tokens, positions, slots) travel through
    // the pinned staging block and device arrays at fixed addresses, so the captured nodes stay valid.
    auto enqueue = [&]() {
        const bool step_begin = c.h_ints_dev != nullptr;
        if (step_begin) {  // block from pinned host memory -> index-table update + embedding + device copy of the block, one launch
            launch_step_begin(c.h_ints_dev, c.d_tokens, (int)((step_bytes + 3) / 4), TC, 3 * TC + LOGIT_ROWS, n, c.d_kv_idx[c.kv_cur ^ 1],
                              c.d_kv_idx[c.kv_cur], m.tok_emb, wt, m.dec_pe, c.d_xd, d, st);
        } else {
            MW_HIP(hipMemcpyAsync(c.d_tokens, hi, step_bytes, hipMemcpyHostToDevice, st));  // the whole step block in one copy
            launch_kv_reorder_dev(c.d_kv_idx[c.kv_cur ^ 1], c.d_kv_idx[c.kv_cur], c.d_rows + LOGIT_ROWS, TC, st);  // no-op without an update
        }
        // launches of the profiled projection class are bracketed by HIP events (mw_profile_begin/end)
        auto gemv = [&](const GemvArgs &g) {
            const bool prof = g.tag == c.prof_tag && c.prof_n + 2 <= (int)c.prof_ev.size();
            if (prof) MW_HIP(hipEventRecord(c.prof_ev[c.prof_n++], st));
            // <= 5 rows on bf16 weights: matrix-core tile kernels (dec_tile.hip); else the VALU weight-streaming path (dec.hip)
            if (c.tile_step && g.tag == GEMV_LOGITS && logits_tile_supported(g)) launch_logits_tile(g, st);
            else if (tile && g.tag != GEMV_LOGITS) launch_proj_wg(g, st);
            else launch_gemv(g, wt, st);
            if (prof) MW_HIP(hipEventRecord(c.prof_ev[c.prof_n++], st));
        };
        // measurement (mw_set_duplicate_class): the class's launches are issued twice; an in-place residual update writes
        // its second result into the cross-attention partial buffer (dead at that point of the layer)
        auto gemv2 = [&](GemvArgs g) {
            gemv(g);
            if (g.tag == c.dup_tag && c.prof_tag < 0) {
                if (g.R && g.out == g.R) g.out = c.d_part;
                gemv(g);
            }
        };
        if (!step_begin) launch_embed(c.d_tokens, c.d_pos, m.tok_emb, wt, m.dec_pe, c.d_xd, n, d, st);
        const float qk_scale = powf(64.0f, -0.25f);
        const int splits = std::min(CROSS_SPLITS_MAX, ceil_div(n_ctx, 128));
        const size_t kv_layer = (size_t)KV_SLOTS * TC * d * ws;
        // <= 5 rows on bf16 K/V: workgroup-wide attention kernels (dec_tile.hip), cross-attention combines in-kernel
        const bool wg_attn = tile && ceil_div(n_ctx, splits) <= 128;
        for (int l = 0; l < hp.n_text_layer; ++l) {
            const DecLayer &e = m.dec[l];
            {
                GemvArgs g;  // LN + Q|K|V; K and V go straight into the cache
                g.x = c.d_xd; g.ldx = d; g.n = n; g.W = e.qkv_w; g.bias = e.qkv_b; g.N = 3 * d; g.K = d;
                g.ln_w = e.ln1_w; g.ln_b = e.ln1_b; g.out = c.d_q; g.ldo = d; g.qkv = 1; g.qk_scale = qk_scale;
                g.kcache = (char *)c.d_kself + l * kv_layer; g.vcache = (char *)c.d_vself + l * kv_layer;
                g.kv_t = wt; g.d_slot = c.d_slot; g.d_pos = c.d_pos; g.n_text_ctx = TC; g.tag = GEMV_QKV;
                gemv2(g);
            }
            {
                const bool prof = c.prof_tag == GEMV_N_TAGS && c.prof_n + 2 <= (int)c.prof_ev.size();
                if (prof) MW_HIP(hipEventRecord(c.prof_ev[c.prof_n++], st));
                if (wg_attn)
                    launch_self_attn_wg(c.d_q, (char *)c.d_kself + l * kv_layer, (char *)c.d_vself + l * kv_layer, c.d_slot, c.d_pos,
                                        c.d_kv_idx[c.kv_cur], c.d_attd, act_t, n, H, d, TC, st);
                else
                    launch_self_attn(c.d_q, (char *)c.d_kself + l * kv_layer, (char *)c.d_vself + l * kv_layer, wt, c.d_slot, c.d_pos,
                                     c.d_kv_idx[c.kv_cur], c.d_attd, act_t, n, H, d, TC, st);
                if (prof) MW_HIP(hipEventRecord(c.prof_ev[c.prof_n++], st));
                if (c.dup_tag == GEMV_N_TAGS && c.prof_tag < 0 && wg_attn)
                    launch_self_attn_wg(c.d_q, (char *)c.d_kself + l * kv_layer, (char *)c.d_vself + l * kv_layer, c.d_slot, c.d_pos,
                                        c.d_kv_idx[c.kv_cur], c.d_attd, act_t, n, H, d, TC, st);
            }
            {
                GemvArgs g;
                g.x = (const float *)c.d_attd; g.x_bf16 = mf; g.ldx = d; g.n = n; g.W = e.out_w; g.bias = e.out_b; g.N = d; g.K = d;
                g.out = c.d_xd; g.ldo = d; g.R = c.d_xd; g.tag = GEMV_ATTN_OUT;
                gemv2(g);
            }
            {
                GemvArgs g;  // LN + cross query (scaled)
                g.x = c.d_xd; g.ldx = d; g.n = n; g.W = e.cq_w; g.bias = e.cq_b; g.N = d; g.K = d;
                g.ln_w = e.lnc_w; g.ln_b = e.lnc_b; g.out = c.d_q; g.ldo = d; g.out_scale = qk_scale; g.tag = GEMV_CROSS_Q;
                gemv2(g);
            }
            const char *ck = (const char *)c.d_cross + (size_t)l * 2 * hp.n_audio_ctx * d * ws;
            {
                const bool prof = c.prof_tag == GEMV_N_TAGS + 1 && c.prof_n + 2 <= (int)c.prof_ev.size();
                if (prof) MW_HIP(hipEventRecord(c.prof_ev[c.prof_n++], st));
                if (wg_attn)
                    launch_cross_attn_wg(c.d_q, ck, ck + (size_t)hp.n_audio_ctx * d * ws, c.d_part, c.d_attd, act_t, c.d_head_cnt, n, H, d,
                                         n_ctx, splits, st);
                else
                    launch_cross_attn(c.d_q, ck, ck + (size_t)hp.n_audio_ctx * d * ws, wt, c.d_part, c.d_attd, act_t, n, H, d, n_ctx,
                                      splits, st);
                if (prof) MW_HIP(hipEventRecord(c.prof_ev[c.prof_n++], st));
                if (c.dup_tag == GEMV_N_TAGS + 1 && c.prof_tag < 0 && wg_attn)
                    launch_cross_attn_wg(c.d_q, ck, ck + (size_t)hp.n_audio_ctx * d * ws, c.d_part, c.d_attd, act_t, c.d_head_cnt, n, H, d,
                                         n_ctx, splits, st);
            }
            {
                GemvArgs g;
                g.x = (const float *)c.d_attd; g.x_bf16 = mf; g.ldx = d; g.n = n; g.W = e.co_w; g.bias = e.co_b; g.N = d; g.K = d;
                g.out = c.d_xd; g.ldo = d; g.R = c.d_xd; g.tag = GEMV_ATTN_OUT;
                gemv2(g);
            }
            {
                GemvArgs g;
                g.x = c.d_xd; g.ldx = d; g.n = n; g.W = e.fc1_w; g.bias = e.fc1_b; g.N = 4 * d; g.K = d;
                g.ln_w = e.ln2_w; g.ln_b = e.ln2_b; g.act = ACT_GELU; g.out = (float *)c.d_ffd; g.out_bf16 = mf; g.ldo = 4 * d; g.tag = GEMV_FC1;
                gemv2(g);
            }
            {
                GemvArgs g;
                g.x = (const float *)c.d_ffd; g.x_bf16 = mf; g.ldx = 4 * d; g.n = n; g.W = e.fc2_w; g.bias = e.fc2_b; g.N = d; g.K = 4 * d;
                g.out = c.d_xd; g.ldo = d; g.R = c.d_xd; g.tag = GEMV_FC2;
                gemv2(g);
            }
        }
        if (n_logit_rows > 0) {
            GemvArgs g;  // final LN + tied-embedding logits, only for the rows that are sampled from
            g.x = c.d_xd; g.ldx = d; g.n = n_logit_rows; g.W = m.tok_emb; g.N = hp.n_vocab; g.K = d;
            g.ln_w = m.dec_ln_w; g.ln_b = m.dec_ln_b; g.out = c.d_logits; g.ldo = hp.n_vocab; g.d_rows = c.d_rows;
            g.tag = GEMV_LOGITS;
            gemv2(g);
        }
        if (fuse) {
            // results (100 bytes per row, write-only in the kernel) are stored straight into pinned host memory: no copy node
            launch_process_logits(c.d_logits, hp.n_vocab, c.d_suppress, c.ps_lp, d_rs_block, c.d_row_res_host, c.ps_n, c.d_work,
                                  nullptr, nullptr, nullptr, st);
        }
    };
    const bool use_graph = c.use_graphs && c.prof_tag < 0;
    if (!use_graph) {
        enqueue();
    } else {
        uint64_t key = ((uint64_t)(c.dup_tag + 1) << 56) | ((uint64_t)n << 40) | ((uint64_t)n_logit_rows << 24) | ((uint64_t)n_ctx << 4) |
                       (uint64_t)c.kv_cur;
        if (fuse) {  // the rules' parameters are kernel arguments of the captured nodes: part of the key (FNV-1a over their bytes)
            uint64_t h = 1469598103934665603ull;
            auto mix = [&](const void *p, size_t nb) {
                for (size_t i = 0; i < nb; ++i) h = (h ^ ((const unsigned char *)p)[i]) * 1099511628211ull;
            };
            mix(&c.ps_lp, sizeof(c.ps_lp));
            mix(&c.ps_n, sizeof(c.ps_n));
            key ^= (h | 1ull) * 0x9e3779b97f4a7c15ull;
        }
        auto it = c.graphs.find(key);
        if (it != c.graphs.end()) {  // most recently used last
            auto pos = std::find(c.graph_lru.begin(), c.graph_lru.end(), key);
            if (pos != c.graph_lru.end() && pos + 1 != c.graph_lru.end()) std::rotate(pos, pos + 1, c.graph_lru.end());
        } else {
            if (c.graphs.size() >= 96) {  // bounded cache (prompt lengths vary): evict the least recently used graph only
                const uint64_t old = c.graph_lru.front();
                c.graph_lru.erase(c.graph_lru.begin());
                auto ot = c.graphs.find(old);
                if (ot != c.graphs.end()) {
                    (void)hipGraphExecDestroy(ot->second);
                    c.graphs.erase(ot);
                }
            }
            hipGraph_t graph = nullptr;
            // one capture (and instantiation) at a time in the process: a precaution, lanes and contexts capture from several
            // threads and a capture is rare (once per step shape); replays of finished graphs run concurrently
            std::lock_guard<std::mutex> capture_lock(g_capture_mu);
            MW_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            try {
                enqueue();
            } catch (...) {
                (void)hipStreamEndCapture(st, &graph);
                if (graph) (void)hipGraphDestroy(graph);
                throw;
            }
            MW_HIP(hipStreamEndCapture(st, &graph));
            hipGraphExec_t exec = nullptr;
            MW_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            MW_HIP(hipGraphDestroy(graph));
            it = c.graphs.emplace(key, exec).first;
            c.graph_lru.push_back(key);
        }
        MW_HIP(hipGraphLaunch(it->second, st));
    }
    MW_HIP(hipEventRecord(c.ev_staging, st));
    c.staging_pending = true;
    c.st_dirty = true;
    c.ps_fused = fuse && use_graph;
    c.n_logit_rows = n_logit_rows;
    c.tm.n_decode++;
    c.tm.decode_ms += now_ms() - t0;
}

// ------------------------------------------------------------------------------------------
// logits rules: static mask (A.5 steps 3-6) built on the host once per parameter set
// ------------------------------------------------------------------------------------------
void upload_suppress_mask(Context &c, const whisper_full_params &p) {
    const Vocab &v = c.model.vocab;
    std::string key = std::string(p.no_timestamps ? "1" : "0") + (p.tdrz_enable ? "1" : "0") +
                      (p.suppress_non_speech_tokens ? "1" : "0") + (c.force_decode_steps > 0 ? "1" : "0") + "|" +
                      (p.suppress_regex ? p.suppress_regex : "");
    if (key == c.suppress_key) return;
    std::vector<unsigned char> mask(v.n_vocab, 0);
    mask[v.not_] = 1;
    if (c.force_decode_steps > 0) mask[v.eot] = 1;  // bench-only extension
    if (p.no_timestamps)
        for (int i = v.beg; i < v.n_vocab; ++i) mask[i] = 1;
    mask[v.sot] = 1;
    mask[v.nosp] = 1;
    if (!p.tdrz_enable) mask[v.solm] = 1;
    mask[v.translate] = 1;
    mask[v.transcribe] = 1;
    mask[v.prev] = 1;
    for (int i = 0; i < N_LANGS; ++i)
        if (v.lang_token(i) < v.n_vocab) mask[v.lang_token(i)] = 1;
    if (p.suppress_regex && p.suppress_regex[0]) {
        std::regex re(p.suppress_regex);
        for (const auto &kv : v.token_to_id)
            if (std::regex_match(kv.first, re)) mask[kv.second] = 1;
    }
    if (p.suppress_non_speech_tokens) {
        for (const char *t : k_non_speech)
            for (const std::string &s : {std::string(t), std::string(" ") + t}) {
                auto it = v.token_to_id.find(s);
                if (it != v.token_to_id.end()) mask[it->second] = 1;
            }
        for (const char *t : {" -", " '"}) {
            auto it = v.token_to_id.find(t);
            if (it != v.token_to_id.end()) mask[it->second] = 1;
        }
    }
    MW_HIP(hipMemcpyAsync(c.d_suppress, mask.data(), mask.size(), hipMemcpyHostToDevice, c.st));
    MW_HIP(hipStreamSynchronize(c.st));  // mask is a stack-lifetime host buffer
    c.suppress_key = key;
}

LogitsParams make_logits_params(const Context &c, const whisper_full_params &p, float temperature, int n_samples) {
    const Vocab &v = c.model.vocab;
    LogitsParams lp;
    lp.n_vocab = v.n_vocab;
    lp.eot = v.eot;
    lp.beg = v.beg;
    auto it = v.token_to_id.find(" ");
    lp.space_id = it != v.token_to_id.end() ? it->second : -1;
    lp.suppress_blank = p.suppress_blank ? 1 : 0;
    lp.temperature = temperature;
    lp.tid0_initial = -1;
    if (p.max_initial_ts > 0.0f) {
        const float precision = float(WHISPER_CHUNK_SIZE) / c.model.hp.n_audio_ctx;
        lp.tid0_initial = (int)std::round(p.max_initial_ts / precision);
    }
    lp.n_samples = n_samples;
    return lp;
}

namespace {

// Row states + parameters of the rules + samplers for the listed decoders (row j serves decoders[idx[j]]), prepared BEFORE
// the decode call that produces their logits so that they can travel in its step block and run inside its graph.
void prepare_sample(Context &c, const whisper_full_params &params, float t_cur, const std::vector<int> &dec_idx,
                    const std::vector<int> &logits_row, int n_samples) {
    const int n = (int)dec_idx.size();
    MW_CHECK(n > 0 && n <= LOGIT_ROWS, "sampler: row count");
    const Vocab &v = c.model.vocab;
    for (int r = 0; r < n; ++r) {
        Decoder &d = c.decoders[dec_idx[r]];
        const auto &tk = d.sequence.tokens;
        LogitsRowState &s = c.h_row_state[r];
        s.logits_row = logits_row[r];
        s.n_sampled = (int)tk.size();
        s.last_is_ts = !tk.empty() && tk.back().id >= v.beg;
        s.penult_is_ts = tk.size() < 2 || tk[tk.size() - 2].id >= v.beg;
        s.has_ts = d.has_ts;
        s.seek_delta = d.seek_delta;
        for (int k = 0; k < n_samples; ++k) s.u[k] = std::generate_canonical<double, 53>(d.rng);
    }
    memset(&c.ps_lp, 0, sizeof(c.ps_lp));  // padding bytes are hashed into the graph key
    c.ps_lp = make_logits_params(c, params, t_cur, n_samples);
    c.ps_n = n;
    c.ps_armed = true;
    c.ps_fused = false;
}

// results of the prepared rows: enqueues the rules + samplers here unless the decode call already did (ps_fused)
void process_and_sample(Context &c, const std::vector<int> &dec_idx) {
    const double t0 = now_ms();
    const int n = (int)dec_idx.size();
    MW_CHECK(c.ps_armed && n == c.ps_n, "sampler: rows were not prepared");
    if (!c.ps_fused) {
        MW_HIP(hipMemcpyAsync(c.d_row_state, c.h_row_state, n * sizeof(LogitsRowState), hipMemcpyHostToDevice, c.st));
        launch_process_logits(c.d_logits, c.model.vocab.n_vocab, c.d_suppress, c.ps_lp, c.d_row_state, c.d_row_res, n, c.d_work,
                              nullptr, nullptr, nullptr, c.st);
        MW_HIP(hipMemcpyAsync(c.h_row_res, c.d_row_res, n * sizeof(LogitsRowResult), hipMemcpyDeviceToHost, c.st));
    }
    MW_HIP(hipStreamSynchronize(c.st));
    c.ps_armed = c.ps_fused = false;
    for (int r = 0; r < n; ++r) c.decoders[dec_idx[r]].sampled = c.h_row_res[r];
    c.tm.n_sample++;
    c.tm.sample_ms += now_ms() - t0;
}

whisper_token_data make_token(const Vocab &v, const LogitsRowResult &r, int k, int tid_default) {
    whisper_token_data t = {r.id[k], r.tid >= 0 ? r.tid : tid_default, r.p[k], r.plog[k], r.pt, r.ptsum, -1, -1, -1, 0.0f};
    if (t.id >= v.beg) {
        t.tid = t.id;
        t.pt = t.p;
    }
    return t;
}

void sequence_score(const whisper_full_params &params, Sequence &s) {
    if (s.result_len == 0) return;
    double result = 0.0;
    for (int i = 0; i < s.result_len; ++i) result += s.tokens[i].plog;
    s.sum_logprobs = result;
    s.avg_logprobs = result / s.result_len;
    double penalty = s.result_len;
    if (params.length_penalty > 0.0f) penalty = pow((5.0 + penalty) / 6.0, params.length_penalty);
    s.score = result / penalty;
    const int n = 32;
    int cnt = 0;
    double entropy = 0.0;
    std::map<int, int> counts;
    for (int i = std::max(0, s.result_len - n); i < s.result_len; ++i) {
        counts[s.tokens[i].id]++;
        cnt++;
    }
    for (const auto &kv : counts) {
        const double p = kv.second / (double)cnt;
        entropy -= p * log(p);
    }
    s.entropy = entropy;
}

bool seq_tokens_equal(const Sequence &a, const Sequence &b) {
    if (a.tokens.size() != b.tokens.size()) return false;
    for (size_t i = 0; i < a.tokens.size(); ++i)
        if (a.tokens[i].id != b.tokens[i].id) return false;
    return true;
}

}  // namespace

int lang_auto_detect(Context &c, int offset_ms, float *lang_probs) {
    const Vocab &v = c.model.vocab;
    const int seek = offset_ms / 10;
    if (seek < 0) return -1;
    if (seek >= c.n_len_org) return -2;
    if (!v.is_multilingual()) return -3;
    encode(c, seek);
    kv_clear(c);
    const int tok = v.sot, pos = 0, slot = 0, row = 0;
    decode(c, &tok, &pos, &slot, 1, &row, 1);
    const int nl = std::min(N_LANGS, v.num_languages());
    std::vector<float> lg(nl);
    MW_HIP(hipMemcpyAsync(lg.data(), c.d_logits + v.lang_token(0), nl * sizeof(float), hipMemcpyDeviceToHost, c.st));
    MW_HIP(hipStreamSynchronize(c.st));
    std::vector<std::pair<double, int>> lid;
    for (int i = 0; i < nl; ++i) lid.emplace_back(lg[i], i);
    std::sort(lid.begin(), lid.end(),
              [](const std::pair<double, int> &a, const std::pair<double, int> &b) { return a.first > b.first; });
    const double mx = lid[0].first;
    double sum = 0;
    for (auto &kv : lid) {
        kv.first = exp(kv.first - mx);
        sum += kv.first;
    }
    for (auto &kv : lid) kv.first /= sum;
    if (lang_probs)
        for (auto &kv : lid) lang_probs[kv.second] = (float)kv.first;
    return lid[0].second;
}

// ------------------------------------------------------------------------------------------
// whisper_full (A.6)
// ------------------------------------------------------------------------------------------
int full(Context &c, whisper_full_params params, const float *samples, int n_samples, bool samples_on_device) {
    const Model &model = c.model;
    const Vocab &vocab = model.vocab;
    const HParams &hp = model.hp;
    const double t_start = now_ms();
    if (params.logits_filter_callback || params.grammar_rules) return -10;
    c.ps_armed = c.ps_fused = false;  // a call that failed between prepare_sample and process_and_sample leaves no armed rows behind
    auto &result_all = c.result_all;
    result_all.clear();
    if (n_samples > 0) pcm_to_mel(c, samples, n_samples, samples_on_device);
    if (c.d_mel == nullptr) return -2;
    std::string language = params.language ? params.language : "";
    if (language.empty() || language == "auto" || params.detect_language) {
        std::vector<float> probs(N_LANGS, 0.0f);
        const int id = lang_auto_detect(c, 0, probs.data());
        if (id < 0) {
            log_msg(LOG_ERROR, "whisper_full: failed to auto-detect language (%d)\n", id);
            return -3;
        }
        c.lang_id = id;
        language = lang_str(id);
        log_msg(LOG_INFO, "whisper_full: auto-detected language: %s (p = %f)\n", language.c_str(), probs[id]);
        if (params.detect_language) return 0;
    }
    const int seek_start = params.offset_ms / 10;
    const int seek_end = params.duration_ms == 0 ? c.n_len : seek_start + params.duration_ms / 10;
    if (seek_end < seek_start + 100) {
        log_msg(LOG_WARN, "whisper_full: input is too short - %d ms < 1000 ms. consider padding the input audio with silence\n",
                (seek_end - seek_start) * 10);
        return 0;
    }
    std::vector<float> temperatures;
    if (params.temperature_inc > 0.0f) {
        for (float t = params.temperature; t < 1.0f + 1e-6f; t += params.temperature_inc) temperatures.push_back(t);
    } else {
        temperatures.push_back(params.temperature);
    }
    int n_decoders = 1;
    if (params.strategy == WHISPER_SAMPLING_GREEDY) n_decoders = params.greedy.best_of;
    else n_decoders = std::max(params.greedy.best_of, params.beam_search.beam_size);
    n_decoders = std::max(1, n_decoders);
    if (n_decoders > MAX_DECODERS) {
        log_msg(LOG_ERROR, "whisper_full: too many decoders requested (%d), max = %d\n", n_decoders, MAX_DECODERS);
        return -4;
    }
    if (params.strategy == WHISPER_SAMPLING_BEAM_SEARCH && params.beam_search.beam_size > MAX_CAND) return -4;
    for (int j = 1; j < n_decoders; ++j) c.decoders[j].rng = std::mt19937(0);
    auto &prompt_past = c.prompt_past;
    if (params.no_context) prompt_past.clear();
    {
        std::vector<int> pt;
        if (!params.prompt_tokens && params.initial_prompt && params.initial_prompt[0]) {
            pt = vocab.tokenize(params.initial_prompt);
        } else if (params.prompt_tokens && params.prompt_n_tokens > 0) {
            pt.assign(params.prompt_tokens, params.prompt_tokens + params.prompt_n_tokens);
        }
        if (!pt.empty()) {
            for (int t : pt) prompt_past.push_back(t);
            std::rotate(prompt_past.begin(), prompt_past.end() - pt.size(), prompt_past.end());
        }
    }
    if (params.audio_ctx > hp.n_audio_ctx) {
        log_msg(LOG_ERROR, "whisper_full: audio_ctx is larger than the maximum allowed (%d > %d)\n", params.audio_ctx,
                hp.n_audio_ctx);
        return -5;
    }
    c.exp_n_audio_ctx = params.audio_ctx;
    std::vector<int> prompt_init = {vocab.sot};
    if (vocab.is_multilingual()) {
        const int id = lang_id(language.c_str());
        if (id < 0 || id >= vocab.num_languages()) {
            log_msg(LOG_ERROR, "whisper_full: unknown language '%s'\n", language.c_str());
            return -9;
        }
        c.lang_id = id;
        prompt_init.push_back(vocab.lang_token(id));
        prompt_init.push_back(params.translate ? vocab.translate : vocab.transcribe);
    }
    if (hp.n_text_layer == 2 && !params.no_timestamps) params.no_timestamps = true;  // distilled models
    if (params.no_timestamps) prompt_init.push_back(vocab.not_);
    upload_suppress_mask(c, params);
    const int fds = c.force_decode_steps;
    int seek = seek_start;
    std::vector<int> prompt;
    // a candidate = (source decoder, one of its sampled tokens); the sequence it stands for — the source's
    // sequence plus that token — is only materialised for the candidates that are kept (whisper.cpp copies the
    // sequence into every candidate: 55 vector copies per beam-5 step, ~20 us of host time between two steps)
    struct beam_candidate {
        int decoder_idx;
        int seek_delta;
        bool has_ts;
        whisper_token_data tk;
        double sum_logprobs_all;  // of the candidate sequence: the sort key
    };
    std::vector<Sequence> old_seq(n_decoders);
    std::vector<std::vector<beam_candidate>> bc_per_dec(n_decoders);
    std::vector<beam_candidate> beam_candidates;
    const bool beam = params.strategy == WHISPER_SAMPLING_BEAM_SEARCH;

    while (true) {
        if (params.progress_callback) {
            const int progress_cur = (100 * (seek - seek_start)) / (seek_end - seek_start);
            params.progress_callback(nullptr, nullptr, progress_cur, params.progress_callback_user_data);
        }
        if (seek + 100 >= seek_end) break;
        if (params.encoder_begin_callback &&
            !params.encoder_begin_callback(nullptr, nullptr, params.encoder_begin_callback_user_data))
            break;
        if (params.abort_callback && params.abort_callback(params.abort_callback_user_data)) return -6;
        encode(c, seek);
        if (seek > seek_start && seek + 500 >= seek_end) prompt_past.clear();
        int best_decoder_id = 0;
        for (int it = 0; it < (int)temperatures.size(); ++it) {
            const float t_cur = temperatures[it];
            int n_decoders_cur = 1;
            if (!beam) {
                if (t_cur > 0.0f) n_decoders_cur = params.greedy.best_of;
            } else {
                n_decoders_cur = t_cur > 0.0f ? params.greedy.best_of : params.beam_search.beam_size;
            }
            n_decoders_cur = std::max(1, n_decoders_cur);
            // number of draws the device sampler makes per decoder row for this temperature
            const int n_draw = beam ? params.beam_search.beam_size : (t_cur < 1e-6f ? 0 : 1);
            for (int j = 0; j < n_decoders_cur; ++j) {
                Decoder &d = c.decoders[j];
                d.sequence.tokens.clear();
                d.sequence.result_len = 0;
                d.sequence.sum_logprobs_all = 0.0;
                d.sequence.sum_logprobs = -INFINITY;
                d.sequence.avg_logprobs = -INFINITY;
                d.sequence.entropy = 0.0;
                d.sequence.score = -INFINITY;
                d.seek_delta = 100 * WHISPER_CHUNK_SIZE;
                d.failed = d.completed = d.has_ts = false;
            }
            {
                prompt.clear();
                if (!prompt_past.empty() && t_cur < 0.5f && params.n_max_text_ctx > 0) {
                    const int n_take = std::min(std::min(params.n_max_text_ctx, hp.n_text_ctx / 2), (int)prompt_past.size());
                    prompt = {vocab.prev};
                    prompt.insert(prompt.begin() + 1, prompt_past.end() - n_take, prompt_past.end());
                }
                prompt.insert(prompt.end(), prompt_init.begin(), prompt_init.end());
                kv_clear(c);
                std::vector<int> pos(prompt.size()), slot(prompt.size(), 0);
                for (size_t i = 0; i < prompt.size(); ++i) pos[i] = (int)i;
                const int last = (int)prompt.size() - 1;
                // every decoder starts from decoder 0's distribution (kv_seq_cp(0, j) == shared index table)
                std::vector<int> idx(n_decoders_cur), rows(n_decoders_cur, 0);
                for (int j = 0; j < n_decoders_cur; ++j) idx[j] = j;
                prepare_sample(c, params, t_cur, idx, rows, n_draw);
                decode(c, prompt.data(), pos.data(), slot.data(), (int)prompt.size(), &last, 1);
                process_and_sample(c, idx);
            }
            for (int i = 0, n_max = hp.n_text_ctx / 2 - 4; i < n_max; ++i) {
                if (beam)
                    for (auto &bc : bc_per_dec) bc.clear();
                for (int j = 0; j < n_decoders_cur; ++j) {
                    Decoder &d = c.decoders[j];
                    if (d.completed || d.failed) continue;
                    if (!beam) {
                        d.sequence.tokens.push_back(make_token(vocab, d.sampled, 0, 0));
                        d.sequence.sum_logprobs_all += d.sequence.tokens.back().plog;
                    } else {
                        for (int k = 0; k < params.beam_search.beam_size; ++k) {
                            const whisper_token_data tk = make_token(vocab, d.sampled, k, vocab.beg);
                            double sum_all = d.sequence.sum_logprobs_all;
                            sum_all += tk.plog;
                            bc_per_dec[j].push_back({j, d.seek_delta, d.has_ts, tk, sum_all});
                        }
                    }
                }
                int src[MAX_DECODERS];
                for (int j = 0; j < MAX_DECODERS; ++j) src[j] = -1;
                if (beam) {
                    beam_candidates.clear();
                    for (const auto &bc : bc_per_dec) beam_candidates.insert(beam_candidates.end(), bc.begin(), bc.end());
                    std::sort(beam_candidates.begin(), beam_candidates.end(),
                              [](const beam_candidate &a, const beam_candidate &b) {
                                  return a.sum_logprobs_all > b.sum_logprobs_all;
                              });
                    for (int j = 0; j < n_decoders_cur; ++j)  // the sources: every live decoder is overwritten below
                        if (!(c.decoders[j].completed || c.decoders[j].failed)) old_seq[j] = std::move(c.decoders[j].sequence);
                    auto same_tokens = [&](const beam_candidate &a, const beam_candidate &b) {
                        return a.tk.id == b.tk.id && seq_tokens_equal(old_seq[a.decoder_idx], old_seq[b.decoder_idx]);
                    };
                    uint32_t cur_c = 0;
                    for (int j = 0; j < n_decoders_cur; ++j) {
                        Decoder &d = c.decoders[j];
                        if (d.completed || d.failed) continue;
                        if (cur_c >= beam_candidates.size()) cur_c = 0;
                        auto &cur = beam_candidates[cur_c++];
                        while (beam_candidates.size() > cur_c && same_tokens(beam_candidates[cur_c], cur) && i > 0) ++cur_c;
                        d.seek_delta = cur.seek_delta;
                        d.has_ts = cur.has_ts;
                        d.sequence = old_seq[cur.decoder_idx];
                        d.sequence.tokens.push_back(cur.tk);
                        d.sequence.sum_logprobs_all = cur.sum_logprobs_all;
                        src[j] = cur.decoder_idx;
                    }
                } else {
                    for (int j = 0; j < n_decoders_cur; ++j)
                        if (!(c.decoders[j].completed || c.decoders[j].failed)) src[j] = j;
                }
                for (int j = 0; j < n_decoders_cur; ++j) {
                    Decoder &d = c.decoders[j];
                    if (d.completed || d.failed) continue;
                    auto &has_ts = d.has_ts;
                    auto &failed = d.failed;
                    auto &completed = d.completed;
                    auto &seek_delta = d.seek_delta;
                    auto &result_len = d.sequence.result_len;
                    {
                        const auto &token = d.sequence.tokens.back();
                        if (token.id > vocab.beg) {
                            const int seek_delta_new = 2 * (token.id - vocab.beg);
                            if (has_ts && seek_delta > seek_delta_new && result_len < i) {
                                failed = true;
                                continue;
                            }
                            seek_delta = seek_delta_new;
                            result_len = i + 1;
                            has_ts = true;
                        }
                        const bool forced = fds > 0 && i + 1 < fds;
                        if (!forced && (token.id == vocab.eot || (params.max_tokens > 0 && i >= params.max_tokens) ||
                                        (has_ts && seek + seek_delta + 100 >= seek_end) || (fds > 0 && i + 1 >= fds))) {
                            if (result_len == 0 && !params.no_timestamps) {
                                if (seek + seek_delta + 100 >= seek_end || fds > 0) {
                                    result_len = i + 1;
                                } else {
                                    failed = true;
                                    continue;
                                }
                            }
                            if (params.single_segment || params.no_timestamps) {
                                result_len = i + 1;
                                seek_delta = 100 * WHISPER_CHUNK_SIZE;
                            }
                            completed = true;
                            continue;
                        }
                    }
                    if (i == n_max - 1 && (result_len == 0 || seek_delta < 100 * WHISPER_CHUNK_SIZE / 2)) {
                        failed = true;
                        continue;
                    }
                }
                {
                    bool completed_all = true;
                    for (int j = 0; j < n_decoders_cur; ++j)
                        if (!(c.decoders[j].completed || c.decoders[j].failed)) completed_all = false;
                    if (completed_all) break;
                }
                {
                    std::vector<int> bt, bp, bs, rows, idx;
                    const int n_past = (int)prompt.size() + i;
                    if (n_past >= hp.n_text_ctx) return -8;
                    // sequences that stopped this step no longer need a cache line
                    for (int j = 0; j < n_decoders_cur; ++j)
                        if (c.decoders[j].failed || c.decoders[j].completed) src[j] = -1;
                    kv_reorder(c, src, n_past);
                    for (int j = 0; j < n_decoders_cur; ++j) {
                        Decoder &d = c.decoders[j];
                        if (d.failed || d.completed) continue;
                        d.i_batch = (int)bt.size();
                        rows.push_back(d.i_batch);
                        idx.push_back(j);
                        bt.push_back(d.sequence.tokens.back().id);
                        bp.push_back(n_past);
                        bs.push_back(j);
                    }
                    prepare_sample(c, params, t_cur, idx, rows, n_draw);
                    decode(c, bt.data(), bp.data(), bs.data(), (int)bt.size(), rows.data(), (int)rows.size());
                    process_and_sample(c, idx);
                }
            }
            {
                double best_score = -INFINITY;
                for (int j = 0; j < n_decoders_cur; ++j) {
                    Decoder &d = c.decoders[j];
                    if (d.failed) continue;
                    d.sequence.tokens.resize(d.sequence.result_len);
                    sequence_score(params, d.sequence);
                    if (d.sequence.result_len > 32 && d.sequence.entropy < params.entropy_thold) {
                        d.failed = true;
                        continue;
                    }
                    if (best_score < d.sequence.score) {
                        best_score = d.sequence.score;
                        best_decoder_id = j;
                    }
                }
            }
            bool success = true;
            if (it != (int)temperatures.size() - 1) {
                const Decoder &d = c.decoders[best_decoder_id];
                if (d.failed || d.sequence.avg_logprobs < params.logprob_thold) success = false;
            }
            if (success) break;
        }
        {
            const Decoder &best = c.decoders[best_decoder_id];
            const int seek_delta = best.seek_delta;
            const int result_len = best.sequence.result_len;
            const auto &tokens_cur = best.sequence.tokens;
            prompt_past.clear();
            if (!prompt.empty() && prompt.front() == vocab.prev)
                prompt_past.insert(prompt_past.end(), prompt.begin() + 1, prompt.end() - prompt_init.size());
            for (int i = 0; i < result_len && i < (int)tokens_cur.size(); ++i) prompt_past.push_back(tokens_cur[i].id);
            if (!tokens_cur.empty()) {
                int i0 = 0;
                int64_t t0 = seek + 2 * (tokens_cur.front().tid - vocab.beg);
                std::string text;
                for (int i = 0; i < (int)tokens_cur.size(); ++i) {
                    if (params.print_special || tokens_cur[i].id < vocab.eot) text += vocab.id_to_token[tokens_cur[i].id];
                    if (tokens_cur[i].id > vocab.beg && !params.single_segment) {
                        const int64_t t1 = seek + 2 * (tokens_cur[i].tid - vocab.beg);
                        if (!text.empty()) {
                            result_all.push_back({t0, t1, text, {}});
                            for (int j = i0; j <= i; ++j) result_all.back().tokens.push_back(tokens_cur[j]);
                            if (params.new_segment_callback)
                                params.new_segment_callback(nullptr, nullptr, 1, params.new_segment_callback_user_data);
                        }
                        text = "";
                        while (i < (int)tokens_cur.size() && tokens_cur[i].id > vocab.beg) i++;
                        i--;
                        t0 = t1;
                        i0 = i + 1;
                    }
                }
                if (!text.empty()) {
                    const int64_t t1 = seek + seek_delta;
                    result_all.push_back({t0, t1, text, {}});
                    for (int j = i0; j < (int)tokens_cur.size(); ++j) result_all.back().tokens.push_back(tokens_cur[j]);
                    if (params.new_segment_callback)
                        params.new_segment_callback(nullptr, nullptr, 1, params.new_segment_callback_user_data);
                }
            }
            // forced decode length (bench-only extension): the window always advances as a whole, so one call over
            // 30 s of audio is exactly one encoder pass + fds decode steps
            seek += fds > 0 ? 100 * WHISPER_CHUNK_SIZE : seek_delta;
        }
    }
    c.tm.total_ms += now_ms() - t_start;
    return 0;
}

}  // namespace mw
