// kernels.hpp — host-side launchers of the hand-written gfx950 kernels (definitions in *.hip).
// Every launcher validates the shapes its kernel and grid assume before launching.
#pragma once

#include "common.hpp"

#include <algorithm>

namespace mw {

// ------------------------------------------------------------------------------------------
// K1 log-mel (mel.hip)
// ------------------------------------------------------------------------------------------
constexpr int MEL_MAX_BINS = 128;
constexpr int MEL_N_FFT = 400;
constexpr int MEL_HOP = 160;
constexpr int MEL_N_FREQ = 201;
constexpr int MEL_MAX_VALS = 2048;  // packed filter taps (groups of 4, zero-padded)

// Filter bank + trig tables as the kernel consumes them (built once per model on the host).
struct MelTables {
    float tw_cos[MEL_N_FFT];
    float tw_sin[MEL_N_FFT];
    float hann[MEL_N_FFT];
    int n_mel;
    int k0[MEL_MAX_BINS];      // first frequency bin of the bin's first tap group (multiple of 4)
    int ngrp[MEL_MAX_BINS];    // number of 4-tap groups
    int voff[MEL_MAX_BINS];    // offset of the bin's taps in vals[]
    float last[MEL_MAX_BINS];  // tap at the Nyquist bin (k = 200), summed separately like the reference
    float vals[MEL_MAX_VALS];
};

// raw[s][frame][bin] = log10(max(mel power, 1e-10)) for frame < n_active, gmax[s] = ordered-int max
void launch_mel_frames(const float *d_pcm, long pcm_stride, int n_samples, int n_streams, const MelTables *d_tab,
                       float *d_raw, long raw_stride, int n_active, int n_mel, unsigned *d_gmax, hipStream_t st);
// out[s][frame][bin] = (max(raw, gmax-8) + 4)/4 for frame < n_active, the clamped floor value beyond
void launch_mel_normalize(const float *d_raw, long raw_stride, float *d_out, long out_stride, int n_len, int n_active,
                          int n_mel, int n_streams, const unsigned *d_gmax, hipStream_t st);
// conv-stem operand: dst[s][1 + i][bin] = mel[s][seek + i][bin] (0 beyond n_len), rows 0 and n_rows+1 zero
void launch_mel_window(const float *d_mel, long mel_stride, int n_len, int seek, int n_rows, int n_mel, int n_streams,
                       void *d_dst, long dst_stride, DType dt, hipStream_t st);
// allocation-free log-mel (mw_pcm_to_mel_noalloc): intermediate rows + maximum word live in a file-scope __device__ array
// of mel.hip; at most one 30 s window (+ LocalVocal's 2 x 320-sample margins) per call
constexpr int MEL_NOALLOC_MAX_SAMPLES = 30 * 16000 + 640;
float *mel_noalloc_init(int device);  // device must be current; returns the array's device address on it
void mel_noalloc(const float *d_pcm, int n_samples, const MelTables *d_tab, int n_mel, float *d_out, size_t out_floats,
                 int *n_len_out, int device, hipStream_t st);
// whisper layout for the getter: out[bin][frame] <- in[frame][bin]
void launch_transpose_f32(const float *d_in, float *d_out, int rows, int cols, hipStream_t st);

// ------------------------------------------------------------------------------------------
// K2/K4/K6 GEMM  C[m][n] = act(alpha * sum_k A[m][k] W[n][k] + bias[n]) + R[m][n]   (gemm.hip)
// ------------------------------------------------------------------------------------------
enum { ACT_NONE = 0, ACT_GELU = 1 };

struct GemmArgs {
    const void *A = nullptr;  // [M][K], row stride lda (elements)
    const void *W = nullptr;  // [N][K], row stride ldw
    void *C = nullptr;        // [M][N] (or [N][M] when trans_c), row stride ldc
    const float *bias = nullptr;
    const float *R = nullptr;  // residual (f32), row stride ldr; may alias C when C is f32
    long lda = 0, ldw = 0, ldc = 0, ldr = 0;
    int M = 0, N = 0, K = 0;
    // batching: blockIdx.z = z0 * nz1 + z1
    int nz0 = 1, nz1 = 1;
    long sA0 = 0, sA1 = 0, sW0 = 0, sW1 = 0, sC0 = 0, sC1 = 0, sB0 = 0, sB1 = 0, sR0 = 0, sR1 = 0;
    float alpha = 1.0f;
    int act = ACT_NONE;
    int trans_c = 0;
    int hb = 0;  // set by launch_gemm: M-tiles per super-row of the XCD-aware tile order (0: tile = blockIdx)
    int epi_vec = 0;                  // set by launch_gemm: shapes / alignment allow the vector epilogue
    const float *epi_zero = nullptr;  // set by launch_gemm: 16 zero bytes on the device (bias stand-in)
    int glds = 0;  // bf16: stage operands with global_load_lds instead of the register ring (Context::gemm_glds)
};
// in: element type of A and W; out: element type of C
void launch_gemm(const GemmArgs &a, DType in, DType out, hipStream_t st);

// ------------------------------------------------------------------------------------------
// MX-fp8 encoder GEMMs (gemm_mx.hip; MW_PRECISION_FP8_MX): OCP e4m3 elements with one E8M0 scale (2^(s - 127)) per block of
// 32 consecutive k, multiplied by v_mfma_scale_f32_16x16x128_f8f6f4.
//   scale(block) = the smallest power of two with absmax / scale <= 448;  element = RNE(value / scale)  (never saturates)
// Elements are plain row-major bytes [rows][K].  Scales are stored in the order the kernel's lanes consume them (one
// 8-byte / 4-byte record per lane and k-step of 128): see mx_sa_off / mx_sw_off.
// ------------------------------------------------------------------------------------------
// activations [M][K]: record = the 8 row tiles (16 rows each) of a 128-row block for one (row & 15, k-block & 3)
__host__ __device__ inline size_t mx_sa_off(int m, int kb, int MB) {
    return (((size_t)(kb >> 2) * MB + (m >> 7)) * 64 + ((kb & 3) * 16 + (m & 15))) * 8 + ((m >> 4) & 7);
}
// weights [N][K]: record = the 4 row tiles of a 64-row block
__host__ __device__ inline size_t mx_sw_off(int n, int kb, int NB) {
    return (((size_t)(kb >> 2) * NB + (n >> 6)) * 64 + ((kb & 3) * 16 + (n & 15))) * 4 + ((n >> 4) & 3);
}
// E8M0 byte of the block scale for a block whose largest magnitude is amax (finite, >= 0): the smallest power of two s
// with amax / s <= 448 = 1.75 x 2^8, clamped at byte 0
__host__ __device__ inline unsigned mx_scale_byte(float amax) {
    const unsigned u = __builtin_bit_cast(unsigned, amax);
    const int e = (int)((u >> 23) & 255u) - ((u & 0x7fffffu) > 0x600000u ? 7 : 8);
    return (unsigned)(e > 0 ? e : 0);
}
inline int mx_row_blocks_a(int M) { return 2 * ((M + 255) / 256); }                       // 128-row blocks, padded to whole 256-row tiles
inline size_t mx_scale_bytes_a(int M, int K) { return (size_t)(K / 128) * mx_row_blocks_a(M) * 512; }
inline size_t mx_scale_bytes_w(int N, int K) { return (size_t)(K / 128) * (N / 64) * 256; }
// bf16 [N][K] (row stride K) -> e4m3 bytes + scales in weight order; K % 128 == 0, N % 64 == 0
void launch_quantize_mx_weights(const void *w_bf16, int N, int K, unsigned char *q, unsigned char *s, hipStream_t st);
// fp32 [M][K] -> e4m3 bytes + scales in activation order (parity entry point; the encoder's producers quantise in place)
void launch_quantize_mx_rows(const float *x, long ldx, int M, int K, unsigned char *q, long ldq, unsigned char *s, hipStream_t st);
void launch_quantize_mx_rows_bf16(const void *x_bf16, long ldx, int M, int K, unsigned char *q, long ldq, unsigned char *s, hipStream_t st);
// LayerNorm (ggml_norm, as launch_layernorm) whose result leaves as MX-fp8 (q, s) and, when out_bf16 is given, as bf16 too
void launch_layernorm_mx(const float *x, long ldx, const float *w, const float *b, void *out_bf16, long ldo, unsigned char *q,
                         long ldq, unsigned char *s, int rows, int d, hipStream_t st);
enum { MXE_BF16 = 0, MXE_F32_RES = 1, MXE_MX_GELU = 2 };
struct GemmMxArgs {
    const unsigned char *A = nullptr, *SA = nullptr;  // [M][K] e4m3, scales in activation order
    const unsigned char *W = nullptr, *SW = nullptr;  // [N][K] e4m3, scales in weight order
    void *C = nullptr;               // MXE_BF16: bf16 [M][N]; MXE_F32_RES: fp32 [M][N] (+= R); MXE_MX_GELU: e4m3 [M][N]
    unsigned char *SC = nullptr;     // MXE_MX_GELU: the output's scales, activation order for K' = N
    const float *bias = nullptr, *R = nullptr;
    long lda = 0, ldw = 0, ldc = 0, ldr = 0;
    int M = 0, N = 0, K = 0;
    float alpha = 1.0f;
    // batch (MXE_BF16 only): member z multiplies the same A with W + z sW (bytes), scales SW + z sSW (bytes), bias + z sBias
    // (floats), into C + z sC (elements) — the cross K / V of all decoder layers in one launch
    int nz = 1;
    long sW = 0, sSW = 0, sC = 0, sBias = 0;
    const float *epi_zero = nullptr;  // set by the launcher
};
// C = epilogue(alpha * A W^T + bias): 256 x 256 persistent tiles, K % 128 == 0, N % 256 == 0
void launch_gemm_mx(const GemmMxArgs &a, int epi, hipStream_t st);
const float *epi_zero_device();  // gemm.hip: 16 zero bytes on the current device

// ------------------------------------------------------------------------------------------
// K3 LayerNorm, softmax, misc (enc_misc.hip)
// ------------------------------------------------------------------------------------------
void launch_layernorm(const float *x, long ldx, const float *w, const float *b, void *out, long ldo, DType out_t,
                      int rows, int d, hipStream_t st);
// rows x n scaled softmax of f32 scores, written as out_t with columns [n, n_pad) zeroed
// fused encoder self-attention, bf16: qk = [n_ctx][2d] (Q | K), vt = V transposed [d][tp] (pad columns finite),
// out = [n_ctx][d]; heads of 64
// n_windows > 1: the windows of a batched encoder pass, slabs back to back (qk [w][n_ctx][2d], vt [w][d][tp], out [w][n_ctx][d])
// ldq: row stride of qk in elements (0: 2 d)
// out_mx: the output as MX-fp8 rows [n_windows * n_ctx][d] with their scales in A-operand order (gemm_mx.hip) instead of `out`
void launch_enc_flash_attn(const void *qk, const void *vt, void *out, int n_ctx, int tp, int d, int H, hipStream_t st,
                           int n_windows = 1, int ldq = 0, unsigned char *out_mx = nullptr, unsigned char *out_mx_scales = nullptr);
// vt[w][n][t] = src[(w n_ctx + t) lds + n] for t < n_ctx, 0 for n_ctx <= t < tp  (bf16; the V columns of a Q|K|V product ->
// the transposed per-window layout the fused attention reads); d % 64 == 0, tp % 64 == 0
void launch_transpose_v(const void *src, long lds_, void *vt, int n_ctx, int tp, int d, int n_windows, hipStream_t st);
void launch_softmax_rows(const float *s, long lds_, void *out, long ldo, DType out_t, long rows, int n, int n_pad,
                         float scale, hipStream_t st);
void launch_fill_zero(void *p, size_t bytes, hipStream_t st);
void launch_convert(const float *src, void *dst, DType dt, size_t n, hipStream_t st);

// ------------------------------------------------------------------------------------------
// K7/K8 decoder step (dec.hip)
// ------------------------------------------------------------------------------------------
constexpr int DEC_MAX_ROWS = 8;      // activation rows one weight pass serves
constexpr int KV_SLOTS = 8;          // physical self-KV slots per stream (= WHISPER_MAX_DECODERS)
constexpr int CROSS_SPLITS_MAX = 16;
// floats per partial softmax record of the decode cross-attention (dec_tile.hip: 66 used, padded to whole 128-byte lines;
// dec.hip packs 66): what the d_part buffers are sized with
constexpr int CROSS_PART_STRIDE = 96;

// A batched step: up to STEP_MAX_ROWS rows that ride in M of one 16-row MFMA tile, taken from up to STEP_MAX_GROUPS
// requests (one request = the rows of one stream context: its beams, or the tokens of a short prompt).  Everything that
// differs per row or per request travels in a StepBlock in device memory, so one captured graph serves any mix of
// contexts with the same (rows, groups, logits rows) shape.
constexpr int STEP_MAX_ROWS = 32;
constexpr int STEP_MAX_GROUPS = 32;
// rows of ONE request in a step (the beams of a stream, or a chunk of a prompt): what the cross-attention kernel, whose
// workgroups serve one request each, is instantiated for
constexpr int STEP_MAX_REQ_ROWS = 16;
struct StepRow {
    bf16_t *kself, *vself;  // self K / V cache of the row's context, layer 0: [L][KV_SLOTS][n_text_ctx][d]
    const int *kv_idx;      // the context's current index table [KV_SLOTS][n_text_ctx]
    int slot, pos, token, group;
};
struct StepGroup {
    const bf16_t *cross;    // cross K | V of the request's context: [L][2][n_audio_ctx][d]
    const int *idx_old;     // beam reshuffle of the index table (reorder_flag): new[j][p] = old[src[j]][p], new[j][n_past] = j
    int *idx_new;
    int row0, nrows, n_ctx;
    int reorder_flag, reorder_n_past;
    int src[KV_SLOTS];
    int pad;
};
// per decoder row, uploaded per step (rules + samplers, logits.hip)
constexpr int MAX_CAND = 8;
struct LogitsRowState {
    int logits_row;       // row of the logits matrix
    int n_sampled;        // tokens sampled so far (0 => initial)
    int last_is_ts, penult_is_ts;
    int has_ts, seek_delta;
    double u[MAX_CAND];   // uniform draws for the samplers
};
struct StepBlock {  // pinned host memory -> device memory by the step's first kernel
    StepRow rows[STEP_MAX_ROWS];
    StepGroup groups[STEP_MAX_GROUPS];
    int logit_rows[STEP_MAX_ROWS];
    LogitsRowState rs[STEP_MAX_ROWS];
    int n_rows, n_groups, pad[2];
};

// x[i][:] = tok_emb[token[i]][:] + pos_emb[pos[i]][:]
void launch_embed(const int *d_tokens, const int *d_pos, const void *tok_emb, DType wt, const float *pos_emb,
                  float *x, int n, int d, hipStream_t st);

// projection classes of the decode step (each is a distinct kernel symbol)
enum { GEMV_QKV = 0, GEMV_ATTN_OUT = 1, GEMV_CROSS_Q = 2, GEMV_FC1 = 3, GEMV_FC2 = 4, GEMV_LOGITS = 5, GEMV_N_TAGS = 6 };

struct GemvArgs {
    int tag = GEMV_ATTN_OUT;
    const float *x = nullptr;  // [n][K] f32 activations (row stride ldx)
    long ldx = 0;
    int n = 0;                 // rows (any count; processed DEC_MAX_ROWS at a time)
    const void *W = nullptr;   // [N][K]
    const float *bias = nullptr;
    int N = 0, K = 0;
    const float *ln_w = nullptr, *ln_b = nullptr;  // when set: LayerNorm(x) is applied first
    int act = ACT_NONE;
    float out_scale = 1.0f;    // applied to (x W^T + b) before the activation
    float *out = nullptr;      // [n][N] f32 (row stride ldo); out = (R ? R : 0) + act(out_scale * (x W^T + b))
    long ldo = 0;
    const float *R = nullptr;  // residual, same layout as out (may alias out)
    // fused self-attention projection epilogue (N = 3d): columns [0,d) -> out (scaled by qk_scale),
    // [d,2d) -> K cache (scaled), [2d,3d) -> V cache, at (slot[i], pos[i])
    int qkv = 0;
    float qk_scale = 1.0f;
    void *kcache = nullptr, *vcache = nullptr;  // [KV_SLOTS][n_text_ctx][d] of this layer
    DType kv_t = DType::BF16;
    const int *d_slot = nullptr, *d_pos = nullptr;
    int n_text_ctx = 0;
    const int *d_rows = nullptr;  // optional row gather: use x row d_rows[i] for output row i
    // bf16 matrix-core tile kernels only (dec_tile.hip): activations already bf16 (residual classes) / bf16 output
    int x_bf16 = 0, out_bf16 = 0;
    // fp8 mode, tile kernels of the batched step only: W again as OCP e4m3 bytes [N][K] + one power-of-two scale per row
    const void *W8 = nullptr;
    const float *w_scale = nullptr;
};
void launch_gemv(const GemvArgs &a, DType wt, hipStream_t st);

// self-attention of n query rows over the indirected cache: row i attends positions 0..pos[i] of
// logical sequence slot[i]; position p of sequence s lives in physical slot kv_idx[s][p].
void launch_self_attn(const float *q, const void *kcache, const void *vcache, DType kv_t, const int *d_slot,
                      const int *d_pos, const int *d_kv_idx, void *out, DType out_t, int n, int H, int d, int n_text_ctx,
                      hipStream_t st);
// cross-attention of n query rows over the stream's encoder K/V [n_ctx][d]; split over keys
void launch_cross_attn(const float *q, const void *kc, const void *vc, DType kv_t, float *partial, void *out, DType out_t,
                       int n, int H, int d, int n_ctx, int splits, hipStream_t st);
// new_idx[j][p] = old_idx[src[j]][p] for p < n_past; new_idx[j][n_past] = j   (beam reshuffle)
struct BeamSrc {
    int src[KV_SLOTS];
};
void launch_kv_reorder(const int *old_idx, int *new_idx, BeamSrc src, int n_past, int n_text_ctx, hipStream_t st);
// same update with its arguments in device memory: args = src[KV_SLOTS], n_past, flag (0: nothing to do) — graph-replayable
void launch_kv_reorder_dev(const int *old_idx, int *new_idx, const int *args, int n_text_ctx, hipStream_t st);
// head of a decode call in one launch: step block from pinned host memory (device view) -> index-table update, token
// embedding of n_rows rows, and the block's copy into device memory (block layout: engine.hip decode())
void launch_step_begin(const int *h_block_dev, int *d_block, int n_ints, int n_text_ctx, int args_off, int n_rows, const int *old_idx,
                       int *new_idx, const void *tok_emb, DType wt, const float *pos_emb, float *x, int d, hipStream_t st);

// ------------------------------------------------------------------------------------------
// the beam step's matrix-core tile kernels (dec_tile.hip): bf16 weights, <= 5 rows
// ------------------------------------------------------------------------------------------
constexpr int TILE_THREADS = 512;
constexpr int TILE_TC_MAX = 448;         // text positions one self-attention pass covers
// workgroup-wide attention kernels for the <= 5-row step (bf16 K/V): every load of a (row, head) or (head, split)
// item is issued up front; the cross kernel's last split per head combines the partial softmaxes itself
// (head_cnt: H zeroed counters, left zeroed)
// logits projection (final LayerNorm + tied embedding) on matrix-core tiles: bf16 weights, <= 5 gathered rows
// ---- batched step (StepBlock-driven; bf16, <= STEP_MAX_ROWS rows) ----
// head of the step: block from pinned host memory (device view) -> per-request index-table updates, token embeddings of
// the rows into x, and the block's copy into device memory
void launch_step_begin_batch(const StepBlock *h_block_dev, StepBlock *d_block, int n_rows, int n_groups, int n_text_ctx,
                             const void *tok_emb, const float *pos_emb, float *x, int d, int *row_phys, hipStream_t st);
// GemvArgs contract of launch_proj_wg for n <= STEP_MAX_ROWS rows; the qkv class writes K / V through blk->rows[i] + kv_layer_off
// (element offset of the layer inside a context's cache)
void launch_proj_batch(const GemvArgs &a, const StepBlock *blk, long kv_layer_off, hipStream_t st);
// row_phys [STEP_MAX_ROWS][TILE_TC_MAX]: every row's physical KV slot per position, written by launch_step_begin_batch
void launch_self_attn_batch(const float *q, const StepBlock *blk, const int *row_phys, long kv_layer_off, void *out, int n, int H, int d,
                            int n_text_ctx, hipStream_t st);
// cross-attention of every request's rows over its own context's K / V (element offsets of the layer's K and V inside
// a context's cross buffer); head_cnt: n_groups * H zeroed counters, left zeroed
void launch_cross_touch_batch(const StepBlock *blk, long k_off, long v_off, int n_groups, int H, int d, int splits, int *sink, hipStream_t st);
void launch_cross_attn_batch(const float *q, const StepBlock *blk, long k_off, long v_off, float *partial, void *out,
                             unsigned *head_cnt, int n_rows, int max_group_rows, int n_groups, int H, int d, int n_ctx_max, int splits,
                             hipStream_t st);
bool logits_tile_supported(const GemvArgs &a);
void launch_logits_tile(const GemvArgs &a, hipStream_t st);
void launch_self_attn_wg(const float *q, const void *kcache, const void *vcache, const int *d_slot, const int *d_pos,
                         const int *d_kv_idx, void *out, DType out_t, int n, int H, int d, int n_text_ctx, hipStream_t st);
void launch_cross_attn_wg(const float *q, const void *kc, const void *vc, float *partial, void *out, DType out_t, unsigned *head_cnt,
                          int n, int H, int d, int n_ctx, int splits, hipStream_t st);
// one decode projection on the matrix cores, workgroup = 16-feature tile (bf16 weights, <= 5 rows; direct classes
// take bf16 rows, fc1 writes bf16 rows)
void launch_proj_wg(const GemvArgs &a, hipStream_t st);

// ------------------------------------------------------------------------------------------
// K9/K10 logits rules + samplers (logits.hip)
// ------------------------------------------------------------------------------------------
struct LogitsParams {
    int n_vocab, eot, beg, space_id;
    int suppress_blank;
    float temperature;
    int tid0_initial;  // round(max_initial_ts / 0.02), <0: rule disabled
    int n_samples;     // 0: greedy argmax ("best"); k>0: draw k samples from the distribution
};
struct LogitsRowResult {
    int tid;
    float pt, ptsum;
    int id[MAX_CAND];
    float p[MAX_CAND], plog[MAX_CAND];
};
// suppress: per-token byte mask (1 = always suppressed).  Optional full-row dumps for parity tests.
// work: process_logits_work_floats(max rows, n_vocab) floats (slab records + the masked rows).
size_t process_logits_work_floats(int n_rows_max, int n_vocab);
void launch_process_logits(const float *logits, long ld, const unsigned char *suppress, const LogitsParams &p,
                           const LogitsRowState *d_rows, LogitsRowResult *d_res, int n_rows, float *work,
                           float *dump_logits, float *dump_logprobs, float *dump_probs, hipStream_t st);

}  // namespace mw
