// model.hpp — GGML-legacy Whisper model files -> HBM-resident weights (K13).
//
// Replaces the model loading whisper.cpp does inside whisper_init_from_file_with_params /
// whisper_init_from_buffer_with_params, called from init_whisper_context
// (/root/reference/src/whisper-utils/whisper-processing.cpp:109,112).  The file is parsed once; every
// tensor is converted to the engine's compute type (bf16, or fp32 in exact mode), re-laid out where a
// kernel wants it (Q|K|V rows concatenated, conv taps k-major, cross K|V of all layers contiguous)
// and uploaded into one device arena.  File layout: SURVEY.md Appendix A.1.
#pragma once

#include "kernels.hpp"

#include <map>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

namespace mw {

struct HParams {
    int n_vocab = 0, n_audio_ctx = 0, n_audio_state = 0, n_audio_head = 0, n_audio_layer = 0;
    int n_text_ctx = 0, n_text_state = 0, n_text_head = 0, n_text_layer = 0, n_mels = 0, ftype = 0;
};

struct Vocab {
    int n_vocab = 0;
    std::vector<std::string> id_to_token;
    std::unordered_map<std::string, int> token_to_id;
    int eot = 50256, sot = 50257, translate = 50357, transcribe = 50358, solm = 50359, prev = 50360, nosp = 50361,
        not_ = 50362, beg = 50363;
    bool is_multilingual() const { return n_vocab >= 51865; }
    int num_languages() const { return n_vocab - 51765 - (is_multilingual() ? 1 : 0); }
    int lang_token(int lang_id) const { return sot + 1 + lang_id; }
    std::vector<int> tokenize(const std::string &text) const;
};

constexpr int N_LANGS = 100;
const char *lang_str(int id);  // nullptr when out of range
int lang_id(const char *code); // -1 when unknown

struct EncLayer {
    float *ln1_w, *ln1_b, *ln2_w, *ln2_b;
    void *qk_w;   // [2d][d]  (query rows, then key rows; the value rows follow: v_w = qk_w + 2 d d)
    float *qk_b;  // [2d]     (key half zero: whisper has no key bias; v_b = qk_b + 2 d)
    void *v_w;
    float *v_b;
    void *out_w;
    float *out_b;
    void *fc1_w;
    float *fc1_b;
    void *fc2_w;
    float *fc2_b;
    // MX-fp8 mode (Model::enc_mx): Q|K|V (qk_w8: [3d][d]), the attention output, fc1 and fc2 again as e4m3 bytes + E8M0 block scales in the order the MX GEMM's lanes
    // read them (kernels.hpp, mx_sw_off)
    unsigned char *qk_w8 = nullptr, *qk_s8 = nullptr, *out_w8 = nullptr, *out_s8 = nullptr, *fc1_w8 = nullptr, *fc1_s8 = nullptr, *fc2_w8 = nullptr,
                  *fc2_s8 = nullptr;
};

struct DecLayer {
    float *ln1_w, *ln1_b, *lnc_w, *lnc_b, *ln2_w, *ln2_b;
    void *qkv_w;   // [3d][d]
    float *qkv_b;  // [3d] (key third zero)
    void *out_w;
    float *out_b;
    void *cq_w;
    float *cq_b;
    void *co_w;
    float *co_b;
    void *fc1_w;
    float *fc1_b;
    void *fc2_w;
    float *fc2_b;
    // fp8 mode (Model::dec_fp8): the same matrices as e4m3 bytes + one power-of-two scale per output row
    unsigned char *qkv_w8 = nullptr, *out_w8 = nullptr, *cq_w8 = nullptr, *co_w8 = nullptr, *fc1_w8 = nullptr, *fc2_w8 = nullptr;
    float *qkv_s = nullptr, *out_s = nullptr, *cq_s = nullptr, *co_s = nullptr, *fc1_s = nullptr, *fc2_s = nullptr;
};

struct Hub;  // engine.hpp: the per-GPU batcher of decode steps over the contexts that share these weights

struct Model {
    HParams hp;
    Vocab vocab;
    DType wt = DType::BF16;  // element type of matrix weights, activations fed to MFMA, and KV caches
    std::vector<float> mel_filters;  // host copy [n_mel][201]
    MelTables *d_mel_tab = nullptr;
    void *conv1_w = nullptr;  // [d][3*n_mel], tap-major: column = k*n_mel + c_in
    float *conv1_b = nullptr;
    void *conv2_w = nullptr;  // [d][3*d]
    float *conv2_b = nullptr;
    float *enc_pe = nullptr;  // [n_audio_ctx][d] f32
    std::vector<EncLayer> enc;
    float *ln_post_w = nullptr, *ln_post_b = nullptr;
    float *dec_pe = nullptr;   // [n_text_ctx][d] f32
    void *tok_emb = nullptr;   // [n_vocab][d]
    std::vector<DecLayer> dec;
    void *ckv_w = nullptr;   // [L][2][d][d]  cross-attention key | value weights of every decoder layer
    float *ckv_b = nullptr;  // [L][2][d]     (key halves zero)
    float *dec_ln_w = nullptr, *dec_ln_b = nullptr;
    // fp8 weights (MW_PRECISION_FP8): every decoder projection and the logits matrix also live as OCP e4m3 bytes with one
    // power-of-two fp32 scale per output row (scale = 2^ceil(log2(absmax / 448)), value = e4m3 x scale); the decode tile
    // kernels stream those (half the bytes).  The bf16 copies are REPLACED by the dequantised values — exact in bf16 because
    // the scale is a power of two — so prompt prefill (bf16 kernels) and the steps compute with one and the same model.
    // The encoder keeps bf16 weights (its GEMMs are not bound by weight bytes).
    bool dec_fp8 = false;
    // MW_PRECISION_FP8_MX: additionally the encoder's Q|K|V, attention-output, fc1 and fc2 products and the cross K / V projections run in MX-fp8 (gemm_mx.hip): e4m3 weights
    // AND activations with one power-of-two scale per block of 32 along k, on v_mfma_scale_f32_16x16x128_f8f6f4.  The
    // bf16 copies stay resident (what the MX weights were quantised from; the conv stem stays bf16).  Needs n_audio_state % 256 == 0.
    bool enc_mx = false;
    unsigned char *ckv_w8 = nullptr, *ckv_s8 = nullptr;  // enc_mx: the cross K | V weights [L][2][d][d] as e4m3 + block scales per matrix
    unsigned char *tok_emb8 = nullptr;
    float *tok_emb_s = nullptr;
    // device arena
    char *arena = nullptr;
    size_t arena_size = 0, arena_used = 0;
    size_t n_params = 0;
    // the device allocations (arena, mel tables) are reference-counted: a copy of a Model is a view of the same
    // HBM-resident weights (one upload per GPU, any number of stream contexts: mw_init_shared)
    std::shared_ptr<void> arena_owner, mel_tab_owner;
    // contexts that share the weights also share the hub that batches their decode steps (null: no batched path)
    std::shared_ptr<Hub> hub;
};

// Parses `size` bytes at `data` (a whole ggml-*.bin image) and uploads to the current HIP device.
// Throws std::runtime_error on malformed input.
// fp8 (with wt = BF16): 1 = additionally quantise the decoder's matrices to e4m3 (Model::dec_fp8); 2 = that and the
// encoder's MX-fp8 weights (Model::enc_mx).
void load_model(Model &m, const unsigned char *data, size_t size, DType wt, hipStream_t st, int fp8 = 0);

}  // namespace mw
