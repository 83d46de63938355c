// model.hip — GGML-legacy file parser + one-time upload/convert into HBM (see model.hpp).
#include "model.hpp"

#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

#include <cmath>
#include <regex>

namespace mw {

namespace {

const char *const k_langs[N_LANGS] = {
    "en", "zh", "de", "es", "ru", "ko", "fr", "ja", "pt", "tr", "pl", "ca", "nl", "ar", "sv", "it", "id",
    "hi", "fi", "vi", "he", "uk", "el", "ms", "cs", "ro", "da", "hu", "ta", "no", "th", "ur", "hr", "bg",
    "lt", "la", "mi", "ml", "cy", "sk", "te", "fa", "lv", "bn", "sr", "az", "sl", "kn", "et", "mk", "br",
    "eu", "is", "hy", "ne", "mn", "bs", "kk", "sq", "sw", "gl", "mr", "pa", "si", "km", "sn", "yo", "so",
    "af", "oc", "ka", "be", "tg", "sd", "gu", "am", "yi", "lo", "uz", "fo", "ht", "ps", "tk", "nn", "mt",
    "sa", "lb", "my", "bo", "tl", "mg", "as", "tt", "haw", "ln", "ha", "ba", "jw", "su", "yue"};

constexpr uint32_t GGML_MAGIC = 0x67676d6c;
enum { TYPE_F32 = 0, TYPE_F16 = 1 };

struct Cursor {
    const unsigned char *p, *end;
    template <typename T> T rd() {
        if (p + sizeof(T) > end) throw std::runtime_error("model file truncated");
        T v;
        memcpy(&v, p, sizeof(T));
        p += sizeof(T);
        return v;
    }
    const unsigned char *take(size_t n) {
        if (n > (size_t)(end - p)) throw std::runtime_error("model file truncated");
        const unsigned char *q = p;
        p += n;
        return q;
    }
};

struct TensorRef {
    int type = 0;
    std::vector<int> ne;  // ggml order (ne[0] fastest)
    size_t n = 0;
    const unsigned char *data = nullptr;
};

// src element i -> dst[perm(i)]; conv weights [co][ci][3] are re-laid as [co][3][ci]
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void convert_kernel(const TS *__restrict__ src, TD *__restrict__ dst, size_t n,
                                                      int conv_ci, float scale) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float v;
        if constexpr (sizeof(TS) == 2) v = __half2float(*reinterpret_cast<const __half *>(&src[i]));
        else v = src[i];
        v *= scale;
        size_t o = i;
        if (conv_ci > 0) {
            const size_t per = (size_t)conv_ci * 3;
            const size_t co = i / per, rem = i % per, ci = rem / 3, k = rem % 3;
            o = co * per + k * conv_ci + ci;
        }
        if constexpr (sizeof(TD) == 2) {
            __hip_bfloat16 h = __float2bfloat16(v);
            dst[o] = *reinterpret_cast<bf16_t *>(&h);
        } else {
            dst[o] = v;
        }
    }
}

// GGML block-quantised tensor types (ggml.h of the whisper.cpp v1.6.2 era): 32 elements per block
//   q4_0 {f16 d; u8 qs[16]}            x = (nibble - 8) * d
//   q4_1 {f16 d; f16 m; u8 qs[16]}     x = nibble * d + m
//   q5_0 {f16 d; u8 qh[4]; u8 qs[16]}  x = ((nibble | bit << 4) - 16) * d
//   q5_1 {f16 d; f16 m; u8 qh[4]; u8 qs[16]}  x = (nibble | bit << 4) * d + m
//   q8_0 {f16 d; i8 qs[32]}            x = qs * d
// element j < 16 of a block takes the low nibble of qs[j], element j + 16 the high nibble; its fifth bit is bit j
// (resp. j + 16) of qh.  Dequantised once at load into the compute type: the engine has no quantised matmul.
enum { TYPE_Q4_0 = 2, TYPE_Q4_1 = 3, TYPE_Q5_0 = 6, TYPE_Q5_1 = 7, TYPE_Q8_0 = 8 };
inline int quant_block_bytes(int type) {
    switch (type) {
    case TYPE_Q4_0: return 18;
    case TYPE_Q4_1: return 20;
    case TYPE_Q5_0: return 22;
    case TYPE_Q5_1: return 24;
    case TYPE_Q8_0: return 34;
    default: return 0;
    }
}

template <typename TD>
__global__ __launch_bounds__(256) void dequant_kernel(const unsigned char *__restrict__ src, TD *__restrict__ dst, size_t n,
                                                      int type, int block_bytes) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const unsigned char *b = src + (i >> 5) * block_bytes;
        const int j = (int)(i & 31);
        const float d = __half2float(*reinterpret_cast<const __half *>(b));
        float v;
        if (type == TYPE_Q8_0) {
            v = d * (float)(signed char)b[2 + j];
        } else {
            const bool has_m = type == TYPE_Q4_1 || type == TYPE_Q5_1, has_h = type == TYPE_Q5_0 || type == TYPE_Q5_1;
            const unsigned char *q = b + 2 + (has_m ? 2 : 0);
            int x;
            if (has_h) {
                const unsigned qh = (unsigned)q[0] | ((unsigned)q[1] << 8) | ((unsigned)q[2] << 16) | ((unsigned)q[3] << 24);
                const unsigned char byte = q[4 + (j & 15)];
                x = (j < 16 ? (byte & 0x0F) : (byte >> 4)) | (int)(((qh >> j) & 1u) << 4);
            } else {
                const unsigned char byte = q[j & 15];
                x = j < 16 ? (byte & 0x0F) : (byte >> 4);
            }
            if (has_m) v = (float)x * d + __half2float(*reinterpret_cast<const __half *>(b + 2));
            else v = (float)(x - (has_h ? 16 : 8)) * d;
        }
        if constexpr (sizeof(TD) == 2) {
            __hip_bfloat16 h = __float2bfloat16(v);
            dst[i] = *reinterpret_cast<bf16_t *>(&h);
        } else {
            dst[i] = v;
        }
    }
}

// fp8 mode: one workgroup per output row.  scale = the power of two that brings the row's largest magnitude under 448 (the
// e4m3 maximum); bytes = RNE conversion by the hardware (v_cvt_pk_fp8_f32, the inverse of the decode the kernels use);
// the bf16 row is overwritten with byte x scale (exact: 4 significant bits times a power of two).
__global__ __launch_bounds__(256) void quantize_rows_fp8_kernel(bf16_t *__restrict__ w, unsigned char *__restrict__ w8,
                                                                float *__restrict__ scale, int K) {
    __shared__ float red[256];
    bf16_t *row = w + (size_t)blockIdx.x * K;
    unsigned char *row8 = w8 + (size_t)blockIdx.x * K;
    float mx = 0.0f;
    for (int k = threadIdx.x; k < K; k += 256) mx = fmaxf(mx, fabsf(__uint_as_float((unsigned)row[k] << 16)));
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    mx = red[0];
    int e = 0;
    if (mx > 0.0f) {
        (void)frexpf(mx / 448.0f, &e);                       // mx / 448 = f x 2^e, f in [0.5, 1)
        if (ldexpf(1.0f, e - 1) * 448.0f >= mx) e -= 1;      // exact power of two: the smaller exponent is enough
    }
    e = max(-100, min(100, e));
    const float sc = ldexpf(1.0f, e), inv = ldexpf(1.0f, -e);
    if (threadIdx.x == 0) scale[blockIdx.x] = sc;
    for (int k = 2 * threadIdx.x; k < K; k += 512) {  // K is even (multiple of 32)
        const float a = __uint_as_float((unsigned)row[k] << 16) * inv, b = __uint_as_float((unsigned)row[k + 1] << 16) * inv;
        const int pk = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);  // bytes 0, 1 of the result
        row8[k] = (unsigned char)(pk & 0xff);
        row8[k + 1] = (unsigned char)((pk >> 8) & 0xff);
        const auto back = __builtin_amdgcn_cvt_pk_f32_fp8(pk, false);
        row[k] = (bf16_t)(__float_as_uint(back[0] * sc) >> 16);
        row[k + 1] = (bf16_t)(__float_as_uint(back[1] * sc) >> 16);
    }
}

struct Loader {
    Model &m;
    std::map<std::string, TensorRef> dir;
    hipStream_t st;
    void *staging = nullptr;
    size_t staging_size = 0;

    void *alloc(size_t bytes) {
        const size_t off = (m.arena_used + 255) & ~(size_t)255;
        if (off + bytes > m.arena_size) throw std::runtime_error("model arena overflow");
        m.arena_used = off + bytes;
        return m.arena + off;
    }
    const TensorRef &get(const std::string &name, size_t expect_n) {
        auto it = dir.find(name);
        if (it == dir.end()) throw std::runtime_error("model file: missing tensor " + name);
        if (it->second.n != expect_n) throw std::runtime_error("model file: wrong size for tensor " + name);
        return it->second;
    }
    // copy tensor `name` (n elements) converted to `dt` at dst (+ element offset)
    void put(const std::string &name, size_t n, void *dst, DType dt, int conv_ci = 0) {
        const TensorRef &t = get(name, n);
        const int qb = quant_block_bytes(t.type);
        const size_t bytes = qb ? n / 32 * qb : n * (t.type == TYPE_F16 ? 2 : 4);
        MW_CHECK(bytes <= staging_size, "staging too small");
        MW_HIP(hipMemcpyAsync(staging, t.data, bytes, hipMemcpyHostToDevice, st));
        dim3 grid((unsigned)std::min<size_t>(ceil_div<size_t>(n, 256), 8192));
        if (qb) {
            MW_CHECK(conv_ci == 0, "quantised convolution weights are not a GGML whisper layout");
            if (dt == DType::BF16)
                hipLaunchKernelGGL(dequant_kernel<bf16_t>, grid, dim3(256), 0, st, (const unsigned char *)staging, (bf16_t *)dst, n,
                                   t.type, qb);
            else
                hipLaunchKernelGGL(dequant_kernel<float>, grid, dim3(256), 0, st, (const unsigned char *)staging, (float *)dst, n,
                                   t.type, qb);
        } else if (t.type == TYPE_F16) {
            if (dt == DType::BF16)
                hipLaunchKernelGGL((convert_kernel<uint16_t, bf16_t>), grid, dim3(256), 0, st, (const uint16_t *)staging,
                                   (bf16_t *)dst, n, conv_ci, 1.0f);
            else
                hipLaunchKernelGGL((convert_kernel<uint16_t, float>), grid, dim3(256), 0, st, (const uint16_t *)staging,
                                   (float *)dst, n, conv_ci, 1.0f);
        } else {
            if (dt == DType::BF16)
                hipLaunchKernelGGL((convert_kernel<float, bf16_t>), grid, dim3(256), 0, st, (const float *)staging,
                                   (bf16_t *)dst, n, conv_ci, 1.0f);
            else
                hipLaunchKernelGGL((convert_kernel<float, float>), grid, dim3(256), 0, st, (const float *)staging,
                                   (float *)dst, n, conv_ci, 1.0f);
        }
        MW_HIP(hipGetLastError());
        // the staging buffer is reused by the next tensor: wait for this conversion first
        MW_HIP(hipStreamSynchronize(st));
        m.n_params += n;
    }
    void *mat(const std::string &name, size_t n, int conv_ci = 0) {
        void *p = alloc(n * dtype_size(m.wt));
        put(name, n, p, m.wt, conv_ci);
        return p;
    }
    // fp8 mode: e4m3 copy + row scales of an uploaded bf16 matrix [N][K] (the bf16 copy becomes the dequantised values)
    void quantize(void *w_bf16, size_t N, size_t K, unsigned char *&w8, float *&scale) {
        w8 = (unsigned char *)alloc(N * K);
        scale = (float *)alloc(N * 4);
        hipLaunchKernelGGL(quantize_rows_fp8_kernel, dim3((unsigned)N), dim3(256), 0, st, (bf16_t *)w_bf16, w8, scale, (int)K);
        MW_HIP(hipGetLastError());
    }
    // MX-fp8 copy (e4m3 bytes + block scales in the GEMM's lane order) of an uploaded bf16 matrix [N][K]; the bf16 copy stays
    void quantize_mx(const void *w_bf16, size_t N, size_t K, unsigned char *&w8, unsigned char *&s8) {
        w8 = (unsigned char *)alloc(N * K);
        s8 = (unsigned char *)alloc(mx_scale_bytes_w((int)N, (int)K));
        launch_quantize_mx_weights(w_bf16, (int)N, (int)K, w8, s8, st);
    }
    float *vec(const std::string &name, size_t n) {
        float *p = (float *)alloc(n * 4);
        put(name, n, p, DType::F32);
        return p;
    }
    float *zeros(size_t n) {
        float *p = (float *)alloc(n * 4);
        MW_HIP(hipMemsetAsync(p, 0, n * 4, st));
        return p;
    }
};

void build_mel_tables(const Model &m, MelTables &t) {
    memset(&t, 0, sizeof(t));
    for (int i = 0; i < MEL_N_FFT; ++i) {
        const double th = (2.0 * M_PI * i) / MEL_N_FFT;
        t.tw_cos[i] = (float)cos(th);
        t.tw_sin[i] = (float)sin(th);
        t.hann[i] = (float)(0.5 * (1.0 - cos(th)));
    }
    const int n_mel = m.hp.n_mels;
    t.n_mel = n_mel;
    int voff = 0;
    for (int j = 0; j < n_mel; ++j) {
        const float *f = &m.mel_filters[(size_t)j * MEL_N_FREQ];
        int lo = -1, hi = -1;
        for (int k = 0; k < MEL_N_FREQ - 1; ++k)
            if (f[k] != 0.0f) {
                if (lo < 0) lo = k;
                hi = k;
            }
        t.last[j] = f[MEL_N_FREQ - 1];
        if (lo < 0) {
            t.k0[j] = 0;
            t.ngrp[j] = 0;
            t.voff[j] = voff;
            continue;
        }
        const int k0 = (lo / 4) * 4, k1 = (hi / 4) * 4 + 4;  // whole groups of 4 taps, as the reference sums them
        t.k0[j] = k0;
        t.ngrp[j] = (k1 - k0) / 4;
        t.voff[j] = voff;
        if (voff + (k1 - k0) > MEL_MAX_VALS) throw std::runtime_error("mel filterbank has too many taps");
        for (int k = k0; k < k1; ++k) t.vals[voff++] = f[k];
    }
}

}  // namespace

const char *lang_str(int id) { return (id >= 0 && id < N_LANGS) ? k_langs[id] : nullptr; }
int lang_id(const char *code) {
    if (!code) return -1;
    for (int i = 0; i < N_LANGS; ++i)
        if (strcmp(code, k_langs[i]) == 0) return i;
    return -1;
}

// whisper.cpp tokenize(): GPT-2 style word split, then greedy longest-match against the vocabulary
std::vector<int> Vocab::tokenize(const std::string &text) const {
    std::vector<std::string> words;
    {
        std::string str = text;
        static const std::regex re(
            R"('s|'t|'re|'ve|'m|'ll|'d| ?[[:alpha:]]+| ?[[:digit:]]+| ?[^\s[:alpha:][:digit:]]+|\s+(?!\S)|\s+)");
        std::smatch mm;
        while (std::regex_search(str, mm, re)) {
            for (auto x : mm) words.push_back(x);
            str = mm.suffix();
        }
    }
    std::vector<int> tokens;
    for (const auto &word : words) {
        const int n = (int)word.size();
        int i = 0;
        while (i < n) {
            int j = n;
            bool found = false;
            for (; j > i; --j) {
                auto it = token_to_id.find(word.substr(i, j - i));
                if (it != token_to_id.end()) {
                    tokens.push_back(it->second);
                    i = j;
                    found = true;
                    break;
                }
            }
            if (!found) ++i;
        }
    }
    return tokens;
}

namespace {
void free_device_block(void *p) {
    if (p) (void)hipFree(p);
}
}  // namespace

void load_model(Model &m, const unsigned char *data, size_t size, DType wt, hipStream_t st, int fp8) {
    if (fp8 && wt != DType::BF16) throw std::runtime_error("fp8 weights go with bf16 activations");
    Cursor c{data, data + size};
    if (c.rd<uint32_t>() != GGML_MAGIC) throw std::runtime_error("bad magic (not a ggml model file)");
    int32_t h[11];
    for (int i = 0; i < 11; ++i) h[i] = c.rd<int32_t>();
    HParams &hp = m.hp;
    hp = {h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9], h[10] % 1000};
    if (hp.n_vocab < 51864 || hp.n_vocab > 60000 || hp.n_audio_ctx <= 0 || hp.n_audio_ctx > 1500 ||
        hp.n_audio_state <= 0 || hp.n_audio_state % 64 || hp.n_audio_head * 64 != hp.n_audio_state ||
        hp.n_text_state != hp.n_audio_state || hp.n_text_head * 64 != hp.n_text_state || hp.n_audio_layer <= 0 ||
        hp.n_text_layer <= 0 || hp.n_text_ctx <= 0 || hp.n_text_ctx > 448 || hp.n_mels <= 0 ||
        hp.n_mels > MEL_MAX_BINS || hp.n_mels % 8 || hp.n_audio_state > 1280)
        throw std::runtime_error("unsupported model hyper-parameters");
    m.wt = wt;
    // mel filters
    const int fm = c.rd<int32_t>(), ff = c.rd<int32_t>();
    if (fm != hp.n_mels || ff != MEL_N_FREQ) throw std::runtime_error("unexpected mel filter shape");
    m.mel_filters.resize((size_t)fm * ff);
    memcpy(m.mel_filters.data(), c.take(m.mel_filters.size() * 4), m.mel_filters.size() * 4);
    // vocabulary
    Vocab &v = m.vocab;
    v.n_vocab = hp.n_vocab;
    const int n_file = c.rd<int32_t>();
    if (n_file < 0 || n_file > hp.n_vocab) throw std::runtime_error("bad vocabulary size");
    v.id_to_token.assign(hp.n_vocab, std::string());
    for (int i = 0; i < n_file; ++i) {
        const uint32_t len = c.rd<uint32_t>();
        const unsigned char *s = c.take(len);
        v.id_to_token[i].assign((const char *)s, len);
        v.token_to_id[v.id_to_token[i]] = i;
    }
    if (v.is_multilingual()) {
        v.eot++;
        v.sot++;
        const int dt = v.num_languages() - 98;
        v.translate += dt; v.transcribe += dt; v.solm += dt; v.prev += dt; v.nosp += dt; v.not_ += dt; v.beg += dt;
    }
    for (int i = n_file; i < hp.n_vocab; ++i) {
        std::string w;
        if (i > v.beg) w = "[_TT_" + std::to_string(i - v.beg) + "]";
        else if (i == v.eot) w = "[_EOT_]";
        else if (i == v.sot) w = "[_SOT_]";
        else if (i == v.translate) w = "[_TRANSLATE_]";
        else if (i == v.transcribe) w = "[_TRANSCRIBE_]";
        else if (i == v.solm) w = "[_SOLM_]";
        else if (i == v.prev) w = "[_PREV_]";
        else if (i == v.nosp) w = "[_NOSP_]";
        else if (i == v.not_) w = "[_NOT_]";
        else if (i == v.beg) w = "[_BEG_]";
        else if (i > v.sot && i <= v.sot + v.num_languages() && lang_str(i - v.sot - 1))
            w = std::string("[_LANG_") + lang_str(i - v.sot - 1) + "]";
        else w = "[_extra_token_" + std::to_string(i) + "]";
        v.id_to_token[i] = w;
        v.token_to_id[w] = i;
    }
    // tensor directory
    Loader L{m, {}, st};
    size_t max_bytes = 0;
    while (c.p < c.end) {
        const int32_t n_dims = c.rd<int32_t>(), name_len = c.rd<int32_t>(), ttype = c.rd<int32_t>();
        if (n_dims < 1 || n_dims > 4 || name_len <= 0 || name_len > 256) throw std::runtime_error("bad tensor header");
        TensorRef t;
        t.type = ttype;
        t.n = 1;
        for (int i = 0; i < n_dims; ++i) {
            const int32_t ne = c.rd<int32_t>();
            if (ne <= 0) throw std::runtime_error("bad tensor shape");
            t.ne.push_back(ne);
            t.n *= (size_t)ne;
        }
        std::string name((const char *)c.take(name_len), name_len);
        const int qb = quant_block_bytes(ttype);
        if (ttype != TYPE_F32 && ttype != TYPE_F16 && qb == 0)
            throw std::runtime_error("tensor " + name + ": unsupported GGML tensor type " + std::to_string(ttype) +
                                     " (f32, f16, q4_0, q4_1, q5_0, q5_1, q8_0 are read)");
        if (qb && (t.ne[0] % 32 != 0)) throw std::runtime_error("tensor " + name + ": quantised row length is not a multiple of 32");
        const size_t bytes = qb ? t.n / 32 * qb : t.n * (ttype == TYPE_F16 ? 2 : 4);
        t.data = c.take(bytes);
        max_bytes = std::max(max_bytes, bytes);
        L.dir[name] = std::move(t);
    }
    const size_t d = hp.n_audio_state, nm = hp.n_mels, ws = dtype_size(wt);
    const size_t Le = hp.n_audio_layer, Ld = hp.n_text_layer;
    // arena size: every matrix in wt, every vector in f32, 256-byte padding per allocation
    size_t mat_elems = d * nm * 3 + d * d * 3 + Le * 12 * d * d + (size_t)hp.n_vocab * d + Ld * 20 * d * d;
    size_t vec_elems = (size_t)hp.n_audio_ctx * d + (size_t)hp.n_text_ctx * d + 2 * d + Le * (2 * d + d + d + 4 * d + d + 4 * d) +
                       Ld * (3 * d + d + d + d + 4 * d + d + 6 * d + 2 * d) + 4 * d + 64;
    m.arena_size = mat_elems * ws + vec_elems * 4 + (64 + 16 * (Le + Ld)) * 256 + (1 << 20);
    if (fp8)  // e4m3 copies + row scales of the decoder projections and the logits matrix: per layer Wq|Wk|Wv (3 d^2), the two
              // attention outputs and cross-q (3 d^2), fc1 and fc2 (8 d^2) = 14 d^2 bytes and 3 + 3 + 4 + 1 = 11 d scales
        m.arena_size += Ld * 14 * d * d + (size_t)hp.n_vocab * d + (Ld * 11 * d + hp.n_vocab) * 4 + (4 + 12 * Ld) * 256;
    m.dec_fp8 = fp8 != 0;
    m.enc_mx = fp8 >= 2;
    if (m.enc_mx && d % 256 != 0) throw std::runtime_error("MX-fp8 encoder: n_audio_state must be a multiple of 256");
    if (m.enc_mx)  // Q|K|V (3 d^2), attention output (d^2), fc1 and fc2 (8 d^2) as bytes + one scale byte per 32
        m.arena_size += Le * (12 * d * d + 12 * d * d / 32 + 8 * 256) + Ld * (2 * d * d + 2 * d * d / 32) + 2 * 256;
    MW_HIP(hipMalloc((void **)&m.arena, m.arena_size));
    m.arena_owner = std::shared_ptr<void>(m.arena, free_device_block);
    L.staging_size = max_bytes;
    MW_HIP(hipMalloc(&L.staging, max_bytes));
    try {
        m.enc_pe = L.vec("encoder.positional_embedding", (size_t)hp.n_audio_ctx * d);
        m.conv1_w = L.mat("encoder.conv1.weight", d * nm * 3, (int)nm);
        m.conv1_b = L.vec("encoder.conv1.bias", d);
        m.conv2_w = L.mat("encoder.conv2.weight", d * d * 3, (int)d);
        m.conv2_b = L.vec("encoder.conv2.bias", d);
        m.enc.resize(Le);
        for (size_t i = 0; i < Le; ++i) {
            const std::string p = "encoder.blocks." + std::to_string(i) + ".";
            EncLayer &e = m.enc[i];
            e.ln1_w = L.vec(p + "attn_ln.weight", d);
            e.ln1_b = L.vec(p + "attn_ln.bias", d);
            e.qk_w = L.alloc(3 * d * d * ws);  // Q | K | V rows, contiguous: the bf16 path reads [2d][d] + [d][d], MX-fp8 all [3d][d]
            L.put(p + "attn.query.weight", d * d, e.qk_w, wt);
            L.put(p + "attn.key.weight", d * d, (char *)e.qk_w + d * d * ws, wt);
            e.v_w = (char *)e.qk_w + 2 * d * d * ws;
            L.put(p + "attn.value.weight", d * d, e.v_w, wt);
            e.qk_b = L.zeros(3 * d);
            L.put(p + "attn.query.bias", d, e.qk_b, DType::F32);
            e.v_b = e.qk_b + 2 * d;
            L.put(p + "attn.value.bias", d, e.v_b, DType::F32);
            e.out_w = L.mat(p + "attn.out.weight", d * d);
            e.out_b = L.vec(p + "attn.out.bias", d);
            e.ln2_w = L.vec(p + "mlp_ln.weight", d);
            e.ln2_b = L.vec(p + "mlp_ln.bias", d);
            e.fc1_w = L.mat(p + "mlp.0.weight", 4 * d * d);
            e.fc1_b = L.vec(p + "mlp.0.bias", 4 * d);
            e.fc2_w = L.mat(p + "mlp.2.weight", 4 * d * d);
            e.fc2_b = L.vec(p + "mlp.2.bias", d);
            if (m.enc_mx) {
                L.quantize_mx(e.qk_w, 3 * d, d, e.qk_w8, e.qk_s8);  // Q | K | V
                L.quantize_mx(e.out_w, d, d, e.out_w8, e.out_s8);
                L.quantize_mx(e.fc1_w, 4 * d, d, e.fc1_w8, e.fc1_s8);
                L.quantize_mx(e.fc2_w, d, 4 * d, e.fc2_w8, e.fc2_s8);
            }
        }
        m.ln_post_w = L.vec("encoder.ln_post.weight", d);
        m.ln_post_b = L.vec("encoder.ln_post.bias", d);
        m.dec_pe = L.vec("decoder.positional_embedding", (size_t)hp.n_text_ctx * d);
        m.tok_emb = L.mat("decoder.token_embedding.weight", (size_t)hp.n_vocab * d);
        m.dec.resize(Ld);
        m.ckv_w = L.alloc(Ld * 2 * d * d * ws);
        m.ckv_b = L.zeros(Ld * 2 * d);
        for (size_t i = 0; i < Ld; ++i) {
            const std::string p = "decoder.blocks." + std::to_string(i) + ".";
            DecLayer &e = m.dec[i];
            e.ln1_w = L.vec(p + "attn_ln.weight", d);
            e.ln1_b = L.vec(p + "attn_ln.bias", d);
            e.qkv_w = L.alloc(3 * d * d * ws);
            L.put(p + "attn.query.weight", d * d, e.qkv_w, wt);
            L.put(p + "attn.key.weight", d * d, (char *)e.qkv_w + d * d * ws, wt);
            L.put(p + "attn.value.weight", d * d, (char *)e.qkv_w + 2 * d * d * ws, wt);
            e.qkv_b = L.zeros(3 * d);
            L.put(p + "attn.query.bias", d, e.qkv_b, DType::F32);
            L.put(p + "attn.value.bias", d, e.qkv_b + 2 * d, DType::F32);
            e.out_w = L.mat(p + "attn.out.weight", d * d);
            e.out_b = L.vec(p + "attn.out.bias", d);
            e.lnc_w = L.vec(p + "cross_attn_ln.weight", d);
            e.lnc_b = L.vec(p + "cross_attn_ln.bias", d);
            e.cq_w = L.mat(p + "cross_attn.query.weight", d * d);
            e.cq_b = L.vec(p + "cross_attn.query.bias", d);
            L.put(p + "cross_attn.key.weight", d * d, (char *)m.ckv_w + (i * 2 + 0) * d * d * ws, wt);
            L.put(p + "cross_attn.value.weight", d * d, (char *)m.ckv_w + (i * 2 + 1) * d * d * ws, wt);
            L.put(p + "cross_attn.value.bias", d, m.ckv_b + (i * 2 + 1) * d, DType::F32);
            e.co_w = L.mat(p + "cross_attn.out.weight", d * d);
            e.co_b = L.vec(p + "cross_attn.out.bias", d);
            e.ln2_w = L.vec(p + "mlp_ln.weight", d);
            e.ln2_b = L.vec(p + "mlp_ln.bias", d);
            e.fc1_w = L.mat(p + "mlp.0.weight", 4 * d * d);
            e.fc1_b = L.vec(p + "mlp.0.bias", 4 * d);
            e.fc2_w = L.mat(p + "mlp.2.weight", 4 * d * d);
            e.fc2_b = L.vec(p + "mlp.2.bias", d);
            if (fp8) {
                L.quantize(e.qkv_w, 3 * d, d, e.qkv_w8, e.qkv_s);
                L.quantize(e.out_w, d, d, e.out_w8, e.out_s);
                L.quantize(e.cq_w, d, d, e.cq_w8, e.cq_s);
                L.quantize(e.co_w, d, d, e.co_w8, e.co_s);
                L.quantize(e.fc1_w, 4 * d, d, e.fc1_w8, e.fc1_s);
                L.quantize(e.fc2_w, d, 4 * d, e.fc2_w8, e.fc2_s);
            }
        }
        if (fp8) L.quantize(m.tok_emb, (size_t)hp.n_vocab, d, m.tok_emb8, m.tok_emb_s);
        if (m.enc_mx) {  // the cross K / V projections read the encoder's output: MX-fp8 as well, one scale array per matrix
            const size_t sb = mx_scale_bytes_w((int)d, (int)d);
            m.ckv_w8 = (unsigned char *)L.alloc(Ld * 2 * d * d);
            m.ckv_s8 = (unsigned char *)L.alloc(Ld * 2 * sb);
            for (size_t i = 0; i < Ld * 2; ++i)
                launch_quantize_mx_weights((const char *)m.ckv_w + i * d * d * ws, (int)d, (int)d, m.ckv_w8 + i * d * d, m.ckv_s8 + i * sb, st);
        }
        m.dec_ln_w = L.vec("decoder.ln.weight", d);
        m.dec_ln_b = L.vec("decoder.ln.bias", d);
        MelTables tab;
        build_mel_tables(m, tab);
        MW_HIP(hipMalloc((void **)&m.d_mel_tab, sizeof(MelTables)));
        m.mel_tab_owner = std::shared_ptr<void>(m.d_mel_tab, free_device_block);
        MW_HIP(hipMemcpyAsync(m.d_mel_tab, &tab, sizeof(tab), hipMemcpyHostToDevice, st));
        MW_HIP(hipStreamSynchronize(st));
    } catch (...) {
        (void)hipFree(L.staging);
        throw;
    }
    (void)hipFree(L.staging);
}

}  // namespace mw
