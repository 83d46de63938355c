// whisper_oracle.cpp — CPU ORACLE (test infrastructure only; see whisper_oracle.h for scope,
// provenance and the "parity unpinned" statement).
//
// Pure fp32 restatement of the whisper.cpp v1.6.2 algorithm LocalVocal calls through
// whisper_full() (/root/reference/src/whisper-utils/whisper-processing.cpp:218-219).  Each block
// cites the SURVEY.md Appendix-A item (and the whisper.cpp routine) it follows.
#include "whisper_oracle.h"

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <random>
#include <regex>
#include <string>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

constexpr int SAMPLE_RATE = 16000;
constexpr int N_FFT = 400;
constexpr int HOP = 160;
constexpr int CHUNK_SIZE = 30;
constexpr int MAX_DECODERS = 8;

// ---------------------------------------------------------------------------------------------
// language table (id order = OpenAI Whisper LANGUAGES; whisper.cpp g_lang)
// ---------------------------------------------------------------------------------------------
const char *const k_langs[] = {
    "en", "zh", "de", "es", "ru", "ko", "fr", "ja", "pt", "tr", "pl", "ca", "nl", "ar", "sv", "it", "id",
    "hi", "fi", "vi", "he", "uk", "el", "ms", "cs", "ro", "da", "hu", "ta", "no", "th", "ur", "hr", "bg",
    "lt", "la", "mi", "ml", "cy", "sk", "te", "fa", "lv", "bn", "sr", "az", "sl", "kn", "et", "mk", "br",
    "eu", "is", "hy", "ne", "mn", "bs", "kk", "sq", "sw", "gl", "mr", "pa", "si", "km", "sn", "yo", "so",
    "af", "oc", "ka", "be", "tg", "sd", "gu", "am", "yi", "lo", "uz", "fo", "ht", "ps", "tk", "nn", "mt",
    "sa", "lb", "my", "bo", "tl", "mg", "as", "tt", "haw", "ln", "ha", "ba", "jw", "su", "yue"};
constexpr int N_LANGS = sizeof(k_langs) / sizeof(k_langs[0]);

// whisper.cpp non_speech_tokens (OpenAI tokenizer.py non_speech_tokens)
const char *const k_non_speech[] = {
    "\"", "#", "(", ")", "*", "+", "/", ":", ";", "<", "=", ">", "@", "[", "\\", "]", "^", "_", "`", "{", "|", "}",
    "~", "「", "」", "『", "』", "<<", ">>", "<<<", ">>>", "--", "---", "-(", "-[", "('", "(\"", "((", "))", "(((",
    ")))", "[[", "]]", "{{", "}}", "♪♪", "♪♪♪", "♩", "♪", "♫", "♬", "♭", "♮", "♯"};

struct Tensor {
    std::vector<int> ne;  // ggml order, ne[0] fastest
    std::vector<float> d;
};

struct Vocab {
    int n_vocab = 51864;
    std::map<std::string, int> token_to_id;
    std::vector<std::string> id_to_token;
    int eot = 50256, sot = 50257, translate = 50357, transcribe = 50358, solm = 50359, prev = 50360, nosp = 50361,
        not_ = 50362, beg = 50363;
    bool is_multilingual() const { return n_vocab >= 51865; }
    int num_languages() const { return n_vocab - 51765 - (is_multilingual() ? 1 : 0); }
};

struct HP {
    int n_vocab, n_audio_ctx, n_audio_state, n_audio_head, n_audio_layer, n_text_ctx, n_text_state, n_text_head,
        n_text_layer, n_mels, ftype;
};

struct Model {
    HP hp;
    int filt_n_mel = 0, filt_n_fft = 0;
    std::vector<float> filters;
    Vocab vocab;
    std::map<std::string, Tensor> t;
    const std::vector<float> &get(const std::string &n) const {
        auto it = t.find(n);
        if (it == t.end()) {
            fprintf(stderr, "oracle: missing tensor %s\n", n.c_str());
            abort();
        }
        return it->second.d;
    }
};

struct Mel {
    int n_len = 0, n_len_org = 0, n_mel = 0;
    std::vector<float> data;
};

struct Sequence {
    std::vector<ow_token_data> tokens;
    int result_len = 0;
    double sum_logprobs_all = 0, sum_logprobs = 0, avg_logprobs = 0, entropy = 0, score = 0;
};

struct Decoder {
    Sequence sequence;
    int seek_delta = 0;
    bool failed = false, completed = false, has_ts = false;
    std::vector<float> probs, logits, logprobs;
    int i_batch = 0;
    std::mt19937 rng;
};

struct Segment {
    int64_t t0, t1;
    std::string text;
    std::vector<ow_token_data> tokens;
};

struct KVSeq {  // one sequence's self-attention cache: [layer][pos][d]
    std::vector<float> k, v;
    int n = 0;
};

float half_to_float(uint16_t h) {
    uint32_t sign = (h >> 15) & 1, exp = (h >> 10) & 0x1f, man = h & 0x3ff, f;
    if (exp == 0) {
        if (man == 0) {
            f = sign << 31;
        } else {
            exp = 127 - 15 + 1;
            while ((man & 0x400) == 0) {
                man <<= 1;
                exp--;
            }
            man &= 0x3ff;
            f = (sign << 31) | (exp << 23) | (man << 13);
        }
    } else if (exp == 31) {
        f = (sign << 31) | 0x7f800000u | (man << 13);
    } else {
        f = (sign << 31) | ((exp - 15 + 127) << 23) | (man << 13);
    }
    float r;
    memcpy(&r, &f, 4);
    return r;
}

// ---------------------------------------------------------------------------------------------
// small dense kernels (fp32, deterministic summation order independent of thread count)
// ---------------------------------------------------------------------------------------------
inline float dot_f32(const float *a, const float *b, int n) {
    float acc[16] = {0};
    int i = 0;
    for (; i + 16 <= n; i += 16)
        for (int j = 0; j < 16; ++j) acc[j] += a[i + j] * b[i + j];
    float s = 0;
    for (int j = 0; j < 16; ++j) s += acc[j];
    for (; i < n; ++i) s += a[i] * b[i];
    return s;
}

// C[m][n] = sum_k A[m][k] W[n][k] (+ bias[n]);  rows strided by lda / ldw / ldc
void matmul_nt(float *C, int ldc, const float *A, int lda, const float *W, int ldw, const float *bias, int M, int N,
               int K) {
#pragma omp parallel for schedule(static) collapse(2)
    for (int m0 = 0; m0 < M; m0 += 4)
        for (int n0 = 0; n0 < N; n0 += 64) {
            int m1 = std::min(M, m0 + 4), n1 = std::min(N, n0 + 64);
            for (int n = n0; n < n1; ++n) {
                const float *w = W + (size_t)n * ldw;
                float b = bias ? bias[n] : 0.0f;
                for (int m = m0; m < m1; ++m) C[(size_t)m * ldc + n] = dot_f32(A + (size_t)m * lda, w, K) + b;
            }
        }
}

void layer_norm(float *out, const float *x, const float *w, const float *b, int rows, int d) {
    // ggml_norm (eps = 1e-5), then * w + b  (A.3)
#pragma omp parallel for schedule(static)
    for (int r = 0; r < rows; ++r) {
        const float *xr = x + (size_t)r * d;
        float *o = out + (size_t)r * d;
        double sum = 0;
        for (int i = 0; i < d; ++i) sum += xr[i];
        float mean = (float)(sum / d);
        double sum2 = 0;
        for (int i = 0; i < d; ++i) {
            float v = xr[i] - mean;
            o[i] = v;
            sum2 += (double)v * v;
        }
        float var = (float)(sum2 / d);
        float scale = 1.0f / sqrtf(var + 1e-5f);
        for (int i = 0; i < d; ++i) o[i] = o[i] * scale * w[i] + b[i];
    }
}

// ---------------------------------------------------------------------------------------------
// OCP microscaling fp8 (MX-fp8, e4m3 elements, one power-of-two scale per block of 32), quantise + dequantise.
// No counterpart in whisper.cpp: this is the arithmetic the engine's MW_PRECISION_FP8_MX mode declares
// (obs-localvocal_amd/csrc/gemm_mx.hip), restated so that the mode can be tested against the same numbers.
//   scale = the smallest power of two s with absmax(block) / s <= 448 (clamped at 2^-127);
//   element = the e4m3 value nearest to value / s, ties to even (subnormals 2^-9 .. 7 x 2^-9 included).
// ---------------------------------------------------------------------------------------------
inline float e4m3_nearest(float x) {  // |x| <= 448
    const float a = fabsf(x);
    if (a == 0.0f) return x;
    int e;
    (void)frexpf(a, &e);                 // a = f x 2^e, f in [0.5, 1): binade 2^(e-1)
    const int q = std::max(e - 1, -6);   // subnormals share the spacing of the lowest binade
    const float step = ldexpf(1.0f, q - 3);
    return copysignf(nearbyintf(a / step) * step, x);  // default rounding mode: to nearest, ties to even
}
void mx_qdq_rows(float *a, size_t rows, int K, size_t ld) {
#pragma omp parallel for schedule(static)
    for (long r = 0; r < (long)rows; ++r)
        for (int k0 = 0; k0 < K; k0 += 32) {
            float *b = a + (size_t)r * ld + k0;
            float amax = 0.0f;
            for (int i = 0; i < 32; ++i) amax = std::max(amax, fabsf(b[i]));
            if (amax == 0.0f) continue;
            int e;
            (void)frexpf(amax / 448.0f, &e);                // amax / 448 = f x 2^e, f in [0.5, 1)
            if (ldexpf(448.0f, e - 1) >= amax) e -= 1;      // exactly a power of two times 448: the smaller scale is enough
            e = std::max(e, -127);
            const float s = ldexpf(1.0f, e), inv = ldexpf(1.0f, -e);
            for (int i = 0; i < 32; ++i) b[i] = e4m3_nearest(b[i] * inv) * s;
        }
}

inline float gelu_tanh(float x) {
    // ggml_gelu: 0.5*x*(1 + tanh(sqrt(2/pi)*x*(1 + 0.044715*x*x)))   (A.3)
    const float c = 0.79788456080286535587989211986876f, a = 0.044715f;
    return 0.5f * x * (1.0f + tanhf(c * x * (1.0f + a * x * x)));
}

}  // namespace

struct ow_context {
    Model model;
    Mel mel;
    int exp_n_audio_ctx = 0;
    int n_ctx_cur = 0;
    std::vector<float> enc_out;                 // [n_ctx][d]
    std::vector<std::vector<float>> cross_k;    // [layer] -> [n_ctx][d] (pre-scaled by dh^-0.25)
    std::vector<std::vector<float>> cross_v;    // [layer] -> [n_ctx][d]
    std::vector<KVSeq> kv;                      // 2*MAX_DECODERS sequences
    std::vector<float> logits;                  // [n][n_vocab] of the last decode
    int n_logits_rows = 0;
    Decoder decoders[MAX_DECODERS];
    std::vector<Segment> result_all;
    std::vector<int> prompt_past;
    int lang_id = 0;
    int n_decode_calls = 0, n_encode_calls = 0;
    std::string tmp_str;
    // MX-fp8 twin of the engine's MW_PRECISION_FP8_MX encoder (ow_set_mx_encoder): the operands of the Q, K, V, attention-output,
    // fc1 and fc2 products of every encoder block and of the cross K / V projections pass through mx_qdq_rows first (weights once, cached here)
    bool mx_encoder = false;
    std::map<std::string, std::vector<float>> mx_weights;
};

namespace {

// ---------------------------------------------------------------------------------------------
// model file (A.1)
// ---------------------------------------------------------------------------------------------
struct Reader {
    FILE *f;
    bool ok = true;
    template <typename T> T rd() {
        T v{};
        if (fread(&v, sizeof(T), 1, f) != 1) ok = false;
        return v;
    }
};

bool load_model(const char *path, Model &m) {
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    Reader r{f};
    uint32_t magic = r.rd<uint32_t>();
    if (magic != 0x67676d6c) {
        fclose(f);
        return false;
    }
    int32_t hp[11];
    for (int i = 0; i < 11; ++i) hp[i] = r.rd<int32_t>();
    m.hp = {hp[0], hp[1], hp[2], hp[3], hp[4], hp[5], hp[6], hp[7], hp[8], hp[9], hp[10] % 1000};
    m.filt_n_mel = r.rd<int32_t>();
    m.filt_n_fft = r.rd<int32_t>();
    m.filters.resize((size_t)m.filt_n_mel * m.filt_n_fft);
    if (fread(m.filters.data(), 4, m.filters.size(), f) != m.filters.size()) r.ok = false;
    int32_t n_file = r.rd<int32_t>();
    Vocab &v = m.vocab;
    v.n_vocab = m.hp.n_vocab;
    v.id_to_token.resize(std::max(n_file, v.n_vocab));
    for (int i = 0; i < n_file && r.ok; ++i) {
        uint32_t len = r.rd<uint32_t>();
        std::string w(len, '\0');
        if (len && fread(&w[0], 1, len, f) != len) r.ok = false;
        v.token_to_id[w] = i;
        v.id_to_token[i] = w;
    }
    if (v.is_multilingual()) {
        v.eot++;
        v.sot++;
        const int dt = v.num_languages() - 98;
        v.translate += dt;
        v.transcribe += dt;
        v.solm += dt;
        v.prev += dt;
        v.nosp += dt;
        v.not_ += dt;
        v.beg += dt;
    }
    for (int i = n_file; i < v.n_vocab; ++i) {
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
        else if (i > v.sot && i <= v.sot + v.num_languages())
            w = std::string("[_LANG_") + ow_lang_str(i - v.sot - 1) + "]";
        else w = "[_extra_token_" + std::to_string(i) + "]";
        v.token_to_id[w] = i;
        v.id_to_token[i] = w;
    }
    while (r.ok) {
        int32_t n_dims = 0;
        if (fread(&n_dims, 4, 1, f) != 1) break;  // EOF
        int32_t name_len = r.rd<int32_t>(), ttype = r.rd<int32_t>();
        Tensor t;
        size_t n = 1;
        for (int i = 0; i < n_dims; ++i) {
            t.ne.push_back(r.rd<int32_t>());
            n *= t.ne.back();
        }
        std::string name(name_len, '\0');
        if (fread(&name[0], 1, name_len, f) != (size_t)name_len) r.ok = false;
        t.d.resize(n);
        if (ttype == 0) {
            if (fread(t.d.data(), 4, n, f) != n) r.ok = false;
        } else if (ttype == 1) {
            std::vector<uint16_t> h(n);
            if (fread(h.data(), 2, n, f) != n) r.ok = false;
            for (size_t i = 0; i < n; ++i) t.d[i] = half_to_float(h[i]);
        } else if (ttype == 2 || ttype == 3 || ttype == 6 || ttype == 7 || ttype == 8) {
            // GGML block-quantised types (32 elements per block), dequantised the way ggml's dequantize_row_q* do:
            // q4_0 {f16 d; qs[16]} (nib-8)*d | q4_1 {d; m; qs} nib*d+m | q5_0 {d; qh[4]; qs} ((nib|bit<<4)-16)*d |
            // q5_1 {d; m; qh; qs} (nib|bit<<4)*d+m | q8_0 {d; i8 qs[32]} qs*d
            const int bb = ttype == 2 ? 18 : ttype == 3 ? 20 : ttype == 6 ? 22 : ttype == 7 ? 24 : 34;
            if (n % 32 != 0) r.ok = false;
            std::vector<unsigned char> raw(n / 32 * bb);
            if (r.ok && fread(raw.data(), 1, raw.size(), f) != raw.size()) r.ok = false;
            for (size_t blk = 0; r.ok && blk < n / 32; ++blk) {
                const unsigned char *b = raw.data() + blk * bb;
                uint16_t dh, mh = 0;
                memcpy(&dh, b, 2);
                const float d = half_to_float(dh);
                float *y = t.d.data() + blk * 32;
                if (ttype == 8) {
                    for (int j = 0; j < 32; ++j) y[j] = (float)(int8_t)b[2 + j] * d;
                    continue;
                }
                const bool has_m = ttype == 3 || ttype == 7, has_h = ttype == 6 || ttype == 7;
                const unsigned char *q = b + 2;
                if (has_m) {
                    memcpy(&mh, q, 2);
                    q += 2;
                }
                const float mn = half_to_float(mh);
                uint32_t qh = 0;
                if (has_h) {
                    memcpy(&qh, q, 4);
                    q += 4;
                }
                for (int j = 0; j < 16; ++j) {
                    int x0 = q[j] & 0x0F, x1 = q[j] >> 4;
                    if (has_h) {
                        x0 |= (int)((qh >> j) & 1u) << 4;
                        x1 |= (int)((qh >> (j + 16)) & 1u) << 4;
                    }
                    if (has_m) {
                        y[j] = (float)x0 * d + mn;
                        y[j + 16] = (float)x1 * d + mn;
                    } else {
                        const int zero = has_h ? 16 : 8;
                        y[j] = (float)(x0 - zero) * d;
                        y[j + 16] = (float)(x1 - zero) * d;
                    }
                }
            }
        } else {
            fprintf(stderr, "oracle: unsupported tensor type %d (%s)\n", ttype, name.c_str());
            r.ok = false;
        }
        m.t[name] = std::move(t);
    }
    fclose(f);
    return r.ok;
}

// ---------------------------------------------------------------------------------------------
// log-mel (A.2; whisper.cpp log_mel_spectrogram + fft/dft)
// ---------------------------------------------------------------------------------------------
float g_sin[N_FFT], g_cos[N_FFT];
bool g_trig_init = false;
void init_trig() {
    if (g_trig_init) return;
    for (int i = 0; i < N_FFT; ++i) {
        double theta = (2 * M_PI * i) / N_FFT;
        g_sin[i] = sinf(theta);
        g_cos[i] = cosf(theta);
    }
    g_trig_init = true;
}

void dft(const float *in, int N, float *out) {
    const int step = N_FFT / N;
    for (int k = 0; k < N; ++k) {
        float re = 0, im = 0;
        for (int n = 0; n < N; ++n) {
            int idx = (k * n * step) % N_FFT;
            re += in[n] * g_cos[idx];
            im -= in[n] * g_sin[idx];
        }
        out[2 * k] = re;
        out[2 * k + 1] = im;
    }
}

void fft(const std::vector<float> &in, std::vector<float> &out) {
    const int N = (int)in.size();
    out.resize(2 * N);
    if (N == 1) {
        out[0] = in[0];
        out[1] = 0;
        return;
    }
    if (N % 2 == 1) {
        dft(in.data(), N, out.data());
        return;
    }
    std::vector<float> even(N / 2), odd(N / 2), ef, of;
    for (int i = 0; i < N; ++i) (i % 2 == 0 ? even[i / 2] : odd[i / 2]) = in[i];
    fft(even, ef);
    fft(odd, of);
    const int step = N_FFT / N;
    for (int k = 0; k < N / 2; ++k) {
        int idx = k * step;
        float re = g_cos[idx], im = -g_sin[idx];
        float re_odd = of[2 * k], im_odd = of[2 * k + 1];
        out[2 * k] = ef[2 * k] + re * re_odd - im * im_odd;
        out[2 * k + 1] = ef[2 * k + 1] + re * im_odd + im * re_odd;
        out[2 * (k + N / 2)] = ef[2 * k] - re * re_odd + im * im_odd;
        out[2 * (k + N / 2) + 1] = ef[2 * k + 1] - re * im_odd - im * re_odd;
    }
}

void pcm_to_mel(const Model &m, const float *samples, int n_samples, Mel &mel) {
    init_trig();
    std::vector<float> hann(N_FFT);
    for (int i = 0; i < N_FFT; ++i) hann[i] = 0.5 * (1.0 - cos((2.0 * M_PI * i) / N_FFT));  // periodic
    const int64_t pad1 = (int64_t)SAMPLE_RATE * CHUNK_SIZE, pad2 = N_FFT / 2;
    std::vector<float> sp(n_samples + pad1 + 2 * pad2, 0.0f);
    std::copy(samples, samples + n_samples, sp.begin() + pad2);
    // reflective pad at the start only: reverse_copy(samples+1, samples+1+pad2, begin)
    for (int i = 0; i < pad2; ++i) sp[i] = (pad2 - i < n_samples) ? samples[pad2 - i] : 0.0f;
    const int n_mel = m.filt_n_mel, n_fft = m.filt_n_fft;
    mel.n_mel = n_mel;
    mel.n_len = (int)((sp.size() - N_FFT) / HOP);
    mel.n_len_org = 1 + (int)((n_samples + pad2 - N_FFT) / HOP);
    mel.data.assign((size_t)n_mel * mel.n_len, 0.0f);
    const int n_real = n_samples + (int)pad2;  // extent of non-padding data in sp
    const int n_active = std::min(n_real / HOP + 1, mel.n_len);
#pragma omp parallel
    {
        std::vector<float> fin(N_FFT), fout(2 * N_FFT);
#pragma omp for schedule(static)
        for (int i = 0; i < n_active; ++i) {
            const int offset = i * HOP;
            const int nv = std::min(N_FFT, n_real - offset);
            for (int j = 0; j < nv; ++j) fin[j] = hann[j] * sp[offset + j];
            for (int j = std::max(nv, 0); j < N_FFT; ++j) fin[j] = 0.0f;
            fft(fin, fout);
            for (int j = 0; j < n_fft; ++j) fout[j] = fout[2 * j] * fout[2 * j] + fout[2 * j + 1] * fout[2 * j + 1];
            for (int j = 0; j < n_mel; ++j) {
                double sum = 0.0;
                const float *fl = &m.filters[(size_t)j * n_fft];
                int k = 0;
                for (; k < n_fft - 3; k += 4)
                    sum += fout[k] * fl[k] + fout[k + 1] * fl[k + 1] + fout[k + 2] * fl[k + 2] + fout[k + 3] * fl[k + 3];
                for (; k < n_fft; ++k) sum += fout[k] * fl[k];
                sum = log10(std::max(sum, 1e-10));
                mel.data[(size_t)j * mel.n_len + i] = (float)sum;
            }
        }
    }
    const float floor_v = (float)log10(1e-10);
    for (int i = n_active; i < mel.n_len; ++i)
        for (int j = 0; j < n_mel; ++j) mel.data[(size_t)j * mel.n_len + i] = floor_v;
    double mmax = -1e20;
    for (float v : mel.data)
        if (v > mmax) mmax = v;
    mmax -= 8.0;
    for (float &v : mel.data) {
        if (v < mmax) v = (float)mmax;
        v = (float)((v + 4.0) / 4.0);
    }
}

// ---------------------------------------------------------------------------------------------
// encoder + cross-KV (A.3; whisper_build_graph_conv / _encoder / _cross)
// ---------------------------------------------------------------------------------------------
void encode(ow_context &c, int mel_offset) {
    const Model &m = c.model;
    const HP &hp = m.hp;
    const int n_ctx = c.exp_n_audio_ctx > 0 ? c.exp_n_audio_ctx : hp.n_audio_ctx;
    const int d = hp.n_audio_state, H = hp.n_audio_head, dh = d / H, n_mel = hp.n_mels;
    const int T2 = 2 * n_ctx;
    c.n_ctx_cur = n_ctx;
    c.n_encode_calls++;
    // mel window [n_mel][2*n_ctx], zero beyond n_len
    std::vector<float> x0((size_t)n_mel * T2, 0.0f);
    const int i0 = std::min(mel_offset, c.mel.n_len), i1 = std::min(mel_offset + T2, c.mel.n_len);
    for (int j = 0; j < n_mel; ++j)
        for (int i = i0; i < i1; ++i) x0[(size_t)j * T2 + (i - i0)] = c.mel.data[(size_t)j * c.mel.n_len + i];
    // conv1 (k3,s1,p1) + bias + GELU -> [d][T2]
    const auto &w1 = m.get("encoder.conv1.weight"), &b1 = m.get("encoder.conv1.bias");
    std::vector<float> x1((size_t)d * T2);
#pragma omp parallel for schedule(static)
    for (int co = 0; co < d; ++co) {
        std::vector<float> acc(T2, b1[co]);
        for (int ci = 0; ci < n_mel; ++ci) {
            const float *w = &w1[((size_t)co * n_mel + ci) * 3];
            const float *xi = &x0[(size_t)ci * T2];
            acc[0] += w[1] * xi[0] + w[2] * xi[1];
            for (int t = 1; t < T2 - 1; ++t) acc[t] += w[0] * xi[t - 1] + w[1] * xi[t] + w[2] * xi[t + 1];
            acc[T2 - 1] += w[0] * xi[T2 - 2] + w[1] * xi[T2 - 1];
        }
        for (int t = 0; t < T2; ++t) x1[(size_t)co * T2 + t] = gelu_tanh(acc[t]);
    }
    // conv2 (k3,s2,p1) + bias + GELU -> transposed [n_ctx][d], + positional embedding
    const auto &w2 = m.get("encoder.conv2.weight"), &b2 = m.get("encoder.conv2.bias");
    const auto &pe = m.get("encoder.positional_embedding");
    std::vector<float> x((size_t)n_ctx * d);
#pragma omp parallel for schedule(static)
    for (int co = 0; co < d; ++co) {
        std::vector<float> acc(n_ctx, b2[co]);
        for (int ci = 0; ci < d; ++ci) {
            const float *w = &w2[((size_t)co * d + ci) * 3];
            const float *xi = &x1[(size_t)ci * T2];
            acc[0] += w[1] * xi[0] + w[2] * xi[1];
            for (int t = 1; t < n_ctx; ++t) acc[t] += w[0] * xi[2 * t - 1] + w[1] * xi[2 * t] + w[2] * xi[2 * t + 1];
        }
        for (int t = 0; t < n_ctx; ++t) x[(size_t)t * d + co] = gelu_tanh(acc[t]) + pe[(size_t)t * d + co];
    }
    std::vector<float> hmx;
    std::vector<float> h((size_t)n_ctx * d), q((size_t)n_ctx * d), k((size_t)n_ctx * d), v((size_t)n_ctx * d),
        att((size_t)n_ctx * d), tmp((size_t)n_ctx * d), ff((size_t)n_ctx * 4 * d);
    const float kq_scale = 1.0f / sqrtf((float)dh);
    // MX twin: weights and the activation operand of the Q, K, V, attention-output, fc1, fc2 (and cross K / V) products on the MX-fp8 grid
    auto mxw = [&](const std::string &name, int K) -> const float * {
        if (!c.mx_encoder) return m.get(name).data();
        auto it = c.mx_weights.find(name);
        if (it == c.mx_weights.end()) {
            it = c.mx_weights.emplace(name, m.get(name)).first;
            mx_qdq_rows(it->second.data(), it->second.size() / K, K, K);
        }
        return it->second.data();
    };
    for (int l = 0; l < hp.n_audio_layer; ++l) {
        const std::string p = "encoder.blocks." + std::to_string(l) + ".";
        layer_norm(h.data(), x.data(), m.get(p + "attn_ln.weight").data(), m.get(p + "attn_ln.bias").data(), n_ctx, d);
        const float *hq = h.data();
        if (c.mx_encoder) {
            hmx = h;
            mx_qdq_rows(hmx.data(), n_ctx, d, d);
            hq = hmx.data();
        }
        matmul_nt(q.data(), d, hq, d, mxw(p + "attn.query.weight", d), d, m.get(p + "attn.query.bias").data(), n_ctx, d, d);
        matmul_nt(k.data(), d, hq, d, mxw(p + "attn.key.weight", d), d, nullptr, n_ctx, d, d);
        matmul_nt(v.data(), d, hq, d, mxw(p + "attn.value.weight", d), d, m.get(p + "attn.value.bias").data(), n_ctx, d, d);
#pragma omp parallel
        {
            std::vector<float> s(n_ctx);
#pragma omp for schedule(static) collapse(2)
            for (int hd = 0; hd < H; ++hd)
                for (int i = 0; i < n_ctx; ++i) {
                    const float *qi = &q[(size_t)i * d + hd * dh];
                    float mx = -INFINITY;
                    for (int j = 0; j < n_ctx; ++j) {
                        s[j] = dot_f32(qi, &k[(size_t)j * d + hd * dh], dh) * kq_scale;
                        mx = std::max(mx, s[j]);
                    }
                    double sum = 0;
                    for (int j = 0; j < n_ctx; ++j) {
                        s[j] = expf(s[j] - mx);
                        sum += s[j];
                    }
                    const float inv = (float)(1.0 / sum);
                    float o[256];
                    for (int e = 0; e < dh; ++e) o[e] = 0;
                    for (int j = 0; j < n_ctx; ++j) {
                        const float pj = s[j] * inv;
                        const float *vj = &v[(size_t)j * d + hd * dh];
                        for (int e = 0; e < dh; ++e) o[e] += pj * vj[e];
                    }
                    for (int e = 0; e < dh; ++e) att[(size_t)i * d + hd * dh + e] = o[e];
                }
        }
        if (c.mx_encoder) mx_qdq_rows(att.data(), n_ctx, d, d);
        matmul_nt(tmp.data(), d, att.data(), d, mxw(p + "attn.out.weight", d), d, m.get(p + "attn.out.bias").data(), n_ctx, d, d);
        for (size_t i = 0; i < x.size(); ++i) x[i] += tmp[i];
        layer_norm(h.data(), x.data(), m.get(p + "mlp_ln.weight").data(), m.get(p + "mlp_ln.bias").data(), n_ctx, d);
        if (c.mx_encoder) mx_qdq_rows(h.data(), n_ctx, d, d);
        matmul_nt(ff.data(), 4 * d, h.data(), d, mxw(p + "mlp.0.weight", d), d, m.get(p + "mlp.0.bias").data(), n_ctx, 4 * d, d);
        for (auto &f : ff) f = gelu_tanh(f);
        if (c.mx_encoder) mx_qdq_rows(ff.data(), n_ctx, 4 * d, 4 * (size_t)d);
        matmul_nt(tmp.data(), d, ff.data(), 4 * d, mxw(p + "mlp.2.weight", 4 * d), 4 * d, m.get(p + "mlp.2.bias").data(), n_ctx, d,
                  4 * d);
        for (size_t i = 0; i < x.size(); ++i) x[i] += tmp[i];
    }
    c.enc_out.resize((size_t)n_ctx * d);
    layer_norm(c.enc_out.data(), x.data(), m.get("encoder.ln_post.weight").data(), m.get("encoder.ln_post.bias").data(),
               n_ctx, d);
    // cross-KV: K = (Wk enc) * dh^-0.25, V = Wv enc + bv
    const int dt = hp.n_text_state, dht = dt / hp.n_text_head;
    const float kscale = powf((float)dht, -0.25f);
    const float *eq = c.enc_out.data();  // MX twin: the cross K / V products read the encoder's output on the MX-fp8 grid
    if (c.mx_encoder) {
        hmx = c.enc_out;
        mx_qdq_rows(hmx.data(), n_ctx, d, d);
        eq = hmx.data();
    }
    c.cross_k.assign(hp.n_text_layer, {});
    c.cross_v.assign(hp.n_text_layer, {});
    for (int l = 0; l < hp.n_text_layer; ++l) {
        const std::string p = "decoder.blocks." + std::to_string(l) + ".cross_attn.";
        c.cross_k[l].resize((size_t)n_ctx * dt);
        c.cross_v[l].resize((size_t)n_ctx * dt);
        matmul_nt(c.cross_k[l].data(), dt, eq, d, mxw(p + "key.weight", d), d, nullptr, n_ctx, dt, d);
        for (auto &f : c.cross_k[l]) f *= kscale;
        matmul_nt(c.cross_v[l].data(), dt, eq, d, mxw(p + "value.weight", d), d, m.get(p + "value.bias").data(), n_ctx, dt, d);
    }
}

// ---------------------------------------------------------------------------------------------
// decoder (A.4; whisper_build_graph_decoder): batch of (token, pos, seq)
// ---------------------------------------------------------------------------------------------
void kv_init(ow_context &c) {
    const HP &hp = c.model.hp;
    c.kv.resize(2 * MAX_DECODERS);
    for (auto &s : c.kv) {
        s.k.assign((size_t)hp.n_text_layer * hp.n_text_ctx * hp.n_text_state, 0.0f);
        s.v.assign((size_t)hp.n_text_layer * hp.n_text_ctx * hp.n_text_state, 0.0f);
        s.n = 0;
    }
}
void kv_clear(ow_context &c) {
    for (auto &s : c.kv) s.n = 0;
}
void kv_seq_cp(ow_context &c, int src, int dst) {
    if (src == dst) return;
    c.kv[dst] = c.kv[src];
}

bool decode(ow_context &c, const std::vector<int> &tokens, const std::vector<int> &pos, const std::vector<int> &seq) {
    const Model &m = c.model;
    const HP &hp = m.hp;
    const int n = (int)tokens.size(), d = hp.n_text_state, H = hp.n_text_head, dh = d / H, n_ctx = c.n_ctx_cur;
    const int n_vocab = hp.n_vocab;
    c.n_decode_calls++;
    for (int i = 0; i < n; ++i)
        if (pos[i] >= hp.n_text_ctx || tokens[i] < 0 || tokens[i] >= n_vocab) return false;
    const auto &te = m.get("decoder.token_embedding.weight"), &pe = m.get("decoder.positional_embedding");
    std::vector<float> x((size_t)n * d), h((size_t)n * d), q((size_t)n * d), k((size_t)n * d), v((size_t)n * d),
        att((size_t)n * d), tmp((size_t)n * d), ff((size_t)n * 4 * d);
    for (int i = 0; i < n; ++i)
        for (int e = 0; e < d; ++e) x[(size_t)i * d + e] = te[(size_t)tokens[i] * d + e] + pe[(size_t)pos[i] * d + e];
    const float kq_scale = powf((float)dh, -0.25f);
    for (int l = 0; l < hp.n_text_layer; ++l) {
        const std::string p = "decoder.blocks." + std::to_string(l) + ".";
        layer_norm(h.data(), x.data(), m.get(p + "attn_ln.weight").data(), m.get(p + "attn_ln.bias").data(), n, d);
        matmul_nt(q.data(), d, h.data(), d, m.get(p + "attn.query.weight").data(), d, m.get(p + "attn.query.bias").data(), n,
                  d, d);
        matmul_nt(k.data(), d, h.data(), d, m.get(p + "attn.key.weight").data(), d, nullptr, n, d, d);
        matmul_nt(v.data(), d, h.data(), d, m.get(p + "attn.value.weight").data(), d, m.get(p + "attn.value.bias").data(), n,
                  d, d);
        for (auto &f : q) f *= kq_scale;
        for (auto &f : k) f *= kq_scale;
        for (int i = 0; i < n; ++i) {
            KVSeq &s = c.kv[seq[i]];
            const size_t off = ((size_t)l * hp.n_text_ctx + pos[i]) * d;
            memcpy(&s.k[off], &k[(size_t)i * d], d * sizeof(float));
            memcpy(&s.v[off], &v[(size_t)i * d], d * sizeof(float));
        }
#pragma omp parallel for schedule(static) collapse(2)
        for (int i = 0; i < n; ++i)
            for (int hd = 0; hd < H; ++hd) {
                const KVSeq &s = c.kv[seq[i]];
                const int np = pos[i] + 1;  // causal: positions 0..pos
                std::vector<float> sc(np);
                float mx = -INFINITY;
                for (int j = 0; j < np; ++j) {
                    sc[j] = dot_f32(&q[(size_t)i * d + hd * dh], &s.k[((size_t)l * hp.n_text_ctx + j) * d + hd * dh], dh);
                    mx = std::max(mx, sc[j]);
                }
                double sum = 0;
                for (int j = 0; j < np; ++j) {
                    sc[j] = expf(sc[j] - mx);
                    sum += sc[j];
                }
                const float inv = (float)(1.0 / sum);
                float o[256];
                for (int e = 0; e < dh; ++e) o[e] = 0;
                for (int j = 0; j < np; ++j) {
                    const float pj = sc[j] * inv;
                    const float *vj = &s.v[((size_t)l * hp.n_text_ctx + j) * d + hd * dh];
                    for (int e = 0; e < dh; ++e) o[e] += pj * vj[e];
                }
                for (int e = 0; e < dh; ++e) att[(size_t)i * d + hd * dh + e] = o[e];
            }
        matmul_nt(tmp.data(), d, att.data(), d, m.get(p + "attn.out.weight").data(), d, m.get(p + "attn.out.bias").data(), n,
                  d, d);
        for (size_t i = 0; i < x.size(); ++i) x[i] += tmp[i];
        // cross attention
        layer_norm(h.data(), x.data(), m.get(p + "cross_attn_ln.weight").data(), m.get(p + "cross_attn_ln.bias").data(), n,
                   d);
        matmul_nt(q.data(), d, h.data(), d, m.get(p + "cross_attn.query.weight").data(), d,
                  m.get(p + "cross_attn.query.bias").data(), n, d, d);
        for (auto &f : q) f *= kq_scale;
        const std::vector<float> &ck = c.cross_k[l], &cv = c.cross_v[l];
#pragma omp parallel for schedule(static) collapse(2)
        for (int i = 0; i < n; ++i)
            for (int hd = 0; hd < H; ++hd) {
                std::vector<float> sc(n_ctx);
                float mx = -INFINITY;
                for (int j = 0; j < n_ctx; ++j) {
                    sc[j] = dot_f32(&q[(size_t)i * d + hd * dh], &ck[(size_t)j * d + hd * dh], dh);
                    mx = std::max(mx, sc[j]);
                }
                double sum = 0;
                for (int j = 0; j < n_ctx; ++j) {
                    sc[j] = expf(sc[j] - mx);
                    sum += sc[j];
                }
                const float inv = (float)(1.0 / sum);
                float o[256];
                for (int e = 0; e < dh; ++e) o[e] = 0;
                for (int j = 0; j < n_ctx; ++j) {
                    const float pj = sc[j] * inv;
                    const float *vj = &cv[(size_t)j * d + hd * dh];
                    for (int e = 0; e < dh; ++e) o[e] += pj * vj[e];
                }
                for (int e = 0; e < dh; ++e) att[(size_t)i * d + hd * dh + e] = o[e];
            }
        matmul_nt(tmp.data(), d, att.data(), d, m.get(p + "cross_attn.out.weight").data(), d,
                  m.get(p + "cross_attn.out.bias").data(), n, d, d);
        for (size_t i = 0; i < x.size(); ++i) x[i] += tmp[i];
        layer_norm(h.data(), x.data(), m.get(p + "mlp_ln.weight").data(), m.get(p + "mlp_ln.bias").data(), n, d);
        matmul_nt(ff.data(), 4 * d, h.data(), d, m.get(p + "mlp.0.weight").data(), d, m.get(p + "mlp.0.bias").data(), n,
                  4 * d, d);
        for (auto &f : ff) f = gelu_tanh(f);
        matmul_nt(tmp.data(), d, ff.data(), 4 * d, m.get(p + "mlp.2.weight").data(), 4 * d, m.get(p + "mlp.2.bias").data(),
                  n, d, 4 * d);
        for (size_t i = 0; i < x.size(); ++i) x[i] += tmp[i];
    }
    layer_norm(h.data(), x.data(), m.get("decoder.ln.weight").data(), m.get("decoder.ln.bias").data(), n, d);
    c.logits.resize((size_t)n * n_vocab);
    c.n_logits_rows = n;
    matmul_nt(c.logits.data(), n_vocab, h.data(), d, te.data(), d, nullptr, n, n_vocab, d);
    return true;
}

// ---------------------------------------------------------------------------------------------
// tokenizer (whisper.cpp tokenize(): regex split + greedy longest match)
// ---------------------------------------------------------------------------------------------
std::vector<int> tokenize(const Vocab &vocab, const std::string &text) {
    std::vector<std::string> words;
    {
        std::string str = text;
        std::string pat =
            R"('s|'t|'re|'ve|'m|'ll|'d| ?[[:alpha:]]+| ?[[:digit:]]+| ?[^\s[:alpha:][:digit:]]+|\s+(?!\S)|\s+)";
        std::regex re(pat);
        std::smatch mm;
        while (std::regex_search(str, mm, re)) {
            for (auto x : mm) words.push_back(x);
            str = mm.suffix();
        }
    }
    std::vector<int> tokens;
    for (const auto &word : words) {
        if (word.empty()) continue;
        int i = 0, n = (int)word.size();
        while (i < n) {
            int j = n;
            bool found = false;
            while (j > i) {
                auto it = vocab.token_to_id.find(word.substr(i, j - i));
                if (it != vocab.token_to_id.end()) {
                    tokens.push_back(it->second);
                    i = j;
                    found = true;
                    break;
                }
                --j;
            }
            if (!found) ++i;
        }
    }
    return tokens;
}

// ---------------------------------------------------------------------------------------------
// logits rules (A.5; whisper_process_logits)
// ---------------------------------------------------------------------------------------------
void process_logits(const ow_context &c, const ow_params &params, float temperature, const float *logits_in,
                    const std::vector<ow_token_data> &tokens_cur, bool has_ts, int seek_delta, std::vector<float> &logits,
                    std::vector<float> &logprobs, std::vector<float> &probs) {
    const Vocab &vocab = c.model.vocab;
    const bool is_initial = tokens_cur.empty();
    const int n_logits = vocab.n_vocab;
    logits.assign(logits_in, logits_in + n_logits);
    if (temperature > 0.0f)
        for (int i = 0; i < n_logits; ++i) logits[i] /= temperature;
    probs.resize(n_logits);
    logprobs.resize(n_logits);
    if (params.suppress_blank && is_initial) {
        logits[vocab.eot] = -INFINITY;
        auto it = vocab.token_to_id.find(" ");
        if (it != vocab.token_to_id.end()) logits[it->second] = -INFINITY;
    }
    if (params.force_decode_steps > 0) logits[vocab.eot] = -INFINITY;  // bench-only extension
    logits[vocab.not_] = -INFINITY;
    if (params.no_timestamps)
        for (int i = vocab.beg; i < n_logits; ++i) logits[i] = -INFINITY;
    logits[vocab.sot] = -INFINITY;
    logits[vocab.nosp] = -INFINITY;
    if (!params.tdrz_enable) logits[vocab.solm] = -INFINITY;
    logits[vocab.translate] = -INFINITY;
    logits[vocab.transcribe] = -INFINITY;
    logits[vocab.prev] = -INFINITY;
    for (int i = 0; i < N_LANGS; ++i) {
        const int id = vocab.sot + 1 + i;
        if (id < n_logits) logits[id] = -INFINITY;
    }
    if (params.suppress_regex != nullptr && params.suppress_regex[0] != 0) {
        std::regex re(params.suppress_regex);
        for (const auto &kv : vocab.token_to_id)
            if (std::regex_match(kv.first, re)) logits[kv.second] = -INFINITY;
    }
    if (params.suppress_non_speech_tokens) {
        for (const char *t : k_non_speech) {
            const std::string forms[2] = {t, std::string(" ") + t};
            for (const auto &s : forms) {
                auto it = vocab.token_to_id.find(s);
                if (it != vocab.token_to_id.end()) logits[it->second] = -INFINITY;
            }
        }
        for (const char *t : {" -", " '"}) {
            auto it = vocab.token_to_id.find(t);
            if (it != vocab.token_to_id.end()) logits[it->second] = -INFINITY;
        }
    }
    {
        const bool last_ts = !tokens_cur.empty() && tokens_cur.back().id >= vocab.beg;
        const bool pen_ts = tokens_cur.size() < 2 || tokens_cur[tokens_cur.size() - 2].id >= vocab.beg;
        if (last_ts) {
            if (pen_ts) {
                for (int i = vocab.beg; i < n_logits; ++i) logits[i] = -INFINITY;
            } else {
                for (int i = 0; i < vocab.eot; ++i) logits[i] = -INFINITY;
            }
        }
    }
    if (is_initial && params.max_initial_ts > 0.0f) {
        const float precision = float(CHUNK_SIZE) / c.model.hp.n_audio_ctx;
        const int tid0 = (int)std::round(params.max_initial_ts / precision);
        for (int i = vocab.beg + tid0 + 1; i < n_logits; ++i) logits[i] = -INFINITY;
    }
    if (has_ts) {
        const int tid0 = seek_delta / 2;
        for (int i = vocab.beg; i < vocab.beg + tid0 && i < n_logits; ++i) logits[i] = -INFINITY;
    }
    {
        const float logit_max = *std::max_element(logits.begin(), logits.end());
        float logsumexp = 0.0f;
        for (int i = 0; i < n_logits; ++i)
            if (logits[i] > -INFINITY) logsumexp += expf(logits[i] - logit_max);
        logsumexp = logf(logsumexp) + logit_max;
        for (int i = 0; i < n_logits; ++i) logprobs[i] = logits[i] > -INFINITY ? logits[i] - logsumexp : -INFINITY;
    }
    {
        float timestamp_logprob = -INFINITY;
        {
            float logsumexp = 0.0f;
            const float logprob_max = *std::max_element(logprobs.begin() + vocab.beg, logprobs.end());
            for (int i = vocab.beg; i < n_logits; ++i)
                if (logprobs[i] > -INFINITY) logsumexp += expf(logprobs[i] - logprob_max);
            if (logsumexp > 0.0f) timestamp_logprob = logf(logsumexp) + logprob_max;
        }
        const float max_text = *std::max_element(logprobs.begin(), logprobs.begin() + vocab.beg);
        if (timestamp_logprob > max_text)
            for (int i = 0; i < vocab.beg; ++i) {
                logits[i] = -INFINITY;
                logprobs[i] = -INFINITY;
            }
    }
    for (int i = 0; i < n_logits; ++i) probs[i] = logits[i] == -INFINITY ? 0.0f : expf(logprobs[i]);
}

// ---------------------------------------------------------------------------------------------
// samplers (A.6; whisper_sample_token / whisper_sample_token_topk)
// ---------------------------------------------------------------------------------------------
ow_token_data sample_token(const Vocab &vocab, Decoder &dec, bool best) {
    ow_token_data r = {0, 0, 0.0f, 0.0f, 0.0f, 0.0f, -1, -1, -1, 0.0f};
    const auto &probs = dec.probs;
    const auto &logprobs = dec.logprobs;
    const int n = vocab.n_vocab;
    {
        double sum_ts = 0.0, max_ts = 0.0;
        for (int i = vocab.beg; i < n; ++i) {
            if (probs[i] == -INFINITY) continue;
            sum_ts += probs[i];
            if (max_ts < probs[i]) {
                max_ts = probs[i];
                r.tid = i;
            }
        }
        r.pt = (float)(max_ts / (sum_ts + 1e-10));
        r.ptsum = (float)sum_ts;
    }
    if (best) {
        for (int i = 0; i < n; ++i)
            if (r.p < probs[i]) {
                r.id = i;
                r.p = probs[i];
                r.plog = logprobs[i];
            }
    } else {
        std::discrete_distribution<> dist(probs.begin(), probs.end());
        r.id = dist(dec.rng);
        r.p = probs[r.id];
        r.plog = logprobs[r.id];
    }
    if (r.id >= vocab.beg) {
        r.tid = r.id;
        r.pt = r.p;
    }
    return r;
}

std::vector<ow_token_data> sample_token_topk(const Vocab &vocab, Decoder &dec, int k) {
    const auto &probs = dec.probs;
    const auto &logprobs = dec.logprobs;
    const int n = vocab.n_vocab;
    std::vector<ow_token_data> result;
    int tid = vocab.beg;
    float pt = 0, ptsum = 0;
    {
        double sum_ts = 0.0, max_ts = 0.0;
        for (int i = vocab.beg; i < n; ++i) {
            if (probs[i] == -INFINITY) continue;
            sum_ts += probs[i];
            if (max_ts < probs[i]) {
                max_ts = probs[i];
                tid = i;
            }
        }
        pt = (float)(max_ts / (sum_ts + 1e-10));
        ptsum = (float)sum_ts;
    }
    std::discrete_distribution<> dist(probs.begin(), probs.end());
    for (int i = 0; i < k; ++i) {
        const int id = dist(dec.rng);
        ow_token_data t = {id, tid, probs[id], logprobs[id], pt, ptsum, -1, -1, -1, 0.0f};
        if (t.id >= vocab.beg) {
            t.tid = t.id;
            t.pt = t.p;
        }
        result.push_back(t);
    }
    return result;
}

void sequence_score(const ow_params &params, Sequence &s) {
    if (s.result_len == 0) return;
    double result = 0.0;
    for (int i = 0; i < s.result_len; ++i) result += s.tokens[i].plog;
    s.sum_logprobs = result;
    s.avg_logprobs = result / s.result_len;
    double penalty = s.result_len;
    if (params.length_penalty > 0.0f) penalty = pow((5.0 + penalty) / 6.0, params.length_penalty);
    s.score = result / penalty;
    {
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
}

bool seq_tokens_equal(const Sequence &a, const Sequence &b) {
    if (a.tokens.size() != b.tokens.size()) return false;
    for (size_t i = 0; i < a.tokens.size(); ++i)
        if (a.tokens[i].id != b.tokens[i].id) return false;
    return true;
}

int lang_auto_detect(ow_context &c, int offset_ms, float *lang_probs) {
    const Vocab &v = c.model.vocab;
    const int seek = offset_ms / 10;
    if (seek < 0) return -1;
    if (seek >= c.mel.n_len_org) return -2;
    if (!v.is_multilingual()) return -3;
    encode(c, seek);
    kv_clear(c);
    if (!decode(c, {v.sot}, {0}, {0})) return -7;
    std::vector<std::pair<double, int>> lid;
    const int nl = std::min(N_LANGS, v.num_languages());
    for (int i = 0; i < nl; ++i) lid.emplace_back(c.logits[v.sot + 1 + i], i);
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

// ---------------------------------------------------------------------------------------------
// whisper_full driver (A.6; whisper_full_with_state)
// ---------------------------------------------------------------------------------------------
int full(ow_context &c, ow_params params, const float *samples, int n_samples) {
    const Model &model = c.model;
    const Vocab &vocab = model.vocab;
    const HP &hp = model.hp;
    auto &result_all = c.result_all;
    result_all.clear();
    if (n_samples > 0) pcm_to_mel(model, samples, n_samples, c.mel);
    std::string language = params.language ? params.language : "";
    if (language.empty() || language == "auto" || params.detect_language) {
        std::vector<float> probs(N_LANGS, 0.0f);
        const int lang_id = lang_auto_detect(c, 0, probs.data());
        if (lang_id < 0) return -3;
        c.lang_id = lang_id;
        language = ow_lang_str(lang_id);
        if (params.detect_language) return 0;
    }
    const int seek_start = params.offset_ms / 10;
    const int seek_end = params.duration_ms == 0 ? c.mel.n_len : seek_start + params.duration_ms / 10;
    if (seek_end < seek_start + 100) return 0;  // "input is too short"
    std::vector<float> temperatures;
    if (params.temperature_inc > 0.0f) {
        for (float t = params.temperature; t < 1.0f + 1e-6f; t += params.temperature_inc) temperatures.push_back(t);
    } else {
        temperatures.push_back(params.temperature);
    }
    int n_decoders = 1;
    if (params.strategy == 0) n_decoders = params.best_of;
    else n_decoders = std::max(params.best_of, params.beam_size);
    n_decoders = std::max(1, n_decoders);
    if (n_decoders > MAX_DECODERS) return -4;
    for (int j = 1; j < n_decoders; ++j) c.decoders[j].rng = std::mt19937(0);
    auto &prompt_past = c.prompt_past;
    if (params.no_context) prompt_past.clear();
    if (params.initial_prompt && params.initial_prompt[0]) {
        std::vector<int> pt = tokenize(vocab, params.initial_prompt);
        for (int t : pt) prompt_past.push_back(t);
        std::rotate(prompt_past.begin(), prompt_past.end() - pt.size(), prompt_past.end());
    }
    if (params.audio_ctx > hp.n_audio_ctx) return -5;
    c.exp_n_audio_ctx = params.audio_ctx;
    std::vector<int> prompt_init = {vocab.sot};
    if (vocab.is_multilingual()) {
        const int lang_id = ow_lang_id(language.c_str());
        if (lang_id < 0) return -9;
        c.lang_id = lang_id;
        prompt_init.push_back(vocab.sot + 1 + lang_id);
        prompt_init.push_back(params.translate ? vocab.translate : vocab.transcribe);
    }
    if (hp.n_text_layer == 2 && !params.no_timestamps) params.no_timestamps = true;  // distilled models
    if (params.no_timestamps) prompt_init.push_back(vocab.not_);
    int seek = seek_start;
    std::vector<int> prompt;
    struct beam_candidate {
        int decoder_idx;
        int seek_delta;
        bool has_ts;
        Sequence sequence;
    };
    std::vector<std::vector<beam_candidate>> bc_per_dec(n_decoders);
    std::vector<beam_candidate> beam_candidates;
    const int fds = params.force_decode_steps;

    while (true) {
        if (seek + 100 >= seek_end) break;
        encode(c, seek);
        if (seek > seek_start && seek + 500 >= seek_end) prompt_past.clear();
        int best_decoder_id = 0;
        for (int it = 0; it < (int)temperatures.size(); ++it) {
            const float t_cur = temperatures[it];
            int n_decoders_cur = 1;
            if (params.strategy == 0) {
                if (t_cur > 0.0f) n_decoders_cur = params.best_of;
            } else {
                n_decoders_cur = t_cur > 0.0f ? params.best_of : params.beam_size;
            }
            n_decoders_cur = std::max(1, n_decoders_cur);
            for (int j = 0; j < n_decoders_cur; ++j) {
                Decoder &d = c.decoders[j];
                d.sequence.tokens.clear();
                d.sequence.result_len = 0;
                d.sequence.sum_logprobs_all = 0.0;
                d.sequence.sum_logprobs = -INFINITY;
                d.sequence.avg_logprobs = -INFINITY;
                d.sequence.entropy = 0.0;
                d.sequence.score = -INFINITY;
                d.seek_delta = 100 * CHUNK_SIZE;
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
                std::vector<int> pos(prompt.size()), seq(prompt.size(), 0);
                for (size_t i = 0; i < prompt.size(); ++i) pos[i] = (int)i;
                if (!decode(c, prompt, pos, seq)) return -7;
                Decoder &d0 = c.decoders[0];
                d0.i_batch = (int)prompt.size() - 1;
                process_logits(c, params, t_cur, &c.logits[(size_t)d0.i_batch * hp.n_vocab], d0.sequence.tokens, d0.has_ts,
                               d0.seek_delta, d0.logits, d0.logprobs, d0.probs);
                for (int j = 1; j < n_decoders_cur; ++j) {
                    kv_seq_cp(c, 0, j);
                    c.decoders[j].probs = d0.probs;
                    c.decoders[j].logits = d0.logits;
                    c.decoders[j].logprobs = d0.logprobs;
                }
            }
            for (int i = 0, n_max = hp.n_text_ctx / 2 - 4; i < n_max; ++i) {
                if (params.strategy == 1)
                    for (auto &bc : bc_per_dec) bc.clear();
                for (int j = 0; j < n_decoders_cur; ++j) {
                    Decoder &d = c.decoders[j];
                    if (d.completed || d.failed) continue;
                    if (params.strategy == 0) {
                        d.sequence.tokens.push_back(sample_token(vocab, d, t_cur < 1e-6f));
                        d.sequence.sum_logprobs_all += d.sequence.tokens.back().plog;
                    } else {
                        const auto tokens_new = sample_token_topk(vocab, d, params.beam_size);
                        for (const auto &tk : tokens_new) {
                            bc_per_dec[j].push_back({j, d.seek_delta, d.has_ts, d.sequence});
                            bc_per_dec[j].back().sequence.tokens.push_back(tk);
                            bc_per_dec[j].back().sequence.sum_logprobs_all += tk.plog;
                        }
                    }
                }
                beam_candidates.clear();
                for (const auto &bc : bc_per_dec) beam_candidates.insert(beam_candidates.end(), bc.begin(), bc.end());
                if (params.strategy == 1) {
                    std::sort(beam_candidates.begin(), beam_candidates.end(),
                              [](const beam_candidate &a, const beam_candidate &b) {
                                  return a.sequence.sum_logprobs_all > b.sequence.sum_logprobs_all;
                              });
                    uint32_t cur_c = 0;
                    std::vector<int> src(n_decoders_cur, -1);
                    for (int j = 0; j < n_decoders_cur; ++j) {
                        Decoder &d = c.decoders[j];
                        if (d.completed || d.failed) continue;
                        if (cur_c >= beam_candidates.size()) cur_c = 0;
                        auto &cur = beam_candidates[cur_c++];
                        while (beam_candidates.size() > cur_c &&
                               seq_tokens_equal(beam_candidates[cur_c].sequence, cur.sequence) && i > 0)
                            ++cur_c;
                        d.seek_delta = cur.seek_delta;
                        d.has_ts = cur.has_ts;
                        d.sequence = cur.sequence;
                        src[j] = cur.decoder_idx;
                        kv_seq_cp(c, cur.decoder_idx, MAX_DECODERS + j);
                    }
                    for (int j = 0; j < n_decoders_cur; ++j) {
                        Decoder &d = c.decoders[j];
                        if (d.completed || d.failed) continue;
                        kv_seq_cp(c, MAX_DECODERS + j, j);
                    }
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
                        const bool forced = fds > 0 && i + 1 < fds;  // bench-only: keep decoding
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
                                seek_delta = 100 * CHUNK_SIZE;
                            }
                            completed = true;
                            continue;
                        }
                    }
                    if (i == n_max - 1 && (result_len == 0 || seek_delta < 100 * CHUNK_SIZE / 2)) {
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
                    std::vector<int> bt, bp, bs;
                    const int n_past = (int)prompt.size() + i;
                    for (int j = 0; j < n_decoders_cur; ++j) {
                        Decoder &d = c.decoders[j];
                        if (d.failed || d.completed) continue;
                        d.i_batch = (int)bt.size();
                        bt.push_back(d.sequence.tokens.back().id);
                        bp.push_back(n_past);
                        bs.push_back(j);
                    }
                    if (!decode(c, bt, bp, bs)) return -8;
                    for (int j = 0; j < n_decoders_cur; ++j) {
                        Decoder &d = c.decoders[j];
                        if (d.failed || d.completed) continue;
                        process_logits(c, params, t_cur, &c.logits[(size_t)d.i_batch * hp.n_vocab], d.sequence.tokens,
                                       d.has_ts, d.seek_delta, d.logits, d.logprobs, d.probs);
                    }
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
                }
            }
            // forced decode length (bench-only extension): the window always advances as a whole, so one call over
            // 30 s of audio is exactly one encoder pass + fds decode steps
            seek += fds > 0 ? 100 * CHUNK_SIZE : seek_delta;
        }
    }
    return 0;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// C interface
// ---------------------------------------------------------------------------------------------
extern "C" {

ow_context *ow_init_from_file(const char *path) {
    ow_context *c = new ow_context();
    if (!load_model(path, c->model)) {
        delete c;
        return nullptr;
    }
    kv_init(*c);
    c->decoders[0].rng = std::mt19937(0);
    c->n_ctx_cur = c->model.hp.n_audio_ctx;
    return c;
}
void ow_free(ow_context *c) { delete c; }
void ow_set_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}
void ow_hparams(ow_context *c, int32_t out[11]) {
    const HP &h = c->model.hp;
    const int v[11] = {h.n_vocab, h.n_audio_ctx, h.n_audio_state, h.n_audio_head, h.n_audio_layer, h.n_text_ctx,
                       h.n_text_state, h.n_text_head, h.n_text_layer, h.n_mels, h.ftype};
    for (int i = 0; i < 11; ++i) out[i] = v[i];
}
void ow_special_tokens(ow_context *c, int32_t out[9]) {
    const Vocab &v = c->model.vocab;
    const int s[9] = {v.eot, v.sot, v.translate, v.transcribe, v.solm, v.prev, v.nosp, v.not_, v.beg};
    for (int i = 0; i < 9; ++i) out[i] = s[i];
}
const char *ow_token_to_str(ow_context *c, int id) {
    if (id < 0 || id >= (int)c->model.vocab.id_to_token.size()) return "";
    return c->model.vocab.id_to_token[id].c_str();
}
int ow_tokenize(ow_context *c, const char *text, int32_t *out, int n_max) {
    auto t = tokenize(c->model.vocab, text);
    if ((int)t.size() > n_max) return -(int)t.size();
    for (size_t i = 0; i < t.size(); ++i) out[i] = t[i];
    return (int)t.size();
}
ow_params ow_default_params(int strategy) {
    ow_params p;
    memset(&p, 0, sizeof(p));
    p.strategy = strategy;
    p.n_threads = 4;
    p.n_max_text_ctx = 16384;
    p.no_context = 1;
    p.suppress_blank = 1;
    p.temperature = 0.0f;
    p.max_initial_ts = 1.0f;
    p.length_penalty = -1.0f;
    p.temperature_inc = 0.2f;
    p.entropy_thold = 2.4f;
    p.logprob_thold = -1.0f;
    p.no_speech_thold = 0.6f;
    p.best_of = strategy == 0 ? 5 : -1;
    p.beam_size = strategy == 1 ? 5 : -1;
    p.patience = -1.0f;
    p.language = "en";
    return p;
}
int ow_pcm_to_mel(ow_context *c, const float *pcm, int n) {
    pcm_to_mel(c->model, pcm, n, c->mel);
    return 0;
}
int ow_mel_n_len(ow_context *c) { return c->mel.n_len; }
int ow_mel_n_len_org(ow_context *c) { return c->mel.n_len_org; }
const float *ow_mel_data(ow_context *c) { return c->mel.data.data(); }
void ow_set_mx_encoder(ow_context *c, int on) { c->mx_encoder = on != 0; }
void ow_mx_qdq(float *a, int rows, int K) { mx_qdq_rows(a, (size_t)rows, K, (size_t)K); }
int ow_encode(ow_context *c, int seek, int audio_ctx) {
    c->exp_n_audio_ctx = audio_ctx;
    encode(*c, seek);
    return 0;
}
const float *ow_encoder_out(ow_context *c) { return c->enc_out.data(); }
const float *ow_cross_kv(ow_context *c, int layer, int is_v) {
    return is_v ? c->cross_v[layer].data() : c->cross_k[layer].data();
}
int ow_decode(ow_context *c, const int32_t *tokens, int n, int n_past, int seq) {
    std::vector<int> t(tokens, tokens + n), p(n), s(n, seq);
    for (int i = 0; i < n; ++i) p[i] = n_past + i;
    return decode(*c, t, p, s) ? 0 : -1;
}
const float *ow_logits(ow_context *c) { return c->logits.data(); }
int ow_process_logits(ow_context *c, const ow_params *p, float temperature, const float *logits_in, const int32_t *history,
                      int n_history, int has_ts, int seek_delta, float *logits_out, float *logprobs_out, float *probs_out) {
    std::vector<ow_token_data> hist(n_history);
    for (int i = 0; i < n_history; ++i) hist[i] = {history[i], 0, 0, 0, 0, 0, -1, -1, -1, 0};
    std::vector<float> lg, lp, pr;
    process_logits(*c, *p, temperature, logits_in, hist, has_ts != 0, seek_delta, lg, lp, pr);
    const size_t n = c->model.vocab.n_vocab;
    if (logits_out) memcpy(logits_out, lg.data(), n * 4);
    if (logprobs_out) memcpy(logprobs_out, lp.data(), n * 4);
    if (probs_out) memcpy(probs_out, pr.data(), n * 4);
    return 0;
}
int ow_full(ow_context *c, const ow_params *p, const float *pcm, int n) { return full(*c, *p, pcm, n); }
int ow_full_n_segments(ow_context *c) { return (int)c->result_all.size(); }
int64_t ow_full_get_segment_t0(ow_context *c, int i) { return c->result_all[i].t0; }
int64_t ow_full_get_segment_t1(ow_context *c, int i) { return c->result_all[i].t1; }
const char *ow_full_get_segment_text(ow_context *c, int i) { return c->result_all[i].text.c_str(); }
int ow_full_n_tokens(ow_context *c, int i) { return (int)c->result_all[i].tokens.size(); }
ow_token_data ow_full_get_token_data(ow_context *c, int i, int j) { return c->result_all[i].tokens[j]; }
void ow_full_get_token_data_p(ow_context *c, int i, int j, ow_token_data *out) { *out = c->result_all[i].tokens[j]; }
int ow_full_lang_id(ow_context *c) { return c->lang_id; }
int ow_lang_auto_detect(ow_context *c, int offset_ms, float *probs) { return lang_auto_detect(*c, offset_ms, probs); }
const char *ow_lang_str(int id) { return (id >= 0 && id < N_LANGS) ? k_langs[id] : nullptr; }
int ow_lang_id(const char *lang) {
    for (int i = 0; i < N_LANGS; ++i)
        if (strcmp(lang, k_langs[i]) == 0) return i;
    return -1;
}
int ow_n_decode_calls(ow_context *c) { return c->n_decode_calls; }
int ow_n_encode_calls(ow_context *c) { return c->n_encode_calls; }

}  // extern "C"
