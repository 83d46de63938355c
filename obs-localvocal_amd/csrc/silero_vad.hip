// silero_vad.hip — Silero VAD (v5, 16 kHz branch) on gfx950: ONNX weight reader + two kernels.  See silero_vad.hpp.
//
// The graph, as read off /root/reference/data/models/silero-vad/silero_vad.onnx (oracle/silero_vad_oracle.py restates it
// on the CPU and is what tests/test_gpu_silero.py checks this file against):
//   Pad(reflect, 64 right) -> Conv(basis [258][256], stride 128): 3 frames -> |.| of (re, im) = channels (c, 129 + c)
//   -> 4 x [Conv1d k 3 pad 1, strides 1 2 2 1, ReLU]: 129 x 3 -> 128 x 3 -> 64 x 2 -> 64 x 1 -> 128 x 1
//   -> LSTM cell (h, c = state; gate rows i, f, g, o) -> ReLU -> 128 -> 1 -> sigmoid.
// One VadIterator::process call scores up to a few hundred windows (10 s = 312): the convolutional front end and the
// input half of the gates (W_ih x + b) have no dependence between windows and run as one workgroup per window; only the
// 128 x 512 recurrent product is sequential, and it runs in ONE workgroup whose 512 threads each keep a row of W_hh in
// registers, so a step costs two LDS barriers and no global weight traffic.  Everything is fp32 (a 0.3 M-parameter model).
#include "silero_vad.hpp"

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <stdexcept>

#include "common.hpp"

namespace mw {

// ------------------------------------------------------------------------------------------
// ONNX protobuf wire format: just enough to find float tensors by name (public onnx.proto3 field numbers)
//   ModelProto.graph = 7;  GraphProto.node = 1, initializer = 5;  NodeProto.output = 2, op_type = 4, attribute = 5;
//   AttributeProto.name = 1, t = 5, g = 6;  TensorProto.dims = 1, data_type = 2, float_data = 4, name = 8, raw_data = 9.
// ------------------------------------------------------------------------------------------
namespace {

struct Span {
    const uint8_t *p, *e;
};
struct Field {
    uint32_t num, wt;
    uint64_t v;  // varint value
    Span s;      // length-delimited payload / fixed bytes
};
struct Tensor {
    std::vector<int64_t> dims;
    std::vector<float> data;
};
using TensorMap = std::map<std::string, Tensor>;

[[noreturn]] void bad(const char *what) { throw std::runtime_error(std::string("silero onnx: ") + what); }

uint64_t varint(Span &s) {
    uint64_t v = 0;
    for (int sh = 0; sh < 64; sh += 7) {
        if (s.p >= s.e) bad("truncated varint");
        const uint8_t c = *s.p++;
        v |= (uint64_t)(c & 0x7f) << sh;
        if (c < 0x80) return v;
    }
    bad("varint too long");
}

bool next_field(Span &s, Field &f) {
    if (s.p >= s.e) return false;
    const uint64_t key = varint(s);
    f.num = (uint32_t)(key >> 3);
    f.wt = (uint32_t)(key & 7);
    f.v = 0;
    f.s = Span{s.p, s.p};
    if (f.wt == 0) {
        f.v = varint(s);
    } else if (f.wt == 1 || f.wt == 5) {
        const size_t n = f.wt == 1 ? 8 : 4;
        if ((size_t)(s.e - s.p) < n) bad("truncated fixed field");
        f.s = Span{s.p, s.p + n};
        s.p += n;
    } else if (f.wt == 2) {
        const uint64_t n = varint(s);
        if (n > (uint64_t)(s.e - s.p)) bad("length-delimited field runs past its message");
        f.s = Span{s.p, s.p + n};
        s.p += n;
    } else {
        bad("unsupported wire type");
    }
    return true;
}

std::string str(const Span &s) { return std::string((const char *)s.p, (size_t)(s.e - s.p)); }

bool parse_tensor(Span s, std::string &name, Tensor &t) {  // false: not a float tensor
    Field f;
    uint64_t dtype = 0;
    Span raw{nullptr, nullptr};
    std::vector<float> fdata;
    while (next_field(s, f)) {
        if (f.num == 1) {
            if (f.wt == 2) {
                Span d = f.s;
                while (d.p < d.e) t.dims.push_back((int64_t)varint(d));
            } else {
                t.dims.push_back((int64_t)f.v);
            }
        } else if (f.num == 2) {
            dtype = f.v;
        } else if (f.num == 4) {
            const size_t n = (size_t)(f.s.e - f.s.p) / 4;
            const size_t at = fdata.size();
            fdata.resize(at + n);
            memcpy(fdata.data() + at, f.s.p, n * 4);
        } else if (f.num == 8) {
            name = str(f.s);
        } else if (f.num == 9) {
            raw = f.s;
        }
    }
    if (dtype != 1) return false;  // FLOAT
    size_t count = 1;
    for (int64_t d : t.dims) {
        if (d < 0 || d > (1 << 24) || (count *= (size_t)d) > ((size_t)1 << 26)) bad("tensor dimensions out of range");
    }
    if (raw.p) {
        if ((size_t)(raw.e - raw.p) != count * 4) bad("raw_data size does not match the tensor's dimensions");
        t.data.resize(count);
        memcpy(t.data.data(), raw.p, count * 4);  // little-endian floats on a little-endian host
    } else {
        if (fdata.size() != count) bad("float_data size does not match the tensor's dimensions");
        t.data.swap(fdata);
    }
    return true;
}

void parse_graph(Span s, TensorMap &out, int depth);

void parse_node(Span s, TensorMap &out, int depth) {
    Field f;
    std::string op, first_output;
    std::vector<Span> attrs;
    while (next_field(s, f)) {
        if (f.num == 2 && first_output.empty()) first_output = str(f.s);
        else if (f.num == 4) op = str(f.s);
        else if (f.num == 5) attrs.push_back(f.s);
    }
    for (Span a : attrs) {
        Field g;
        std::string aname;
        Span t{nullptr, nullptr}, sub{nullptr, nullptr};
        while (next_field(a, g)) {
            if (g.num == 1) aname = str(g.s);
            else if (g.num == 5) t = g.s;
            else if (g.num == 6) sub = g.s;
        }
        if (op == "Constant" && aname == "value" && t.p && !first_output.empty()) {
            std::string tn;
            Tensor tt;
            if (parse_tensor(t, tn, tt)) out.emplace(first_output, std::move(tt));
        }
        if (sub.p) parse_graph(sub, out, depth + 1);
    }
}

void parse_graph(Span s, TensorMap &out, int depth) {
    if (depth > 16) bad("sub-graphs nested too deeply");
    Field f;
    while (next_field(s, f)) {
        if (f.num == 1 && f.wt == 2) {
            parse_node(f.s, out, depth);
        } else if (f.num == 5 && f.wt == 2) {
            std::string tn;
            Tensor tt;
            if (parse_tensor(f.s, tn, tt) && !tn.empty()) out.emplace(tn, std::move(tt));
        }
    }
}

bool ends_with(const std::string &s, const std::string &suffix) {
    return s.size() >= suffix.size() && s.compare(s.size() - suffix.size(), suffix.size(), suffix) == 0;
}

std::vector<float> take(const TensorMap &m, const std::string &name, std::initializer_list<int64_t> dims) {
    auto it = m.find(name);
    if (it == m.end()) bad(("tensor not found: " + name).c_str());
    if (it->second.dims != std::vector<int64_t>(dims)) bad(("unexpected shape of " + name).c_str());
    return it->second.data;
}

}  // namespace

SileroWeights load_silero_onnx(const std::string &path) {
    FILE *fp = fopen(path.c_str(), "rb");
    if (!fp) bad(("cannot open " + path).c_str());
    std::vector<uint8_t> buf;
    {
        uint8_t chunk[1 << 16];
        size_t n;
        while ((n = fread(chunk, 1, sizeof(chunk), fp)) > 0) {
            buf.insert(buf.end(), chunk, chunk + n);
            if (buf.size() > ((size_t)64 << 20)) {
                fclose(fp);
                bad("file larger than 64 MB");
            }
        }
        fclose(fp);
    }
    TensorMap m;
    Span s{buf.data(), buf.data() + buf.size()};
    Field f;
    bool have_graph = false;
    while (next_field(s, f)) {
        if (f.num == 7 && f.wt == 2) {
            parse_graph(f.s, m, 0);
            have_graph = true;
        }
    }
    if (!have_graph) bad("no graph in the file");
    // the 16 kHz branch is the one whose STFT basis is [258][1][256]; its tensors share that tensor's name prefix
    const std::string basis_suffix = "stft.forward_basis_buffer";
    std::string prefix;
    int n_found = 0;
    for (const auto &kv : m) {
        if (ends_with(kv.first, basis_suffix) && kv.second.dims == std::vector<int64_t>{258, 1, 256}) {
            prefix = kv.first.substr(0, kv.first.size() - basis_suffix.size());
            n_found += 1;
        }
    }
    if (n_found != 1) bad("expected exactly one stft.forward_basis_buffer of shape [258, 1, 256]");
    SileroWeights w;
    w.basis = take(m, prefix + basis_suffix, {258, 1, 256});
    const int cin[4] = {129, 128, 64, 64}, cout[4] = {128, 64, 64, 128};
    for (int k = 0; k < 4; ++k) {
        const std::string base = prefix + "encoder." + std::to_string(k) + ".reparam_conv.";
        w.c_w[k] = take(m, base + "weight", {cout[k], cin[k], 3});
        w.c_b[k] = take(m, base + "bias", {cout[k]});
    }
    w.w_ih = take(m, prefix + "decoder.rnn.weight_ih", {512, 128});
    w.w_hh = take(m, prefix + "decoder.rnn.weight_hh", {512, 128});
    w.b_ih = take(m, prefix + "decoder.rnn.bias_ih", {512});
    w.b_hh = take(m, prefix + "decoder.rnn.bias_hh", {512});
    w.out_w = take(m, prefix + "decoder.decoder.2.weight", {1, 128, 1});
    w.out_b = take(m, prefix + "decoder.decoder.2.bias", {1})[0];
    return w;
}

// ------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------
namespace {

struct FrontW {  // device pointers, every matrix transposed so that consecutive threads read consecutive floats
    const float *basisT;            // [256][258]
    const float *cwT[4], *cb[4];    // [(i * 3 + k)][o]
    const float *w_ihT;             // [128][512]
    const float *bg;                // [512]  b_ih + b_hh
};

// one window per workgroup: 512 samples -> gx[512] = W_ih . features + b_ih + b_hh
__global__ __launch_bounds__(256) void silero_frontend_kernel(const float *__restrict__ wav, FrontW W, float *__restrict__ gx, int n_windows) {
    __shared__ float xp[576], spec[258 * 3], mag[129 * 3], y0[128 * 3], y1[64 * 2], y2[64], y3[128];
    const int tid = threadIdx.x, w = blockIdx.x;
    if (w >= n_windows) return;
    const float *x = wav + (size_t)w * 512;
    for (int i = tid; i < 576; i += 256) xp[i] = i < 512 ? x[i] : x[510 - (i - 512)];  // reflect: 64 samples on the right
    __syncthreads();
    for (int c = tid; c < 258; c += 256) {
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
#pragma unroll 8
        for (int k = 0; k < 256; ++k) {
            const float b = W.basisT[k * 258 + c];
            a0 = fmaf(b, xp[k], a0);
            a1 = fmaf(b, xp[128 + k], a1);
            a2 = fmaf(b, xp[256 + k], a2);
        }
        spec[c * 3] = a0;
        spec[c * 3 + 1] = a1;
        spec[c * 3 + 2] = a2;
    }
    __syncthreads();
    for (int i = tid; i < 129 * 3; i += 256) {
        const float re = spec[i], im = spec[129 * 3 + i];
        mag[i] = sqrtf(re * re + im * im);
    }
    __syncthreads();
    if (tid < 128) {  // encoder.0: 129 -> 128, stride 1, 3 -> 3 frames
        const float b = W.cb[0][tid];
        float a[3] = {b, b, b};
        for (int i = 0; i < 129; ++i) {
            const float w0 = W.cwT[0][(i * 3) * 128 + tid], w1 = W.cwT[0][(i * 3 + 1) * 128 + tid], w2 = W.cwT[0][(i * 3 + 2) * 128 + tid];
            const float m0 = mag[i * 3], m1 = mag[i * 3 + 1], m2 = mag[i * 3 + 2];
            a[0] = fmaf(w1, m0, fmaf(w2, m1, a[0]));
            a[1] = fmaf(w0, m0, fmaf(w1, m1, fmaf(w2, m2, a[1])));
            a[2] = fmaf(w0, m1, fmaf(w1, m2, a[2]));
        }
        for (int t = 0; t < 3; ++t) y0[tid * 3 + t] = fmaxf(a[t], 0.0f);
    }
    __syncthreads();
    if (tid < 128) {  // encoder.1: 128 -> 64, stride 2, 3 -> 2 frames: frame t reads inputs 2t - 1, 2t, 2t + 1
        const int o = tid & 63, t = tid >> 6;
        float a = W.cb[1][o];
        for (int i = 0; i < 128; ++i) {
            const float w0 = W.cwT[1][(i * 3) * 64 + o], w1 = W.cwT[1][(i * 3 + 1) * 64 + o], w2 = W.cwT[1][(i * 3 + 2) * 64 + o];
            if (t == 0) a = fmaf(w1, y0[i * 3], fmaf(w2, y0[i * 3 + 1], a));
            else a = fmaf(w0, y0[i * 3 + 1], fmaf(w1, y0[i * 3 + 2], a));
        }
        y1[o * 2 + t] = fmaxf(a, 0.0f);
    }
    __syncthreads();
    if (tid < 64) {  // encoder.2: 64 -> 64, stride 2, 2 -> 1 frame: reads inputs -1 (padding), 0, 1
        float a = W.cb[2][tid];
        for (int i = 0; i < 64; ++i) a = fmaf(W.cwT[2][(i * 3 + 1) * 64 + tid], y1[i * 2], fmaf(W.cwT[2][(i * 3 + 2) * 64 + tid], y1[i * 2 + 1], a));
        y2[tid] = fmaxf(a, 0.0f);
    }
    __syncthreads();
    if (tid < 128) {  // encoder.3: 64 -> 128, stride 1, 1 frame: only the middle tap meets data
        float a = W.cb[3][tid];
        for (int i = 0; i < 64; ++i) a = fmaf(W.cwT[3][(i * 3 + 1) * 128 + tid], y2[i], a);
        y3[tid] = fmaxf(a, 0.0f);
    }
    __syncthreads();
    for (int j = tid; j < 512; j += 256) {
        float a = W.bg[j];
#pragma unroll 8
        for (int k = 0; k < 128; ++k) a = fmaf(W.w_ihT[k * 512 + j], y3[k], a);
        gx[(size_t)w * 512 + j] = a;
    }
}

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

struct StreamDesc {  // one stream of a recurrent launch: its windows inside gx / probs and its state
    float *state;
    int win0, n_windows;
};

// the recurrent cell over a stream's windows in order: ONE workgroup per stream (blockIdx.x; `many` null: the single stream
// `one`), thread j = gate row j (i: 0..127, f, g, o)
__global__ __launch_bounds__(512) void silero_lstm_kernel(const float *__restrict__ gx, const float *__restrict__ w_hhT,
                                                         const float *__restrict__ out_w, float out_b, StreamDesc one,
                                                         const StreamDesc *__restrict__ many, float *__restrict__ probs) {
    __shared__ float hs[128], ga[512], red[2];
    const int j = threadIdx.x;
    const StreamDesc D = many ? many[blockIdx.x] : one;
    float *__restrict__ state = D.state;
    const int n_windows = D.n_windows;
    gx += (size_t)D.win0 * 512;
    probs += D.win0;
    float wr[128];
#pragma unroll
    for (int k = 0; k < 128; ++k) wr[k] = w_hhT[k * 512 + j];
    float c = 0.0f, ow = 0.0f;
    if (j < 128) {
        hs[j] = state[j];
        c = state[128 + j];
        ow = out_w[j];
    }
    float g_next = n_windows > 0 ? gx[j] : 0.0f;
    __syncthreads();
    for (int t = 0; t < n_windows; ++t) {  // bounded by the argument: every wave leaves the loop together
        float g = g_next;
        g_next = gx[(size_t)min(t + 1, n_windows - 1) * 512 + j];
#pragma unroll
        for (int k = 0; k < 128; ++k) g = fmaf(wr[k], hs[k], g);
        ga[j] = (j >> 7) == 2 ? tanhf(g) : sigmoid_f(g);
        __syncthreads();
        if (j < 128) {
            c = fmaf(ga[128 + j], c, ga[j] * ga[256 + j]);
            const float h = ga[384 + j] * tanhf(c);
            hs[j] = h;
            float v = fmaxf(h, 0.0f) * ow;
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
            if ((j & 63) == 0) red[j >> 6] = v;
        }
        __syncthreads();
        if (j == 0) probs[t] = sigmoid_f(red[0] + red[1] + out_b);
    }
    if (j < 128) {
        state[j] = hs[j];
        state[128 + j] = c;
    }
}

std::vector<float> transpose(const std::vector<float> &a, int rows, int cols) {  // [rows][cols] -> [cols][rows]
    std::vector<float> t((size_t)rows * cols);
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) t[(size_t)c * rows + r] = a[(size_t)r * cols + c];
    return t;
}

}  // namespace

SileroVad::SileroVad(const SileroWeights &w, int device) : device_(device) {
    MW_CHECK(w.basis.size() == 258 * 256 && w.w_ih.size() == 512 * 128 && w.w_hh.size() == 512 * 128 && w.b_ih.size() == 512 &&
                 w.b_hh.size() == 512 && w.out_w.size() == 128,
             "silero: weight shapes");
    const int cin[4] = {129, 128, 64, 64}, cout[4] = {128, 64, 64, 128};
    for (int k = 0; k < 4; ++k)
        MW_CHECK(w.c_w[k].size() == (size_t)cout[k] * cin[k] * 3 && w.c_b[k].size() == (size_t)cout[k], "silero: encoder shapes");
    MW_HIP(hipSetDevice(device_));
    std::vector<float> all;
    auto put = [&](const std::vector<float> &v) {
        const size_t at = all.size();
        all.insert(all.end(), v.begin(), v.end());
        all.resize((all.size() + 63) / 64 * 64);  // 256-byte aligned pieces
        return at;
    };
    o_basis_ = put(transpose(w.basis, 258, 256));
    for (int k = 0; k < 4; ++k) {
        o_cw_[k] = put(transpose(w.c_w[k], cout[k], cin[k] * 3));
        o_cb_[k] = put(w.c_b[k]);
    }
    o_wih_ = put(transpose(w.w_ih, 512, 128));
    o_whh_ = put(transpose(w.w_hh, 512, 128));
    std::vector<float> bg(512);
    for (int j = 0; j < 512; ++j) bg[j] = w.b_ih[j] + w.b_hh[j];
    o_bg_ = put(bg);
    o_outw_ = put(w.out_w);
    out_b_ = w.out_b;
    weights_id_ = 1469598103934665603ull;
    for (float v : all) {
        unsigned u;
        memcpy(&u, &v, 4);
        weights_id_ = (weights_id_ ^ u) * 1099511628211ull;
    }
    {
        unsigned u;
        memcpy(&u, &out_b_, 4);
        weights_id_ = (weights_id_ ^ u) * 1099511628211ull;
    }
    MW_HIP(hipStreamCreateWithFlags(&st_, hipStreamNonBlocking));
    MW_HIP(hipMalloc(&d_desc_, MAX_BATCH_STREAMS * sizeof(StreamDesc)));
    MW_HIP(hipHostMalloc(&h_desc_, MAX_BATCH_STREAMS * sizeof(StreamDesc)));
    MW_HIP(hipMalloc((void **)&d_w_, all.size() * sizeof(float)));
    MW_HIP(hipMemcpy(d_w_, all.data(), all.size() * sizeof(float), hipMemcpyHostToDevice));
    MW_HIP(hipMalloc((void **)&d_state_, 256 * sizeof(float)));
    MW_HIP(hipMemset(d_state_, 0, 256 * sizeof(float)));
}

SileroVad::~SileroVad() {
    (void)hipSetDevice(device_);
    if (st_) (void)hipStreamSynchronize(st_);
    (void)hipFree(d_w_);
    (void)hipFree(d_state_);
    (void)hipFree(d_wav_);
    (void)hipFree(d_gx_);
    (void)hipFree(d_probs_);
    (void)hipFree(d_desc_);
    if (h_desc_) (void)hipHostFree(h_desc_);
    if (h_pin_) (void)hipHostFree(h_pin_);
    if (st_) (void)hipStreamDestroy(st_);
}

void SileroVad::reserve(int n_windows) {
    if (n_windows <= cap_) return;
    const int cap = std::max(n_windows, std::max(2 * cap_, 512));
    (void)hipFree(d_wav_);
    (void)hipFree(d_gx_);
    (void)hipFree(d_probs_);
    if (h_pin_) (void)hipHostFree(h_pin_);
    d_wav_ = d_gx_ = d_probs_ = h_pin_ = nullptr;
    cap_ = 0;
    MW_HIP(hipMalloc((void **)&d_wav_, (size_t)cap * WINDOW * sizeof(float)));
    MW_HIP(hipMalloc((void **)&d_gx_, (size_t)cap * 512 * sizeof(float)));
    MW_HIP(hipMalloc((void **)&d_probs_, (size_t)cap * sizeof(float)));
    MW_HIP(hipHostMalloc((void **)&h_pin_, (size_t)cap * (WINDOW + 1) * sizeof(float)));
    cap_ = cap;
}

void SileroVad::process(const float *wav, int n_windows, bool reset, float *probs) {
    MW_CHECK(n_windows >= 0 && (n_windows == 0 || (wav && probs)), "silero: process arguments");
    MW_HIP(hipSetDevice(device_));
    if (reset) MW_HIP(hipMemsetAsync(d_state_, 0, 256 * sizeof(float), st_));
    if (n_windows == 0) {
        MW_HIP(hipStreamSynchronize(st_));
        return;
    }
    reserve(n_windows);
    const size_t n = (size_t)n_windows * WINDOW;
    memcpy(h_pin_, wav, n * sizeof(float));
    MW_HIP(hipMemcpyAsync(d_wav_, h_pin_, n * sizeof(float), hipMemcpyHostToDevice, st_));
    FrontW W;
    W.basisT = d_w_ + o_basis_;
    for (int k = 0; k < 4; ++k) {
        W.cwT[k] = d_w_ + o_cw_[k];
        W.cb[k] = d_w_ + o_cb_[k];
    }
    W.w_ihT = d_w_ + o_wih_;
    W.bg = d_w_ + o_bg_;
    hipLaunchKernelGGL(silero_frontend_kernel, dim3(n_windows), dim3(256), 0, st_, (const float *)d_wav_, W, d_gx_, n_windows);
    MW_HIP(hipGetLastError());
    hipLaunchKernelGGL(silero_lstm_kernel, dim3(1), dim3(512), 0, st_, (const float *)d_gx_, (const float *)(d_w_ + o_whh_),
                       (const float *)(d_w_ + o_outw_), out_b_, StreamDesc{d_state_, 0, n_windows}, (const StreamDesc *)nullptr, d_probs_);
    MW_HIP(hipGetLastError());
    float *h_probs = h_pin_ + (size_t)cap_ * WINDOW;
    MW_HIP(hipMemcpyAsync(h_probs, d_probs_, (size_t)n_windows * sizeof(float), hipMemcpyDeviceToHost, st_));
    MW_HIP(hipStreamSynchronize(st_));
    memcpy(probs, h_probs, (size_t)n_windows * sizeof(float));
}

void SileroVad::process_batch(const SileroBatchItem *items, int n) {
    MW_CHECK(items && n >= 1 && n <= MAX_BATCH_STREAMS, "silero: batch of 1..256 streams");
    SileroVad &L = *items[0].vad;  // the leader: its stream, staging and weights serve the whole batch
    long total = 0;
    for (int i = 0; i < n; ++i) {
        const SileroBatchItem &it = items[i];
        MW_CHECK(it.vad && it.n_windows >= 0 && (it.n_windows == 0 || (it.wav && it.probs)), "silero: batch item");
        MW_CHECK(it.vad->device_ == L.device_ && it.vad->weights_id_ == L.weights_id_, "silero: a batch shares one model on one device");
        for (int k = 0; k < i; ++k) MW_CHECK(items[k].vad != it.vad, "silero: a stream appears once per batch");
        total += it.n_windows;
    }
    MW_CHECK(total <= (1 << 24), "silero: batch too large");
    MW_HIP(hipSetDevice(L.device_));
    for (int i = 0; i < n; ++i) {
        // another stream's own launches (an earlier process() on its own HIP stream) must be done with its state
        if (items[i].vad != &L) MW_HIP(hipStreamSynchronize(items[i].vad->st_));
        if (items[i].reset) MW_HIP(hipMemsetAsync(items[i].vad->d_state_, 0, 256 * sizeof(float), L.st_));
    }
    if (total == 0) {
        MW_HIP(hipStreamSynchronize(L.st_));
        return;
    }
    L.reserve((int)total);
    StreamDesc *hd = static_cast<StreamDesc *>(L.h_desc_);
    long at = 0;
    for (int i = 0; i < n; ++i) {
        hd[i] = StreamDesc{items[i].vad->d_state_, (int)at, items[i].n_windows};
        if (items[i].n_windows) memcpy(L.h_pin_ + at * WINDOW, items[i].wav, (size_t)items[i].n_windows * WINDOW * sizeof(float));
        at += items[i].n_windows;
    }
    MW_HIP(hipMemcpyAsync(L.d_wav_, L.h_pin_, (size_t)total * WINDOW * sizeof(float), hipMemcpyHostToDevice, L.st_));
    MW_HIP(hipMemcpyAsync(L.d_desc_, hd, (size_t)n * sizeof(StreamDesc), hipMemcpyHostToDevice, L.st_));
    FrontW W;
    W.basisT = L.d_w_ + L.o_basis_;
    for (int k = 0; k < 4; ++k) {
        W.cwT[k] = L.d_w_ + L.o_cw_[k];
        W.cb[k] = L.d_w_ + L.o_cb_[k];
    }
    W.w_ihT = L.d_w_ + L.o_wih_;
    W.bg = L.d_w_ + L.o_bg_;
    hipLaunchKernelGGL(silero_frontend_kernel, dim3((unsigned)total), dim3(256), 0, L.st_, (const float *)L.d_wav_, W, L.d_gx_, (int)total);
    MW_HIP(hipGetLastError());
    hipLaunchKernelGGL(silero_lstm_kernel, dim3(n), dim3(512), 0, L.st_, (const float *)L.d_gx_, (const float *)(L.d_w_ + L.o_whh_),
                       (const float *)(L.d_w_ + L.o_outw_), L.out_b_, StreamDesc{nullptr, 0, 0}, (const StreamDesc *)L.d_desc_, L.d_probs_);
    MW_HIP(hipGetLastError());
    float *h_probs = L.h_pin_ + (size_t)L.cap_ * WINDOW;
    MW_HIP(hipMemcpyAsync(h_probs, L.d_probs_, (size_t)total * sizeof(float), hipMemcpyDeviceToHost, L.st_));
    MW_HIP(hipStreamSynchronize(L.st_));
    at = 0;
    for (int i = 0; i < n; ++i) {
        if (items[i].n_windows) memcpy(items[i].probs, h_probs + at, (size_t)items[i].n_windows * sizeof(float));
        at += items[i].n_windows;
    }
}

void SileroVad::get_state(float *hc256) {
    MW_HIP(hipSetDevice(device_));
    MW_HIP(hipStreamSynchronize(st_));
    MW_HIP(hipMemcpy(hc256, d_state_, 256 * sizeof(float), hipMemcpyDeviceToHost));
}

}  // namespace mw
