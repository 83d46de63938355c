// silero_vad.hpp — the Silero VAD (v5) network of LocalVocal's VadIterator on the GPU.
//
// Replaces the ONNX Runtime session VadIterator::predict_one runs once per 32 ms window
// (/root/reference/src/whisper-utils/silero-vad-onnx.cpp:111-140; session set-up :47-93; model file
// /root/reference/data/models/silero-vad/silero_vad.onnx, named by the caller as in vad-processing.cpp:466-478).  The
// weights are read from that same .onnx file by a protobuf wire-format reader (nothing in the file is executed); the
// 16 kHz branch of the graph is restated as two kernels (silero_vad.hip) that score ALL windows of a VadIterator::process
// call in one pass: the window-parallel front end, then the recurrent cell over the windows in order.
#pragma once

#include <hip/hip_runtime.h>

#include <string>
#include <vector>

namespace mw {

struct SileroWeights {
    // shapes of the 16 kHz branch (row-major, as stored in the file)
    std::vector<float> basis;         // [258][256]      stft.forward_basis_buffer
    std::vector<float> c_w[4], c_b[4];  // encoder.k.reparam_conv: [128][129][3], [64][128][3], [64][64][3], [128][64][3]
    std::vector<float> w_ih, w_hh;    // [512][128]      decoder.rnn (stored gate order i, f, g, o)
    std::vector<float> b_ih, b_hh;    // [512]
    std::vector<float> out_w;         // [128]           decoder.decoder.2 (Conv1d 128 -> 1, k = 1)
    float out_b = 0.0f;
};

// Host only (no GPU needed).  Throws std::runtime_error naming what is wrong with the file.
SileroWeights load_silero_onnx(const std::string &path);

class SileroVad;
// one stream's share of a batched pass (SileroVad::process_batch)
struct SileroBatchItem {
    SileroVad *vad;      // the stream's network state (all items of a batch: the same weights, the same device)
    const float *wav;    // host
    int n_windows;
    bool reset;
    float *probs;        // host, n_windows
};

class SileroVad {
public:
    static constexpr int WINDOW = 512;  // 32 ms at 16 kHz (silero-vad-onnx.cpp:333-335)
    SileroVad(const SileroWeights &w, int device);
    ~SileroVad();
    SileroVad(const SileroVad &) = delete;
    SileroVad &operator=(const SileroVad &) = delete;
    // Probabilities of n_windows consecutive windows of `wav` (host memory); the recurrent state carries over from the
    // previous call unless `reset` (VadIterator::reset_states(true)).
    void process(const float *wav, int n_windows, bool reset, float *probs);
    void get_state(float *hc256);  // h[128], c[128]: the model's `state` tensor (tests)
    // The windows of SEVERAL streams in one pass (SURVEY.md §8 f2 "batched across streams"): one front-end launch over all
    // windows, one recurrent launch with a workgroup per stream, each on its own state.  Same results as process() per
    // stream.  The handles must not be in use by another thread during the call; launches go on items[0].vad's stream.
    static void process_batch(const SileroBatchItem *items, int n);
    static constexpr int MAX_BATCH_STREAMS = 256;

private:
    void reserve(int n_windows);
    int device_ = 0;
    hipStream_t st_ = nullptr;
    float *d_w_ = nullptr;  // all weights, one allocation (offsets below, in floats)
    size_t o_basis_ = 0, o_cw_[4] = {0, 0, 0, 0}, o_cb_[4] = {0, 0, 0, 0}, o_wih_ = 0, o_whh_ = 0, o_bg_ = 0, o_outw_ = 0;
    float out_b_ = 0.0f;
    float *d_wav_ = nullptr, *d_gx_ = nullptr, *d_probs_ = nullptr, *d_state_ = nullptr;
    float *h_pin_ = nullptr;  // pinned staging: windows in, probabilities out
    int cap_ = 0;
    unsigned long long weights_id_ = 0;  // FNV-1a over the weights: a batch shares the leader's copy
    void *d_desc_ = nullptr, *h_desc_ = nullptr;  // per-stream descriptors of a batched pass (device, pinned host)
};

}  // namespace mw
