"""CPU restatement of the Silero VAD (v5) network LocalVocal runs per 32 ms window — TEST INFRASTRUCTURE ONLY.

Only tests/ may import this module: it is the checker for obs-localvocal_amd/csrc/silero_vad.hip, never the product.

What it restates: the 16 kHz branch of the graph in /root/reference/data/models/silero-vad/silero_vad.onnx (inputs
`input` [1, 512], `state` [2, 1, 128], `sr`; outputs `output` [1, 1], `stateN`), called window by window by
VadIterator::predict_one (/root/reference/src/whisper-utils/silero-vad-onnx.cpp:111-140) from VadIterator::process
(:262-284, consecutive non-overlapping windows of window_size_samples = 32 ms * 16 = 512, :333-335).  Read off the
graph's nodes (tools/onnx_wire.py walks the protobuf; nothing in the file is executed):

  Pad(reflect, 64 samples on the right) -> Conv(stft.forward_basis_buffer [258, 1, 256], stride 128) -> 3 frames
  -> sqrt(re^2 + im^2) of channels [0, 129) and [129, 258)
  -> encoder.{0,1,2,3}.reparam_conv: Conv1d k = 3, pad 1, strides 1, 2, 2, 1 (129 -> 128 -> 64 -> 64 -> 128), ReLU each
     (3 -> 3 -> 2 -> 1 -> 1 frames)
  -> LSTM cell on the single remaining frame with h = state[0], c = state[1]; the graph re-orders decoder.rnn.weight_*
     rows [0:128], [384:512], [128:384] into ONNX's i, o, f, c order, i.e. the stored order is i, f, g, o
  -> ReLU(h') -> Conv1d(128 -> 1, k = 1) -> Sigmoid -> mean over the (single) frame = `output`; stateN = (h', c').

PARITY UNPINNED: ONNX Runtime is not in this image and the reference holds no expected probabilities for any input,
so this restatement is checked only for internal consistency (tests/test_silero_cpu.py) and the HIP path is checked
against it; that shows two readings of the same graph agree, not that either equals an ONNX Runtime session.
"""
from __future__ import annotations

import numpy as np

from tools import onnx_wire

WINDOW = 512
from tools.silero_synth import SUFFIXES, synth_weights, write_onnx  # noqa: E402,F401  (tests take them from here)


def _constants(graph, found):
    """Every tensor the graph (and its sub-graphs) holds as an initializer or a Constant node's value, by name."""
    for k, a in graph["init"].items():
        found.setdefault(k, a)
    for n in graph["nodes"]:
        if n["op"] == "Constant" and "value" in n["attr"] and n["output"]:
            found.setdefault(n["output"][0], n["attr"]["value"])
        for v in n["attr"].values():
            if isinstance(v, dict):
                _constants(v, found)
    return found


def load_weights(path: str) -> dict:
    """The 16 kHz branch's tensors: the branch is the one whose stft.forward_basis_buffer is [258, 1, 256] (the 8 kHz branch
    holds the same names under another prefix); the other tensors are taken under that tensor's name prefix."""
    consts = _constants(onnx_wire.load(path), {})
    b_suffix, b_shape = SUFFIXES["basis"]
    prefixes = [k[:-len(b_suffix)] for k, a in consts.items() if k.endswith(b_suffix) and tuple(a.shape) == b_shape]
    if len(prefixes) != 1:
        raise ValueError(f"{path}: {len(prefixes)} tensors named *{b_suffix} of shape {b_shape}")
    w = {}
    for key, (suffix, shape) in SUFFIXES.items():
        a = consts.get(prefixes[0] + suffix)
        if a is None or tuple(a.shape) != shape:
            raise ValueError(f"{path}: no tensor {prefixes[0] + suffix} of shape {shape}")
        w[key] = np.ascontiguousarray(a, dtype=np.float32)
    return w


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def _conv3(x, w, b, stride):
    """Conv1d, kernel 3, zero padding 1.  x [C_in, T] -> [C_out, T_out], float32 accumulation in float64 order-free form."""
    cin, t = x.shape
    xp = np.zeros((cin, t + 2), np.float32)
    xp[:, 1:t + 1] = x
    t_out = (t + 2 - 3) // stride + 1
    out = np.empty((w.shape[0], t_out), np.float32)
    for j in range(t_out):
        seg = xp[:, stride * j:stride * j + 3]                      # [C_in, 3]
        out[:, j] = (np.einsum("oik,ik->o", w.astype(np.float64), seg.astype(np.float64)) + b).astype(np.float32)
    return out


def frontend(w: dict, window: np.ndarray) -> np.ndarray:
    """One 512-sample window -> the 128 encoder features the LSTM cell sees."""
    x = np.asarray(window, np.float32)
    assert x.shape == (WINDOW,)
    xp = np.concatenate([x, x[-2:-66:-1]])                           # reflect: xp[512 + i] = x[510 - i], i < 64
    basis = w["basis"][:, 0, :].astype(np.float64)                   # [258, 256]
    frames = np.stack([xp[128 * t:128 * t + 256] for t in range(3)], 1).astype(np.float64)  # [256, 3]
    spec = (basis @ frames).astype(np.float32)                       # [258, 3]
    mag = np.sqrt(spec[:129] ** 2 + spec[129:] ** 2).astype(np.float32)
    y = np.maximum(_conv3(mag, w["c0_w"], w["c0_b"], 1), 0.0)        # [128, 3]
    y = np.maximum(_conv3(y, w["c1_w"], w["c1_b"], 2), 0.0)          # [64, 2]
    y = np.maximum(_conv3(y, w["c2_w"], w["c2_b"], 2), 0.0)          # [64, 1]
    y = np.maximum(_conv3(y, w["c3_w"], w["c3_b"], 1), 0.0)          # [128, 1]
    return y[:, 0].astype(np.float32)


def step(w: dict, feat: np.ndarray, h: np.ndarray, c: np.ndarray):
    """LSTM cell (stored gate order i, f, g, o) + the 128 -> 1 head.  Returns (probability, h', c')."""
    g = (w["w_ih"].astype(np.float64) @ feat + w["b_ih"] + w["w_hh"].astype(np.float64) @ h + w["b_hh"]).astype(np.float32)
    i, f, gg, o = _sigmoid(g[0:128]), _sigmoid(g[128:256]), np.tanh(g[256:384]), _sigmoid(g[384:512])
    c2 = (f * c + i * gg).astype(np.float32)
    h2 = (o * np.tanh(c2)).astype(np.float32)
    z = float(w["out_w"][0, :, 0].astype(np.float64) @ np.maximum(h2, 0.0) + w["out_b"][0])
    return float(np.float32(_sigmoid(z))), h2, c2


class SileroOracle:
    """predict_one's contract: probability of one window, recurrent state carried between calls, zeroed on reset."""

    def __init__(self, path_or_weights):
        self.w = load_weights(path_or_weights) if isinstance(path_or_weights, str) else path_or_weights
        self.reset()

    def reset(self):
        self.h = np.zeros(128, np.float32)
        self.c = np.zeros(128, np.float32)

    def predict_one(self, window: np.ndarray) -> float:
        p, self.h, self.c = step(self.w, frontend(self.w, window), self.h, self.c)
        return p

    def process(self, wav: np.ndarray, reset: bool = True) -> np.ndarray:
        if reset:
            self.reset()
        n = len(wav) // WINDOW
        return np.array([self.predict_one(wav[WINDOW * j:WINDOW * (j + 1)]) for j in range(n)], np.float32)
