"""Synthetic Silero-VAD-shaped model files (tests and development tools; the GPU box has no /root/reference).

`SUFFIXES`: tensor name suffix and shape of the 15 tensors of the 16 kHz branch of the Silero v5 graph
(/root/reference/data/models/silero-vad/silero_vad.onnx).  `synth_weights` draws weights of those shapes that make the
network behave like a VAD on synthetic audio; `write_onnx` writes them into a file with the real file's nesting
(Constant(16000), Equal, If{then: 16 kHz constants, else: 8 kHz shapes}) through tools/onnx_wire.py.
"""
from __future__ import annotations

import numpy as np

from tools import onnx_wire

SUFFIXES = {
    "basis": ("stft.forward_basis_buffer", (258, 1, 256)),
    "c0_w": ("encoder.0.reparam_conv.weight", (128, 129, 3)), "c0_b": ("encoder.0.reparam_conv.bias", (128,)),
    "c1_w": ("encoder.1.reparam_conv.weight", (64, 128, 3)), "c1_b": ("encoder.1.reparam_conv.bias", (64,)),
    "c2_w": ("encoder.2.reparam_conv.weight", (64, 64, 3)), "c2_b": ("encoder.2.reparam_conv.bias", (64,)),
    "c3_w": ("encoder.3.reparam_conv.weight", (128, 64, 3)), "c3_b": ("encoder.3.reparam_conv.bias", (128,)),
    "w_ih": ("decoder.rnn.weight_ih", (512, 128)), "w_hh": ("decoder.rnn.weight_hh", (512, 128)),
    "b_ih": ("decoder.rnn.bias_ih", (512,)), "b_hh": ("decoder.rnn.bias_hh", (512,)),
    "out_w": ("decoder.decoder.2.weight", (1, 128, 1)), "out_b": ("decoder.decoder.2.bias", (1,)),
}


def synth_weights(seed: int = 0, gain: float = 2.5) -> dict:
    """Random weights of the graph's shapes (the GPU box has no /root/reference), built so that the probability follows
    the window's energy the way a VAD's does (near 0.05 on silence, near 1 on loud input, everything between on the way):
    a windowed-Fourier-like basis, mostly positive convolutions, input and output gates closed at rest."""
    r = np.random.default_rng(seed)
    w = {}
    k = np.arange(256)
    win = 0.5 - 0.5 * np.cos(2 * np.pi * k / 256)
    f = np.arange(129)[:, None]
    basis = np.concatenate([np.cos(2 * np.pi * f * k / 256) * win, -np.sin(2 * np.pi * f * k / 256) * win], 0)
    w["basis"] = (basis[:, None, :] + 0.01 * r.standard_normal((258, 1, 256))).astype(np.float32)
    for i in range(4):
        shape = SUFFIXES[f"c{i}_w"][1]
        fan = shape[1] * 3
        w[f"c{i}_w"] = (np.abs(r.standard_normal(shape)) * (gain * 1.2 / fan) + 0.3 * r.standard_normal(shape) / np.sqrt(fan)).astype(np.float32)
        w[f"c{i}_b"] = (0.01 * r.standard_normal(shape[0])).astype(np.float32)
    w["w_ih"] = (0.15 * r.standard_normal((512, 128)) + 0.05).astype(np.float32)
    w["w_hh"] = (0.08 * r.standard_normal((512, 128))).astype(np.float32)
    b = 0.1 * r.standard_normal(512)
    b[0:128] -= 2.0
    b[384:512] -= 2.0
    w["b_ih"] = b.astype(np.float32)
    w["b_hh"] = (0.05 * r.standard_normal(512)).astype(np.float32)
    w["out_w"] = (0.12 + 0.05 * r.standard_normal((1, 128, 1))).astype(np.float32)
    w["out_b"] = np.array([-3.0], np.float32)
    return w


def write_onnx(path: str, w16: dict, seed: int = 1):
    """A file with the nesting of the real one: Constant(16000), Equal, If{then: 16 kHz constants, else: 8 kHz shapes}."""
    r = np.random.default_rng(seed)

    def branch(prefix, tensors):
        return {"name": prefix, "nodes": [{"op": "Constant", "input": [], "output": [prefix + "__Inline_0__" + SUFFIXES[k][0]],
                                           "attr": {"value": a}} for k, a in tensors.items()],
                "outputs": [prefix + "_out0", prefix + "_out1"]}

    shapes8 = {"basis": (130, 1, 128), "c0_w": (128, 65, 3)}
    w8 = {k: (0.1 * r.standard_normal(shapes8.get(k, SUFFIXES[k][1]))).astype(np.float32) for k in SUFFIXES}
    g = {"name": "synthetic_silero", "inputs": ["input", "state", "sr"], "outputs": ["output", "stateN"],
         "nodes": [{"op": "Constant", "input": [], "output": ["Constant_0_output"], "attr": {"value": np.array(16000, np.int64)}},
                   {"op": "Equal", "input": ["sr", "Constant_0_output"], "output": ["Equal_0_C"]},
                   {"op": "If", "input": ["Equal_0_C"], "output": ["output", "stateN"],
                    "attr": {"else_branch": branch("If_0_else_branch", w8), "then_branch": branch("If_0_then_branch", w16)}}]}
    onnx_wire.save(path, g)
