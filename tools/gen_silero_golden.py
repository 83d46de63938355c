#!/usr/bin/env python3
"""Writes tests/golden/silero_v5_16k.npz: the 16 kHz branch's tensors of the reference's Silero VAD model file
(/root/reference/data/models/silero-vad/silero_vad.onnx, read with tools/onnx_wire.py — a protobuf wire reader that
executes nothing from the file) plus the probabilities the CPU restatement (oracle/silero_vad_oracle.py) gives on three
seeded clips.  Run in the build container (the GPU box has no /root/reference); tests/test_gpu_silero.py rebuilds a model
file from these tensors and sends them through the HIP kernels.  PARITY UNPINNED: the expected probabilities come from the
restatement, not from ONNX Runtime (absent here)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import silero_vad_oracle as so  # noqa: E402

REAL = "/root/reference/data/models/silero-vad/silero_vad.onnx"


def clips():
    sr = 16000
    t = np.arange(4 * sr) / sr
    f0 = 120 + 20 * np.sin(2 * np.pi * 0.7 * t)
    ph = np.cumsum(2 * np.pi * f0 / sr)
    voiced = sum(np.sin(k * ph) / k for k in range(1, 20)) * (0.5 + 0.5 * np.sin(2 * np.pi * 4 * t)) ** 2   # a vowel-like buzz
    voiced = (0.2 * voiced / np.abs(voiced).max()).astype(np.float32)
    rng = np.random.default_rng(3)
    gated = voiced.copy()
    gated[sr:2 * sr] = (0.001 * rng.standard_normal(sr)).astype(np.float32)      # a second of near-silence inside
    noise = (0.05 * rng.standard_normal(4 * sr)).astype(np.float32)
    return {"voiced": voiced, "gated": gated, "noise": noise}


def main():
    w = so.load_weights(REAL)
    out = {"w_" + k: np.ascontiguousarray(v, dtype=np.float32) for k, v in w.items()}
    for name, x in clips().items():
        out["clip_" + name] = x
        out["prob_" + name] = so.SileroOracle(w).process(x, reset=True).astype(np.float32)
    path = os.path.join(ROOT, "tests", "golden", "silero_v5_16k.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", {k: (float(v.min()), float(v.max())) for k, v in out.items() if k.startswith("prob_")})


if __name__ == "__main__":
    main()
