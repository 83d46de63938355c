#!/usr/bin/env python3
"""SURVEY.md §8(d) configuration 5: streaming partial transcription latency through the LocalVocal worker.

A 64 s synthetic "speech-like" signal (noise bursts of 0.5-2.5 s separated by 0.3-1 s of near-silence, seed 7) is fed
to the worker thread in real-time-ordered 10 ms packets (lock-step, no sleeping); VAD mode ACTIVE with the Silero network
on the GPU (`--vad silero`, default: a synthetic model file of the real graph's shapes whose probability follows the
signal's energy, tools/silero_synth.py; `--vad gate`: the energy gate), partial transcription every partial_latency = 1100 ms
(transcription-filter-properties.cpp:710); beam search as the plugin defaults.  Reported: per-call latency of
run_whisper_inference (request observed -> result callback), p50 / p95 / max, split into partial and final calls.
Weights: seeded synthetic GGML file of the chosen dims (bf16, or `--precision fp8`: e4m3 decoder weights);
decode length forced to `--tokens` per call (random weights would stop after 1-2 tokens).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import ggml_io, silero_synth, synth_large  # noqa: E402
from tools.offline_test import load_package  # noqa: E402


def speechlike(seed, seconds, sr=16000):
    rng = np.random.RandomState(seed)
    n = int(seconds * sr)
    x = (1e-4 * rng.randn(n)).astype(np.float32)
    t = int(rng.uniform(0.0, 0.5) * sr)
    while t < n:
        b = int(rng.uniform(0.5, 2.5) * sr)
        x[t:t + b] += (0.1 * rng.randn(min(b, n - t))).astype(np.float32)
        t += b + int(rng.uniform(0.3, 1.0) * sr)
    return x


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="large-v3")
    ap.add_argument("--seconds", type=float, default=64.0)
    ap.add_argument("--tokens", type=int, default=12)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp8", "fp8mx"])
    ap.add_argument("--vad", default="silero", choices=["silero", "gate"])
    ap.add_argument("--model-dir", default=os.environ.get("MI355_WHISPER_BENCH_DIR", "/tmp/mi355_whisper_bench"))
    a = ap.parse_args()
    pkg = load_package()
    os.makedirs(a.model_dir, exist_ok=True)
    path = os.path.join(a.model_dir, f"ggml-{a.model}-synth-s1234.bin")
    if not os.path.exists(path):
        synth_large.write_large(path, a.model, seed=1234, quiet_special=True)
    f = pkg.Filter(precision={"bf16": pkg.PRECISION_BF16, "fp8": pkg.PRECISION_FP8, "fp8mx": pkg.PRECISION_FP8_MX}[a.precision])
    t_req, lat = {}, []

    def on_request(pcm, t0, t1, vs):
        t_req["t"], t_req["vs"], t_req["n"] = time.perf_counter(), vs, len(pcm)

    def on_text(d):
        if "t" in t_req:
            lat.append((time.perf_counter() - t_req.pop("t"), t_req["vs"], t_req["n"]))

    try:
        if a.vad == "silero":
            vad_path = os.path.join(a.model_dir, "silero_synth_s11.onnx")
            if not os.path.exists(vad_path):
                silero_synth.write_onnx(vad_path, silero_synth.synth_weights(11))
            assert f.set_silero_model(vad_path) == 0
        f.setup_audio(16000, 1)
        f.L.lv_filter_set_vad_mode(f.f, pkg.VAD_MODE_ACTIVE)
        f.L.lv_filter_set_partial(f.f, 1, 1100)
        f.L.lv_filter_set_thresholds(f.f, 0.0, 1e6)
        f.params.temperature_inc = 0.0
        f.set_callbacks(on_text=on_text, on_request=on_request)
        assert f.start(path) == 0
        f.engine().set_force_decode_steps(a.tokens)
        x = speechlike(7, a.seconds)
        for i in range(0, len(x), 160):
            t0 = time.time()
            while f.L.lv_filter_input_bytes(f.f) != 0 and time.time() - t0 < 30.0:
                time.sleep(0.0002)
            f.push_packet([x[i:i + 160]], 1_000_000_000 + i * 1_000_000_000 // 16000)
        f.push_packet(32000, 1_000_000_000 + len(x) * 1_000_000_000 // 16000)
        time.sleep(1.5)
        f.stop()
    finally:
        f.close()

    def stats(v):
        v = np.sort(np.asarray(v)) * 1000.0
        return {"calls": int(len(v)), "p50_ms": round(float(np.percentile(v, 50)), 2), "p95_ms": round(float(np.percentile(v, 95)), 2),
                "max_ms": round(float(v[-1]), 2)} if len(v) else {"calls": 0}

    out = {"workload": f"{a.model} { {'fp8': 'e4m3 decoder weights, bf16 activations', 'fp8mx': 'e4m3 decoder weights + MX-fp8 encoder products', 'bf16': 'bf16'}[a.precision] } beam 5, {a.seconds:.0f} s speech-like stream, VAD active ({'Silero network on the GPU, synthetic weights' if a.vad == 'silero' else 'energy gate'}), partials every 1100 ms, "
                       f"{a.tokens} forced tokens per call",
           "all": stats([l for l, _, _ in lat]), "partial": stats([l for l, vs, _ in lat if vs == 3]),
           "final": stats([l for l, vs, _ in lat if vs != 3]),
           "mean_chunk_s": round(float(np.mean([n for _, _, n in lat]) / 16000.0), 2) if lat else 0.0}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
