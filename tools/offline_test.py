#!/usr/bin/env python3
"""Linux equivalent of LocalVocal's offline test harness, driving the MI355X engine through the C ABI.

Follows /root/reference/src/tests/localvocal-offline-test.cpp (Windows/macOS only there): read an audio file and a
JSON config, build the filter state with the harness's parameter set (`create_context`, :96-207), push the audio in
25 ms packets in lock-step with the worker thread (:499-567), append two seconds of silence (:568-582), wait for the
input to drain (:587-602), and write

    output.txt      one line per transcribed sentence (result text after the English trim)
    segments.json   [{"start_time": s, "end_time": s, "segment_label": text}, ...] from the audio-chunk callback (:215-248)

`wer()` restates src/tests/evaluate_output.py:33-36 (Levenshtein over lower-cased whitespace tokens, divided by the
longer token count) without the Levenshtein package.

usage: offline_test.py <audio.wav> <config.json> [--outdir DIR]
config keys honoured: whisper_model_path, whisper_language, whisper_sampling_method, temperature, no_context,
enable_audio_chunks_callback, vad_mode, partial_transcription, sentence_psum_accept_thresh, duration_filter_threshold, temperature_inc (extensions),
precision ("bf16"/"fp32", extension), silero_vad_model_file (the Silero network on the GPU when the file exists, else the
energy gate, as the library logs).  ct2_model_folder, source/target_language are accepted and ignored (no CTranslate2 in
this build).
"""
from __future__ import annotations

import argparse
import importlib.util
import json
import os
import re
import sys
import time
import unicodedata
import wave

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_package():
    if "obs_localvocal_amd" in sys.modules:
        return sys.modules["obs_localvocal_amd"]
    spec = importlib.util.spec_from_file_location("obs_localvocal_amd", os.path.join(ROOT, "obs-localvocal_amd", "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["obs_localvocal_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def read_wav(path):
    """-> (sample_rate, [channel arrays float32 in [-1, 1]]); PCM 8/16/32-bit and (via numpy) IEEE float32 files."""
    with open(path, "rb") as fh:
        head = fh.read(44)
    fmt_tag = int.from_bytes(head[20:22], "little") if head[:4] == b"RIFF" else 1
    if fmt_tag == 3:  # IEEE float: the wave module refuses it
        with open(path, "rb") as fh:
            buf = fh.read()
        pos = buf.index(b"fmt ") + 8
        ch = int.from_bytes(buf[pos + 2:pos + 4], "little")
        sr = int.from_bytes(buf[pos + 4:pos + 8], "little")
        dpos = buf.index(b"data") + 8
        n = int.from_bytes(buf[dpos - 4:dpos], "little")
        x = np.frombuffer(buf, "<f4", n // 4, dpos).reshape(-1, ch)
    else:
        with wave.open(path, "rb") as w:
            sr, ch, sw, n = w.getframerate(), w.getnchannels(), w.getsampwidth(), w.getnframes()
            raw = w.readframes(n)
        if sw == 2:
            x = np.frombuffer(raw, "<i2").astype(np.float32) / 32768.0
        elif sw == 4:
            x = np.frombuffer(raw, "<i4").astype(np.float32) / 2147483648.0
        elif sw == 1:
            x = (np.frombuffer(raw, np.uint8).astype(np.float32) - 128.0) / 128.0
        else:
            raise ValueError("unsupported sample width %d" % sw)
        x = x.reshape(-1, ch)
    return sr, [np.ascontiguousarray(x[:, c], np.float32) for c in range(x.shape[1])]


def write_wav(path, sr, x):
    with wave.open(path, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(sr)
        w.writeframes((np.clip(x, -1, 1) * 32767.0).astype("<i2").tobytes())


# ---- evaluate_output.py semantics ----
def tokenize(text, should_remove_accents=False, remove_punctuation=False):
    if should_remove_accents:
        text = "".join(c for c in unicodedata.normalize("NFD", text) if unicodedata.category(c) != "Mn")
        text = re.sub(r"\b(\w+?)(a)\b", r"\1e", text)
    if remove_punctuation:
        text = re.sub(r"\s+", " ", re.sub(r"[^\w\s]", "", text)).strip()
    return text.lower().split()


def levenshtein(a, b):
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i]
        for j, y in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
        prev = cur
    return prev[-1]


def wer(ref_tokens, hyp_tokens):
    return levenshtein(ref_tokens, hyp_tokens) / max(len(ref_tokens), len(hyp_tokens), 1)


# ---- the harness ----
def run(audio_path, config, outdir=".", pkg=None, realtime=False, requests_out=None):
    """requests_out: a list that receives (pcm, t0, t1, vad_state) of every inference request (tests replay them on the oracle)."""
    pkg = pkg or load_package()
    sr, chans = read_wav(audio_path)
    prec = {"bf16": pkg.PRECISION_BF16, "fp32": pkg.PRECISION_FP32}.get(config.get("precision", "bf16"), pkg.PRECISION_BF16)
    f = pkg.Filter(precision=prec)
    out_txt, seg_json = os.path.join(outdir, "output.txt"), os.path.join(outdir, "segments.json")
    open(out_txt, "w").close()
    if os.path.exists(seg_json):
        os.remove(seg_json)
    segments, results = [], []

    def on_text(d):
        results.append(d)
        if d["result"] == 2 and d["text"]:  # DETECTION_RESULT_SPEECH
            with open(out_txt, "a", encoding="utf-8", errors="ignore") as fh:
                fh.write(d["text"].decode("utf-8", "ignore") + "\n")

    def on_chunk(pcm, vad_state, d):
        segments.append({"start_time": d["start"] / 1000.0, "end_time": d["end"] / 1000.0,
                         "segment_label": d["text"].decode("utf-8", "ignore")})
        with open(seg_json, "w") as fh:
            json.dump(segments, fh, indent=4)

    try:
        if config.get("silero_vad_model_file"):
            f.set_silero_model(config["silero_vad_model_file"])   # (localvocal-offline-test.cpp passes it to start_whisper_thread_with_path)
        f.setup_audio(sr, len(chans))
        # create_context(): the harness's parameter set
        p = f.L.whisper_full_default_params(int(config.get("whisper_sampling_method", 0)))
        ctypes_params = f.L.lv_filter_params(f.f).contents
        for name in [n for n, _ in type(p)._fields_]:
            setattr(ctypes_params, name, getattr(p, name))
        q = ctypes_params
        q.duration_ms = 3000
        q.detect_language = False
        q.n_threads = 4
        q.n_max_text_ctx = 16384
        q.translate = False
        q.no_context = bool(config.get("no_context", False))
        q.single_segment = True
        q.print_special = q.print_progress = q.print_realtime = q.print_timestamps = False
        q.token_timestamps = False
        q.thold_pt = q.thold_ptsum = 0.01
        q.max_len = 0
        q.split_on_word = False
        q.max_tokens = 0
        q.suppress_blank = True
        q.suppress_non_speech_tokens = True
        q.temperature = float(config.get("temperature", 0.0))
        if "temperature_inc" in config:   # extension: 0 switches the temperature fallback off
            q.temperature_inc = float(config["temperature_inc"])
        q.max_initial_ts = 1.0
        q.length_penalty = -1.0
        f.L.lv_filter_set_language(f.f, str(config.get("whisper_language", "en")).encode())
        f.L.lv_filter_set_thresholds(f.f, float(config.get("sentence_psum_accept_thresh", 0.4)),
                                     float(config.get("duration_filter_threshold", 2.25)))
        f.L.lv_filter_set_vad_mode(f.f, int(config.get("vad_mode", pkg.VAD_MODE_ACTIVE)))
        f.L.lv_filter_set_partial(f.f, int(bool(config.get("partial_transcription", False))), 1100)
        f.L.lv_filter_enable_audio_chunks_callback(f.f, int(bool(config.get("enable_audio_chunks_callback", True))))
        f.set_callbacks(on_text=on_text, on_chunk=on_chunk,
                        on_request=None if requests_out is None else (lambda pcm, t0, t1, vs: requests_out.append((pcm, t0, t1, vs))))
        rc = f.start(config["whisper_model_path"])
        if rc != 0:
            raise RuntimeError("failed to start the whisper thread: %d (%s)" % (rc, pkg.last_error()))
        n = len(chans[0])
        win = sr * 25 // 1000
        t_start = time.time()
        for k, i in enumerate(range(0, n, win)):
            # lock-step with the worker: wait until the previous packet was taken, so buffers build up like in OBS
            t0 = time.time()
            while f.L.lv_filter_input_bytes(f.f) != 0 and time.time() - t0 < 30.0:
                time.sleep(0.0005)
            if realtime:
                lag = t_start + k * 0.025 - time.time()
                if lag > 0:
                    time.sleep(lag)
            f.push_packet([c[i:i + win] for c in chans], 1_000_000_000 + i * 1_000_000_000 // sr)
        f.push_packet(2 * sr, 1_000_000_000 + n * 1_000_000_000 // sr)  # two seconds of silence
        t0 = time.time()
        while f.L.lv_filter_input_bytes(f.f) >= sr // 2 * 4 and time.time() - t0 < 60.0:
            time.sleep(0.1)
        time.sleep(0.6)  # let the worker finish the call in flight (it holds the context mutex)
        f.stop()
    finally:
        f.close()
    return results, segments


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("audio")
    ap.add_argument("config")
    ap.add_argument("--outdir", default=".")
    ap.add_argument("--ref", help="reference transcript: prints the WER like evaluate_output.py")
    a = ap.parse_args()
    config = json.load(open(a.config))
    results, segments = run(a.audio, config, a.outdir)
    print("sentences: %d, segments: %d" % (sum(r["result"] == 2 for r in results), len(segments)))
    if a.ref:
        ref = " ".join(s.strip() for s in open(a.ref, encoding="utf-8", errors="ignore"))
        hyp = " ".join(s.strip() for s in open(os.path.join(a.outdir, "output.txt"), encoding="utf-8", errors="ignore"))
        print('"%s" WER: "%.2g"' % (a.ref, wer(tokenize(ref), tokenize(hyp))))


if __name__ == "__main__":
    main()
