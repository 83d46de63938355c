"""The offline-test-equivalent harness (tools/offline_test.py; reference src/tests/localvocal-offline-test.cpp and
src/tests/evaluate_output.py — SURVEY.md §8 f3)."""
import json
import os

import numpy as np
import pytest

from tools import offline_test as ot
from tools import ggml_io


def test_wer_semantics_of_evaluate_output():
    # Levenshtein over lower-cased tokens / max(len(ref), len(hyp))   (evaluate_output.py:33-36)
    assert ot.wer(ot.tokenize("the cat sat"), ot.tokenize("The cat sat")) == 0.0
    assert ot.wer(ot.tokenize("the cat sat on the mat"), ot.tokenize("the cat sit on mat")) == pytest.approx(2 / 6)
    assert ot.wer(ot.tokenize("a b"), ot.tokenize("a b c d")) == pytest.approx(2 / 4)
    assert ot.tokenize("Hello, World!  ok", remove_punctuation=True) == ["hello", "world", "ok"]
    assert ot.tokenize("niña está", should_remove_accents=True) == ["nine", "este"]
    assert ot.levenshtein("kitten", "sitting") == 3


def test_wav_round_trip(tmp_path):
    x = (0.3 * np.sin(np.arange(16000) * 0.05)).astype(np.float32)
    p = str(tmp_path / "a.wav")
    ot.write_wav(p, 16000, x)
    sr, ch = ot.read_wav(p)
    assert sr == 16000 and len(ch) == 1 and np.abs(ch[0] - x).max() < 1e-4


@pytest.mark.gpu
def test_harness_end_to_end(pkg, model_file, tmp_path):
    """The files the harness writes, and every sentence in them against the CPU oracle: whisper_full (greedy, the harness's
    create_context parameters, localvocal-offline-test.cpp:135-176) + the reference's token filter
    (whisper-processing.cpp:246-323) on the PCM each inference request carried."""
    import ctypes
    from oracle import pyoracle
    from test_gpu_parity import _reference_post_filter
    from test_segmentation_cpu import speechlike
    wav = str(tmp_path / "in.wav")
    ot.write_wav(wav, 16000, speechlike(5, 10.0) * 3.0)
    path = model_file("micro.en")
    cfg = {"whisper_model_path": path, "whisper_language": "en", "whisper_sampling_method": 0,
           "silero_vad_model_file": "unused.onnx", "ct2_model_folder": "", "source_language": "none", "target_language": "none",
           "log_level": "info", "enable_audio_chunks_callback": True, "precision": "fp32",
           "sentence_psum_accept_thresh": 0.0, "duration_filter_threshold": 1e6, "temperature_inc": 0.0}   # random weights: accept all
    libc = ctypes.CDLL(None)
    libc.rand.restype = ctypes.c_int
    libc.srand(4321)
    requests = []
    results, segments = ot.run(wav, cfg, str(tmp_path), pkg=pkg, requests_out=requests)
    assert len(segments) >= 2 and os.path.exists(tmp_path / "segments.json")
    on_disk = json.load(open(tmp_path / "segments.json"))
    assert on_disk == segments and all(s["end_time"] >= s["start_time"] for s in segments)
    lines = [l.rstrip("\n") for l in open(tmp_path / "output.txt", encoding="utf-8")]
    spoken = [r["text"].decode("utf-8", "ignore") for r in results if r["result"] == 2 and r["text"]]
    assert lines == spoken

    # ---- the oracle on the same requests ----
    assert len(requests) == len(results) >= 2
    orc = pyoracle.Oracle(path)
    libc.srand(4321)   # replay the worker's rand() stream (short requests are centred in rand() noise)
    n_speech = 0
    for (pcm, t0, t1, vs), d in zip(requests, results):
        if t1 - t0 < 50:
            assert d["result"] == 0
            continue
        fed = pcm
        if len(pcm) < 16000:   # whisper-processing.cpp:162-180
            n_new = int(np.float32(1.01) * np.float32(16000))
            fed = np.array([np.float32(0.01) * (np.float32(libc.rand()) / np.float32(2147483647) * np.float32(2.0) - np.float32(1.0))
                            for _ in range(n_new)], np.float32)
            off = (n_new - len(pcm)) // 2
            fed[off:off + len(pcm)] = pcm
        po = orc.default_params(0, temperature_inc=0.0, suppress_blank=1, suppress_non_speech_tokens=1, single_segment=1,
                                no_context=0, max_initial_ts=1.0, length_penalty=-1.0, n_max_text_ctx=16384, language="en",
                                duration_ms=int(len(fed) * 1000 // 16000))
        assert orc.full(po, fed) == 0
        res, text, kept = _reference_post_filter(orc, orc.segments(), len(pcm) * 1000 // 16000, 0.0, 1e6, vs == 3)
        assert d["result"] == res
        want = pkg.remove_leading_trailing_nonalpha(text) if res in (2, 5) else b""
        assert d["text"] == want and d["n_tokens"] == len(kept) and (d["start"], d["end"]) == (t0, t1)
        n_speech += res == 2
    assert n_speech >= 1 and n_speech == len(lines)
