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
    from test_segmentation_cpu import speechlike
    wav = str(tmp_path / "in.wav")
    ot.write_wav(wav, 16000, speechlike(5, 10.0) * 3.0)
    cfg = {"whisper_model_path": model_file("micro.en"), "whisper_language": "en", "whisper_sampling_method": 0,
           "silero_vad_model_file": "unused.onnx", "ct2_model_folder": "", "source_language": "none", "target_language": "none",
           "log_level": "info", "enable_audio_chunks_callback": True, "precision": "fp32",
           "sentence_psum_accept_thresh": 0.0}
    results, segments = ot.run(wav, cfg, str(tmp_path), pkg=pkg)
    assert len(segments) >= 2 and os.path.exists(tmp_path / "segments.json")
    on_disk = json.load(open(tmp_path / "segments.json"))
    assert on_disk == segments and all(s["end_time"] >= s["start_time"] for s in segments)
    lines = [l.rstrip("\n") for l in open(tmp_path / "output.txt", encoding="utf-8")]
    spoken = [r["text"].decode("utf-8", "ignore") for r in results if r["result"] == 2 and r["text"]]
    assert lines == spoken
