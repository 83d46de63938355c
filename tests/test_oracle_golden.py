"""CPU: pins the oracle (oracle/whisper_oracle.cpp) against the committed golden vectors.

The reference's own tests hold no vectors for this path; tests/golden/ was produced by
tools/gen_golden.py from HuggingFace transformers' independent Whisper implementation (same
architecture, seeded weights).  Parity with whisper.cpp itself stays UNPINNED (library absent).
"""
import os

import numpy as np
import pytest

from oracle import pyoracle
from tools import ggml_io

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("n_mels,preset", [(80, "micro.en"), (128, "micro")])
@pytest.mark.parametrize("seed,secs", [(0, 30.0), (1, 3.0), (2, 0.9)])
def test_mel_matches_hf_feature_extractor(model_file, n_mels, preset, seed, secs):
    g = np.load(os.path.join(GOLD, f"mel_{n_mels}_seed{seed}.npz"))
    orc = pyoracle.Oracle(model_file(preset))
    mel = orc.mel(ggml_io.synth_pcm(seed, secs))
    n_samples = int(round(secs * 16000))
    assert mel.shape == (n_mels, (n_samples + 480000) // 160)
    stride = int(g["stride"])
    ours = mel[:, :3000][:, ::stride]
    ref = g["mel"]
    # whisper.cpp pads the tail with zeros where HF reflect-pads: frames touching the last 200
    # samples differ by construction -> compare frames strictly inside the audio,
    # and the all-padding frames (both are the clamped floor).
    n_valid = (n_samples - 200) // 160
    fr = np.arange(0, 3000, stride)
    inside = fr < n_valid
    # tolerance: north_star "mel within 1e-4 relative" (values are O(1) after (x+4)/4)
    assert np.abs(ours[:, inside] - ref[:, inside]).max() < 1e-4
    outside = fr > (n_samples + 400) // 160 + 1
    if outside.any():
        assert np.abs(ours[:, outside] - ref[:, outside]).max() < 1e-4


@pytest.mark.parametrize("preset", ["micro.en", "micro"])
def test_encoder_decoder_match_hf(model_file, preset):
    g = np.load(os.path.join(GOLD, f"encdec_{preset}.npz"))
    orc = pyoracle.Oracle(model_file(preset))
    mel = orc.mel(ggml_io.synth_pcm(0, 30.0))
    # HF consumed its own feature extractor's mel; identical to ours except the last 2 frames
    assert np.abs(mel[:, :3000][:, ::50] - g["mel_in"]).max() < 1e-4
    enc = orc.encode(0)
    assert enc.shape == (1500, orc.n_audio_state)
    # fp32 vs fp32, different summation order; last rows see the tail-padding difference
    assert np.abs(enc[::25][:-1] - g["enc_rows"][:-1]).max() < 2e-3
    assert abs(enc.mean() - g["enc_mean"]) < 1e-3 and abs(enc.std() - g["enc_std"]) < 1e-3
    toks = g["tokens"]
    logits = orc.decode(toks, n_past=0, seq=0)
    cols = g["cols"]
    assert np.abs(logits[:, cols] - g["logits"]).max() < 2e-3
    assert np.abs(logits.max(-1) - g["logits_max"]).max() < 2e-3
    assert (logits.argmax(-1) == g["argmax"]).all()
    # incremental decoding with the KV cache reproduces the teacher-forced rows
    orc.decode(toks[:5], n_past=0, seq=1)
    for i in range(5, len(toks)):
        row = orc.decode(toks[i:i + 1], n_past=i, seq=1)[0]
        assert np.abs(row - logits[i]).max() < 1e-4


@pytest.mark.parametrize("n_vocab,preset", [(51864, "micro.en"), (51866, "micro")])
def test_timestamp_rules_match_hf_processor(model_file, n_vocab, preset):
    g = np.load(os.path.join(GOLD, f"rules_{n_vocab}.npz"))
    orc = pyoracle.Oracle(model_file(preset))
    beg, eot = int(g["beg"]), int(g["eot"])
    assert (orc.beg, orc.eot, orc.n_vocab) == (beg, eot, n_vocab)
    p = orc.default_params(0, suppress_blank=0, max_initial_ts=1.0)
    for hi, hist in enumerate(g["hists"]):
        h = [int(t) for t in hist if t >= 0]
        rs = np.random.RandomState(1000 + hi)
        logits = rs.standard_normal(n_vocab).astype(np.float32) * 3.0
        if hi == 4:
            logits[beg:] += 4.0
        ts = [t for t in h if t >= beg]
        has_ts = len(ts) > 0
        seek_delta = 2 * (ts[-1] - beg) if has_ts else 3000
        lg, lp, pr = orc.process_logits(p, logits, h, has_ts=has_ts, seek_delta=seek_delta)
        ours = np.isneginf(lg)
        ref = np.unpackbits(g["masks"][hi])[:n_vocab].astype(bool)
        # documented differences between whisper.cpp and the OpenAI/HF rule set:
        #  * whisper.cpp additionally suppresses sot/lang/task tokens every step (HF: separate list)
        #  * HF forces the FIRST sampled token to be a timestamp; whisper.cpp only caps it
        #  * monotonic rule: whisper.cpp allows ts == last ts, HF forbids it unless closing a pair
        cmp = np.ones(n_vocab, bool)
        cmp[eot:beg] = False
        if not h:
            cmp[:beg] = False
        if has_ts:
            cmp[ts[-1]] = False
        assert (ours[cmp] == ref[cmp]).all(), hi
        # probabilities are a proper distribution over the unsuppressed tokens
        assert abs(pr.sum() - 1.0) < 1e-3 or bool(g["ts_forced"][hi])
        if hi == 4:  # timestamp mass rule fired: every text token is suppressed
            assert ours[:beg].all() and bool(g["ts_forced"][hi])
