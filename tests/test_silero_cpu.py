"""CPU: the Silero VAD weight reader and the CPU restatement.

The reference runs /root/reference/data/models/silero-vad/silero_vad.onnx through ONNX Runtime
(/root/reference/src/whisper-utils/silero-vad-onnx.cpp:47-140).  Neither ONNX Runtime nor `onnx` is in this image: the
library reads the weights with its own protobuf wire reader (csrc/silero_vad.hip) and tools/onnx_wire.py is the Python
twin the restatement (oracle/silero_vad_oracle.py) loads them with.  PARITY UNPINNED: no expected probabilities exist
anywhere in the reference; what is checked here is that the two readers take the same tensors from the same file
(synthetic everywhere, the reference's own file where /root/reference exists) and that the restatement on the real
weights behaves like a voice-activity detector.
"""
import os

import numpy as np
import pytest

from oracle import silero_vad_oracle as so
from tools import onnx_wire

REAL = "/root/reference/data/models/silero-vad/silero_vad.onnx"
ORDER = ["basis", "c0_w", "c0_b", "c1_w", "c1_b", "c2_w", "c2_b", "c3_w", "c3_b", "w_ih", "w_hh", "b_ih", "b_hh", "out_w", "out_b"]


def weighted_sums(w):
    out = []
    for k in ORDER:
        v = w[k].ravel().astype(np.float64)
        out.append(float(((np.arange(v.size) % 7 + 1) * v).sum()))
    return out


def test_wire_round_trip(tmp_path):
    w = so.synth_weights(5)
    path = str(tmp_path / "syn.onnx")
    so.write_onnx(path, w)
    g = onnx_wire.load(path)
    assert [n["op"] for n in g["nodes"]] == ["Constant", "Equal", "If"] and g["inputs"] == ["input", "state", "sr"]
    back = so.load_weights(path)   # picks the 16 kHz branch although the 8 kHz branch holds the same names
    assert all(np.array_equal(w[k], back[k]) for k in w)


def test_library_reader_takes_the_same_tensors(pkg, tmp_path):
    w = so.synth_weights(6)
    path = str(tmp_path / "syn.onnx")
    so.write_onnx(path, w)
    assert pkg.silero_inspect(path) == pytest.approx(weighted_sums(w), rel=1e-12, abs=1e-9)


@pytest.mark.skipif(not os.path.exists(REAL), reason="the reference's silero_vad.onnx is not on this machine")
def test_readers_agree_on_the_reference_file(pkg):
    w = so.load_weights(REAL)
    assert {k: v.shape for k, v in w.items()} == {k: shape for k, (_, shape) in so.SUFFIXES.items()}
    assert pkg.silero_inspect(REAL) == pytest.approx(weighted_sums(w), rel=1e-12, abs=1e-9)


@pytest.mark.skipif(not os.path.exists(REAL), reason="the reference's silero_vad.onnx is not on this machine")
def test_restatement_on_the_reference_weights_behaves_like_a_vad():
    o = so.SileroOracle(REAL)
    sr = 16000
    t = np.arange(2 * sr) / sr
    silence = (0.001 * np.random.default_rng(0).standard_normal(2 * sr)).astype(np.float32)
    f0 = 120 + 20 * np.sin(2 * np.pi * 0.7 * t)
    ph = np.cumsum(2 * np.pi * f0 / sr)
    voiced = sum(np.sin(k * ph) / k for k in range(1, 20)) * (0.5 + 0.5 * np.sin(2 * np.pi * 4 * t)) ** 2   # a vowel-like buzz
    voiced = (0.2 * voiced / np.abs(voiced).max()).astype(np.float32)
    tone = (0.2 * np.sin(2 * np.pi * 1000 * t)).astype(np.float32)
    p_sil, p_voiced, p_tone = o.process(silence), o.process(voiced), o.process(tone)
    assert p_sil[8:].max() < 0.1 and p_tone[8:].max() < 0.1       # (the first windows carry the zero state's transient)
    assert p_voiced.mean() > 0.3 and p_voiced.max() > 0.6
    # the state is what carries between windows: a reset reproduces, no reset does not
    again = o.process(voiced, reset=True)
    assert np.array_equal(again, p_voiced)
    assert not np.array_equal(o.process(voiced, reset=False), p_voiced)


def test_malformed_files_are_refused(pkg, tmp_path):
    w = so.synth_weights(7)
    good = str(tmp_path / "good.onnx")
    so.write_onnx(good, w)
    blob = open(good, "rb").read()
    cut = str(tmp_path / "cut.onnx")
    open(cut, "wb").write(blob[:len(blob) // 3])
    with pytest.raises(RuntimeError):
        pkg.silero_inspect(cut)
    noise = str(tmp_path / "noise.onnx")
    open(noise, "wb").write(np.random.default_rng(0).integers(0, 256, 4096, dtype=np.uint8).tobytes())
    with pytest.raises(RuntimeError):
        pkg.silero_inspect(noise)
    # a graph whose 16 kHz tensors are not there (only the 8 kHz shapes)
    wrong = dict(w)
    wrong["basis"] = w["basis"][:130, :, :128].copy()
    bad = str(tmp_path / "bad.onnx")
    so.write_onnx(bad, {k: v for k, v in wrong.items()})
    with pytest.raises(RuntimeError):
        pkg.silero_inspect(bad)
    with pytest.raises(RuntimeError):
        pkg.silero_inspect(str(tmp_path / "absent.onnx"))


def test_filter_without_the_model_file_keeps_the_energy_gate(pkg, tmp_path):
    f = pkg.Filter()
    try:
        assert f.set_silero_model(str(tmp_path / "absent.onnx")) == 1 and f.L.lv_filter_vad_is_silero(f.f) == 0
        x = np.concatenate([np.zeros(16000, np.float32), 0.1 * np.random.RandomState(3).randn(16000).astype(np.float32)])
        st = f.vad_speech_timestamps(x, True)
        assert st and st[-1][1] == len(x)
        assert len(f.vad_probabilities()) == len(x) // 512
    finally:
        f.close()


@pytest.mark.skipif(not os.path.exists(REAL), reason="the reference's silero_vad.onnx is not on this machine")
def test_golden_fixture_holds_the_reference_files_tensors():
    # tests/golden/silero_v5_16k.npz (tools/gen_silero_golden.py) is what the GPU test sends through the kernels
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "silero_v5_16k.npz"))
    w = so.load_weights(REAL)
    assert sorted(k[2:] for k in g.files if k.startswith("w_")) == sorted(w)
    assert all(np.array_equal(g["w_" + k], np.asarray(v, dtype=np.float32)) for k, v in w.items())
    for name in ("voiced", "gated", "noise"):
        assert np.abs(so.SileroOracle(w).process(g["clip_" + name]) - g["prob_" + name]).max() < 1e-6
