"""GGML block-quantised model files (q4_0, q4_1, q5_0, q5_1, q8_0 — SURVEY.md §8 f4: most catalogue models are
quantised, /root/reference/data/models/models_directory.json:362-371).

The engine and the oracle both dequantise at load (neither has a quantised matmul; whisper.cpp's own q-dot also
quantises the activations, so real-whisper.cpp parity on these files is approximate by construction and unpinned
here).  What is checked: (CPU) the oracle's C++ block decoder against the numpy restatement in tools/ggml_io.py,
bit for bit through the decoder logits; (GPU) the HIP loader's dequantisation kernel against the oracle on the
same file.
"""
import numpy as np
import pytest

from oracle import pyoracle
from tools import ggml_io

FTYPES = {"q4_0": 2, "q4_1": 3, "q8_0": 7, "q5_0": 8, "q5_1": 9}


def _logits(orc, pcm):
    orc.mel(pcm)
    orc.encode(0)
    toks = [orc.sot if hasattr(orc, "sot") else 50257, 50362]
    return np.asarray(orc.decode(toks, 0))


@pytest.mark.parametrize("q", sorted(FTYPES))
def test_oracle_block_decoder_matches_numpy(model_file, tmp_path, q):
    path = model_file("micro.en", ftype=FTYPES[q])
    hp, filt, vocab, tensors = ggml_io.read_ggml(path)          # numpy dequantisation
    raw = ggml_io.synthetic_tensors(hp, 1234)
    w = "decoder.blocks.0.mlp.0.weight"
    err = float(np.abs(tensors[w] - raw[w]).max())
    assert 0 < err < {"q4_0": 0.2, "q4_1": 0.12, "q5_0": 0.1, "q5_1": 0.06, "q8_0": 0.01}[q] * float(np.abs(raw[w]).max()) + 1e-6
    # the same values written back as a plain f32 file must give the oracle bit-identical logits
    hp32 = ggml_io.HParams(*hp.as_list())
    hp32.ftype = 0
    p32 = str(tmp_path / "deq.bin")
    ggml_io.write_ggml(p32, hp32, tensors, vocab, filt)
    pcm = ggml_io.synth_pcm(0, 3.0)
    a, b = pyoracle.Oracle(path), pyoracle.Oracle(p32)
    la, lb = _logits(a, pcm), _logits(b, pcm)
    assert la.shape == lb.shape and np.array_equal(la, lb)
    a.close()
    b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("q", sorted(FTYPES))
def test_engine_reads_quantised_files(pkg, model_file, q):
    path = model_file("micro.en", ftype=FTYPES[q])
    eng, orc = pkg.Engine(path, precision=pkg.PRECISION_FP32), pyoracle.Oracle(path)
    try:
        pcm = ggml_io.synth_pcm(1, 5.0)
        p = eng.default_params(0, temperature_inc=0.0, language="en", duration_ms=5000)
        po = orc.default_params(0, temperature_inc=0.0, language="en", duration_ms=5000)
        assert eng.full(p, pcm) == 0 and orc.full(po, pcm) == 0
        assert eng.token_ids() == orc.token_ids() and len(orc.token_ids()) > 0
        eng.mel(pcm); orc.mel(pcm)
        eo, oo = np.asarray(eng.encode(0)), np.asarray(orc.encode(0))
        assert np.abs(eo - oo).max() < 2e-3
    finally:
        eng.close()
        orc.close()


def test_unknown_tensor_type_is_rejected(pkg, model_file, tmp_path):
    # a type id outside {f32, f16, q4_0, q4_1, q5_0, q5_1, q8_0} must fail loudly, not load garbage
    import torch
    src = open(model_file("micro.en"), "rb").read()
    hp, filt, vocab, tensors = ggml_io.read_ggml(model_file("micro.en"))
    # the first tensor header follows the vocabulary: patch its ttype field to 12 (q2_K in later ggml versions)
    name = b"encoder.positional_embedding"
    at = src.index(name) - 4 * 2 - 4   # [n_dims][name_len][ttype][ne0][ne1] name
    bad = bytearray(src)
    assert int.from_bytes(bad[at:at + 4], "little") == 0
    bad[at:at + 4] = (12).to_bytes(4, "little")
    p = tmp_path / "bad.bin"
    p.write_bytes(bytes(bad))
    if torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            pkg.Engine(str(p))
    with pytest.raises(Exception):
        pyoracle.Oracle(str(p))
