"""The route by which parity becomes PINNED (SURVEY.md §8 c/d): a real whisper.cpp build and real model files.

Neither exists in this image or on the GPU box, nothing can be fetched, and /root/reference holds no whisper.cpp source
(cmake/BuildWhispercpp.cmake:118 downloads v1.6.2 at build time) — so these tests SKIP with "unverified" wherever they
run today.  On a machine that has them:

  WHISPER_CPP_LIB=/path/to/libwhisper.so      a whisper.cpp v1.6.2 shared library (the ABI of include/whisper.h)
  MI355_WHISPER_MODEL_DIR=/path/to/models     ggml-*.bin files (e.g. ggml-tiny.en.bin, ggml-base.en.bin: BASELINE configs[0/1])

the CPU test pins the oracle (token ids, greedy, temperature fallback off, the synthetic clips every other test uses) and
the GPU test then compares this library in fp32 with the same run.  The library named by WHISPER_CPP_LIB is loaded only
because the operator named it; nothing inside /root/reference is loaded or executed.
"""
import ctypes as C
import glob
import os

import numpy as np
import pytest

from tools import ggml_io

LIB, MODEL_DIR = os.environ.get("WHISPER_CPP_LIB"), os.environ.get("MI355_WHISPER_MODEL_DIR")
MODELS = sorted(glob.glob(os.path.join(MODEL_DIR, "ggml-*.bin"))) if MODEL_DIR else []
needs_reference_build = pytest.mark.skipif(not (LIB and os.path.exists(LIB) and MODELS),
                                          reason="unverified: no whisper.cpp build (WHISPER_CPP_LIB) / model files (MI355_WHISPER_MODEL_DIR)")


def _whisper_cpp_tokens(pkg, model, pcm):
    L = pkg.bind_whisper_abi(C.CDLL(LIB))
    cp = L.whisper_context_default_params()
    cp.use_gpu = False
    ctx = L.whisper_init_from_file_with_params(model.encode(), cp)
    assert ctx, model
    try:
        p = L.whisper_full_default_params(0)
        p.temperature_inc = 0.0
        p.print_progress = p.print_realtime = p.print_timestamps = p.print_special = False
        p.language = b"en"
        p.n_threads = 8
        assert L.whisper_full(ctx, p, pcm.ctypes.data_as(C.POINTER(C.c_float)), len(pcm)) == 0
        return [[L.whisper_full_get_token_data(ctx, i, j).id for j in range(L.whisper_full_n_tokens(ctx, i))]
                for i in range(L.whisper_full_n_segments(ctx))]
    finally:
        L.whisper_free(ctx)


@needs_reference_build
@pytest.mark.parametrize("model", MODELS or ["-"])
def test_oracle_equals_whisper_cpp(pkg, model):
    from oracle import pyoracle
    pcm = ggml_io.synth_pcm(7, 9.0)
    want = _whisper_cpp_tokens(pkg, model, pcm)
    orc = pyoracle.Oracle(model)
    po = orc.default_params(0, temperature_inc=0.0, language="en", duration_ms=0)
    assert orc.full(po, pcm) == 0
    got = [[t["id"] for t in s["tokens"]] for s in orc.segments()]
    assert got == want


@pytest.mark.gpu
@needs_reference_build
@pytest.mark.parametrize("model", MODELS or ["-"])
def test_engine_equals_whisper_cpp(pkg, model):
    pcm = ggml_io.synth_pcm(7, 9.0)
    want = _whisper_cpp_tokens(pkg, model, pcm)
    eng = pkg.Engine(model, precision=pkg.PRECISION_FP32)
    try:
        p = eng.default_params(0, temperature_inc=0.0, language="en")
        assert eng.full(p, pcm) == 0
        L = eng.L
        got = [[L.whisper_full_get_token_data(eng.ctx, i, j).id for j in range(L.whisper_full_n_tokens(eng.ctx, i))]
               for i in range(L.whisper_full_n_segments(eng.ctx))]
        assert got == want
    finally:
        eng.close()
