"""GPU: the LocalVocal worker end to end on the HIP engine — audio tap -> whisper_loop thread -> segmentation
-> run_whisper_inference -> set_text_callback — and its agreement with (a) the segmentation restatement and
(b) the CPU oracle's whisper_full on the PCM each request carried (through the reference's post-filter rules).
"""
import ctypes
import time

import numpy as np
import pytest

from oracle import pyoracle
from oracle import segmentation_oracle as so
from tools import ggml_io
from test_gpu_parity import _reference_post_filter
from test_segmentation_cpu import speechlike

pytestmark = pytest.mark.gpu
SR = 16000


def _wait(cond, timeout_s):
    t0 = time.time()
    while time.time() - t0 < timeout_s:
        if cond():
            return True
        time.sleep(0.02)
    return False


@pytest.mark.parametrize("mode,partial", [(so.VAD_MODE_DISABLED, True), (so.VAD_MODE_ACTIVE, False), (so.VAD_MODE_HYBRID, True)])
def test_worker_thread_transcribes_what_segmentation_sends(pkg, model_file, mode, partial):
    path = model_file("micro.en")
    f = pkg.Filter(precision=pkg.PRECISION_FP32)
    requests, texts = [], []
    f.set_callbacks(on_text=lambda d: texts.append(d),
                    on_request=lambda pcm, t0, t1, vs: requests.append((pcm, t0, t1, vs)))
    try:
        f.setup_audio(SR, 1)
        f.L.lv_filter_set_vad_mode(f.f, mode)
        f.L.lv_filter_set_partial(f.f, int(partial), 1100)
        f.L.lv_filter_set_segment_duration(f.f, 3000)
        f.L.lv_filter_set_thresholds(f.f, 0.0, 1e6)  # random weights: accept every sentence, whatever its time tokens
        f.L.lv_filter_set_max_sub_duration(f.f, 60000)
        f.params.temperature_inc = 0.0
        x = speechlike(21, 8.0)
        ctypes.CDLL(None).srand(1234)
        orc = so.SegmenterOracle(vad_mode=mode, partial_transcription=partial, partial_latency=1100, segment_duration=3000)
        if mode == so.VAD_MODE_DISABLED:
            # With VAD disabled an idle pass flushes the whisper buffer (vad-processing.cpp:133-143), so a
            # lock-step feed would depend on timing; everything is queued before the worker starts instead:
            # its first pass takes all 8 s (<= the 10 s cap) and sends them as one full segment.
            for i in range(0, len(x), 4000):
                ts = 1_000_000_000 + i * 1_000_000_000 // SR
                assert f.push_packet([x[i:i + 4000]], ts) == 0
                orc.push_packet(x[i:i + 4000], ts)
            orc.step()
            assert f.start(path) == 0
            assert _wait(lambda: len(texts) >= len(orc.requests), 30.0)
        else:
            assert f.start(path) == 0
            assert f.start(path) == -2  # "cannot init whisper: whisper_context is not null"
            # feed in lock-step with the worker (as the offline harness does, localvocal-offline-test.cpp:527-541):
            # the next packet goes in once the previous one was drained, so the packet->pass grouping is fixed
            # (these two modes do nothing on an idle pass)
            for i in range(0, len(x), 4000):
                pkt = x[i:i + 4000]
                ts = 1_000_000_000 + i * 1_000_000_000 // SR
                assert f.push_packet([pkt], ts) == 0
                orc.push_packet(pkt, ts)
                orc.step()
                assert _wait(lambda: f.L.lv_filter_input_bytes(f.f) == 0, 20.0), "worker did not drain the input buffer"
                assert _wait(lambda: len(requests) >= len(orc.requests) and len(texts) >= len(orc.requests), 30.0)
        f.stop()
        assert len(orc.requests) >= (1 if mode == so.VAD_MODE_DISABLED else 2)
        assert len(requests) == len(orc.requests) == len(texts)
        for (pcm, t0, t1, vs), w in zip(requests, orc.requests):
            assert (t0, t1, vs) == (w.t0, w.t1, w.vad_state) and np.array_equal(pcm, w.pcm)

        # every callback result equals the oracle's whisper_full + the reference's token filter on that PCM
        orc_w = pyoracle.Oracle(path)
        libc = ctypes.CDLL(None)
        n_speech = 0
        libc.rand.restype = ctypes.c_int
        libc.srand(1234)  # replay the worker's rand() stream (seeded the same way before it started)
        for (pcm, t0, t1, vs), d in zip(requests, texts):
            if t1 - t0 < 50:   # guard at whisper-processing.cpp:150-157: no whisper_full, no rand()
                assert d["result"] == 0
                continue
            fed = pcm
            if len(pcm) < SR:  # whisper-processing.cpp:162-180: centred in 1.01 s of +-0.01 rand() noise
                n_new = int(np.float32(1.01) * np.float32(SR))
                fed = np.array([np.float32(0.01) * (np.float32(libc.rand()) / np.float32(2147483647) * np.float32(2.0)
                                                    - np.float32(1.0)) for _ in range(n_new)], np.float32)
                off = (n_new - len(pcm)) // 2
                fed[off:off + len(pcm)] = pcm
            # the same call sequence as the worker made: beam search draws its candidates from decoder 0's
            # mt19937, which persists across whisper_full calls of one context
            po = orc_w.default_params(1, temperature_inc=0.0, suppress_blank=0, language="en",
                                      duration_ms=int(len(fed) * 1000 // SR))
            assert orc_w.full(po, fed) == 0
            res, text, kept = _reference_post_filter(orc_w, orc_w.segments(), len(pcm) * 1000 // SR, 0.0, 1e6, vs == 3)
            assert d["result"] == res
            want = pkg.remove_leading_trailing_nonalpha(text) if res in (2, 5) else b""
            assert d["text"] == want and d["n_tokens"] == len(kept)
            assert (d["start"], d["end"]) == (t0, t1)
            n_speech += res in (2, 5)
        assert n_speech >= 1
    finally:
        f.close()


def test_audio_tap_buffers_once_a_model_is_loaded(pkg, model_file):
    f = pkg.Filter(precision=pkg.PRECISION_FP32)
    try:
        f.setup_audio(SR, 1)
        f.L.lv_filter_set_vad_mode(f.f, so.VAD_MODE_DISABLED)
        got = []
        f.set_callbacks(on_request=lambda pcm, t0, t1, vs: got.append(len(pcm)))
        assert f.filter_audio([np.zeros(1600, np.float32)]) == 0   # no model yet: dropped
        assert f.start(model_file("micro.en")) == 0
        assert f.filter_audio([ggml_io.synth_pcm(0, 0.5)]) == 1
        assert _wait(lambda: f.L.lv_filter_input_bytes(f.f) == 0, 10.0)
        assert _wait(lambda: len(got) >= 1, 10.0)   # idle pass flushes the half second as a final segment
        assert got[0] == 8000 + 320
        f.stop()
        assert f.filter_audio([np.zeros(1600, np.float32)]) == 0   # model gone again
    finally:
        f.close()
