"""GPU: model lifecycle of a filter, decision for decision as update_whisper_model makes them
(/root/reference/src/whisper-utils/whisper-model-utils.cpp:12-142, shutdown_whisper_thread / start_whisper_thread_with_path
whisper-utils.cpp:9-51): initial load, same-name no-op, A -> B switch while the worker is parked in its wait, unknown and
not-downloaded catalogue names, external files, the DTW-flag restart (which, as in the reference, restarts with the path
shutdown_whisper_thread has just cleared), and whisper_free racing a worker that is parked or mid-inference.
"""
import time

import numpy as np
import pytest

from test_segmentation_cpu import speechlike

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]
SR = 16000


def _wait(cond, timeout_s):
    t0 = time.time()
    while time.time() - t0 < timeout_s:
        if cond():
            return True
        time.sleep(0.02)
    return False


def _new_filter(pkg, texts, requests):
    f = pkg.Filter(precision=pkg.PRECISION_FP32)
    f.set_callbacks(on_text=lambda d: texts.append(d), on_request=lambda pcm, t0, t1, vs: requests.append((pcm, t0, t1, vs)))
    f.setup_audio(SR, 1)
    f.L.lv_filter_set_vad_mode(f.f, 0)             # VAD_MODE_ACTIVE on the energy gate
    f.L.lv_filter_set_thresholds(f.f, 0.0, 1e6)   # random weights: accept every sentence
    f.L.lv_filter_set_max_sub_duration(f.f, 60000)
    f.params.temperature_inc = 0.0
    return f


def _feed(f, x):
    for i in range(0, len(x), 4000):
        assert f.push_packet([x[i:i + 4000]], 1_000_000_000 + i * 1_000_000_000 // SR) == 0


def test_update_whisper_model_lifecycle(pkg, model_file):
    file_a, file_b = model_file("micro.en", seed=1234), model_file("micro.en", seed=77)
    texts, requests = [], []
    f = _new_filter(pkg, texts, requests)
    try:
        f.register_model("Micro A", file_a)
        f.register_model("Micro B", file_b)
        f.register_model("Listed, not downloaded", None)

        # no model named yet: "Whisper model path is empty", nothing happens
        assert f.update_model() == 1 and not f.model_state()["context"]

        # initial load
        f.set_setting("whisper_model_path", "Micro A")
        assert f.update_model() == 0
        st = f.model_state()
        assert st == dict(model_path="Micro A", file_loaded=file_a, loaded_new=False, worker_running=True, dtw=False, context=True)
        ctx_a = f.L.lv_filter_context(f.f)

        # same name: "Model path did not change" — same context, same worker
        assert f.update_model() == 0
        assert f.L.lv_filter_context(f.f) == ctx_a and f.model_state() == st

        # the worker serves model A
        x = speechlike(5, 3.0)
        _feed(f, x)
        assert _wait(lambda: len(texts) >= 1, 30.0)

        # A -> B while the worker is parked in its 250 ms wait: shutdown frees the context under the context mutex, wakes
        # and joins the worker, the new worker starts on B; prompt (no 250 ms multiples stacked up, no deadlock)
        # (really parked: no packet waiting and no inference in flight — every request has had its result — for three worker
        # periods in a row; VAD is active here, so a burst still open at the end of the feed may stay in the whisper buffer)
        def parked():
            return f.L.lv_filter_input_bytes(f.f) == 0 and len(requests) == len(texts)

        def parked_for(seconds):
            t_end = time.time() + seconds
            while time.time() < t_end:
                if not parked():
                    return False
                time.sleep(0.02)
            return True

        assert _wait(lambda: parked() and parked_for(0.8), 30.0)
        f.set_setting("whisper_model_path", "Micro B")
        t0 = time.time()
        assert f.update_model() == 0
        assert time.time() - t0 < 20.0
        n_a = len(texts)                                   # (the old worker is joined: nothing more arrives from model A)
        st = f.model_state()
        assert st == dict(model_path="Micro B", file_loaded=file_b, loaded_new=True, worker_running=True, dtw=False, context=True)

        # ... the context the filter now holds computes with B's weights (first call on it, against the first call of a
        # filter that loaded B's file directly; a third one on A's file answers differently), and the new worker serves it
        got_b = f.run_whisper_inference(x, 0, 3000, 1)
        g = pkg.Filter(precision=pkg.PRECISION_FP32)
        try:
            g.L.lv_filter_set_thresholds(g.f, 0.0, 1e6)
            g.params.temperature_inc = 0.0
            assert g.load_model(file_b) == 0
            want = g.run_whisper_inference(x, 0, 3000, 1)
            assert got_b == want and len(want["tokens"]) > 0
            g.L.lv_filter_unload_model(g.f)
            assert g.load_model(file_a) == 0
            assert g.run_whisper_inference(x, 0, 3000, 1)["tokens"] != want["tokens"]
        finally:
            g.close()
        _feed(f, x)
        assert _wait(lambda: len(texts) >= n_a + 1, 30.0)

        # unknown catalogue name: the reference shuts the worker down FIRST, then fails the lookup -> filter left stopped
        f.set_setting("whisper_model_path", "No Such Model")
        assert f.update_model() == 1
        st = f.model_state()
        assert (st["context"], st["worker_running"], st["model_path"]) == (False, False, "")
        # listed but not on disk (the reference opens its download dialog here)
        f.set_setting("whisper_model_path", "Listed, not downloaded")
        assert f.update_model() == 1 and not f.model_state()["context"]

        # external file
        f.set_setting("whisper_model_path", "!!!external!!!")
        assert f.update_model() == 1                      # "External model file path is empty"
        f.set_setting("whisper_model_path_external", file_a)
        assert f.update_model() == 0
        st = f.model_state()
        assert (st["model_path"], st["file_loaded"], st["worker_running"]) == ("!!!external!!!", file_a, True)
        ctx_ext = f.L.lv_filter_context(f.f)
        assert f.update_model() == 0 and f.L.lv_filter_context(f.f) == ctx_ext   # "External model file is already loaded"
        f.set_setting("whisper_model_path_external", file_b)                      # another external file: restart
        assert f.update_model() == 0 and f.model_state()["file_loaded"] == file_b

        # DTW flag on a catalogue model: the reference restarts with gf->whisper_model_path, which shutdown_whisper_thread
        # has just cleared -> "Failed to initialize whisper context", filter stopped; the NEXT call reloads with the flag set
        f.set_setting("whisper_model_path", "Micro A")
        assert f.update_model() == 0 and f.model_state()["loaded_new"]
        f.set_setting("dtw_token_timestamps", True)
        assert f.update_model() == 1
        st = f.model_state()
        assert (st["context"], st["worker_running"], st["model_path"], st["dtw"]) == (False, False, "", True)
        assert f.update_model() == 0
        st = f.model_state()
        assert (st["model_path"], st["file_loaded"], st["loaded_new"], st["worker_running"], st["dtw"]) == \
            ("Micro A", file_a, False, True, True)

        # whisper_free against a worker that is in the middle of whisper_full: stop() takes the context mutex, so it
        # returns only after the running inference and frees nothing under it
        n0 = len(requests)
        _feed(f, speechlike(9, 6.0))
        assert _wait(lambda: len(requests) > n0, 30.0)     # the worker has taken the audio and is (or was just) transcribing
        f.stop()
        st = f.model_state()
        assert (st["context"], st["worker_running"], st["model_path"]) == (False, False, "")
        # ... and against a parked worker, immediately after a start
        assert f.start(file_a) == 0
        f.stop()
        assert not f.model_state()["context"]
    finally:
        f.close()
