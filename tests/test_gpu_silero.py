"""GPU: the Silero VAD network as HIP kernels (csrc/silero_vad.hip) against its CPU restatement
(oracle/silero_vad_oracle.py) — SURVEY.md §8 f2; the reference side is VadIterator::predict_one / process,
/root/reference/src/whisper-utils/silero-vad-onnx.cpp:111-140, 262-284, driven from vad_based_segmentation
(/root/reference/src/whisper-utils/vad-processing.cpp:202-367).

Weights: synthetic, of the graph's shapes, in an .onnx file of the real file's nesting (the GPU box has no
/root/reference; tests/test_silero_cpu.py checks both readers on the reference's own file where it exists).
Tolerance: fp32 both sides, different summation orders (the restatement accumulates in float64): |dp| < 1e-4 on the
probabilities, 1e-3 on the recurrent state after 250 windows.  PARITY UNPINNED (no ONNX Runtime here, no expected
outputs in the reference).
"""
import numpy as np
import pytest

from oracle import segmentation_oracle as sg
from oracle import silero_vad_oracle as so
from test_segmentation_cpu import drive, speechlike

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(300)]
W = 512


@pytest.fixture(scope="module")
def model(tmp_path_factory):
    w = so.synth_weights(11)
    path = str(tmp_path_factory.mktemp("silero") / "silero_synth.onnx")
    so.write_onnx(path, w)
    return path, w


def test_probabilities_and_state_match_the_restatement(pkg, model):
    path, w = model
    vad = pkg.SileroVad(path)
    try:
        x = speechlike(21, 8.0)
        got = vad.process(x, reset=True)
        orc = so.SileroOracle(w)
        want = orc.process(x, reset=True)
        assert len(got) == len(x) // W == len(want)
        assert want.min() < 0.2 and want.max() > 0.9            # the signal drives the detector both ways
        err = float(np.abs(got - want).max())
        print(f"silero probabilities: max |gpu - restatement| = {err:.2e} over {len(got)} windows")
        assert err < 1e-4
        state = vad.state()
        assert np.abs(state[:128] - orc.h).max() < 1e-3 and np.abs(state[128:] - orc.c).max() < 1e-3
    finally:
        vad.close()


def test_the_reference_models_own_weights_through_the_kernels(pkg, tmp_path):
    """The tensors of the reference's data/models/silero-vad/silero_vad.onnx (16 kHz branch; tests/golden/silero_v5_16k.npz,
    written by tools/gen_silero_golden.py in the build container) rebuilt into a model file and run through the HIP
    kernels: probabilities within 1e-4 of the restatement's on the same weights (fixture) and of the restatement run here;
    the network answers like a VAD on them (a vowel-like buzz high, noise low).  The reference side is
    VadIterator::predict_one, /root/reference/src/whisper-utils/silero-vad-onnx.cpp:111-140.  PARITY UNPINNED (no ONNX
    Runtime output to compare with)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "silero_v5_16k.npz"))
    w = {k[2:]: g[k] for k in g.files if k.startswith("w_")}
    assert {k: v.shape for k, v in w.items()} == {k: shape for k, (_, shape) in so.SUFFIXES.items()}
    path = str(tmp_path / "silero_real_weights.onnx")
    so.write_onnx(path, w)
    vad = pkg.SileroVad(path)
    try:
        for name in ("voiced", "gated", "noise"):
            x, want = g["clip_" + name], g["prob_" + name]
            got = vad.process(x, reset=True)
            here = so.SileroOracle(w).process(x, reset=True)
            err = float(np.abs(got - want).max())
            print(f"silero, reference weights, clip {name}: max |gpu - restatement| = {err:.2e}, p in [{got.min():.3f}, {got.max():.3f}]")
            assert len(got) == len(x) // W == len(want) and err < 1e-4 and np.abs(got - here).max() < 1e-4
        assert vad.process(g["clip_voiced"], reset=True).max() > 0.6 and vad.process(g["clip_noise"], reset=True)[8:].max() < 0.15
        p = vad.process(g["clip_gated"], reset=True)
        assert p[:25].max() > 0.5 and p[45:60].max() < 0.1      # speech, then the silent second (windows 31 .. 62)
    finally:
        vad.close()


def test_state_carries_between_calls_and_reset_clears_it(pkg, model):
    path, _ = model
    vad = pkg.SileroVad(path)
    try:
        x = speechlike(22, 6.0)
        whole = vad.process(x, reset=True)
        cut = 57 * W
        first = vad.process(x[:cut], reset=True)
        second = vad.process(x[cut:], reset=False)
        assert np.array_equal(np.concatenate([first, second]), whole)        # the same arithmetic, window by window
        assert not np.array_equal(vad.process(x[cut:], reset=True), second)  # ... which the state is part of
        assert np.array_equal(vad.process(x, reset=True), whole)
        # ragged tail ignored (VadIterator::process stops at the last whole window), nothing to do for less than a window
        assert np.array_equal(vad.process(x[:cut + 100], reset=True), first)
        before = vad.state()
        assert len(vad.process(x[:100], reset=False)) == 0 and np.array_equal(vad.state(), before)
        assert len(vad.process(x[:0], reset=True)) == 0 and not vad.state().any()
        # many windows in one call (10 minutes): the recurrent kernel's loop is bounded by its argument
        long = np.tile(x, 100)
        p = vad.process(long, reset=True)
        assert len(p) == len(long) // W and np.isfinite(p).all() and np.array_equal(p[:len(whole)], whole)
    finally:
        vad.close()


def test_batched_streams_equal_single_stream_calls(pkg, model):
    # SURVEY.md §8 f2 "batched across streams": the windows of several streams in one front-end launch + one recurrent launch
    # with a workgroup per stream; per stream exactly what lv_silero_process returns, state carried per stream
    path, _ = model
    n = 9
    solo = [pkg.SileroVad(path) for _ in range(n)]
    together = [pkg.SileroVad(path) for _ in range(n)]
    try:
        rng = np.random.RandomState(5)
        for rnd in range(3):
            wavs = [speechlike(100 + 10 * rnd + i, float(rng.uniform(0.2, 3.0))) for i in range(n)]
            wavs[2] = wavs[2][:100]                              # less than a window: nothing to do for this stream
            resets = [rnd == 0 or bool(rng.randint(0, 2)) for _ in range(n)]
            want = [v.process(w, reset=r) for v, w, r in zip(solo, wavs, resets)]
            got = pkg.silero_process_batch(together, wavs, resets)
            for i in range(n):
                assert np.array_equal(got[i], want[i]), (rnd, i)
                assert np.array_equal(together[i].state(), solo[i].state()), (rnd, i)
        assert any(len(w) for w in want)
        # a stream twice in one batch, or streams of different models, are refused
        with pytest.raises(RuntimeError):
            pkg.silero_process_batch([together[0], together[0]], [wavs[0], wavs[1]], [True, True])
    finally:
        for v in solo + together:
            v.close()


def test_vad_iterator_on_the_network_matches_the_restatement(pkg, model):
    path, w = model
    f = pkg.Filter()
    try:
        assert f.set_silero_model(path) == 0 and f.L.lv_filter_vad_is_silero(f.f) == 1
        for seed in (70, 72, 73, 76, 79):   # (signals on which the restatement's probabilities keep the margin asserted below)
            x = speechlike(seed, 12.0, burst=(0.05, 1.5), gap=(0.05, 0.6))
            orc = sg.VadIteratorOracle(prob_source=so.SileroOracle(w))
            want = orc.process(x, True)
            # the comparison is of decisions: it needs every probability clear of both thresholds by more than the tolerance
            assert min(np.abs(orc.probs - 0.5).min(), np.abs(orc.probs - 0.35).min()) > 1e-3
            assert f.vad_speech_timestamps(x, True) == want and len(want) >= 2
            assert np.abs(f.vad_probabilities() - orc.probs).max() < 1e-4
        # without a reset the network's state carries into the next buffer, on both sides
        orc = sg.VadIteratorOracle(prob_source=so.SileroOracle(w))
        a, b = speechlike(90, 4.0), speechlike(91, 4.0)
        assert f.vad_speech_timestamps(a, True) == orc.process(a, True)
        assert f.vad_speech_timestamps(b, False) == orc.process(b, False)
        assert np.abs(f.vad_probabilities() - orc.probs).max() < 1e-4
    finally:
        f.close()


@pytest.mark.parametrize("mode", [sg.VAD_MODE_ACTIVE, sg.VAD_MODE_HYBRID])
def test_segmentation_driven_by_the_network_matches_the_restatement(pkg, model, mode):
    path, w = model
    got, want, states = drive(pkg, mode, True, 3, seconds=20.0, silero=(path, so.SileroOracle(w)))
    for a, b in states:
        assert a == b
    assert len(got) == len(want) >= 2
    for (pcm, t0, t1, vs), r in zip(got, want):
        assert (t0, t1, vs) == (r.t0, r.t1, r.vad_state) and np.array_equal(pcm, r.pcm)


def test_worker_thread_with_the_network_as_its_vad(pkg, model, model_file):
    # whisper_loop end to end (audio tap -> vad_based_segmentation on the device network -> run_whisper_inference on the
    # same GPU): the requests the worker sends are those of the restatement driven by the restated network
    import time
    path, w = model
    f = pkg.Filter(precision=pkg.PRECISION_FP32)
    requests, texts = [], []
    f.set_callbacks(on_text=lambda d: texts.append(d), on_request=lambda pcm, t0, t1, vs: requests.append((pcm, t0, t1, vs)))
    try:
        assert f.set_silero_model(path) == 0
        f.setup_audio(16000, 1)
        f.L.lv_filter_set_vad_mode(f.f, sg.VAD_MODE_ACTIVE)
        f.L.lv_filter_set_partial(f.f, 0, 1100)
        f.L.lv_filter_set_segment_duration(f.f, 3000)
        f.L.lv_filter_set_thresholds(f.f, 0.0, 1e6)
        f.params.temperature_inc = 0.0
        assert f.start(model_file("micro.en")) == 0
        assert f.L.lv_filter_vad_is_silero(f.f) == 1          # start_whisper_thread_with_path built the same iterator
        orc = sg.SegmenterOracle(vad_mode=sg.VAD_MODE_ACTIVE, partial_transcription=False, partial_latency=1100, segment_duration=3000,
                                 vad=sg.VadIteratorOracle(prob_source=so.SileroOracle(w)))
        x = speechlike(3, 10.0)

        def wait(cond, timeout_s):
            t0 = time.time()
            while time.time() - t0 < timeout_s:
                if cond():
                    return True
                time.sleep(0.01)
            return False

        for i in range(0, len(x), 4000):   # lock-step with the worker, as the offline harness feeds (one pass per packet)
            ts = 1_000_000_000 + i * 1_000_000_000 // 16000
            assert f.push_packet([x[i:i + 4000]], ts) == 0
            orc.push_packet(x[i:i + 4000], ts)
            orc.step()
            assert wait(lambda: f.L.lv_filter_input_bytes(f.f) == 0, 20.0)
            assert wait(lambda: len(requests) >= len(orc.requests) and len(texts) >= len(orc.requests), 30.0)
        f.stop()
        assert len(orc.requests) >= 2 and len(requests) == len(orc.requests) == len(texts)
        for (pcm, t0, t1, vs), r in zip(requests, orc.requests):
            assert (t0, t1, vs) == (r.t0, r.t1, r.vad_state) and np.array_equal(pcm, r.pcm)
    finally:
        f.close()


def test_a_file_that_is_not_the_graph_is_refused(pkg, model, tmp_path):
    path, _ = model
    cut = str(tmp_path / "cut.onnx")
    blob = open(path, "rb").read()
    open(cut, "wb").write(blob[:len(blob) // 2])
    f = pkg.Filter()
    try:
        with pytest.raises(RuntimeError):
            f.set_silero_model(cut)
    finally:
        f.close()
    with pytest.raises(RuntimeError):
        pkg.SileroVad(cut)
