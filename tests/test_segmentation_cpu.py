"""CPU: the callers of the hot path — settings mapping, audio tap, ring buffers,
VAD iterator, the three segmentation modes — against the Python restatement in oracle/segmentation_oracle.py.

No model is loaded here: every inference request then ends in DETECTION_RESULT_UNKNOWN ("whisper context is
null", whisper-processing.cpp:188-192), which is exactly what makes the segmentation observable on a box
without a GPU: the request observer sees the PCM and the timestamps each call would have transcribed.
Parity unpinned at the LocalVocal boundary (no reference fixtures exist for segmentation; see the oracle header).
"""
import numpy as np
import pytest

from oracle import segmentation_oracle as so

SR = 16000


def speechlike(seed, seconds, burst=(0.4, 2.5), gap=(0.3, 1.0)):
    """Noise bursts (amp 0.1) separated by near-silence (amp 1e-4): far from the gate on both sides."""
    rng = np.random.RandomState(seed)
    n = int(seconds * SR)
    x = (1e-4 * rng.randn(n)).astype(np.float32)
    t = int(rng.uniform(0.0, 0.5) * SR)
    while t < n:
        b = int(rng.uniform(*burst) * SR)
        x[t:t + b] += (0.1 * rng.randn(min(b, n - t))).astype(np.float32)
        t += b + int(rng.uniform(*gap) * SR)
    return x


def drive(pkg, mode, partial, seed, seconds=24.0, segment_duration=3000, packet_ms=(10, 400), steps_every=(1, 6), silero=None):
    """silero = (onnx path, probability source for the restatement): the filter's VAD runs the Silero network on the GPU."""
    x = speechlike(seed, seconds)
    f = pkg.Filter()
    assert f.f
    if silero is not None:
        assert f.set_silero_model(silero[0]) == 0
    got = []
    f.set_callbacks(on_request=lambda pcm, t0, t1, vs: got.append((pcm, t0, t1, vs)))
    f.setup_audio(SR, 1)
    f.L.lv_filter_set_vad_mode(f.f, mode)
    f.L.lv_filter_set_partial(f.f, int(partial), 1100)
    f.L.lv_filter_set_segment_duration(f.f, segment_duration)
    orc = so.SegmenterOracle(vad_mode=mode, partial_transcription=partial, partial_latency=1100,
                             segment_duration=segment_duration,
                             **({} if silero is None else {"vad": so.VadIteratorOracle(prob_source=silero[1])}))
    rng = np.random.RandomState(seed + 1000)
    pos, k, states = 0, 0, []
    base_ns = 1_000_000_000
    while pos < len(x):
        n = min(int(rng.randint(packet_ms[0], packet_ms[1] + 1) * SR / 1000), len(x) - pos)
        ts = base_ns + pos * 1_000_000_000 // SR
        f.push_packet([x[pos:pos + n]], ts)
        orc.push_packet(x[pos:pos + n], ts)
        pos += n
        k += 1
        if k % int(rng.randint(steps_every[0], steps_every[1] + 1)) == 0:
            st = f.segmentation_step()
            so_st = orc.step()
            states.append(((st["vad_on"], st["start"], st["end"], st["last_partial_end"]), so_st.as_tuple()))
    for _ in range(3):  # drain
        st = f.segmentation_step()
        so_st = orc.step()
        states.append(((st["vad_on"], st["start"], st["end"], st["last_partial_end"]), so_st.as_tuple()))
    f.close()
    return got, orc.requests, states


@pytest.mark.parametrize("mode", [so.VAD_MODE_ACTIVE, so.VAD_MODE_HYBRID, so.VAD_MODE_DISABLED])
@pytest.mark.parametrize("partial", [False, True])
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_segmentation_matches_restatement(pkg, mode, partial, seed):
    got, want, states = drive(pkg, mode, partial, seed)
    for a, b in states:
        assert a == b
    assert len(got) == len(want)
    assert len(got) >= 2, "the scripted stream must actually trigger inference requests"
    for (pcm, t0, t1, vs), w in zip(got, want):
        assert (t0, t1, vs) == (w.t0, w.t1, w.vad_state)
        assert len(pcm) == len(w.pcm)
        assert np.array_equal(pcm, w.pcm)  # byte-exact: buffers only move samples
        # 10 ms of zeros either side (whisper-processing.cpp:331-341)
        assert not pcm[:160].any() and not pcm[-160:].any()
    if partial:
        assert any(vs == so.VAD_STATE_PARTIAL for _, _, _, vs in got)


def test_vad_iterator_matches_restatement(pkg):
    f = pkg.Filter()
    for seed in range(6):
        x = speechlike(seed + 50, 12.0, burst=(0.05, 1.5), gap=(0.05, 0.6))
        want = so.VadIteratorOracle().process(x, True)
        assert f.vad_speech_timestamps(x, True) == want
        assert len(want) >= 2
    # speech open at the end of the buffer is closed at the buffer length (silero-vad-onnx.cpp:275-283)
    x = np.concatenate([np.zeros(SR, np.float32), 0.1 * np.random.RandomState(3).randn(SR).astype(np.float32)])
    st = f.vad_speech_timestamps(x, True)
    assert st and st[-1][1] == len(x)
    # silence only / shorter than a window
    assert f.vad_speech_timestamps(np.zeros(SR, np.float32)) == []
    assert f.vad_speech_timestamps(np.zeros(100, np.float32)) == []
    f.close()


def test_ten_second_cap_and_timestamp_reset(pkg):
    f = pkg.Filter()
    got = []
    f.set_callbacks(on_request=lambda pcm, t0, t1, vs: got.append((len(pcm), t0, t1, vs)))
    f.setup_audio(SR, 1)
    f.L.lv_filter_set_vad_mode(f.f, so.VAD_MODE_DISABLED)
    f.L.lv_filter_set_segment_duration(f.f, 7000)
    orc = so.SegmenterOracle(vad_mode=so.VAD_MODE_DISABLED, segment_duration=7000)
    x = speechlike(9, 12.5)
    for i in range(0, len(x), 8000):  # 25 packets of 0.5 s
        ts = 5_000_000_000 + i * 1_000_000_000 // SR
        f.push_packet([x[i:i + 8000]], ts)
        orc.push_packet(x[i:i + 8000], ts)
    f.segmentation_step()
    orc.step()
    # one pass takes at most 10 s of input (vad-processing.cpp:44-65); the rest waits for the next pass
    assert got[0][0] == 10 * SR + 320 and f.L.lv_filter_input_bytes(f.f) == (len(x) - 10 * SR) * 4
    f.segmentation_step()
    orc.step()
    f.segmentation_step()  # nothing new: the leftover in the whisper buffer goes out as a final segment
    orc.step()
    assert [(n, a, b, c) for n, a, b, c in got] == [(len(r.pcm), r.t0, r.t1, r.vad_state) for r in orc.requests]
    assert f.L.lv_filter_input_bytes(f.f) == 0 and f.L.lv_filter_whisper_bytes(f.f) == 0
    # silence packets through the NULL-data push
    assert f.push_packet(160, 1) == 0 and f.L.lv_filter_input_bytes(f.f) == 640
    f.close()


def test_audio_tap_drops_without_model(pkg):
    f = pkg.Filter()
    f.setup_audio(SR, 1)
    # transcription-filter.cpp:86-89: no whisper context -> the packet passes through unbuffered
    assert f.filter_audio([np.zeros(160, np.float32)]) == 0
    assert f.L.lv_filter_input_bytes(f.f) == 0
    f.close()


def test_settings_map_onto_whisper_full_params(pkg):
    f = pkg.Filter()
    # defaults: plugin strategy BEAM_SEARCH, suppress_blank/non_speech false (whisper-params.cpp:59-60,86-87)
    assert f.apply_settings() == 0
    p = f.params
    assert p.strategy == pkg.SAMPLING_BEAM_SEARCH and p.beam_search.beam_size == 5
    assert not p.suppress_blank and not p.suppress_non_speech_tokens and not p.print_progress
    assert p.language == b"en" and not p.detect_language
    for k, v in {"strategy": 0, "n_threads": 3, "n_max_text_ctx": 99, "offset_ms": 7, "duration_ms": 2500,
                 "max_len": 40, "max_tokens": 31, "audio_ctx": 750, "greedy.best_of": 2, "beam_search.beam_size": 3}.items():
        f.set_setting(k, v)
    for k in ["whisper_translate", "no_timestamps", "single_segment", "token_timestamps", "split_on_word",
              "tdrz_enable", "suppress_blank", "suppress_non_speech_tokens", "print_special"]:
        f.set_setting(k, True)
    f.set_setting("no_context", False)
    for k, v in {"thold_pt": 0.02, "thold_ptsum": 0.03, "temperature": 0.25, "max_initial_ts": 0.5, "length_penalty": 0.75,
                 "temperature_inc": 0.125, "entropy_thold": 2.0, "logprob_thold": -0.5, "no_speech_thold": 0.375,
                 "beam_search.patience": 1.5, "sentence_psum_accept_thresh": 0.3}.items():
        f.set_setting(k, float(v))
    f.set_setting("suppress_regex", "ab+")
    f.set_setting("initial_prompt", "hello there")
    f.set_setting("whisper_language_select", "")
    f.set_setting("n_context_sentences", 2)
    assert f.apply_settings() == 0
    p = f.params
    assert (p.strategy, p.n_threads, p.n_max_text_ctx, p.offset_ms, p.duration_ms) == (0, 3, 99, 7, 2500)
    assert (p.max_len, p.max_tokens, p.audio_ctx, p.greedy.best_of, p.beam_search.beam_size) == (40, 31, 750, 2, 3)
    assert p.translate and p.no_timestamps and p.single_segment and p.token_timestamps and p.split_on_word
    assert p.tdrz_enable and p.suppress_blank and p.suppress_non_speech_tokens and p.print_special and not p.no_context
    assert [round(v, 6) for v in (p.thold_pt, p.thold_ptsum, p.temperature, p.max_initial_ts, p.length_penalty,
                                  p.temperature_inc, p.entropy_thold, p.logprob_thold, p.no_speech_thold,
                                  p.beam_search.patience)] == [0.02, 0.03, 0.25, 0.5, 0.75, 0.125, 2.0, -0.5, 0.375, 1.5]
    assert p.suppress_regex == b"ab+" and p.initial_prompt == b"hello there"
    # no language selected -> "auto" + detect_language (transcription-filter.cpp:427-433)
    assert p.language == b"auto" and p.detect_language
    f.close()


def test_remove_leading_trailing_nonalpha_observable_behaviour(pkg):
    r = pkg.remove_leading_trailing_nonalpha
    assert r(b"") == b"" and r(b"a") == b"a" and r(b".") == b"" and r(b" a") == b"a" and r(b"a.") == b"a"
    assert r(b"ab") == b"ab" and r(b"..") == b""
    # three bytes and more: exactly the first byte goes (see the note in segmentation.hip)
    assert r(b" Hello there.") == b"Hello there." and r(b"Hello") == b"ello" and r(b" a ") == b"a "


def test_resampler(pkg):
    f = pkg.Filter()
    x = speechlike(4, 1.0)
    f.setup_audio(SR, 1)
    assert np.array_equal(f.resample([x]), x)  # 16 kHz mono is the identity, bit for bit
    # 48 kHz stereo: a 440 Hz tone on both channels comes out as the same tone at 16 kHz, packet by packet
    f.setup_audio(48000, 2)
    t = np.arange(48000 * 2) / 48000.0
    s = (0.5 * np.sin(2 * np.pi * 440.0 * t)).astype(np.float32)
    out = np.concatenate([f.resample([s[i:i + 4800], s[i:i + 4800]]) for i in range(0, len(s), 4800)])
    assert abs(len(out) - 32000) <= 64
    ref = 0.5 * np.sin(2 * np.pi * 440.0 * np.arange(len(out)) / 16000.0)
    assert np.max(np.abs(out[200:] - ref[200:])) < 2e-3
    # content above the new Nyquist is rejected
    hf = (0.5 * np.sin(2 * np.pi * 11000.0 * t)).astype(np.float32)
    f.setup_audio(48000, 1)
    out = np.concatenate([f.resample([hf[i:i + 4800]]) for i in range(0, len(hf), 4800)])
    assert np.max(np.abs(out[200:])) < 5e-3
    f.close()
