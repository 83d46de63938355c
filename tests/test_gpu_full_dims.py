"""GPU: the benchmarked configurations at their OWN dimensions against the oracle, teacher-forced.

  BASELINE configs[2]  Whisper large-v3 dims (d=1280, 32+32 layers, 128 mel bins, vocab 51866), bf16, 5 rows per step
  BASELINE configs[3]  large-v3-turbo dims (4 decoder layers), bf16, 64 concurrent contexts on one upload of the weights
  BASELINE configs[4]  large-v3 dims, e4m3 decoder weights, Silero-VAD-gated 3 s chunks with partials through whisper_loop

Seeded synthetic weights in the GGML container (tools/synth_large; no checkpoints exist offline).  The bf16 engine cannot be
token-exact against the fp32 oracle, so the comparison is teacher-forced: the ORACLE's token histories are fed to both
sides, one step at a time (5 rows of 5 different sequences per engine step = the beam-5 shape the bench runs), and every
step is checked on its own:
  * encoder output: max |engine - oracle| <= ENC_TOL x max |oracle|
  * logits of every row: max |engine - oracle| <= LOGIT_TOL x (max - min of the oracle's row)   (stated bf16 tolerance)
  * top-1 agreement on every row whose oracle top-2 margin exceeds twice that row's tolerance
A 12-token prompt on a sixth sequence covers the > 5-row prefill path (VALU kernels on bf16 weights) at full width.
"""
import threading

import numpy as np
import pytest

from oracle import pyoracle
from tools import ggml_io, synth_large

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]

ENC_TOL = 0.02     # bf16 operands through 32 encoder layers (measured 0.008)
LOGIT_TOL = 0.01   # of the row's logit spread, through up to 32 decoder layers (measured 0.005)


@pytest.fixture(scope="module")
def big_model(tmp_path_factory):
    d = tmp_path_factory.mktemp("full_dims")
    cache = {}

    def get(preset, fp8_decoder=False):
        key = (preset, fp8_decoder)
        if key not in cache:
            path = str(d / f"ggml-{preset}-synth-s1234{'-e4m3' if fp8_decoder else ''}.bin")
            synth_large.write_large(path, preset, seed=1234, quiet_special=True, fp8_decoder=fp8_decoder)
            cache[key] = path
        return cache[key]

    return get


def _teacher_forced(pkg, path, n_rows=5, steps=8, precision=None):
    pyoracle.lib().ow_set_threads(16)
    pcm = ggml_io.synth_pcm(100, 30.0)
    orc = pyoracle.Oracle(path)
    eng = pkg.Engine(path, precision=pkg.PRECISION_BF16 if precision is None else precision)
    try:
        orc.mel(pcm)
        eng.mel(pcm)
        eo, eg = orc.encode(0), eng.encode(0)
        enc_rel = float(np.abs(eg - eo).max() / np.abs(eo).max())
        print(f"encoder [{eo.shape[0]} x {eo.shape[1]}], {orc.n_audio_layer} layers: max |bf16 - fp32 oracle| / max |oracle| = {enc_rel:.4f}")
        assert np.isfinite(eg).all() and enc_rel < ENC_TOL
        lang_en = orc.sot + 1
        prompt = [orc.sot, lang_en, orc.transcribe]
        worst, checked_top1, agree_top1 = 0.0, 0, 0
        margins = []

        def compare(ref_row, got_row, what):
            nonlocal worst, checked_top1, agree_top1
            spread = float(ref_row.max() - ref_row.min())
            err = float(np.abs(got_row - ref_row).max())
            worst = max(worst, err / spread)
            assert err <= LOGIT_TOL * spread, f"{what}: |d logits| {err:.4f} > {LOGIT_TOL} x spread {spread:.3f}"
            top2 = np.partition(ref_row, -2)[-2:]
            margins.append(float(top2[1] - top2[0]) / spread)
            if float(top2[1] - top2[0]) > 2.0 * LOGIT_TOL * spread:
                checked_top1 += 1
                agree_top1 += int(np.argmax(got_row) == np.argmax(ref_row))
                assert np.argmax(got_row) == np.argmax(ref_row), what

        # prompt prefill per sequence (3 rows in one call), then row r continues with the oracle's (r+1)-th best text token
        hist = []
        for r in range(n_rows):
            ref = orc.decode(prompt, 0, r)
            got = eng.decode(prompt, 0, r)
            for i in range(len(prompt)):
                compare(ref[i], got[i], f"prefill seq {r} row {i}")
            order = np.argsort(-ref[-1][:orc.eot])
            hist.append([int(order[r])])
        n_past = len(prompt)
        for s in range(steps):
            toks = [h[-1] for h in hist]
            got = eng.decode_rows(toks, [n_past] * n_rows, list(range(n_rows)))   # ONE engine step, n_rows sequences
            for r in range(n_rows):
                ref = orc.decode([toks[r]], n_past, r)[0]
                compare(ref, got[r], f"step {s} row {r}")
                hist[r].append(int(np.argmax(ref[:orc.eot])))                     # teacher forcing: the ORACLE's choice
            n_past += 1
        # > 5 rows in one call: a 12-token prompt on another sequence (prefill path)
        long_prompt = [orc.prev] + [int(t) for t in hist[0][:8]] + prompt
        ref = orc.decode(long_prompt, 0, n_rows)
        got = eng.decode(long_prompt, 0, n_rows)
        for i in (0, 5, len(long_prompt) - 1):
            compare(ref[i], got[i], f"12-token prefill row {i}")
        print(f"{orc.n_text_layer} decoder layers, {n_rows} rows x {steps} teacher-forced steps: worst |d logits| / spread = {worst:.4f}; "
              f"top-1 equal on {agree_top1}/{checked_top1} rows with a decisive oracle margin")
        print("oracle top-2 margins / spread: median %.4f, max %.4f, %d rows" % (float(np.median(margins)), max(margins), len(margins)))
    finally:
        eng.close()
        orc.close()


def test_large_v3_dims_bf16_teacher_forced_beam5_rows(pkg, big_model):
    _teacher_forced(pkg, big_model("large-v3"), n_rows=5, steps=8)


def test_turbo_dims_bf16_teacher_forced_and_64_shared_weight_contexts(pkg, big_model):
    path = big_model("large-v3-turbo")
    _teacher_forced(pkg, path, n_rows=5, steps=8)
    # configs[3] at its own size: 64 concurrent streams on ONE upload of the weights, decode steps and encoder windows of
    # different streams batched by the hub, give each stream the tokens it gets alone
    n_streams, steps = 64, 24
    clips = [ggml_io.synth_pcm(100 + i, 30.0) for i in range(n_streams)]
    first = pkg.Engine(path, precision=pkg.PRECISION_BF16)
    engs = [first] + [pkg.Engine(path, precision=pkg.PRECISION_BF16, share=first) for _ in range(n_streams - 1)]
    try:
        params = []
        for e in engs:
            e.set_force_decode_steps(steps)
            params.append(e.default_params(pkg.SAMPLING_GREEDY, temperature_inc=0.0, duration_ms=30000, language="en"))
        solo = []
        for i, e in enumerate(engs):
            assert e.full(params[i], clips[i]) == 0
            solo.append(e.token_ids())
        # (random weights: a timestamp token may close the segment long before the forced length)
        assert all(len(t) >= 1 for t in solo) and sum(len(t) >= 4 for t in solo) >= n_streams // 2
        assert len({tuple(t) for t in solo}) > 1          # the clips do lead to different transcripts
        out, errs = [None] * n_streams, []
        s0 = first.hub_stats()

        def work(i):
            try:
                for _ in range(2):
                    assert engs[i].full(params[i], clips[i]) == 0
                    out[i] = engs[i].token_ids()
            except Exception as ex:  # noqa: BLE001
                errs.append(repr(ex))

        th = [threading.Thread(target=work, args=(i,)) for i in range(n_streams)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        assert out == solo
        s1 = first.hub_stats()
        d_steps, d_rows, d_req, d_pass, d_win = (s1[k] - s0[k] for k in range(5))
        print(f"hub, 64 turbo streams: {d_rows / d_steps:.1f} rows per step, {d_win / d_pass:.1f} windows per encoder pass")
        assert d_rows / d_steps > 4 and d_win / d_pass > 2   # the streams did share steps and passes
    finally:
        for e in engs[1:] + engs[:1]:
            e.close()


def test_turbo_dims_fp8_weights_teacher_forced_and_bit_equal_to_bf16_on_the_same_weights(pkg, big_model):
    """BASELINE configs[4] precision: decoder projections + logits matrix as e4m3 bytes with a power-of-two scale per row.
    The file's decoder weights are written already rounded to that grid (tools/ggml_io.e4m3_quantize_dequantize_rows), so
    the fp32 oracle, the bf16 engine and the fp8 engine all hold the SAME weights and the comparison isolates the kernels:
      * against the oracle: the bf16 tolerances of the tests above, unchanged;
      * against the bf16 engine: the logits are equal to the bit (e4m3 -> bf16 is exact, the scale is a power of two applied
        to the finished fp32 sum) — every fp8 kernel variant (qkv, cross-q, fc1, attention outputs, fc2, logits; 1, 5 and 16
        rows) is checked element by element this way.
    On a file that is NOT pre-rounded the fp8 engine differs from the bf16 engine by the quantisation itself (printed)."""
    path = big_model("large-v3-turbo", fp8_decoder=True)
    _teacher_forced(pkg, path, n_rows=5, steps=8, precision=pkg.PRECISION_FP8)
    pcm = ggml_io.synth_pcm(100, 30.0)

    def run(p, precision):
        eng = pkg.Engine(p, precision=precision)
        try:
            eng.mel(pcm)
            eng.encode(0)
            prompt = [eng.sot, eng.sot + 1, eng.L.whisper_token_transcribe(eng.ctx)]
            out = []
            for r in range(8):
                out.append(eng.decode(prompt, 0, r))       # 3-row calls (5-row kernels)
            out.append(eng.decode_rows([100 + r for r in range(8)], [3] * 8, list(range(8))))       # 8 rows -> 16-row kernels
            out.append(eng.decode_rows([7], [4], [0]))                                              # 1-row kernels
            out.append(eng.decode_rows([200 + r for r in range(5)], [4] * 5, [1, 2, 3, 4, 5]))      # 5-row kernels
            out.append(eng.decode([eng.sot - 1 - r for r in range(12)], 5, 6))                      # 12-token call, 16-row kernels
            eng.set_force_decode_steps(24)
            par = eng.default_params(pkg.SAMPLING_BEAM_SEARCH, temperature_inc=0.0, duration_ms=30000, language="en")
            assert eng.full(par, pcm) == 0
            return out, eng.token_ids()
        finally:
            eng.close()

    l8, t8 = run(path, pkg.PRECISION_FP8)
    l16, t16 = run(path, pkg.PRECISION_BF16)
    for a, b in zip(l8, l16):
        assert np.isfinite(a).all() and np.array_equal(a, b)
    assert t8 == t16 and len(t8) >= 1
    # the price of the quantisation itself, on weights that are not on the e4m3 grid
    raw = big_model("large-v3-turbo")
    q8, _ = run(raw, pkg.PRECISION_FP8)
    q16, _ = run(raw, pkg.PRECISION_BF16)
    worst = max(float(np.abs(a - b).max() / (b.max() - b.min())) for a, b in zip(q8, q16))
    print(f"e4m3 weights vs bf16 weights of an unrounded file: worst |d logits| / spread = {worst:.4f}")
    assert worst < 0.25


def test_large_v3_dims_long_prompt_prefill_in_chunks_of_16_rows(pkg, big_model):
    """LocalVocal's context-sentence path hands whisper_full an initial_prompt / prompt_past of up to 224 tokens on every call
    (/root/reference/src/whisper-utils/whisper-processing.cpp:191-199; the offline harness runs no_context=false,
    localvocal-offline-test.cpp:184).  A 200-token prompt goes through the hub's 16-row matrix-core kernels in 13 steps (one
    pass over the weights per 16 tokens): the logits of its last 32 rows against the fp32 oracle (bf16 tolerance of the
    tests above), the same call on a context that takes the per-context path (MI355_WHISPER_NO_BATCH: VALU kernels) within
    the same bound of it, and a decode step behind the prompt sees all 200 cached positions."""
    import os
    path = big_model("large-v3")
    pyoracle.lib().ow_set_threads(16)
    pcm = ggml_io.synth_pcm(100, 30.0)
    orc = pyoracle.Oracle(path)
    eng = pkg.Engine(path, precision=pkg.PRECISION_BF16)
    try:
        orc.mel(pcm)
        eng.mel(pcm)
        orc.encode(0)
        eng.encode(0)
        rng = np.random.RandomState(5)
        prompt = [orc.prev] + [int(t) for t in rng.randint(0, orc.eot, 196)] + [orc.sot, orc.sot + 1, orc.transcribe]
        assert len(prompt) == 200
        s0 = eng.hub_stats()
        ref = orc.decode(prompt, 0, 0)[-32:]
        got = eng.decode(prompt, 0, 0)
        s1 = eng.hub_stats()
        assert s1[0] - s0[0] == 13 and s1[1] - s0[1] == 200         # 13 steps of <= 16 rows on the hub
        worst = 0.0
        for i in range(32):
            spread = float(ref[i].max() - ref[i].min())
            err = float(np.abs(got[i] - ref[i]).max())
            worst = max(worst, err / spread)
            assert err <= LOGIT_TOL * spread, f"prompt row {168 + i}"
        nxt = int(np.argmax(ref[-1][:orc.eot]))
        ref1 = orc.decode([nxt], 200, 0)[0]
        got1 = eng.decode_rows([nxt], [200], [0])[0]
        spread = float(ref1.max() - ref1.min())
        assert float(np.abs(got1 - ref1).max()) <= LOGIT_TOL * spread
        print(f"200-token prompt in 13 hub steps: worst |d logits| / spread over the last 32 rows = {worst:.4f}")
    finally:
        eng.close()
        orc.close()


def test_large_v3_dims_fp8_weights_teacher_forced(pkg, big_model):
    """BASELINE configs[4], the model half at its own size: 32 decoder layers of e4m3 projections + the e4m3 logits matrix
    against the fp32 oracle on the same (pre-rounded) weights, tolerances of the bf16 tests unchanged."""
    _teacher_forced(pkg, big_model("large-v3", fp8_decoder=True), n_rows=5, steps=8, precision=pkg.PRECISION_FP8)


def _config4_runner(pkg, path, tmp_path, mode):
    """-> (run(precision) -> (requests, results), sg): one filter life through whisper_loop on a 20 s speech-like stream; run()
    asserts that the inference requests are those of the segmentation restatement driven by the restated Silero network"""
    import time

    from oracle import segmentation_oracle as sg
    from oracle import silero_vad_oracle as so
    from test_segmentation_cpu import speechlike

    w = so.synth_weights(11)
    vad_path = str(tmp_path / "silero_synth.onnx")
    so.write_onnx(vad_path, w)
    x = speechlike(7, 20.0)
    vad_mode = sg.VAD_MODE_ACTIVE if mode == "active" else sg.VAD_MODE_HYBRID

    def wait(cond, timeout_s):
        t0 = time.time()
        while time.time() - t0 < timeout_s:
            if cond():
                return True
            time.sleep(0.005)
        return False

    def run(precision):
        import ctypes
        libc = ctypes.CDLL(None)
        f = pkg.Filter(precision=precision)
        requests, results = [], []

        def on_text(d):
            d["all_tokens"] = f.engine().token_ids()   # on the worker thread, behind its whisper_full: nobody else uses the context
            results.append(d)

        f.set_callbacks(on_text=on_text, on_request=lambda pcm, t0, t1, vs: requests.append((pcm, t0, t1, vs)))
        try:
            assert f.set_silero_model(vad_path) == 0
            f.setup_audio(16000, 1)
            f.L.lv_filter_set_vad_mode(f.f, vad_mode)
            f.L.lv_filter_set_partial(f.f, 1, 1100)
            f.L.lv_filter_set_segment_duration(f.f, 3000)
            f.L.lv_filter_set_thresholds(f.f, 0.0, 1e6)
            f.params.temperature_inc = 0.0
            assert f.start(path) == 0 and f.L.lv_filter_vad_is_silero(f.f) == 1
            f.engine().set_force_decode_steps(12)
            orc = sg.SegmenterOracle(vad_mode=vad_mode, partial_transcription=True, partial_latency=1100,
                                     segment_duration=3000, vad=sg.VadIteratorOracle(prob_source=so.SileroOracle(w)))
            for i in range(0, len(x), 4000):   # lock-step with the worker, as the offline harness feeds (one pass per packet)
                ts = 1_000_000_000 + i * 1_000_000_000 // 16000
                # requests under 1 s are centred in rand() noise (whisper-processing.cpp:162-180).  rand() is process-global
                # and the HIP runtime draws from it too while it initialises (a first run in a fresh process differed from
                # the second at the first short request): re-seeded before every packet, so every run pads alike
                libc.srand(1234)
                assert f.push_packet([x[i:i + 4000]], ts) == 0
                orc.push_packet(x[i:i + 4000], ts)
                orc.step()
                assert wait(lambda: f.L.lv_filter_input_bytes(f.f) == 0, 30.0)
                assert wait(lambda: len(requests) >= len(orc.requests) and len(results) >= len(orc.requests), 60.0)
            f.stop()
            assert len(requests) == len(orc.requests) == len(results)
            for (pcm, t0, t1, vs), r in zip(requests, orc.requests):
                assert (t0, t1, vs) == (r.t0, r.t1, r.vad_state) and np.array_equal(pcm, r.pcm)
            return requests, results
        finally:
            f.close()

    return run, sg


@pytest.mark.parametrize("mode", ["active", "hybrid"])
def test_config4_streaming_partials_large_v3_fp8_through_whisper_loop(pkg, big_model, tmp_path, mode):
    """BASELINE configs[4] as ONE thing: large-v3 dims, e4m3 decoder weights, the Silero network on the GPU gating 3 s
    segments with partials every 1100 ms, everything through the worker thread (whisper_loop ->
    vad_based_segmentation -> run_inference_and_callbacks -> run_whisper_inference -> whisper_full;
    /root/reference/src/whisper-utils/whisper-processing.cpp:326-431, vad-processing.cpp:202-367).
      * the inference requests (PCM bytes, timestamps, VAD state) are those of the segmentation restatement driven by the
        restated network;
      * per request the fp8 filter returns what a bf16 filter returns on the same file: result enum, text, kept tokens with
        their probabilities and the full whisper_full token sequence, equal to the bit (e4m3 -> bf16 is exact and the row
        scales are powers of two; the model arithmetic itself is pinned by the teacher-forced test above).
    12 forced tokens per call, beam 5 (the plugin's default strategy), temperature fallback off.  VAD mode ACTIVE cuts at the
    network's speech ends (segment_duration is not read in that mode, vad-processing.cpp:202-367); HYBRID adds the 3 s cap
    of segment_duration (vad-processing.cpp:369-454)."""
    run, sg = _config4_runner(pkg, big_model("large-v3", fp8_decoder=True), tmp_path, mode)
    req8, res8 = run(pkg.PRECISION_FP8)
    req16, res16 = run(pkg.PRECISION_BF16)
    n_partial = sum(vs == sg.VAD_STATE_PARTIAL for _, _, _, vs in req8)
    print(f"configs[4]: {len(req8)} inference requests ({n_partial} partial), "
          f"{sum(len(r['all_tokens']) for r in res8)} tokens; mean chunk {np.mean([len(p) for p, _, _, _ in req8]) / 16000:.2f} s")
    assert len(req8) >= 6 and (mode == "hybrid" or 0 < n_partial < len(req8))
    assert all(len(p) <= 16000 * 11 for p, _, _, _ in req8)                  # MAX_MS_WORK_BUFFER
    # (random weights: a timestamp token may close a call's segment before the forced length, and a chunk under 1 s is padded)
    assert all(len(r["all_tokens"]) >= 1 for r in res8) and sum(len(r["all_tokens"]) >= 4 for r in res8) >= len(res8) // 2
    assert len({tuple(r["all_tokens"]) for r in res8}) > 1                   # ... and the chunks lead to different transcripts
    for k, (a, b) in enumerate(zip(res8, res16)):
        assert a == b, f"request {k} ({len(req8[k][0])} samples, VAD state {req8[k][3]})"


def test_config4_streaming_partials_with_the_mx_fp8_encoder_through_whisper_loop(pkg, big_model, tmp_path):
    """configs[4] with MW_PRECISION_FP8_MX (e4m3 decoder weights + the encoder's large products in MX-fp8): the same stream
    through the same worker.  The requests are the restatement's (asserted by the runner); the results cannot equal the
    bf16 filter's (the encoder computes in another format: tests/test_gpu_mx_encoder.py holds it to its own rule) — what
    is asserted is that the mode is deterministic through the worker (two filter lives give the same results: a window
    encodes to the same bits whichever tile height its pass chose) and that it produces transcripts."""
    run, sg = _config4_runner(pkg, big_model("large-v3", fp8_decoder=True), tmp_path, "active")
    req_a, res_a = run(pkg.PRECISION_FP8_MX)
    req_b, res_b = run(pkg.PRECISION_FP8_MX)
    assert len(req_a) == len(req_b) >= 6
    assert all(len(r["all_tokens"]) >= 1 for r in res_a) and len({tuple(r["all_tokens"]) for r in res_a}) > 1
    for k, (a, b) in enumerate(zip(res_a, res_b)):
        assert a == b, f"request {k}"
