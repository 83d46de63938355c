"""GPU: the BASELINE.json / SURVEY.md §8(d) configurations that are parity cases rather than bench lines, at the real
model dimensions (seeded synthetic weights in the GGML container; no checkpoints exist offline).

  configs[0]  tiny.en dims, 30 s, greedy, the offline-test parameter set (localvocal-offline-test.cpp:176-200)
  configs[1]  base.en dims, greedy fp32, one 30 s window: token ids equal to the oracle's
  configs[3]  several independent streams per GPU (one context per stream), greedy, partitioned i mod n
With random weights a window would stop after a few tokens, so the decode length is forced on both sides.
"""
import numpy as np
import pytest

from oracle import pyoracle
from tools import ggml_io, stream_shard

pytestmark = pytest.mark.gpu


def _forced_tokens(pkg, path, pcm, steps, precision, strategy=0, **kw):
    eng = pkg.Engine(path, precision=precision)
    try:
        eng.set_force_decode_steps(steps)
        p = eng.default_params(strategy, temperature_inc=0.0, duration_ms=int(len(pcm) * 1000 // 16000), **kw)
        assert eng.full(p, pcm) == 0
        return eng.token_ids(), eng.segments()
    finally:
        eng.close()


@pytest.mark.parametrize("preset,steps", [("tiny.en", 64), ("base.en", 48)])
def test_real_dims_greedy_fp32_tokens_equal_oracle(pkg, model_file, preset, steps):
    path = model_file(preset, ftype=1)          # f16 matrices, like the catalogue's ggml-*.bin files
    pcm = ggml_io.synth_pcm(0, 30.0)            # chirps + AM + noise, seed 0, 480 000 samples
    # offline-test parameter set: greedy, no_context, single language "en", timestamps on
    kw = dict(language="en", no_context=True, single_segment=False, suppress_blank=True)
    got, segs = _forced_tokens(pkg, path, pcm, steps, pkg.PRECISION_FP32, **kw)
    orc = pyoracle.Oracle(path)
    try:
        po = orc.default_params(0, temperature_inc=0.0, duration_ms=30000, force_decode_steps=steps,
                                **{k: (int(v) if isinstance(v, bool) else v) for k, v in kw.items()})
        assert orc.full(po, pcm) == 0
        want = orc.token_ids()
    finally:
        orc.close()
    assert len(want) >= steps // 2
    assert got == want
    # bf16 mode on the same file: same length, match rate reported (not required to be exact)
    got16, _ = _forced_tokens(pkg, path, pcm, steps, pkg.PRECISION_BF16, **kw)
    agree = sum(a == b for a, b in zip(got16, want))
    print(f"{preset}: bf16 tokens equal to the fp32 oracle at {agree}/{len(want)} positions")
    assert len(got16) == len(want)


def test_independent_streams_one_context_each(pkg, model_file):
    # configs[3] in miniature: 6 streams, clip seeds 100+i, greedy, forced 24 steps; stream i -> rank i mod n.
    # Contexts are independent: running them interleaved on one GPU gives each stream the tokens it gets alone.
    path = model_file("micro", ftype=1)
    n_streams, steps = 6, 24
    clips = [ggml_io.synth_pcm(100 + i, 10.0) for i in range(n_streams)]
    alone = [_forced_tokens(pkg, path, clips[i], steps, pkg.PRECISION_FP32, language="en")[0] for i in range(n_streams)]
    engs = [pkg.Engine(path, precision=pkg.PRECISION_FP32) for _ in range(n_streams)]
    try:
        for e in engs:
            e.set_force_decode_steps(steps)
        out = [None] * n_streams
        for rnd in range(2):  # two rounds, interleaved across contexts
            for i in (range(n_streams) if rnd == 0 else reversed(range(n_streams))):
                p = engs[i].default_params(0, temperature_inc=0.0, duration_ms=10000, language="en")
                assert engs[i].full(p, clips[i]) == 0
                out[i] = engs[i].token_ids()
        assert out == alone
    finally:
        for e in engs:
            e.close()
    # the partition the bench uses over ranks covers every stream exactly once
    for world in (1, 2, 4, 8):
        seen = sorted(s for r in range(world) for s in stream_shard.assign_streams(n_streams, world, r))
        assert seen == list(range(n_streams))


def test_contexts_are_reentrant_across_threads(pkg, model_file):
    # SURVEY.md §8b threading: one context per OBS source, different sources run concurrently on their own worker
    # threads; results must not depend on what other contexts are doing.
    import threading
    path = model_file("micro.en", ftype=1)
    clips = [ggml_io.synth_pcm(200 + i, 8.0) for i in range(4)]
    alone = [_forced_tokens(pkg, path, c, 20, pkg.PRECISION_BF16, strategy=1, language="en")[0] for c in clips]
    out, errs = [None] * 4, []

    def work(i):
        try:
            for _ in range(3):
                out[i] = _forced_tokens(pkg, path, clips[i], 20, pkg.PRECISION_BF16, strategy=1, language="en")[0]
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    assert out == alone


def test_every_environment_switch_gives_the_same_results(pkg, model_file, monkeypatch):
    # The library reads its environment switches once, when a context is created; each selects other kernels / another
    # launch structure of the SAME arithmetic.  One context per switch, compared with a default context:
    #  * MI355_WHISPER_NO_FUSED_SAMPLE: rules + samplers as launches of their own behind the decode graph (the form the
    #    bench's launch-by-launch roofline leg also runs) -> identical tokens, beam search included;
    #  * MI355_WHISPER_NO_STEP_BEGIN: copy node + index-table and embedding kernels instead of the one kernel that reads the
    #    step block from pinned host memory -> identical tokens;
    #  * MI355_WHISPER_NO_GRAPH: the step as eager launches instead of a replayed hipGraph -> identical tokens;
    #  * MI355_WHISPER_GEMM_GLDS=1: encoder GEMM operands staged by global_load_lds instead of the register ring; both
    #    accumulate the same k-steps in the same order -> bit-identical encoder output;
    #  * MI355_WHISPER_NO_FLASH_ATTN: encoder attention as QK^T, softmax, PV launches with bf16 probabilities instead of the
    #    fused kernel -> encoder output within the bf16 tolerance of test_encoder_bf16 (relative error < 1 %).
    path = model_file("tiny.en", ftype=1, n_audio_layer=2, n_text_layer=2, n_audio_state=512, n_audio_head=8,
                      n_text_state=512, n_text_head=8)
    pcm = ggml_io.synth_pcm(11, 12.0)
    base_tok, base_seg = _forced_tokens(pkg, path, pcm, 24, pkg.PRECISION_BF16, strategy=1, language="en")
    assert len(base_tok) >= 20
    #  * MI355_WHISPER_NO_BATCH: the context runs its own steps and encoder pass instead of the hub's -> identical tokens;
    #  * MI355_WHISPER_LANES=1: one step lane -> identical tokens;
    #  * MI355_WHISPER_PRECISION=fp8 is a precision, not a kernel switch: covered by test_gpu_full_dims.py;
    #  * MI355_WHISPER_CROSS_PREFETCH (measurement only): a parallel branch in the step graph that reads each
    #    layer's cross K / V ahead of its cross-attention -> identical tokens.
    for switch in ("MI355_WHISPER_NO_FUSED_SAMPLE", "MI355_WHISPER_NO_STEP_BEGIN", "MI355_WHISPER_NO_GRAPH", "MI355_WHISPER_NO_BATCH",
                   "MI355_WHISPER_LANES", "MI355_WHISPER_CROSS_PREFETCH"):
        monkeypatch.setenv(switch, "1")
        tok, seg = _forced_tokens(pkg, path, pcm, 24, pkg.PRECISION_BF16, strategy=1, language="en")
        monkeypatch.delenv(switch)
        assert tok == base_tok and seg == base_seg, switch

    def encoder_out(**env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        eng = pkg.Engine(path, precision=pkg.PRECISION_BF16)
        for k in env:
            monkeypatch.delenv(k)
        try:
            eng.mel(pcm)
            return eng.encode(0).copy()
        finally:
            eng.close()

    ref = encoder_out()
    assert np.isfinite(ref).all()
    assert np.array_equal(encoder_out(MI355_WHISPER_GEMM_GLDS="1"), ref)
    unfused = encoder_out(MI355_WHISPER_NO_FLASH_ATTN="1")
    rel = float(np.abs(unfused - ref).max() / np.abs(ref).max())
    print(f"unfused vs fused encoder attention: max rel diff {rel:.4f}")
    assert rel < 0.01


def test_stream_contexts_share_one_weight_upload(pkg, model_file, monkeypatch):
    # configs[3]: many streams per GPU, weights resident once.  mw_init_shared (and whisper_init_from_file on a file that
    # is already resident) gives a context of its own on the same weights; the weights outlive the context that uploaded
    # them for as long as another context uses them.
    import threading
    path = model_file("micro", ftype=1)
    clips = [ggml_io.synth_pcm(300 + i, 6.0) for i in range(3)]
    monkeypatch.setenv("MI355_WHISPER_NO_SHARED_WEIGHTS", "1")     # reference run: every context uploads its own copy
    alone = [_forced_tokens(pkg, path, c, 16, pkg.PRECISION_BF16, language="en")[0] for c in clips]
    monkeypatch.delenv("MI355_WHISPER_NO_SHARED_WEIGHTS")
    first = pkg.Engine(path, precision=pkg.PRECISION_BF16)
    second = pkg.Engine(path, precision=pkg.PRECISION_BF16, share=first)   # explicit
    third = pkg.Engine(path, precision=pkg.PRECISION_BF16)                 # same file, same device: shared by the loader
    engs = [first, second, third]
    out, errs = [None] * 3, []

    def work(i, e):
        try:
            e.set_force_decode_steps(16)
            p = e.default_params(0, temperature_inc=0.0, duration_ms=6000, language="en")
            assert e.full(p, clips[i]) == 0
            out[i] = e.token_ids()
        except Exception as ex:  # noqa: BLE001
            errs.append(repr(ex))

    try:
        th = [threading.Thread(target=work, args=(i, engs[i])) for i in range(3)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        assert out == alone
        first.close()            # the uploader goes first; the views keep the weights alive
        work(1, second)
        work(2, third)
        assert not errs, errs
        assert out == alone
    finally:
        for e in engs:
            e.close()


@pytest.mark.parametrize("width,heads", [(384, 6), (512, 8), (768, 12), (1024, 16), (1280, 20)])
def test_tile_kernels_every_model_width_bf16(pkg, model_file, width, heads):
    # The beam-step tile kernels are specialised per width (k-steps per wave, chunks per row, vocabulary chunks of the
    # logits kernel); tiny/base/small/medium/large widths with 2 layers each, 5 rows at once (the beam-5 shape),
    # bf16 logits against the fp32 oracle on the same file, stated tolerance as test_decoder_bf16_within_tolerance.
    path = model_file("tiny.en", ftype=1, n_audio_state=width, n_audio_head=heads, n_audio_layer=2,
                      n_text_state=width, n_text_head=heads, n_text_layer=2)
    pcm = ggml_io.synth_pcm(3, 4.0)
    orc = pyoracle.Oracle(path)
    eng = pkg.Engine(path, precision=pkg.PRECISION_BF16)
    try:
        orc.mel(pcm)
        eng.mel(pcm)
        orc.encode(0)
        eng.encode(0)
        toks = [orc.sot, 11, 22, 33, 44]
        ref, got = orc.decode(toks, 0, 0), eng.decode(toks, 0, 0)   # 5 rows in one call
        assert got.shape == ref.shape
        # bf16 operands: the error grows with the logits' scale (which grows with the width); 0.6 % of their spread (measured: 0.30-0.37 %)
        tol = 0.006 * float(ref.max() - ref.min())
        err = float(np.abs(got - ref).max())
        print(f"width {width}: max |bf16 - fp32 oracle| = {err:.4f}, logits spread {ref.max() - ref.min():.2f}, tol {tol:.4f}")
        assert err < tol
        r1, g1 = orc.decode([55], 5, 0)[0], eng.decode([55], 5, 0)[0]   # single row (greedy shape)
        assert np.abs(g1 - r1).max() < tol
    finally:
        eng.close()
        orc.close()


def test_batched_steps_across_streams_equal_solo_runs(pkg, model_file):
    # Decode steps of contexts that share one weight upload are batched by the hub: rows of different streams ride in ONE
    # weight pass (up to 16 rows / 16 requests per step).  12 streams on a 512-wide model (tile kernels), beam search (5 rows
    # each, so a step holds up to 3 streams) and greedy, every stream with its own clip: tokens, segments and token
    # probabilities equal to the stream's solo run, and the hub's counters show that steps really were shared.
    import threading
    path = model_file("tiny.en", ftype=1, n_audio_layer=2, n_text_layer=2, n_audio_state=512, n_audio_head=8,
                      n_text_state=512, n_text_head=8)
    n_streams, steps, rounds = 12, 20, 2
    clips = [ggml_io.synth_pcm(400 + i, 6.0 + 0.5 * (i % 3)) for i in range(n_streams)]

    def make():
        first = pkg.Engine(path, precision=pkg.PRECISION_BF16)
        engs = [first] + [pkg.Engine(path, precision=pkg.PRECISION_BF16, share=first) for _ in range(n_streams - 1)]
        params = []
        for i, e in enumerate(engs):
            e.set_force_decode_steps(steps)
            strategy = pkg.SAMPLING_BEAM_SEARCH if i % 2 == 0 else pkg.SAMPLING_GREEDY
            params.append(e.default_params(strategy, temperature_inc=0.0, duration_ms=int(len(clips[i]) * 1000 // 16000),
                                           language="en"))
        return engs, params

    # solo reference: fresh contexts, one stream at a time (beam search draws from the context's persistent mt19937, so the
    # k-th call of a context is compared with the k-th call of a fresh context)
    engs, params = make()
    try:
        assert engs[0].hub_stats() is not None
        solo = []
        for i, e in enumerate(engs):
            rounds_i = []
            for _ in range(rounds):
                assert e.full(params[i], clips[i]) == 0
                rounds_i.append(e.segments())
            solo.append(rounds_i)
        s0 = engs[0].hub_stats()
        assert s0[1] >= s0[0] and s0[2] == s0[0]            # alone: one request per step
    finally:
        for e in engs[1:] + engs[:1]:
            e.close()
    engs, params = make()
    try:
        s0 = engs[0].hub_stats()
        out, errs = [[] for _ in range(n_streams)], []
        gate = threading.Barrier(n_streams)

        def work(i):
            try:
                gate.wait()
                for _ in range(rounds):
                    assert engs[i].full(params[i], clips[i]) == 0
                    out[i].append(engs[i].segments())
            except Exception as ex:  # noqa: BLE001
                errs.append(repr(ex))

        th = [threading.Thread(target=work, args=(i,)) for i in range(n_streams)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        for i in range(n_streams):
            assert out[i] == solo[i], f"stream {i}"
        s1 = engs[0].hub_stats()
        d_steps, d_rows, d_req = (s1[k] - s0[k] for k in range(3))
        print(f"hub: {d_req} requests in {d_steps} steps ({d_req / d_steps:.2f} requests, {d_rows / d_steps:.2f} rows per step)")
        assert d_req > d_steps                               # requests of different streams shared steps
        d_pass, d_win = s1[3] - s0[3], s1[4] - s0[4]
        print(f"hub: {d_win} encoder windows in {d_pass} passes")
        assert d_win >= n_streams * rounds and d_pass < d_win  # ... and windows of different streams shared encoder passes
    finally:
        for e in engs[1:] + engs[:1]:
            e.close()


def test_successive_calls_with_different_audio_on_hub_contexts_equal_solo_runs(pkg, model_file):
    # The hub's encoder lane reads a context's mel on ITS stream: it has to wait for the mel kernels of THIS call on the
    # context's own stream (engine.hip pcm_to_mel sets st_dirty; round 2 left the flag clear after a context's first call, so
    # a later call's encoder could read the previous call's mel).  10 streams alternate between two very different clips
    # each, 4 calls in a row, all at once; greedy + no_context, so every call depends on its own audio only and must equal
    # the solo transcript of that clip.
    import threading
    path = model_file("tiny.en", ftype=1, n_audio_layer=2, n_text_layer=2, n_audio_state=512, n_audio_head=8,
                      n_text_state=512, n_text_head=8)
    n_streams, steps, calls = 10, 12, 4
    clips = [[ggml_io.synth_pcm(700 + 2 * i, 29.0), 0.2 * ggml_io.synth_pcm(701 + 2 * i, 4.0 + 0.3 * i)] for i in range(n_streams)]
    first = pkg.Engine(path, precision=pkg.PRECISION_BF16)
    engs = [first] + [pkg.Engine(path, precision=pkg.PRECISION_BF16, share=first) for _ in range(n_streams - 1)]
    try:
        params = []
        for e in engs:
            e.set_force_decode_steps(steps)
        def par(e, clip):
            return e.default_params(pkg.SAMPLING_GREEDY, temperature_inc=0.0, duration_ms=int(len(clip) * 1000 // 16000), language="en")
        solo = []
        for i, e in enumerate(engs):
            row = []
            for k in range(2):
                assert e.full(par(e, clips[i][k]), clips[i][k]) == 0
                row.append(e.segments())
            assert row[0] != row[1]                       # the two clips of a stream do lead to different transcripts
            solo.append(row)
        out, errs = [[] for _ in range(n_streams)], []
        gate = threading.Barrier(n_streams)

        def work(i):
            try:
                gate.wait()
                for k in range(calls):
                    c = clips[i][k % 2]
                    assert engs[i].full(par(engs[i], c), c) == 0
                    out[i].append(engs[i].segments())
            except Exception as ex:  # noqa: BLE001
                errs.append(repr(ex))

        th = [threading.Thread(target=work, args=(i,)) for i in range(n_streams)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        for i in range(n_streams):
            assert out[i] == [solo[i][k % 2] for k in range(calls)], f"stream {i}"
    finally:
        for e in engs[1:] + engs[:1]:
            e.close()


def test_mel_is_bit_exact_while_other_contexts_encode(pkg, model_file):
    # Regression: built with hipcc's SLP vectoriser (packed-f32 VALU), mel_frames_kernel returned a few wrong values per call
    # whenever kernels of OTHER contexts shared its compute units (alone it was bit-exact) — 150 of 240 calls differed.
    # The library is built with -fno-slp-vectorize (csrc/Makefile); every call must equal the stream's solo result.
    import threading
    path = model_file("tiny.en", ftype=1, n_audio_layer=2, n_text_layer=2, n_audio_state=512, n_audio_head=8,
                      n_text_state=512, n_text_head=8)
    n = 8
    clips = [ggml_io.synth_pcm(400 + i, 6.0 + 0.5 * (i % 3)) for i in range(n)]
    first = pkg.Engine(path, precision=pkg.PRECISION_BF16)
    engs = [first] + [pkg.Engine(path, precision=pkg.PRECISION_BF16, share=first) for _ in range(n - 1)]
    try:
        for i in range(n):
            engs[i].mel(clips[i])
            engs[i].encode(0)
        solo = [engs[i].mel(clips[i]).copy() for i in range(4)]
        stop, bad = [False], [0]

        def noisy(i):
            while not stop[0]:
                engs[i].encode(0)

        def checker(i):
            for _ in range(25):
                if not np.array_equal(engs[i].mel(clips[i]), solo[i]):
                    bad[0] += 1

        tn = [threading.Thread(target=noisy, args=(i,)) for i in range(4, n)]
        tc = [threading.Thread(target=checker, args=(i,)) for i in range(4)]
        for t in tn + tc:
            t.start()
        for t in tc:
            t.join()
        stop[0] = True
        for t in tn:
            t.join()
        assert bad[0] == 0, f"{bad[0]} of 100 mel calls differ from the solo result"
    finally:
        for e in engs[1:] + engs[:1]:
            e.close()
