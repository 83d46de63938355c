"""GPU: the benchmarked configurations at their OWN dimensions against the oracle, teacher-forced.

  BASELINE configs[2]  Whisper large-v3 dims (d=1280, 32+32 layers, 128 mel bins, vocab 51866), bf16, 5 rows per step
  BASELINE configs[3]  large-v3-turbo dims (4 decoder layers), bf16, 64 concurrent contexts on one upload of the weights

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
