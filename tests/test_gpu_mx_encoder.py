"""GPU: MW_PRECISION_FP8_MX — the encoder's Q|K|V, attention-output, fc1 and fc2 products (and the cross K / V projections) in MX-fp8 on the fp8 matrix cores (gemm_mx.hip).

The arithmetic the mode declares is restated in the oracle (oracle/whisper_oracle.cpp: mx_qdq_rows, ow_set_mx_encoder):
per block of 32 along k the smallest power-of-two scale that brings the block under 448, e4m3 elements rounded to nearest
even; products and everything else in fp32.  Parity is therefore checked at three levels:
  * the quantisers, bit for bit, against that rule (activations, weights, the LayerNorm producer, the GELU producer);
  * the product, against a float64 product of the DECODED operands the library hands back: what differs is the matrix
    core's own summation (the products of one 128-deep instruction are aligned to a common exponent before they are
    added: measured up to 4e-5 of sum |a w| on blocks of very different scales; fp32 between instructions) — tolerance
    PROD_TOL of sum |a w|, for every tile height the launcher chooses and a ragged last tile;
  * the whole encoder (and teacher-forced decoder steps on its cross K / V) at large-v3 dims against the fp32 oracle AND
    the oracle's MX twin.  An e4m3 element changes by 6-12 % when its input crosses a rounding boundary, and the engine's
    inputs differ from the twin's by the bf16 rounding of everything that stays bf16 (conv stem, attention, attention output): a few per
    cent of the elements of every block round the other way, so after 32 blocks engine and twin are two realisations of
    the same quantisation noise rather than equal numbers (measured: engine - twin 0.055, engine - fp32 0.047, twin - fp32
    0.047 of the largest activation).  What the test asserts is therefore: the engine is no further from the fp32 oracle
    than the declared arithmetic itself is (a wrong scale or operand map anywhere would show here), and no further from
    the twin than two independent realisations can be.
No reference-held vectors exist for this mode (the reference has no fp8 path): parity unpinned, as for the rest of the model
arithmetic.
"""
import threading

import numpy as np
import pytest

from oracle import pyoracle
from tools import ggml_io, synth_large

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]

ENC_TOL = 0.02       # the bf16 path's own bound (V, attention, attention output stay bf16)
LOGIT_TOL_MX = 0.02  # of the row's logit spread, decoder on cross K / V of the MX encoder (two noise realisations; measured 0.004)
PROD_TOL = 2e-4      # of sum |a w| per output element (see above)
MX_COST_TOL = 0.15   # engine (MX-fp8) vs the plain fp32 oracle, max |d| / max |oracle| after 32 blocks (measured 0.09)


def _gelu64(x):
    return 0.5 * x * (1.0 + np.tanh(0.79788456080286535588 * x * (1.0 + 0.044715 * x * x)))


def _operands(rng, M, K, N):
    A = rng.standard_normal((M, K)).astype(np.float32) * np.exp(rng.standard_normal((M, K // 32, 1)) * 1.5).repeat(32, 2).reshape(M, K).astype(np.float32)
    A[0, :32] = 0.0                          # an all-zero block
    A[1, 0:32] = np.linspace(-448.0, 448.0, 32)  # absmax exactly 448: scale 1
    A[2, 0:32] = 449.0                       # just above: scale 2
    A[3, 0] = 1.0
    A[3, 1:32] = 2.0 ** -10                  # elements in the subnormal range of the block's scale
    A[4, 32:64] = 1.0e-30                    # a very small block
    W = (rng.standard_normal((N, K)) * 0.04).astype(np.float32)
    return A, W


@pytest.mark.parametrize("M,K,N", [(300, 256, 256), (1500, 1280, 2560), (1500, 5120, 1280), (3000, 1280, 2560), (6200, 1280, 2560)])
def test_mx_quantisers_bit_exact_and_product_against_decoded_operands(pkg, M, K, N):
    rng = np.random.default_rng(M + K + N)
    A, W = _operands(rng, M, K, N)
    bias = (rng.standard_normal(N) * 0.3).astype(np.float32)
    R = rng.standard_normal((M, N)).astype(np.float32)
    C, Ad, Wd = pkg.mx_gemm(A, W, bias=bias, R=R, epilogue=pkg.MXE_F32_RES)
    # quantisers: the rule, bit for bit (weights are quantised from their bf16 rounding, as the loader does)
    assert np.array_equal(Ad, pyoracle.mx_qdq(A))
    Wb = (np.ascontiguousarray(W).view(np.uint32) + 0x7FFF + ((W.view(np.uint32) >> 16) & 1)) & 0xFFFF0000
    assert np.array_equal(Wd, pyoracle.mx_qdq(Wb.view(np.float32)))
    # product: float64 on the decoded operands
    ref = Ad.astype(np.float64) @ Wd.astype(np.float64).T
    mag = np.abs(Ad).astype(np.float64) @ np.abs(Wd).astype(np.float64).T
    full = ref + bias + R
    err = np.abs(C - full)
    assert np.isfinite(C).all()
    assert (err <= PROD_TOL * mag + 1e-6).all(), f"fp32 rows: worst err / sum |a w| = {(err / (mag + 1e-30)).max():.2e}"
    # bf16 rows
    Cb, _, _ = pkg.mx_gemm(A, W, bias=bias, epilogue=pkg.MXE_BF16)
    errb = np.abs(Cb - (ref + bias))
    assert (errb <= np.abs(ref + bias) / 256.0 * 1.001 + PROD_TOL * mag + 1e-6).all()
    # GELU rows re-quantised to MX-fp8 for the next product
    Cg, _, _ = pkg.mx_gemm(A, W, bias=bias, epilogue=pkg.MXE_MX_GELU)
    y = _gelu64(ref + bias).astype(np.float32)
    want = pyoracle.mx_qdq(y)
    same = float(np.mean(Cg == want))
    amax = np.abs(y).reshape(M, N // 32, 32).max(axis=2, keepdims=True).repeat(32, 2).reshape(M, N)
    # the pre-activation differs from float64 by the matrix core's summation (dz <= PROD_TOL sum |a w| / 4 as measured,
    # |GELU'| <= 1.13); where that moves a value across a rounding boundary it lands on the neighbouring e4m3 value (one
    # step: at most |y| / 8, or one subnormal step of the block's scale)
    dz = 1.13 * (PROD_TOL / 4) * mag
    assert (np.abs(Cg - want) <= (np.abs(y) + dz) / 8.0 + dz + amax / 224.0 * 2.0 ** -9 + 1e-30).all()
    assert same > 0.97, same
    print(f"M={M} K={K} N={N}: fp32 rows worst err / sum|a w| = {(err / (mag + 1e-30)).max():.2e}; GELU rows equal to the rule on {100 * same:.2f} % of the elements")


@pytest.mark.parametrize("M,K,N", [(6200, 1280, 2560), (2100, 128, 1024), (1500, 5120, 1280)])
def test_every_tile_height_gives_the_same_bits(pkg, monkeypatch, M, K, N):
    """The launcher picks 256-, 128- or 64-row tiles by an estimated launch time; an output element is the same chain of
    MFMAs whatever the tile.  Forced heights (MI355_WHISPER_MX_TILE_ROWS) on shapes with several tiles per workgroup on
    the small-tile kernel (its loads run two k-steps ahead, across tile ends), a single k-step, and a ragged last tile."""
    rng = np.random.default_rng(M * 7 + K + N)
    A, W = _operands(rng, M, K, N)
    bias = (rng.standard_normal(N) * 0.3).astype(np.float32)
    R = rng.standard_normal((M, N)).astype(np.float32)
    out = {}
    for rows in ("256", "128", "64"):
        monkeypatch.setenv("MI355_WHISPER_MX_TILE_ROWS", rows)
        out[rows] = (pkg.mx_gemm(A, W, bias=bias, R=R, epilogue=pkg.MXE_F32_RES)[0], pkg.mx_gemm(A, W, bias=bias, epilogue=pkg.MXE_BF16)[0],
                     pkg.mx_gemm(A, W, bias=bias, epilogue=pkg.MXE_MX_GELU)[0])
    monkeypatch.delenv("MI355_WHISPER_MX_TILE_ROWS")
    chosen = pkg.mx_gemm(A, W, bias=bias, R=R, epilogue=pkg.MXE_F32_RES)[0]
    for rows in ("128", "64"):
        for a, b in zip(out["256"], out[rows]):
            assert np.array_equal(a, b), rows
    assert np.array_equal(chosen, out["256"][0])


def test_layernorm_mx_rows_follow_the_rule(pkg):
    rng = np.random.default_rng(5)
    rows, d = 203, 1280
    x = rng.standard_normal((rows, d)).astype(np.float32) * 3.0
    x[:, 77] *= 40.0                 # an outlier channel, as real activations have
    w = (1.0 + 0.05 * rng.standard_normal(d)).astype(np.float32)
    b = (0.02 * rng.standard_normal(d)).astype(np.float32)
    dec, hb = pkg.mx_layernorm(x, w, b, with_bf16=True)
    x64 = x.astype(np.float64)
    mu = x64.mean(axis=1, keepdims=True)
    var = ((x64 - mu) ** 2).mean(axis=1, keepdims=True)
    y = ((x64 - mu) / np.sqrt(var + 1e-5) * w + b).astype(np.float32)
    assert np.abs(hb - y).max() <= np.abs(y).max() / 256.0 * 1.01     # the bf16 copy V reads
    want = pyoracle.mx_qdq(y)
    same = float(np.mean(dec == want))
    amax = np.abs(y).reshape(rows, d // 32, 32).max(axis=2, keepdims=True).repeat(32, 2).reshape(rows, d)
    assert (np.abs(dec - want) <= np.abs(y) / 8.0 + amax / 224.0 * 2.0 ** -9 + 1e-30).all()
    assert same > 0.999, same   # fp32 statistics on the device vs float64 here: a handful of values sit on a rounding boundary
    dec2, none = pkg.mx_layernorm(x, w, b, with_bf16=False)
    assert none is None and np.array_equal(dec, dec2)


@pytest.fixture(scope="module")
def mx_model(tmp_path_factory):
    d = tmp_path_factory.mktemp("mx_dims")
    path = str(d / "ggml-large-v3-turbo-synth-s1234-mx.bin")
    # decoder tensors on the e4m3 row grid, encoder Q / K / fc1 / fc2 on the MX grid: the oracle (fp32 reader) and the
    # engine (which quantises at load) hold exactly the same weights
    synth_large.write_large(path, "large-v3-turbo", seed=1234, quiet_special=True, fp8_decoder=True, mx_encoder=True)
    return path


def test_encoder_and_teacher_forced_steps_against_the_oracles_mx_twin(pkg, mx_model):
    pyoracle.lib().ow_set_threads(16)
    pcm = ggml_io.synth_pcm(100, 30.0)
    orc = pyoracle.Oracle(mx_model)
    eng = pkg.Engine(mx_model, precision=pkg.PRECISION_FP8_MX)
    try:
        orc.mel(pcm)
        eng.mel(pcm)
        plain = orc.encode(0)                     # fp32 oracle on the same weights
        orc.set_mx_encoder(True)
        twin = orc.encode(0)                      # the declared arithmetic; its cross K / V feed the decode steps below
        got = eng.encode(0)
        assert np.isfinite(got).all()
        rel_twin = float(np.abs(got - twin).max() / np.abs(twin).max())
        rel_plain = float(np.abs(got - plain).max() / np.abs(plain).max())
        cost = float(np.abs(twin - plain).max() / np.abs(plain).max())
        print(f"encoder, 32 blocks, MX-fp8 Q|K|V / fc1 / fc2: engine vs the oracle's MX twin {rel_twin:.4f}; engine vs fp32 oracle {rel_plain:.4f}; "
              f"twin vs fp32 oracle (the format's own cost) {cost:.4f}   (max |d| / max |ref|)")
        assert rel_plain < MX_COST_TOL
        assert rel_plain <= 1.25 * cost + ENC_TOL / 4      # as close to the truth as the declared arithmetic
        assert rel_twin <= 1.5 * cost + ENC_TOL / 4        # sqrt(2) x the noise of one realisation, with margin
        # decoder on the MX encoder's cross K / V, teacher-forced with the twin's tokens: 5 rows per step
        prompt = [orc.sot, orc.sot + 1, orc.transcribe]
        worst = 0.0
        hist = []
        for r in range(5):
            ref = orc.decode(prompt, 0, r)
            out = eng.decode(prompt, 0, r)
            for i in range(len(prompt)):
                spread = float(ref[i].max() - ref[i].min())
                worst = max(worst, float(np.abs(out[i] - ref[i]).max()) / spread)
            hist.append([int(np.argsort(-ref[-1][:orc.eot])[r])])
        n_past = len(prompt)
        for s in range(6):
            toks = [h[-1] for h in hist]
            out = eng.decode_rows(toks, [n_past] * 5, list(range(5)))
            for r in range(5):
                ref = orc.decode([toks[r]], n_past, r)[0]
                spread = float(ref.max() - ref.min())
                worst = max(worst, float(np.abs(out[r] - ref).max()) / spread)
                hist[r].append(int(np.argmax(ref[:orc.eot])))
            n_past += 1
        print(f"teacher-forced logits on the MX encoder's cross K / V: worst |d logits| / spread = {worst:.4f}")
        assert worst < LOGIT_TOL_MX
    finally:
        eng.close()
        orc.close()


def test_a_window_encodes_to_the_same_bits_alone_and_inside_a_batched_pass(pkg, mx_model):
    """The launcher picks 64-, 128- or 256-row tiles by M; an element's chain of MFMAs over k is the same in all of them."""
    n = 6
    clips = [ggml_io.synth_pcm(300 + i, 30.0) for i in range(n)]
    first = pkg.Engine(mx_model, precision=pkg.PRECISION_FP8_MX)
    engs = [first] + [pkg.Engine(mx_model, precision=pkg.PRECISION_FP8_MX, share=first) for _ in range(n - 1)]
    try:
        solo = []
        for e, c in zip(engs, clips):
            e.mel(c)
            solo.append(e.encode(0))
        assert not np.array_equal(solo[0], solo[1])
        s0 = first.hub_stats()
        out, errs = [None] * n, []
        gate = threading.Barrier(n)

        def work(i):
            try:
                gate.wait()
                out[i] = engs[i].encode(0)
            except Exception as ex:  # noqa: BLE001
                errs.append(repr(ex))

        th = [threading.Thread(target=work, args=(i,)) for i in range(n)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        s1 = first.hub_stats()
        assert s1[4] - s0[4] == n and s1[3] - s0[3] < n      # the windows did share passes
        for i in range(n):
            assert np.array_equal(out[i], solo[i]), i
    finally:
        for e in engs[1:] + engs[:1]:
            e.close()


def test_base_dims_shorter_audio_context_and_a_model_the_mode_does_not_fit(pkg, tmp_path):
    """d = 512 (base.en: two 256-column tiles, 4 k-steps, 80 mel bins), the full and a shorter audio context (n_ctx = 500: the
    last 16-row slice, the 64-column pad of V^T and the scale records of a ragged 128-row block), against the fp32 oracle
    and its MX twin; tiny.en (d = 384, not a multiple of 256) is refused at load with a message, not computed some other way."""
    path = str(tmp_path / "base-mx.bin")
    synth_large.write_large(path, "base.en", seed=77, quiet_special=True, fp8_decoder=True, mx_encoder=True)
    pcm = ggml_io.synth_pcm(5, 30.0)
    orc = pyoracle.Oracle(path)
    eng = pkg.Engine(path, precision=pkg.PRECISION_FP8_MX)
    try:
        orc.mel(pcm)
        eng.mel(pcm)
        for audio_ctx in (0, 500):
            orc.set_mx_encoder(False)
            plain = orc.encode(0, audio_ctx)
            k_plain = orc.cross_kv(0, False, audio_ctx or None)
            orc.set_mx_encoder(True)
            twin = orc.encode(0, audio_ctx)
            got = eng.encode(0, audio_ctx)
            assert got.shape == twin.shape and np.isfinite(got).all()
            cost = float(np.abs(twin - plain).max() / np.abs(plain).max())
            rel_plain = float(np.abs(got - plain).max() / np.abs(plain).max())
            rel_twin = float(np.abs(got - twin).max() / np.abs(twin).max())
            print(f"base.en dims, audio_ctx {audio_ctx or 1500}: engine vs twin {rel_twin:.4f}, engine vs fp32 oracle {rel_plain:.4f}, twin vs fp32 oracle {cost:.4f}")
            assert rel_plain <= 1.25 * cost + ENC_TOL / 4 and rel_twin <= 1.5 * cost + ENC_TOL / 4
            k_eng, k_orc = eng.cross_kv(0, False, audio_ctx or None), orc.cross_kv(0, False, audio_ctx or None)
            assert k_eng.shape == k_orc.shape
            cost_k = float(np.abs(k_orc - k_plain).max() / np.abs(k_plain).max())   # the cross K product quantises once more
            assert float(np.abs(k_eng - k_orc).max() / np.abs(k_orc).max()) <= 1.5 * cost_k + ENC_TOL
            assert float(np.abs(k_eng - k_plain).max() / np.abs(k_plain).max()) <= 1.25 * cost_k + ENC_TOL
    finally:
        eng.close()
        orc.close()
    tiny = str(tmp_path / "tiny-mx.bin")
    synth_large.write_large(tiny, "tiny.en", seed=78, quiet_special=True)
    with pytest.raises(Exception) as ei:
        pkg.Engine(tiny, precision=pkg.PRECISION_FP8_MX)
    assert "256" in str(ei.value) or "MX" in str(ei.value), str(ei.value)
    ok = pkg.Engine(tiny, precision=pkg.PRECISION_FP8)     # the e4m3-decoder mode has no such limit
    ok.close()
