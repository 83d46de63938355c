"""CPU: the MX-fp8 rule (blocks of 32 along k, power-of-two scale bringing the block under 448, e4m3 elements, round to
nearest even) as the oracle states it (oracle/whisper_oracle.cpp: mx_qdq_rows) against known answers and against an
independent numpy statement (tools/ggml_io.mxfp8_quantize_dequantize).  The GPU tests hold the engine's MX-fp8 kernels to
this rule bit for bit (tests/test_gpu_mx_encoder.py)."""
import numpy as np

from oracle import pyoracle
from tools import ggml_io


def _block(vals, fill=0.0):
    b = np.full((1, 32), fill, np.float32)
    b[0, :len(vals)] = vals
    return b


def test_known_answers():
    q = pyoracle.mx_qdq
    # absmax 448 -> scale 1: integers up to 16 are exact, ties go to the even mantissa, steps double per binade
    got = q(_block([448, 17, 19, 21, 23, 33, 100, 0.001, 2.0 ** -9, 2.0 ** -10, 3 * 2.0 ** -10, -17]))[0]
    assert list(got[:12]) == [448, 16, 20, 20, 24, 32, 96, 2.0 ** -9, 2.0 ** -9, 0.0, 2.0 ** -8, -16]
    # absmax just above 448 -> scale 2: 224.5 -> 224; 1.5 and 2.5 are on the grid; 2^-9 / 2 is half the smallest subnormal -> 0
    got = q(_block([449, 3, 5, 2.0 ** -9, 3 * 2.0 ** -9]))[0]
    assert list(got[:5]) == [448, 3, 5, 0.0, 2.0 ** -7]
    # exactly a power of two times 448 keeps the smaller scale
    got = q(_block([896, 2, 6]))[0]
    assert list(got[:3]) == [896, 2, 6]
    # a block of zeros stays zeros; tiny blocks keep their relative precision
    assert not q(np.zeros((1, 32), np.float32)).any()
    tiny = q(_block([1.0e-30, 0.5e-30]))[0]
    assert abs(tiny[0] - 1.0e-30) <= 1.0e-30 / 16 and abs(tiny[1] - 0.5e-30) <= 0.5e-30 / 16
    # blocks are independent
    two = np.concatenate([_block([448, 17]), _block([7, 17 / 64])], axis=1)   # second block: scale 2^-6
    got = q(two)[0]
    assert got[1] == 16 and got[32] == 7 and got[33] == 0.25


def test_every_e4m3_value_times_a_power_of_two_is_a_fixed_point_and_results_stay_on_the_grid():
    vals = []
    for e in range(16):
        for m in range(8):
            if e == 15 and m == 7:
                continue
            vals.append(m * 2.0 ** -9 if e == 0 else (1 + m / 8.0) * 2.0 ** (e - 7))
    grid = np.array(sorted(set(vals)), np.float32)
    assert len(grid) == 127 and grid[-1] == 448
    for k in (-20, -3, 0, 5):
        for i in range(0, len(grid) - 1, 31):
            blk = _block(list(grid[i:i + 31] * 2.0 ** k) + [448 * 2.0 ** k])
            assert np.array_equal(pyoracle.mx_qdq(blk), blk)
    rng = np.random.default_rng(0)
    x = (rng.standard_normal((64, 256)) * np.exp(2 * rng.standard_normal((64, 256)))).astype(np.float32)
    y = pyoracle.mx_qdq(x)
    assert np.array_equal(pyoracle.mx_qdq(y), y)                       # idempotent
    rel = np.abs(y - x) / np.maximum(np.abs(x), 1e-30)
    amax = np.abs(x).reshape(64, 8, 32).max(axis=2, keepdims=True).repeat(32, 2).reshape(64, 256)
    assert ((rel <= 1 / 16 + 1e-6) | (np.abs(y - x) <= amax / 224 * 2.0 ** -10 * 1.001)).all()   # half a step, normal or subnormal
    assert np.array_equal(y.astype(np.float16).astype(np.float32), y)  # representable in the GGML file's f16


def test_oracle_rule_equals_the_numpy_statement():
    rng = np.random.default_rng(1)
    x = (rng.standard_normal((200, 1280)) * np.exp(1.5 * rng.standard_normal((200, 40, 1))).repeat(32, 2).reshape(200, 1280)).astype(np.float32)
    x[0, :32] = 0
    x[1, :32] = np.linspace(-448, 448, 32)
    assert np.array_equal(pyoracle.mx_qdq(x), ggml_io.mxfp8_quantize_dequantize(x))


def test_the_twin_is_off_by_default_and_changes_the_encoder_when_on(tmp_path):
    path = str(tmp_path / "m.bin")
    ggml_io.make_model_file(path, "micro.en", seed=3, n_audio_state=256, n_audio_head=4, n_text_state=256, n_text_head=4)
    orc = pyoracle.Oracle(path)
    try:
        orc.mel(ggml_io.synth_pcm(7, 3.0))
        a = orc.encode(0)
        orc.set_mx_encoder(True)
        b = orc.encode(0)
        c = orc.encode(0)
        orc.set_mx_encoder(False)
        d = orc.encode(0)
        assert np.array_equal(a, d) and np.array_equal(b, c) and not np.array_equal(a, b)
        assert np.abs(b - a).max() < 0.2 * np.abs(a).max()
    finally:
        orc.close()
