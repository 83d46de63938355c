"""GGML-legacy (`ggml-*.bin`) Whisper model files: writer, reader and seeded synthetic models.

Test / bench tooling only (never on the device path).  The container has no real Whisper weights
(SURVEY.md §8c), so parity and throughput runs use seeded synthetic weights written in the exact
container format the engine's loader parses (magic 0x67676d6c, 11 int32
hparams, mel filters, vocab, then tensors until EOF).

The file layout restates what whisper.cpp v1.6.2's loader reads; the reference fetches that library
at build time (`/root/reference/cmake/BuildWhispercpp.cmake:118`) and loads such files through
`whisper_init_from_file_with_params` (`/root/reference/src/whisper-utils/whisper-processing.cpp:112`).
"""
from __future__ import annotations

import struct
from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np

GGML_MAGIC = 0x67676D6C
GGML_TYPE_F32 = 0
GGML_TYPE_F16 = 1


@dataclass
class HParams:
    n_vocab: int = 51864
    n_audio_ctx: int = 1500
    n_audio_state: int = 384
    n_audio_head: int = 6
    n_audio_layer: int = 4
    n_text_ctx: int = 448
    n_text_state: int = 384
    n_text_head: int = 6
    n_text_layer: int = 4
    n_mels: int = 80
    ftype: int = 1  # 0 = all f32, 1 = 2-D weights f16

    def as_list(self) -> List[int]:
        return [self.n_vocab, self.n_audio_ctx, self.n_audio_state, self.n_audio_head,
                self.n_audio_layer, self.n_text_ctx, self.n_text_state, self.n_text_head,
                self.n_text_layer, self.n_mels, self.ftype]


# Named dimension presets (standard Whisper sizes) + small test-only shapes.
PRESETS: Dict[str, HParams] = {
    "tiny.en": HParams(51864, 1500, 384, 6, 4, 448, 384, 6, 4, 80, 1),
    "base.en": HParams(51864, 1500, 512, 8, 6, 448, 512, 8, 6, 80, 1),
    "tiny": HParams(51865, 1500, 384, 6, 4, 448, 384, 6, 4, 80, 1),
    "base": HParams(51865, 1500, 512, 8, 6, 448, 512, 8, 6, 80, 1),
    "large-v3": HParams(51866, 1500, 1280, 20, 32, 448, 1280, 20, 32, 128, 1),
    "large-v3-turbo": HParams(51866, 1500, 1280, 20, 32, 448, 1280, 20, 4, 128, 1),
    # test-only: smallest shapes that still exercise every code path (dh = 64 like real models)
    "micro.en": HParams(51864, 1500, 128, 2, 2, 448, 128, 2, 3, 80, 1),
    "micro": HParams(51866, 1500, 128, 2, 2, 448, 128, 2, 3, 128, 1),
}


# --------------------------------------------------------------------------------------------
# mel filterbank (Slaney scale, Slaney area normalisation) — the table OpenAI ships as mel_filters.npz
# --------------------------------------------------------------------------------------------
def _hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)


def mel_filterbank(n_mels: int, n_fft: int = 400, sr: int = 16000) -> np.ndarray:
    """[n_mels, 1 + n_fft/2] float32, librosa.filters.mel(sr, n_fft, n_mels) semantics."""
    n_freqs = 1 + n_fft // 2
    fftfreqs = np.linspace(0, sr / 2, n_freqs)
    mel_pts = np.linspace(_hz_to_mel(0.0), _hz_to_mel(sr / 2.0), n_mels + 2)
    hz_pts = _mel_to_hz(mel_pts)
    fdiff = np.diff(hz_pts)
    ramps = hz_pts[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, n_freqs), dtype=np.float64)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (hz_pts[2:n_mels + 2] - hz_pts[:n_mels])
    w *= enorm[:, None]
    return w.astype(np.float32)


# --------------------------------------------------------------------------------------------
# vocabulary
# --------------------------------------------------------------------------------------------
def _gpt2_byte_order() -> List[int]:
    """Byte value of GPT-2 token ids 0..255 (printable bytes first), so id 13 = '.', id 220 = ' '."""
    bs = list(range(33, 127)) + list(range(161, 173)) + list(range(174, 256))
    rest = [b for b in range(256) if b not in bs]
    return bs + rest


def synthetic_vocab(n_file: int, seed: int = 4321) -> List[bytes]:
    """n_file unique byte-level pieces: ids 0..255 single bytes in GPT-2 order, the rest random
    lowercase word pieces (about half with a leading space), deterministic in `seed`."""
    order = _gpt2_byte_order()
    toks: List[bytes] = [bytes([b]) for b in order]
    seen = set(toks)
    rng = np.random.RandomState(seed)
    letters = b"abcdefghijklmnopqrstuvwxyz"
    while len(toks) < n_file:
        n = int(rng.randint(2, 7))
        w = bytes(letters[int(i)] for i in rng.randint(0, 26, size=n))
        if rng.randint(0, 2):
            w = b" " + w
        if w in seen:
            continue
        seen.add(w)
        toks.append(w)
    return toks[:n_file]


def n_vocab_in_file(n_vocab: int) -> int:
    # English-only files carry ids < eot (50256); multilingual ones 50257 (ids <= eot)
    return 50257 if n_vocab >= 51865 else 50256


# --------------------------------------------------------------------------------------------
# tensor naming / shapes
# --------------------------------------------------------------------------------------------
def tensor_shapes(hp: HParams) -> List[Tuple[str, Tuple[int, ...], bool]]:
    """(name, numpy shape [row-major, last dim fastest], is_matrix_weight) in file order."""
    da, dt = hp.n_audio_state, hp.n_text_state
    out: List[Tuple[str, Tuple[int, ...], bool]] = []
    out.append(("encoder.positional_embedding", (hp.n_audio_ctx, da), False))
    out.append(("encoder.conv1.weight", (da, hp.n_mels, 3), True))
    out.append(("encoder.conv1.bias", (da, 1), False))
    out.append(("encoder.conv2.weight", (da, da, 3), True))
    out.append(("encoder.conv2.bias", (da, 1), False))
    for i in range(hp.n_audio_layer):
        p = f"encoder.blocks.{i}."
        out += [(p + "attn_ln.weight", (da,), False), (p + "attn_ln.bias", (da,), False),
                (p + "attn.query.weight", (da, da), True), (p + "attn.query.bias", (da,), False),
                (p + "attn.key.weight", (da, da), True),
                (p + "attn.value.weight", (da, da), True), (p + "attn.value.bias", (da,), False),
                (p + "attn.out.weight", (da, da), True), (p + "attn.out.bias", (da,), False),
                (p + "mlp_ln.weight", (da,), False), (p + "mlp_ln.bias", (da,), False),
                (p + "mlp.0.weight", (4 * da, da), True), (p + "mlp.0.bias", (4 * da,), False),
                (p + "mlp.2.weight", (da, 4 * da), True), (p + "mlp.2.bias", (da,), False)]
    out += [("encoder.ln_post.weight", (da,), False), ("encoder.ln_post.bias", (da,), False)]
    out.append(("decoder.positional_embedding", (hp.n_text_ctx, dt), False))
    out.append(("decoder.token_embedding.weight", (hp.n_vocab, dt), True))
    for i in range(hp.n_text_layer):
        p = f"decoder.blocks.{i}."
        out += [(p + "attn_ln.weight", (dt,), False), (p + "attn_ln.bias", (dt,), False),
                (p + "attn.query.weight", (dt, dt), True), (p + "attn.query.bias", (dt,), False),
                (p + "attn.key.weight", (dt, dt), True),
                (p + "attn.value.weight", (dt, dt), True), (p + "attn.value.bias", (dt,), False),
                (p + "attn.out.weight", (dt, dt), True), (p + "attn.out.bias", (dt,), False),
                (p + "cross_attn_ln.weight", (dt,), False), (p + "cross_attn_ln.bias", (dt,), False),
                (p + "cross_attn.query.weight", (dt, dt), True), (p + "cross_attn.query.bias", (dt,), False),
                (p + "cross_attn.key.weight", (dt, dt), True),
                (p + "cross_attn.value.weight", (dt, dt), True), (p + "cross_attn.value.bias", (dt,), False),
                (p + "cross_attn.out.weight", (dt, dt), True), (p + "cross_attn.out.bias", (dt,), False),
                (p + "mlp_ln.weight", (dt,), False), (p + "mlp_ln.bias", (dt,), False),
                (p + "mlp.0.weight", (4 * dt, dt), True), (p + "mlp.0.bias", (4 * dt,), False),
                (p + "mlp.2.weight", (dt, 4 * dt), True), (p + "mlp.2.bias", (dt,), False)]
    out += [("decoder.ln.weight", (dt,), False), ("decoder.ln.bias", (dt,), False)]
    return out


def synthetic_tensors(hp: HParams, seed: int = 1234, sigma: float = 0.02,
                      quiet_special: bool = False) -> Dict[str, np.ndarray]:
    """Seeded N(0, sigma) matrices / embeddings, LayerNorm weight 1 (+small noise) bias small.

    `quiet_special`: scale the token-embedding rows of the special / timestamp tokens (id >= eot)
    so their tied-output logits stay near zero.  Used by the throughput bench only, where it keeps
    the timestamp-mass rule from firing on every step with random weights.
    """
    rng = np.random.RandomState(seed)
    t: Dict[str, np.ndarray] = {}
    for name, shape, is_mat in tensor_shapes(hp):
        if name.endswith("ln.weight") or name.endswith("ln_post.weight"):
            a = 1.0 + 0.05 * rng.standard_normal(shape)
        elif name.endswith(".bias"):
            a = 0.02 * rng.standard_normal(shape)
        elif name.endswith("positional_embedding"):
            a = 0.05 * rng.standard_normal(shape)
        elif name == "decoder.token_embedding.weight":
            a = 0.05 * rng.standard_normal(shape)
        else:
            fan_in = int(np.prod(shape[1:]))
            a = rng.standard_normal(shape) * max(sigma, 1.0 / np.sqrt(fan_in))
        t[name] = a.astype(np.float32)
    if quiet_special:
        t["decoder.token_embedding.weight"][50256:] *= 0.01
    return t


# --------------------------------------------------------------------------------------------
# GGML block quantisation (32 elements per block), as the catalogue's q4/q5/q8 model files use it
# --------------------------------------------------------------------------------------------
GGML_TYPE_Q4_0, GGML_TYPE_Q4_1, GGML_TYPE_Q5_0, GGML_TYPE_Q5_1, GGML_TYPE_Q8_0 = 2, 3, 6, 7, 8
QUANT_BLOCK_BYTES = {GGML_TYPE_Q4_0: 18, GGML_TYPE_Q4_1: 20, GGML_TYPE_Q5_0: 22, GGML_TYPE_Q5_1: 24, GGML_TYPE_Q8_0: 34}
# hparams.ftype (enum ggml_ftype) -> tensor type of the quantised matrices
FTYPE_TO_QTYPE = {2: GGML_TYPE_Q4_0, 3: GGML_TYPE_Q4_1, 7: GGML_TYPE_Q8_0, 8: GGML_TYPE_Q5_0, 9: GGML_TYPE_Q5_1}


def quantize_blocks(a: np.ndarray, qtype: int) -> bytes:
    """Reference-style quantisation of a float32 array (size % 32 == 0) into GGML blocks."""
    x = np.ascontiguousarray(a, np.float32).reshape(-1, 32)
    nblk = x.shape[0]
    out = np.zeros((nblk, QUANT_BLOCK_BYTES[qtype]), np.uint8)
    if qtype == GGML_TYPE_Q8_0:
        d = np.abs(x).max(axis=1) / 127.0
        inv = np.where(d > 0, 1.0 / np.where(d > 0, d, 1), 0).astype(np.float32)
        q = np.clip(np.round(x * inv[:, None]), -127, 127).astype(np.int8)
        out[:, 0:2] = d.astype("<f2").view(np.uint8).reshape(nblk, 2)
        out[:, 2:] = q.view(np.uint8)
        return out.tobytes()
    sym = qtype in (GGML_TYPE_Q4_0, GGML_TYPE_Q5_0)
    levels = 16 if qtype in (GGML_TYPE_Q4_0, GGML_TYPE_Q4_1) else 32
    if sym:  # d = (value of largest magnitude) / -(levels/2)
        idx = np.abs(x).argmax(axis=1)
        mx = x[np.arange(nblk), idx]
        d = mx / -(levels // 2)
        inv = np.where(d != 0, 1.0 / np.where(d != 0, d, 1), 0).astype(np.float32)
        q = np.minimum(levels - 1, (x * inv[:, None] + (levels // 2 + 0.5)).astype(np.int32)).clip(0, levels - 1)
        head = d.astype("<f2").view(np.uint8).reshape(nblk, 2)
        pos = 2
    else:
        mn, mxv = x.min(axis=1), x.max(axis=1)
        d = (mxv - mn) / (levels - 1)
        inv = np.where(d != 0, 1.0 / np.where(d != 0, d, 1), 0).astype(np.float32)
        q = np.minimum(levels - 1, ((x - mn[:, None]) * inv[:, None] + 0.5).astype(np.int32)).clip(0, levels - 1)
        head = np.concatenate([d.astype("<f2").view(np.uint8).reshape(nblk, 2), mn.astype("<f2").view(np.uint8).reshape(nblk, 2)], axis=1)
        pos = 4
    out[:, :pos] = head
    if levels == 32:
        bits = ((q >> 4) & 1).astype(np.uint32)
        qh = (bits << np.arange(32, dtype=np.uint32)[None, :]).sum(axis=1).astype("<u4")
        out[:, pos:pos + 4] = qh.view(np.uint8).reshape(nblk, 4)
        pos += 4
    out[:, pos:pos + 16] = ((q[:, :16] & 0xF) | ((q[:, 16:] & 0xF) << 4)).astype(np.uint8)
    return out.tobytes()


def dequantize_blocks(buf, qtype: int, n: int, off: int = 0) -> np.ndarray:
    """GGML blocks -> float32 (numpy restatement of ggml's dequantize_row_q*)."""
    bb = QUANT_BLOCK_BYTES[qtype]
    nblk = n // 32
    b = np.frombuffer(buf, np.uint8, nblk * bb, off).reshape(nblk, bb)
    d = b[:, 0:2].copy().view("<f2").astype(np.float32)  # [nblk, 1]
    if qtype == GGML_TYPE_Q8_0:
        return (b[:, 2:].copy().view(np.int8).astype(np.float32) * d).reshape(-1)
    has_m = qtype in (GGML_TYPE_Q4_1, GGML_TYPE_Q5_1)
    has_h = qtype in (GGML_TYPE_Q5_0, GGML_TYPE_Q5_1)
    pos = 2
    m = 0.0
    if has_m:
        m = b[:, 2:4].copy().view("<f2").astype(np.float32)
        pos = 4
    hi = np.zeros((nblk, 32), np.int32)
    if has_h:
        qh = b[:, pos:pos + 4].copy().view("<u4").astype(np.uint32)  # [nblk, 1]
        hi = ((qh >> np.arange(32, dtype=np.uint32)[None, :]) & 1).astype(np.int32) << 4
        pos += 4
    qs = b[:, pos:pos + 16].astype(np.int32)
    x = np.concatenate([qs & 0xF, qs >> 4], axis=1) | hi
    if has_m:
        return (x.astype(np.float32) * d + m).reshape(-1)
    return ((x - (16 if has_h else 8)).astype(np.float32) * d).reshape(-1)


# --------------------------------------------------------------------------------------------
# writer / reader
# --------------------------------------------------------------------------------------------
def write_ggml(path: str, hp: HParams, tensors: Dict[str, np.ndarray], vocab: List[bytes] | None = None,
               mel_filters: np.ndarray | None = None) -> None:
    if mel_filters is None:
        mel_filters = mel_filterbank(hp.n_mels)
    if vocab is None:
        vocab = synthetic_vocab(n_vocab_in_file(hp.n_vocab))
    with open(path, "wb") as f:
        f.write(struct.pack("<I", GGML_MAGIC))
        f.write(struct.pack("<11i", *hp.as_list()))
        n_mel, n_fft = mel_filters.shape
        f.write(struct.pack("<ii", n_mel, n_fft))
        f.write(np.ascontiguousarray(mel_filters, dtype="<f4").tobytes())
        f.write(struct.pack("<i", len(vocab)))
        for w in vocab:
            f.write(struct.pack("<I", len(w)))
            f.write(w)
        for name, shape, is_mat in tensor_shapes(hp):
            a = np.ascontiguousarray(tensors[name], dtype=np.float32).reshape(shape)
            qtype = FTYPE_TO_QTYPE.get(hp.ftype % 1000)
            # quantised files: every 2-d weight is quantised, the (3-d) conv weights stay f16, vectors f32
            quant = qtype is not None and is_mat and len(shape) == 2 and shape[-1] % 32 == 0
            use_f16 = is_mat and not quant and hp.ftype % 1000 != 0
            ttype = qtype if quant else (GGML_TYPE_F16 if use_f16 else GGML_TYPE_F32)
            ne = list(reversed(shape))  # ggml order: ne[0] is the fastest dimension
            nb = name.encode()
            f.write(struct.pack("<iii", len(ne), len(nb), ttype))
            f.write(struct.pack("<%di" % len(ne), *ne))
            f.write(nb)
            f.write(quantize_blocks(a, qtype) if quant else a.astype("<f2" if use_f16 else "<f4").tobytes())


def read_ggml(path: str):
    """Returns (HParams, mel_filters, vocab list[bytes], {name: float32 ndarray})."""
    with open(path, "rb") as f:
        buf = f.read()
    off = 0
    (magic,) = struct.unpack_from("<I", buf, off); off += 4
    assert magic == GGML_MAGIC, hex(magic)
    hp = HParams(*struct.unpack_from("<11i", buf, off)); off += 44
    n_mel, n_fft = struct.unpack_from("<ii", buf, off); off += 8
    filt = np.frombuffer(buf, "<f4", n_mel * n_fft, off).reshape(n_mel, n_fft).copy(); off += 4 * n_mel * n_fft
    (nv,) = struct.unpack_from("<i", buf, off); off += 4
    vocab = []
    for _ in range(nv):
        (ln,) = struct.unpack_from("<I", buf, off); off += 4
        vocab.append(buf[off:off + ln]); off += ln
    tensors = {}
    while off < len(buf):
        n_dims, name_len, ttype = struct.unpack_from("<iii", buf, off); off += 12
        ne = struct.unpack_from("<%di" % n_dims, buf, off); off += 4 * n_dims
        name = buf[off:off + name_len].decode(); off += name_len
        n = int(np.prod(ne))
        if ttype in QUANT_BLOCK_BYTES:
            a = dequantize_blocks(buf, ttype, n, off); off += n // 32 * QUANT_BLOCK_BYTES[ttype]
        elif ttype == GGML_TYPE_F16:
            a = np.frombuffer(buf, "<f2", n, off).astype(np.float32); off += 2 * n
        else:
            a = np.frombuffer(buf, "<f4", n, off).astype(np.float32); off += 4 * n
        tensors[name] = a.reshape(tuple(reversed(ne)))
    return hp, filt, vocab, tensors


def f16_round(t: Dict[str, np.ndarray], hp: HParams) -> Dict[str, np.ndarray]:
    """The tensor values a loader sees after the f16 storage of matrix weights (ftype 1)."""
    if hp.ftype != 1:
        return t
    mats = {n for n, _, m in tensor_shapes(hp) if m}
    return {k: (v.astype(np.float16).astype(np.float32) if k in mats else v) for k, v in t.items()}


# --------------------------------------------------------------------------------------------
# OCP e4m3 (fp8) with one power-of-two scale per row: the engine's MW_PRECISION_FP8 weight format
# --------------------------------------------------------------------------------------------
FP8_DECODER_SUFFIXES = (".attn.query.weight", ".attn.key.weight", ".attn.value.weight", ".attn.out.weight",
                        ".cross_attn.query.weight", ".cross_attn.out.weight", ".mlp.0.weight", ".mlp.2.weight")


def is_fp8_decoder_tensor(name: str) -> bool:
    """The tensors MW_PRECISION_FP8 quantises: decoder projections + the (tied) token embedding; not the cross K/V weights."""
    return name == "decoder.token_embedding.weight" or (name.startswith("decoder.blocks.") and name.endswith(FP8_DECODER_SUFFIXES))


def e4m3_quantize_dequantize_rows(a: np.ndarray) -> np.ndarray:
    """a [N][K] float32 -> the nearest values of the form e4m3 x 2^s(row) (round to nearest even, saturating at 448),
    s(row) the smallest exponent with 448 x 2^s >= max |row|.  Idempotent; every result is exact in bf16 and f16."""
    a = np.asarray(a, dtype=np.float64)
    mx = np.abs(a).max(axis=1, keepdims=True)
    with np.errstate(divide="ignore"):
        e = np.where(mx > 0, np.ceil(np.log2(mx / 448.0)), 0.0)
    e = np.where(448.0 * 2.0 ** (e - 1) >= mx, e - 1, e)       # guard the log2 rounding at exact powers of two
    e = np.where(448.0 * 2.0 ** e < mx, e + 1, e)
    y = a / 2.0 ** e
    ay = np.abs(y)
    with np.errstate(divide="ignore"):
        ex = np.where(ay > 0, np.floor(np.log2(ay)), -6.0)
    ex = np.maximum(ex, -6.0)                                   # below 2^-6: subnormals, fixed step 2^-9
    step = 2.0 ** (ex - 3.0)
    q = np.minimum(np.rint(ay / step) * step, 448.0)
    return (np.sign(y) * q * 2.0 ** e).astype(np.float32)


# --------------------------------------------------------------------------------------------
# MX-fp8 (OCP microscaling: e4m3 elements, one power-of-two scale per block of 32 along k): the format of the encoder
# products MW_PRECISION_FP8_MX runs on the fp8 matrix cores (obs-localvocal_amd/csrc/gemm_mx.hip)
# --------------------------------------------------------------------------------------------
MX_ENCODER_SUFFIXES = (".attn.query.weight", ".attn.key.weight", ".attn.value.weight", ".attn.out.weight", ".mlp.0.weight", ".mlp.2.weight")


def is_mx_encoder_tensor(name: str) -> bool:
    """The matrices MW_PRECISION_FP8_MX multiplies in MX-fp8: Q, K, V, attention output, fc1, fc2 of every encoder block and the decoder layers'
    cross K / V projections of the encoder's output."""
    return (name.startswith("encoder.blocks.") and name.endswith(MX_ENCODER_SUFFIXES)) or \
        (name.startswith("decoder.blocks.") and name.endswith((".cross_attn.key.weight", ".cross_attn.value.weight")))


def mxfp8_quantize_dequantize(a: np.ndarray) -> np.ndarray:
    """a [..., K] float32, K % 32 == 0 -> the nearest MX-fp8 values: per block of 32 along the last axis the smallest
    power-of-two scale that brings the block under 448, e4m3 elements rounded to nearest even.  Idempotent; exact in bf16/f16."""
    a = np.asarray(a, dtype=np.float32)
    assert a.shape[-1] % 32 == 0
    return e4m3_quantize_dequantize_rows(a.reshape(-1, 32)).reshape(a.shape)


def make_model_file(path: str, preset: str = "micro.en", seed: int = 1234, ftype: int | None = None,
                    quiet_special: bool = False, **overrides) -> HParams:
    hp = HParams(**{**PRESETS[preset].__dict__, **overrides})
    if ftype is not None:
        hp.ftype = ftype
    write_ggml(path, hp, synthetic_tensors(hp, seed, quiet_special=quiet_special))
    return hp


# --------------------------------------------------------------------------------------------
# synthetic PCM (chirps + AM + noise)
# --------------------------------------------------------------------------------------------
def synth_pcm(seed: int = 0, seconds: float = 30.0, sr: int = 16000) -> np.ndarray:
    rng = np.random.RandomState(seed)
    n = int(round(seconds * sr))
    t = np.arange(n, dtype=np.float64) / sr
    x = np.zeros(n, dtype=np.float64)
    for _ in range(5):
        f0 = rng.uniform(100, 2000)
        f1 = rng.uniform(f0, 4000)
        ph = rng.uniform(0, 2 * np.pi)
        x += np.sin(ph + 2 * np.pi * (f0 * t + 0.5 * (f1 - f0) * t * t / max(seconds, 1e-9)))
    env = 0.5 * (1.0 + np.sin(2 * np.pi * 4.0 * t + rng.uniform(0, 2 * np.pi)))
    x = 0.1 * x / 5.0 * env + 0.005 * rng.standard_normal(n)
    return x.astype(np.float32)


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--preset", default="micro.en")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--ftype", type=int, default=None)
    a = ap.parse_args()
    hp = make_model_file(a.out, a.preset, a.seed, a.ftype)
    print(hp)
