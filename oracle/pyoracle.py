"""ctypes binding of the CPU oracle (oracle/liboracle_whisper.so).

TEST INFRASTRUCTURE ONLY — importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never from the product package (see oracle/whisper_oracle.h).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle_whisper.so")


class OwParams(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "strategy", "n_threads", "n_max_text_ctx", "offset_ms", "duration_ms", "translate", "no_context",
        "no_timestamps", "single_segment", "print_special", "max_tokens", "audio_ctx", "tdrz_enable",
        "detect_language", "suppress_blank", "suppress_non_speech_tokens")] + [
        (n, C.c_float) for n in ("temperature", "max_initial_ts", "length_penalty", "temperature_inc",
                                 "entropy_thold", "logprob_thold", "no_speech_thold")] + [
        ("best_of", C.c_int32), ("beam_size", C.c_int32), ("patience", C.c_float),
        ("force_decode_steps", C.c_int32),
        ("suppress_regex", C.c_char_p), ("initial_prompt", C.c_char_p), ("language", C.c_char_p)]


class OwTokenData(C.Structure):
    _fields_ = [("id", C.c_int32), ("tid", C.c_int32), ("p", C.c_float), ("plog", C.c_float),
                ("pt", C.c_float), ("ptsum", C.c_float), ("t0", C.c_int64), ("t1", C.c_int64),
                ("t_dtw", C.c_int64), ("vlen", C.c_float)]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "whisper_oracle.cpp")
    if force or not os.path.exists(_LIB_PATH) or (
            os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(_LIB_PATH)):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        P = C.c_void_p
        L.ow_init_from_file.restype = P; L.ow_init_from_file.argtypes = [C.c_char_p]
        L.ow_free.argtypes = [P]
        L.ow_set_threads.argtypes = [C.c_int]
        L.ow_hparams.argtypes = [P, C.POINTER(C.c_int32)]
        L.ow_special_tokens.argtypes = [P, C.POINTER(C.c_int32)]
        L.ow_token_to_str.restype = C.c_char_p; L.ow_token_to_str.argtypes = [P, C.c_int]
        L.ow_tokenize.argtypes = [P, C.c_char_p, C.POINTER(C.c_int32), C.c_int]
        L.ow_default_params.restype = OwParams; L.ow_default_params.argtypes = [C.c_int]
        L.ow_pcm_to_mel.argtypes = [P, C.POINTER(C.c_float), C.c_int]
        L.ow_mel_n_len.argtypes = [P]; L.ow_mel_n_len_org.argtypes = [P]
        L.ow_mel_data.restype = C.POINTER(C.c_float); L.ow_mel_data.argtypes = [P]
        L.ow_encode.argtypes = [P, C.c_int, C.c_int]
        L.ow_encoder_out.restype = C.POINTER(C.c_float); L.ow_encoder_out.argtypes = [P]
        L.ow_cross_kv.restype = C.POINTER(C.c_float); L.ow_cross_kv.argtypes = [P, C.c_int, C.c_int]
        L.ow_decode.argtypes = [P, C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int]
        L.ow_logits.restype = C.POINTER(C.c_float); L.ow_logits.argtypes = [P]
        FP = C.POINTER(C.c_float)
        L.ow_set_mx_encoder.argtypes = [P, C.c_int]; L.ow_set_mx_encoder.restype = None
        L.ow_mx_qdq.argtypes = [FP, C.c_int, C.c_int]; L.ow_mx_qdq.restype = None
        L.ow_process_logits.argtypes = [P, C.POINTER(OwParams), C.c_float, FP, C.POINTER(C.c_int32), C.c_int,
                                        C.c_int, C.c_int, FP, FP, FP]
        L.ow_full.argtypes = [P, C.POINTER(OwParams), FP, C.c_int]
        L.ow_full_n_segments.argtypes = [P]
        L.ow_full_get_segment_t0.restype = C.c_int64; L.ow_full_get_segment_t0.argtypes = [P, C.c_int]
        L.ow_full_get_segment_t1.restype = C.c_int64; L.ow_full_get_segment_t1.argtypes = [P, C.c_int]
        L.ow_full_get_segment_text.restype = C.c_char_p; L.ow_full_get_segment_text.argtypes = [P, C.c_int]
        L.ow_full_n_tokens.argtypes = [P, C.c_int]
        L.ow_full_get_token_data_p.argtypes = [P, C.c_int, C.c_int, C.POINTER(OwTokenData)]
        L.ow_full_lang_id.argtypes = [P]
        L.ow_lang_auto_detect.argtypes = [P, C.c_int, FP]
        L.ow_lang_str.restype = C.c_char_p; L.ow_lang_str.argtypes = [C.c_int]
        L.ow_lang_id.argtypes = [C.c_char_p]
        L.ow_n_decode_calls.argtypes = [P]; L.ow_n_encode_calls.argtypes = [P]
        _lib = L
    return _lib


def _fp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class Oracle:
    def __init__(self, model_path: str):
        self.L = lib()
        self.ctx = self.L.ow_init_from_file(model_path.encode())
        if not self.ctx:
            raise RuntimeError("oracle: failed to load " + model_path)
        hp = (C.c_int32 * 11)()
        self.L.ow_hparams(self.ctx, hp)
        (self.n_vocab, self.n_audio_ctx, self.n_audio_state, self.n_audio_head, self.n_audio_layer,
         self.n_text_ctx, self.n_text_state, self.n_text_head, self.n_text_layer, self.n_mels, self.ftype) = list(hp)
        sp = (C.c_int32 * 9)()
        self.L.ow_special_tokens(self.ctx, sp)
        (self.eot, self.sot, self.translate, self.transcribe, self.solm, self.prev, self.nosp, self.not_,
         self.beg) = list(sp)
        self._keep = []

    def close(self):
        if self.ctx:
            self.L.ow_free(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def default_params(self, strategy: int = 0, **kw) -> OwParams:
        p = self.L.ow_default_params(strategy)
        for k, v in kw.items():
            if isinstance(v, str):
                v = v.encode()
            if isinstance(v, bytes):
                self._keep.append(v)
            setattr(p, k, v)
        return p

    def token_to_str(self, i: int) -> bytes:
        return self.L.ow_token_to_str(self.ctx, i)

    def tokenize(self, text: str):
        buf = (C.c_int32 * 2048)()
        n = self.L.ow_tokenize(self.ctx, text.encode(), buf, 2048)
        return list(buf[:max(n, 0)])

    def mel(self, pcm: np.ndarray) -> np.ndarray:
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        self.L.ow_pcm_to_mel(self.ctx, _fp(pcm), len(pcm))
        n_len = self.L.ow_mel_n_len(self.ctx)
        return np.ctypeslib.as_array(self.L.ow_mel_data(self.ctx), (self.n_mels, n_len)).copy()

    def set_mx_encoder(self, on: bool = True) -> None:
        """the MX-fp8 twin of the engine's PRECISION_FP8_MX encoder (whisper_oracle.cpp: mx_qdq_rows)"""
        self.L.ow_set_mx_encoder(self.ctx, 1 if on else 0)

    def encode(self, seek: int = 0, audio_ctx: int = 0) -> np.ndarray:
        self.L.ow_encode(self.ctx, seek, audio_ctx)
        n_ctx = audio_ctx if audio_ctx > 0 else self.n_audio_ctx
        return np.ctypeslib.as_array(self.L.ow_encoder_out(self.ctx), (n_ctx, self.n_audio_state)).copy()

    def cross_kv(self, layer: int, is_v: bool, n_ctx: int | None = None) -> np.ndarray:
        n_ctx = n_ctx or self.n_audio_ctx
        return np.ctypeslib.as_array(self.L.ow_cross_kv(self.ctx, layer, int(is_v)), (n_ctx, self.n_text_state)).copy()

    def decode(self, tokens, n_past: int = 0, seq: int = 0) -> np.ndarray:
        t = np.ascontiguousarray(tokens, dtype=np.int32)
        rc = self.L.ow_decode(self.ctx, t.ctypes.data_as(C.POINTER(C.c_int32)), len(t), n_past, seq)
        if rc != 0:
            raise RuntimeError("oracle decode failed")
        return np.ctypeslib.as_array(self.L.ow_logits(self.ctx), (len(t), self.n_vocab)).copy()

    def process_logits(self, params: OwParams, logits: np.ndarray, history, temperature=0.0, has_ts=False,
                       seek_delta=3000):
        lg = np.ascontiguousarray(logits, dtype=np.float32)
        h = np.ascontiguousarray(history, dtype=np.int32)
        o1 = np.empty(self.n_vocab, np.float32); o2 = np.empty_like(o1); o3 = np.empty_like(o1)
        self.L.ow_process_logits(self.ctx, C.byref(params), float(temperature), _fp(lg),
                                 h.ctypes.data_as(C.POINTER(C.c_int32)), len(h), int(has_ts), int(seek_delta),
                                 _fp(o1), _fp(o2), _fp(o3))
        return o1, o2, o3

    def full(self, params: OwParams, pcm: np.ndarray) -> int:
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        return self.L.ow_full(self.ctx, C.byref(params), _fp(pcm), len(pcm))

    def segments(self):
        out = []
        for i in range(self.L.ow_full_n_segments(self.ctx)):
            toks = []
            for j in range(self.L.ow_full_n_tokens(self.ctx, i)):
                td = OwTokenData()
                self.L.ow_full_get_token_data_p(self.ctx, i, j, C.byref(td))
                toks.append(dict(id=td.id, tid=td.tid, p=td.p, plog=td.plog, pt=td.pt, ptsum=td.ptsum))
            out.append(dict(t0=self.L.ow_full_get_segment_t0(self.ctx, i), t1=self.L.ow_full_get_segment_t1(self.ctx, i),
                            text=self.L.ow_full_get_segment_text(self.ctx, i), tokens=toks))
        return out

    def token_ids(self):
        return [t["id"] for s in self.segments() for t in s["tokens"]]

    def lang_auto_detect(self, offset_ms: int = 0):
        probs = np.zeros(100, np.float32)
        lid = self.L.ow_lang_auto_detect(self.ctx, offset_ms, _fp(probs))
        return lid, probs


def mx_qdq(a: np.ndarray) -> np.ndarray:
    """MX-fp8 quantise + dequantise of the rows of a 2-D array (blocks of 32 along the last axis): the oracle's statement of
    the rule the engine's MX-fp8 kernels implement."""
    out = np.ascontiguousarray(a, dtype=np.float32).copy()
    assert out.ndim == 2 and out.shape[1] % 32 == 0
    lib().ow_mx_qdq(_fp(out), out.shape[0], out.shape[1])
    return out
