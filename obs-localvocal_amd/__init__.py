"""obs-localvocal_amd — thin ctypes binding of libmi355_whisper.so (the C ABI in include/*.h).

The product is the shared library; this module only declares its entry points for Python callers
(tests, bench.py, __graft_entry__.py).  It contains no arithmetic and no fallback: if the library is
missing or no MI355X is usable, loading a model raises.

The directory name carries a hyphen (repo contract), so import it through `load_package()` in
tests/conftest.py or `importlib` (see __graft_entry__.py); the module registers itself as
`obs_localvocal_amd`.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MI355_WHISPER_LIB: another BUILD of this same library (development A/B runs of two kernel variants on one box); no other
# library can stand in, every symbol of include/*.h is bound below
LIB_PATH = os.environ.get("MI355_WHISPER_LIB") or os.path.join(_HERE, "libmi355_whisper.so")
CSRC = os.path.join(_HERE, "csrc")

PRECISION_BF16 = 0
PRECISION_FP32 = 1
PRECISION_FP8 = 2
PRECISION_FP8_MX = 3
SAMPLING_GREEDY = 0
SAMPLING_BEAM_SEARCH = 1


class TokenData(C.Structure):  # whisper_token_data
    _fields_ = [("id", C.c_int32), ("tid", C.c_int32), ("p", C.c_float), ("plog", C.c_float),
                ("pt", C.c_float), ("ptsum", C.c_float), ("t0", C.c_int64), ("t1", C.c_int64),
                ("t_dtw", C.c_int64), ("vlen", C.c_float)]


class _Greedy(C.Structure):
    _fields_ = [("best_of", C.c_int)]


class _Beam(C.Structure):
    _fields_ = [("beam_size", C.c_int), ("patience", C.c_float)]


class FullParams(C.Structure):  # whisper_full_params (include/whisper.h)
    _fields_ = [
        ("strategy", C.c_int), ("n_threads", C.c_int), ("n_max_text_ctx", C.c_int), ("offset_ms", C.c_int),
        ("duration_ms", C.c_int),
        ("translate", C.c_bool), ("no_context", C.c_bool), ("no_timestamps", C.c_bool), ("single_segment", C.c_bool),
        ("print_special", C.c_bool), ("print_progress", C.c_bool), ("print_realtime", C.c_bool),
        ("print_timestamps", C.c_bool),
        ("token_timestamps", C.c_bool), ("thold_pt", C.c_float), ("thold_ptsum", C.c_float), ("max_len", C.c_int),
        ("split_on_word", C.c_bool), ("max_tokens", C.c_int),
        ("speed_up", C.c_bool), ("debug_mode", C.c_bool), ("audio_ctx", C.c_int),
        ("tdrz_enable", C.c_bool),
        ("suppress_regex", C.c_char_p), ("initial_prompt", C.c_char_p),
        ("prompt_tokens", C.POINTER(C.c_int32)), ("prompt_n_tokens", C.c_int),
        ("language", C.c_char_p), ("detect_language", C.c_bool),
        ("suppress_blank", C.c_bool), ("suppress_non_speech_tokens", C.c_bool),
        ("temperature", C.c_float), ("max_initial_ts", C.c_float), ("length_penalty", C.c_float),
        ("temperature_inc", C.c_float), ("entropy_thold", C.c_float), ("logprob_thold", C.c_float),
        ("no_speech_thold", C.c_float),
        ("greedy", _Greedy), ("beam_search", _Beam),
        ("new_segment_callback", C.c_void_p), ("new_segment_callback_user_data", C.c_void_p),
        ("progress_callback", C.c_void_p), ("progress_callback_user_data", C.c_void_p),
        ("encoder_begin_callback", C.c_void_p), ("encoder_begin_callback_user_data", C.c_void_p),
        ("abort_callback", C.c_void_p), ("abort_callback_user_data", C.c_void_p),
        ("logits_filter_callback", C.c_void_p), ("logits_filter_callback_user_data", C.c_void_p),
        ("grammar_rules", C.c_void_p), ("n_grammar_rules", C.c_size_t), ("i_start_rule", C.c_size_t),
        ("grammar_penalty", C.c_float),
    ]


class ContextParams(C.Structure):  # whisper_context_params
    _fields_ = [("use_gpu", C.c_bool), ("flash_attn", C.c_bool), ("gpu_device", C.c_int),
                ("dtw_token_timestamps", C.c_bool), ("dtw_aheads_preset", C.c_int), ("dtw_n_top", C.c_int),
                ("dtw_aheads_n_heads", C.c_size_t), ("dtw_aheads_heads", C.c_void_p), ("dtw_mem_size", C.c_size_t)]


class Timings(C.Structure):
    _fields_ = [("mel_ms", C.c_double), ("encode_ms", C.c_double), ("decode_ms", C.c_double),
                ("sample_ms", C.c_double), ("total_ms", C.c_double), ("n_encode", C.c_int), ("n_decode", C.c_int),
                ("n_sample", C.c_int)]


class LvDetection(C.Structure):
    _fields_ = [("result", C.c_int), ("text", C.c_char_p), ("start_timestamp_ms", C.c_uint64),
                ("end_timestamp_ms", C.c_uint64), ("tokens", C.POINTER(TokenData)), ("n_tokens", C.c_int),
                ("language", C.c_char_p)]


# every symbol include/whisper.h, include/mi355_whisper.h and include/localvocal_hotpath.h declare
EXPORTED_SYMBOLS = [
    "whisper_log_set", "whisper_context_default_params", "whisper_context_default_params_by_ref",
    "whisper_free_context_params", "whisper_init_from_file_with_params", "whisper_init_from_buffer_with_params",
    "whisper_print_system_info", "whisper_free", "whisper_full_default_params", "whisper_full_default_params_by_ref",
    "whisper_free_params", "whisper_full", "whisper_full_n_segments", "whisper_full_n_tokens",
    "whisper_full_get_token_data", "whisper_token_to_str", "whisper_lang_auto_detect", "whisper_lang_str",
    "whisper_full_get_segment_t0", "whisper_full_get_segment_t1", "whisper_full_get_segment_text",
    "whisper_full_get_token_id", "whisper_full_get_token_p", "whisper_full_lang_id", "whisper_lang_id",
    "whisper_lang_max_id", "whisper_is_multilingual", "whisper_n_vocab", "whisper_n_text_ctx", "whisper_n_audio_ctx",
    "whisper_n_len", "whisper_token_eot", "whisper_token_sot", "whisper_token_solm", "whisper_token_prev",
    "whisper_token_nosp", "whisper_token_not", "whisper_token_beg", "whisper_token_lang", "whisper_token_translate",
    "whisper_token_transcribe", "whisper_tokenize",
    "mw_full_device", "mw_profile_begin", "mw_profile_end",
    "mw_init_from_file", "mw_init_from_buffer", "mw_init_shared", "mw_hparams", "mw_pcm_to_mel", "mw_pcm_to_mel_device", "mw_pcm_to_mel_noalloc", "mw_mel_n_len", "mw_mx_gemm", "mw_mx_layernorm",
    "mw_get_mel", "mw_encode", "mw_get_encoder_out", "mw_get_cross_kv", "mw_decode", "mw_decode_rows", "mw_get_logits", "mw_kv_clear",
    "mw_process_logits", "mw_get_timings", "mw_reset_timings", "mw_stream", "mw_synchronize",
    "mw_set_force_decode_steps", "mw_hub_stats", "mw_set_duplicate_class", "mw_bench_decode_step", "mw_bench_mel", "mw_bench_encode", "mw_last_error",
    "lv_filter_create", "lv_filter_destroy", "lv_filter_params", "lv_filter_set_language", "lv_filter_set_thresholds",
    "lv_filter_set_context_sentences", "lv_filter_push_sentence", "lv_filter_set_gpu_device", "lv_filter_load_model",
    "lv_filter_unload_model", "lv_filter_context", "lv_run_whisper_inference", "lv_to_timestamp",
    "lv_settings_set_int", "lv_settings_set_double", "lv_settings_set_bool", "lv_settings_set_string",
    "lv_settings_get_int", "lv_settings_get_double", "lv_settings_get_bool", "lv_filter_apply_settings",
    "lv_filter_setup_audio", "lv_filter_set_vad_mode", "lv_filter_set_partial", "lv_filter_set_segment_duration",
    "lv_filter_set_max_sub_duration", "lv_filter_enable_audio_chunks_callback", "lv_filter_set_vad_gate",
    "lv_filter_set_callbacks", "lv_filter_filter_audio", "lv_filter_push_packet", "lv_filter_clear_buffers",
    "lv_filter_input_bytes", "lv_filter_whisper_bytes", "lv_filter_segmentation_step", "lv_filter_reset_vad_state",
    "lv_filter_start", "lv_filter_stop", "lv_filter_update_model", "lv_filter_register_model", "lv_filter_model_state",
    "lv_filter_load_catalogue", "lv_filter_verify_model", "lv_model_file_sha256",
    "lv_vad_speech_timestamps", "lv_remove_leading_trailing_nonalpha",
    "lv_filter_set_silero_model", "lv_filter_vad_is_silero", "lv_filter_vad_probabilities",
    "lv_silero_create", "lv_silero_process", "lv_silero_process_batch", "lv_silero_state", "lv_silero_destroy", "lv_silero_inspect",
    "lv_resample",
]

LV_TEXT_CB = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(LvDetection))
LV_CHUNK_CB = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_float), C.c_size_t, C.c_int, C.POINTER(LvDetection))
LV_REQUEST_CB = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_float), C.c_size_t, C.c_uint64, C.c_uint64, C.c_int)
VAD_MODE_ACTIVE, VAD_MODE_HYBRID, VAD_MODE_DISABLED = 0, 1, 2
VAD_STATE_WAS_ON, VAD_STATE_WAS_OFF, VAD_STATE_IS_OFF, VAD_STATE_PARTIAL = 0, 1, 2, 3


def build(force: bool = False) -> str:
    """Compile libmi355_whisper.so for gfx950 (hipcc cross-compiles; no GPU needed)."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean", "-s"])
    subprocess.check_call(["make", "-C", CSRC, "-j8", "-s"])
    return LIB_PATH


_lib = None


def bind_whisper_abi(L):
    """ctypes signatures of the whisper.h symbols LocalVocal binds (include/whisper.h), on any library that exports them:
    this one, or a whisper.cpp v1.6.2 build named by WHISPER_CPP_LIB (tests/test_pinning_hooks.py)."""
    P = C.c_void_p
    FP = C.POINTER(C.c_float)
    IP = C.POINTER(C.c_int32)
    L.whisper_context_default_params.restype = ContextParams
    L.whisper_init_from_file_with_params.restype = P
    L.whisper_init_from_file_with_params.argtypes = [C.c_char_p, ContextParams]
    L.whisper_init_from_buffer_with_params.restype = P
    L.whisper_init_from_buffer_with_params.argtypes = [P, C.c_size_t, ContextParams]
    L.whisper_print_system_info.restype = C.c_char_p
    L.whisper_free.argtypes = [P]
    L.whisper_full_default_params.restype = FullParams
    L.whisper_full_default_params.argtypes = [C.c_int]
    L.whisper_full.argtypes = [P, FullParams, FP, C.c_int]
    L.whisper_full_n_segments.argtypes = [P]
    L.whisper_full_n_tokens.argtypes = [P, C.c_int]
    L.whisper_full_get_token_data.restype = TokenData
    L.whisper_full_get_token_data.argtypes = [P, C.c_int, C.c_int]
    L.whisper_token_to_str.restype = C.c_char_p
    L.whisper_token_to_str.argtypes = [P, C.c_int]
    L.whisper_lang_auto_detect.argtypes = [P, C.c_int, C.c_int, FP]
    L.whisper_lang_str.restype = C.c_char_p
    L.whisper_lang_str.argtypes = [C.c_int]
    L.whisper_lang_id.argtypes = [C.c_char_p]
    L.whisper_full_get_segment_t0.restype = C.c_int64
    L.whisper_full_get_segment_t0.argtypes = [P, C.c_int]
    L.whisper_full_get_segment_t1.restype = C.c_int64
    L.whisper_full_get_segment_t1.argtypes = [P, C.c_int]
    L.whisper_full_get_segment_text.restype = C.c_char_p
    L.whisper_full_get_segment_text.argtypes = [P, C.c_int]
    L.whisper_full_lang_id.argtypes = [P]
    L.whisper_tokenize.argtypes = [P, C.c_char_p, IP, C.c_int]
    for n in ("whisper_token_eot", "whisper_token_sot", "whisper_token_solm", "whisper_token_prev", "whisper_token_nosp",
              "whisper_token_not", "whisper_token_beg", "whisper_token_translate", "whisper_token_transcribe",
              "whisper_n_vocab", "whisper_n_text_ctx", "whisper_n_audio_ctx", "whisper_n_len", "whisper_is_multilingual"):
        getattr(L, n).argtypes = [P]
    L.whisper_token_lang.argtypes = [P, C.c_int]
    return L


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libmi355_whisper.so is not built (run __graft_entry__.build()); there is no fallback path")
    L = bind_whisper_abi(C.CDLL(LIB_PATH))
    P = C.c_void_p
    FP = C.POINTER(C.c_float)
    IP = C.POINTER(C.c_int32)
    L.mw_init_from_file.restype = P
    L.mw_init_from_file.argtypes = [C.c_char_p, C.c_int, C.c_int]
    L.mw_init_from_buffer.restype = P
    L.mw_init_shared.restype = P
    L.mw_init_shared.argtypes = [P]
    L.mw_init_from_buffer.argtypes = [P, C.c_size_t, C.c_int, C.c_int]
    L.mw_hparams.argtypes = [P, IP]
    L.mw_pcm_to_mel.argtypes = [P, FP, C.c_int]
    L.mw_pcm_to_mel_device.argtypes = [P, P, C.c_int]
    L.mw_pcm_to_mel_noalloc.argtypes = [P, P, C.c_int, P, C.c_size_t, C.POINTER(C.c_int)]
    FPm = C.POINTER(C.c_float)
    L.mw_mx_gemm.argtypes = [C.c_int, FPm, C.c_int, C.c_int, FPm, C.c_int, FPm, FPm, C.c_int, FPm, FPm, FPm]
    L.mw_mx_layernorm.argtypes = [C.c_int, FPm, FPm, FPm, C.c_int, C.c_int, FPm, FPm]
    L.mw_mel_n_len.argtypes = [P]
    L.mw_full_device.argtypes = [P, FullParams, P, C.c_int]
    L.mw_get_mel.argtypes = [P, FP]
    L.mw_encode.argtypes = [P, C.c_int, C.c_int]
    L.mw_get_encoder_out.argtypes = [P, FP]
    L.mw_get_cross_kv.argtypes = [P, C.c_int, C.c_int, FP]
    L.mw_decode.argtypes = [P, IP, C.c_int, C.c_int, C.c_int]
    L.mw_decode_rows.argtypes = [P, IP, IP, IP, C.c_int]
    L.mw_get_logits.argtypes = [P, FP]
    L.mw_kv_clear.argtypes = [P]
    L.mw_process_logits.argtypes = [P, C.POINTER(FullParams), C.c_float, FP, IP, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.POINTER(C.c_double), FP, FP, FP, C.POINTER(TokenData)]
    L.mw_get_timings.argtypes = [P, C.POINTER(Timings)]
    L.mw_reset_timings.argtypes = [P]
    L.mw_stream.restype = P
    L.mw_stream.argtypes = [P]
    L.mw_synchronize.argtypes = [P]
    L.mw_set_force_decode_steps.argtypes = [P, C.c_int]
    L.mw_set_duplicate_class.argtypes = [P, C.c_int]
    L.mw_hub_stats.argtypes = [P, C.POINTER(C.c_ulonglong)]
    L.mw_bench_decode_step.argtypes = [P, C.c_int, C.c_int, C.c_int, FP]
    L.mw_bench_mel.argtypes = [P, C.c_int, C.c_int, FP]
    L.mw_bench_encode.argtypes = [P, C.c_int, FP]
    L.mw_last_error.restype = C.c_char_p
    L.mw_profile_begin.argtypes = [P, C.c_int, C.c_int]
    L.mw_profile_end.argtypes = [P, C.POINTER(C.c_int), FP]
    L.lv_filter_create.restype = P
    L.lv_filter_destroy.argtypes = [P]
    L.lv_filter_params.restype = C.POINTER(FullParams)
    L.lv_filter_params.argtypes = [P]
    L.lv_filter_set_language.argtypes = [P, C.c_char_p]
    L.lv_filter_set_thresholds.argtypes = [P, C.c_float, C.c_float]
    L.lv_filter_set_context_sentences.argtypes = [P, C.c_int]
    L.lv_filter_push_sentence.argtypes = [P, C.c_char_p]
    L.lv_filter_set_gpu_device.argtypes = [P, C.c_int, C.c_int]
    L.lv_filter_load_model.argtypes = [P, C.c_char_p]
    L.lv_filter_unload_model.argtypes = [P]
    L.lv_filter_context.restype = P
    L.lv_filter_context.argtypes = [P]
    L.lv_run_whisper_inference.argtypes = [P, FP, C.c_size_t, C.c_uint64, C.c_uint64, C.c_int, C.POINTER(LvDetection)]
    L.lv_to_timestamp.argtypes = [C.c_uint64, C.c_char_p, C.c_size_t]
    FPP = C.POINTER(C.POINTER(C.c_float))
    L.lv_settings_set_int.argtypes = [P, C.c_char_p, C.c_longlong]
    L.lv_settings_set_double.argtypes = [P, C.c_char_p, C.c_double]
    L.lv_settings_set_bool.argtypes = [P, C.c_char_p, C.c_int]
    L.lv_settings_set_string.argtypes = [P, C.c_char_p, C.c_char_p]
    L.lv_settings_get_int.restype = C.c_longlong
    L.lv_settings_get_int.argtypes = [P, C.c_char_p]
    L.lv_settings_get_double.restype = C.c_double
    L.lv_settings_get_double.argtypes = [P, C.c_char_p]
    L.lv_settings_get_bool.argtypes = [P, C.c_char_p]
    L.lv_filter_apply_settings.argtypes = [P]
    L.lv_filter_setup_audio.argtypes = [P, C.c_uint32, C.c_uint32]
    L.lv_filter_set_vad_mode.argtypes = [P, C.c_int]
    L.lv_filter_set_partial.argtypes = [P, C.c_int, C.c_int]
    L.lv_filter_set_segment_duration.argtypes = [P, C.c_int]
    L.lv_filter_set_max_sub_duration.argtypes = [P, C.c_int]
    L.lv_filter_enable_audio_chunks_callback.argtypes = [P, C.c_int]
    L.lv_filter_set_vad_gate.argtypes = [P, C.c_float, C.c_float, C.c_float]
    L.lv_filter_set_callbacks.argtypes = [P, LV_TEXT_CB, LV_CHUNK_CB, LV_REQUEST_CB, C.c_void_p]
    L.lv_filter_filter_audio.argtypes = [P, FPP, C.c_uint32]
    L.lv_filter_push_packet.argtypes = [P, FPP, C.c_uint32, C.c_uint64]
    L.lv_filter_clear_buffers.argtypes = [P]
    L.lv_filter_input_bytes.restype = C.c_size_t
    L.lv_filter_input_bytes.argtypes = [P]
    L.lv_filter_whisper_bytes.restype = C.c_size_t
    L.lv_filter_whisper_bytes.argtypes = [P]
    L.lv_filter_segmentation_step.argtypes = [P, C.POINTER(C.c_uint64)]
    L.lv_filter_reset_vad_state.argtypes = [P]
    L.lv_filter_start.argtypes = [P, C.c_char_p]
    L.lv_filter_stop.argtypes = [P]
    L.lv_filter_update_model.argtypes = [P]
    L.lv_filter_load_catalogue.argtypes = [P, C.c_char_p, C.c_char_p, C.c_char_p]
    L.lv_filter_verify_model.argtypes = [P, C.c_char_p]
    L.lv_model_file_sha256.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
    L.lv_filter_register_model.argtypes = [P, C.c_char_p, C.c_char_p]
    L.lv_filter_model_state.argtypes = [P, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                        C.POINTER(C.c_int)]
    L.lv_vad_speech_timestamps.argtypes = [P, FP, C.c_size_t, C.c_int, C.POINTER(C.c_int), C.c_int]
    L.lv_remove_leading_trailing_nonalpha.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
    L.lv_filter_set_silero_model.argtypes = [P, C.c_char_p]
    L.lv_filter_vad_is_silero.argtypes = [P]
    L.lv_filter_vad_probabilities.argtypes = [P, FP, C.c_int]
    L.lv_silero_create.argtypes = [C.c_char_p, C.c_int]
    L.lv_silero_create.restype = C.c_void_p
    L.lv_silero_process.argtypes = [C.c_void_p, FP, C.c_int, C.c_int, FP]
    L.lv_silero_state.argtypes = [C.c_void_p, FP]
    L.lv_silero_process_batch.argtypes = [C.POINTER(C.c_void_p), C.POINTER(FP), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(FP), C.c_int]
    L.lv_silero_destroy.argtypes = [C.c_void_p]
    L.lv_silero_destroy.restype = None
    L.lv_silero_inspect.argtypes = [C.c_char_p, C.POINTER(C.c_double), C.c_int]
    L.lv_resample.argtypes = [P, FPP, C.c_uint32, FP, C.c_int]
    _lib = L
    return L


def _fp(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def last_error() -> str:
    return (lib().mw_last_error() or b"").decode(errors="replace")


class SileroVad:
    """The Silero VAD network on one GPU (VadIterator::predict_one over consecutive 512-sample windows)."""
    WINDOW = 512

    def __init__(self, onnx_path: str, device: int = 0):
        self.L = lib()
        self.h = self.L.lv_silero_create(onnx_path.encode(), device)
        if not self.h:
            raise RuntimeError("lv_silero_create: " + last_error())

    def process(self, wav, reset: bool = True) -> np.ndarray:
        wav = np.ascontiguousarray(wav, dtype=np.float32)
        n = len(wav) // self.WINDOW
        probs = np.zeros(max(n, 1), dtype=np.float32)
        if self.L.lv_silero_process(self.h, _fp(wav), n, int(reset), _fp(probs)) != 0:
            raise RuntimeError("lv_silero_process: " + last_error())
        return probs[:n].copy()

    def state(self) -> np.ndarray:
        out = np.zeros(256, dtype=np.float32)
        if self.L.lv_silero_state(self.h, _fp(out)) != 0:
            raise RuntimeError("lv_silero_state: " + last_error())
        return out

    def close(self):
        if self.h:
            self.L.lv_silero_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def silero_process_batch(vads, wavs, resets):
    """The windows of several SileroVad streams in one pass; returns one probability array per stream."""
    n = len(vads)
    wavs = [np.ascontiguousarray(w, dtype=np.float32) for w in wavs]
    nw = [len(w) // SileroVad.WINDOW for w in wavs]
    outs = [np.zeros(max(k, 1), dtype=np.float32) for k in nw]
    H = (C.c_void_p * n)(*[v.h for v in vads])
    FP = C.POINTER(C.c_float)
    Wp = (FP * n)(*[_fp(w) for w in wavs])
    Pp = (FP * n)(*[_fp(o) for o in outs])
    N = (C.c_int * n)(*nw)
    R = (C.c_int * n)(*[int(bool(r)) for r in resets])
    if lib().lv_silero_process_batch(H, Wp, N, R, Pp, n) != 0:
        raise RuntimeError("lv_silero_process_batch: " + last_error())
    return [o[:k].copy() for o, k in zip(outs, nw)]


def silero_inspect(onnx_path: str):
    """Host only: position-weighted sums of the 15 tensors the library's ONNX reader takes from the file."""
    sums = (C.c_double * 15)()
    n = lib().lv_silero_inspect(onnx_path.encode(), sums, 15)
    if n != 15:
        raise RuntimeError("lv_silero_inspect: " + last_error())
    return [sums[i] for i in range(15)]


class Engine:
    """One whisper_context on one GPU."""

    def __init__(self, model_path: str, precision: int = PRECISION_BF16, device: int = 0, ctx=None, share: "Engine | None" = None):
        self.L = lib()
        self._own = ctx is None
        if ctx is not None:
            self.ctx = ctx
        elif share is not None:  # one more stream context on the weights `share` already uploaded
            self.ctx = self.L.mw_init_shared(share.ctx)
        else:
            self.ctx = self.L.mw_init_from_file(model_path.encode(), device, precision)
        if not self.ctx:
            raise RuntimeError("mi355_whisper: failed to create context: " + last_error())
        hp = (C.c_int32 * 11)()
        self.L.mw_hparams(self.ctx, hp)
        (self.n_vocab, self.n_audio_ctx, self.n_audio_state, self.n_audio_head, self.n_audio_layer, self.n_text_ctx,
         self.n_text_state, self.n_text_head, self.n_text_layer, self.n_mels, self.ftype) = list(hp)
        self.eot = self.L.whisper_token_eot(self.ctx)
        self.sot = self.L.whisper_token_sot(self.ctx)
        self.beg = self.L.whisper_token_beg(self.ctx)
        self._keep = []

    def close(self):
        if self.ctx and self._own:
            self.L.whisper_free(self.ctx)
        self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {last_error()}")

    # ---- params ----
    def default_params(self, strategy: int = SAMPLING_GREEDY, **kw) -> FullParams:
        p = self.L.whisper_full_default_params(strategy)
        for k, v in kw.items():
            if k == "best_of":
                p.greedy.best_of = v
            elif k == "beam_size":
                p.beam_search.beam_size = v
            else:
                if isinstance(v, str):
                    v = v.encode()
                if isinstance(v, bytes):
                    self._keep.append(v)
                setattr(p, k, v)
        return p

    # ---- stages ----
    def mel(self, pcm: np.ndarray) -> np.ndarray:
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        self._check(self.L.mw_pcm_to_mel(self.ctx, _fp(pcm), len(pcm)), "mw_pcm_to_mel")
        out = np.empty((self.n_mels, self.L.mw_mel_n_len(self.ctx)), np.float32)
        self._check(self.L.mw_get_mel(self.ctx, _fp(out)), "mw_get_mel")
        return out

    def encode(self, seek: int = 0, audio_ctx: int = 0) -> np.ndarray:
        self._check(self.L.mw_encode(self.ctx, seek, audio_ctx), "mw_encode")
        n_ctx = audio_ctx if audio_ctx > 0 else self.n_audio_ctx
        out = np.empty((n_ctx, self.n_audio_state), np.float32)
        self._check(self.L.mw_get_encoder_out(self.ctx, _fp(out)), "mw_get_encoder_out")
        return out

    def cross_kv(self, layer: int, is_v: bool, n_ctx: int | None = None) -> np.ndarray:
        out = np.empty((n_ctx or self.n_audio_ctx, self.n_text_state), np.float32)
        self._check(self.L.mw_get_cross_kv(self.ctx, layer, int(is_v), _fp(out)), "mw_get_cross_kv")
        return out

    def kv_clear(self):
        self._check(self.L.mw_kv_clear(self.ctx), "mw_kv_clear")

    def decode(self, tokens, n_past: int = 0, seq: int = 0) -> np.ndarray:
        t = np.ascontiguousarray(tokens, dtype=np.int32)
        self._check(self.L.mw_decode(self.ctx, t.ctypes.data_as(C.POINTER(C.c_int32)), len(t), n_past, seq), "mw_decode")
        out = np.empty((min(len(t), 32), self.n_vocab), np.float32)
        self._check(self.L.mw_get_logits(self.ctx, _fp(out)), "mw_get_logits")
        return out

    def decode_rows(self, tokens, pos, seq) -> np.ndarray:
        """One step over len(tokens) rows of different sequences (beam-step shape); logits of every row."""
        t, p, s = (np.ascontiguousarray(x, dtype=np.int32) for x in (tokens, pos, seq))
        ip = C.POINTER(C.c_int32)
        self._check(self.L.mw_decode_rows(self.ctx, t.ctypes.data_as(ip), p.ctypes.data_as(ip), s.ctypes.data_as(ip), len(t)),
                    "mw_decode_rows")
        out = np.empty((len(t), self.n_vocab), np.float32)
        self._check(self.L.mw_get_logits(self.ctx, _fp(out)), "mw_get_logits")
        return out

    def process_logits(self, params: FullParams, logits, history, temperature=0.0, has_ts=False, seek_delta=3000,
                       n_draws=0, u=None):
        lg = np.ascontiguousarray(logits, dtype=np.float32)
        h = np.ascontiguousarray(history, dtype=np.int32)
        o1 = np.empty(self.n_vocab, np.float32)
        o2 = np.empty_like(o1)
        o3 = np.empty_like(o1)
        toks = (TokenData * max(1, n_draws))()
        uu = np.ascontiguousarray(u if u is not None else np.zeros(max(1, n_draws)), dtype=np.float64)
        self._check(self.L.mw_process_logits(self.ctx, C.byref(params), float(temperature), _fp(lg),
                                             h.ctypes.data_as(C.POINTER(C.c_int32)), len(h), int(has_ts), int(seek_delta),
                                             n_draws, uu.ctypes.data_as(C.POINTER(C.c_double)), _fp(o1), _fp(o2), _fp(o3),
                                             toks), "mw_process_logits")
        return o1, o2, o3, [dict(id=t.id, tid=t.tid, p=t.p, plog=t.plog, pt=t.pt, ptsum=t.ptsum) for t in toks]

    # ---- whisper_full ----
    def full(self, params: FullParams, pcm: np.ndarray) -> int:
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        return self.L.whisper_full(self.ctx, params, _fp(pcm), len(pcm))

    def full_device(self, params: FullParams, pcm_device_ptr: int, n_samples: int) -> int:
        """whisper_full with the PCM already in HBM (pcm_device_ptr: e.g. torch tensor .data_ptr())."""
        return self.L.mw_full_device(self.ctx, params, C.c_void_p(pcm_device_ptr), n_samples)

    def segments(self):
        out = []
        for i in range(self.L.whisper_full_n_segments(self.ctx)):
            toks = []
            for j in range(self.L.whisper_full_n_tokens(self.ctx, i)):
                td = self.L.whisper_full_get_token_data(self.ctx, i, j)
                toks.append(dict(id=td.id, tid=td.tid, p=td.p, plog=td.plog, pt=td.pt, ptsum=td.ptsum))
            out.append(dict(t0=self.L.whisper_full_get_segment_t0(self.ctx, i),
                            t1=self.L.whisper_full_get_segment_t1(self.ctx, i),
                            text=self.L.whisper_full_get_segment_text(self.ctx, i), tokens=toks))
        return out

    def token_ids(self):
        return [t["id"] for s in self.segments() for t in s["tokens"]]

    def token_to_str(self, i: int) -> bytes:
        return self.L.whisper_token_to_str(self.ctx, i)

    def lang_auto_detect(self, offset_ms: int = 0):
        probs = np.zeros(100, np.float32)
        return self.L.whisper_lang_auto_detect(self.ctx, offset_ms, 1, _fp(probs)), probs

    def set_force_decode_steps(self, n: int):
        self._check(self.L.mw_set_force_decode_steps(self.ctx, n), "mw_set_force_decode_steps")

    def hub_stats(self):
        """(steps, rows, requests, encoder passes, encoder windows) of the batcher shared with the other contexts on
        these weights, or None."""
        out = (C.c_ulonglong * 5)()
        rc = self.L.mw_hub_stats(self.ctx, out)
        return None if rc != 0 else tuple(int(x) for x in out)

    def timings(self) -> Timings:
        t = Timings()
        self.L.mw_get_timings(self.ctx, C.byref(t))
        return t

    def reset_timings(self):
        self.L.mw_reset_timings(self.ctx)


class Filter:
    """LocalVocal-side mirror (include/localvocal_hotpath.h): transcription_filter_data + run_whisper_inference."""

    def __init__(self, device: int = 0, precision: int = -1):
        self.L = lib()
        self.f = self.L.lv_filter_create()
        if not self.f:
            raise RuntimeError("lv_filter_create failed")
        self.L.lv_filter_set_gpu_device(self.f, device, precision)

    def close(self):
        if self.f:
            self.L.lv_filter_destroy(self.f)
            self.f = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def params(self) -> FullParams:
        return self.L.lv_filter_params(self.f).contents

    def load_model(self, path: str) -> int:
        return self.L.lv_filter_load_model(self.f, path.encode())

    def engine(self) -> Engine:
        return Engine("", ctx=self.L.lv_filter_context(self.f))

    def run_whisper_inference(self, pcm, t0=0, t1=0, vad_state=1):
        out = LvDetection()
        if pcm is None:
            rc = self.L.lv_run_whisper_inference(self.f, None, 0, t0, t1, vad_state, C.byref(out))
        else:
            pcm = np.ascontiguousarray(pcm, dtype=np.float32)
            rc = self.L.lv_run_whisper_inference(self.f, _fp(pcm), len(pcm), t0, t1, vad_state, C.byref(out))
        if rc != 0:
            raise RuntimeError("lv_run_whisper_inference failed")
        toks = [dict(id=out.tokens[i].id, p=out.tokens[i].p) for i in range(out.n_tokens)]
        return dict(result=out.result, text=out.text or b"", start=out.start_timestamp_ms, end=out.end_timestamp_ms,
                    tokens=toks, language=(out.language or b"").decode())

    # ---- callers of the path: settings, audio tap, segmentation, worker (localvocal_hotpath.h, second half) ----
    def set_setting(self, key: str, value):
        k = key.encode()
        if isinstance(value, bool):
            self.L.lv_settings_set_bool(self.f, k, int(value))
        elif isinstance(value, int):
            self.L.lv_settings_set_int(self.f, k, value)
        elif isinstance(value, float):
            self.L.lv_settings_set_double(self.f, k, value)
        else:
            self.L.lv_settings_set_string(self.f, k, str(value).encode())

    def apply_settings(self) -> int:
        return self.L.lv_filter_apply_settings(self.f)

    def setup_audio(self, sample_rate: int, channels: int) -> int:
        self.channels = channels
        return self.L.lv_filter_setup_audio(self.f, sample_rate, channels)

    def set_callbacks(self, on_text=None, on_chunk=None, on_request=None):
        """Python callables receiving plain dicts / numpy copies; kept alive on the instance."""
        def det(d):
            d = d.contents
            return dict(result=d.result, text=(d.text or b""), start=d.start_timestamp_ms, end=d.end_timestamp_ms,
                        n_tokens=d.n_tokens, language=(d.language or b"").decode(),
                        tokens=[(d.tokens[i].id, d.tokens[i].p) for i in range(d.n_tokens)])
        self._cb_text = LV_TEXT_CB((lambda u, d: on_text(det(d))) if on_text else 0)
        self._cb_chunk = LV_CHUNK_CB((lambda u, pcm, n, vs, d: on_chunk(np.ctypeslib.as_array(pcm, (n,)).copy(), vs, det(d)))
                                     if on_chunk else 0)
        self._cb_req = LV_REQUEST_CB((lambda u, pcm, n, t0, t1, vs: on_request(np.ctypeslib.as_array(pcm, (n,)).copy(), t0, t1, vs))
                                     if on_request else 0)
        self.L.lv_filter_set_callbacks(self.f, self._cb_text, self._cb_chunk, self._cb_req, None)

    def _planar(self, data):
        chans = [np.ascontiguousarray(c, dtype=np.float32) for c in data]
        arr = (C.POINTER(C.c_float) * len(chans))(*[_fp(c) for c in chans])
        return chans, arr

    def push_packet(self, data, timestamp_offset_ns: int) -> int:
        """data: list of per-channel float arrays, or an int frame count for silence."""
        if isinstance(data, int):
            return self.L.lv_filter_push_packet(self.f, None, data, timestamp_offset_ns)
        chans, arr = self._planar(data)
        return self.L.lv_filter_push_packet(self.f, arr, len(chans[0]), timestamp_offset_ns)

    def filter_audio(self, data) -> int:
        chans, arr = self._planar(data)
        return self.L.lv_filter_filter_audio(self.f, arr, len(chans[0]))

    def segmentation_step(self):
        st = (C.c_uint64 * 4)()
        rc = self.L.lv_filter_segmentation_step(self.f, st)
        if rc != 0:
            raise RuntimeError(f"lv_filter_segmentation_step failed: {rc}")
        return dict(vad_on=bool(st[0]), start=st[1], end=st[2], last_partial_end=st[3])

    def vad_speech_timestamps(self, pcm, reset_state=True, cap=256):
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        pairs = (C.c_int * (2 * cap))()
        n = self.L.lv_vad_speech_timestamps(self.f, _fp(pcm), len(pcm), int(reset_state), pairs, cap)
        if n < 0:
            raise RuntimeError("lv_vad_speech_timestamps failed")
        return [(pairs[2 * i], pairs[2 * i + 1]) for i in range(min(n, cap))]

    def set_silero_model(self, path) -> int:
        """0: the filter's VAD is the Silero network on the GPU; 1: no such file (energy gate); raises on a malformed file."""
        rc = self.L.lv_filter_set_silero_model(self.f, (path or "").encode())
        if rc < 0:
            raise RuntimeError("lv_filter_set_silero_model: " + last_error())
        return rc

    def vad_probabilities(self):
        n = self.L.lv_filter_vad_probabilities(self.f, None, 0)
        out = np.zeros(max(n, 1), dtype=np.float32)
        self.L.lv_filter_vad_probabilities(self.f, _fp(out), n)
        return out[:max(n, 0)].copy()

    def resample(self, data):
        chans, arr = self._planar(data)
        out = np.zeros(len(chans[0]) * 2 + 64, dtype=np.float32)
        n = self.L.lv_resample(self.f, arr, len(chans[0]), _fp(out), len(out))
        if n < 0:
            raise RuntimeError("lv_resample failed")
        return out[:n].copy()

    def start(self, model_path: str) -> int:
        return self.L.lv_filter_start(self.f, model_path.encode())

    def stop(self):
        self.L.lv_filter_stop(self.f)

    def register_model(self, name: str, file: str | None):
        """models_info() entry + find_model_bin_file result for update_model (None: listed, not on disk)."""
        return self.L.lv_filter_register_model(self.f, name.encode(), file.encode() if file else None)

    def load_catalogue(self, json_path: str, data_models_dir: str | None = None, config_models_dir: str | None = None) -> int:
        """load_models_info + find_model_bin_file over a models_directory.json: registers every transcription model."""
        enc = lambda v: v.encode() if v else None  # noqa: E731
        return self.L.lv_filter_load_catalogue(self.f, json_path.encode(), enc(data_models_dir), enc(config_models_dir))

    def verify_model(self, friendly_name: str) -> int:
        """0: the file on disk has the catalogue's sha256; 1: it differs; 2: nothing to compare."""
        return self.L.lv_filter_verify_model(self.f, friendly_name.encode())

    def update_model(self) -> int:
        """update_whisper_model(gf): 0 = a context is loaded afterwards, 1 = the filter is left stopped."""
        return self.L.lv_filter_update_model(self.f)

    def model_state(self) -> dict:
        a, b = C.create_string_buffer(1024), C.create_string_buffer(1024)
        new, run, dtw = C.c_int(0), C.c_int(0), C.c_int(0)
        rc = self.L.lv_filter_model_state(self.f, a, len(a), b, len(b), C.byref(new), C.byref(run), C.byref(dtw))
        if rc != 0:
            raise RuntimeError("lv_filter_model_state failed")
        return dict(model_path=a.value.decode(), file_loaded=b.value.decode(), loaded_new=bool(new.value),
                    worker_running=bool(run.value), dtw=bool(dtw.value), context=bool(self.L.lv_filter_context(self.f)))


def model_file_sha256(path: str) -> str:
    buf = C.create_string_buffer(80)
    if lib().lv_model_file_sha256(path.encode(), buf, len(buf)) != 0:
        raise RuntimeError("lv_model_file_sha256 failed")
    return buf.value.decode()


def remove_leading_trailing_nonalpha(s: bytes) -> bytes:
    buf = C.create_string_buffer(len(s) + 8)
    lib().lv_remove_leading_trailing_nonalpha(s, buf, len(buf))
    return buf.value



MXE_BF16, MXE_F32_RES, MXE_MX_GELU = 0, 1, 2


def mx_gemm(A, W, bias=None, R=None, epilogue=MXE_BF16, device=0):
    """The MX-fp8 product of MW_PRECISION_FP8_MX on host arrays (mw_mx_gemm): returns (C [M][N] fp32, A decoded, W decoded)."""
    L = lib()
    A = np.ascontiguousarray(A, dtype=np.float32)
    W = np.ascontiguousarray(W, dtype=np.float32)
    M, K = A.shape
    N = W.shape[0]
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None  # noqa: E731
    b = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32)
    r = None if R is None else np.ascontiguousarray(R, dtype=np.float32)
    out, ad, wd = np.empty((M, N), np.float32), np.empty((M, K), np.float32), np.empty((N, K), np.float32)
    rc = L.mw_mx_gemm(device, fp(A), M, K, fp(W), N, fp(b), fp(r), epilogue, fp(out), fp(ad), fp(wd))
    if rc != 0:
        raise RuntimeError(f"mw_mx_gemm failed ({rc}): {last_error()}")
    return out, ad, wd


def mx_layernorm(x, w, b, with_bf16=True, device=0):
    """LayerNorm with MX-fp8 rows out (mw_mx_layernorm): returns (decoded MX rows, bf16 rows or None)."""
    L = lib()
    x = np.ascontiguousarray(x, dtype=np.float32)
    w = np.ascontiguousarray(w, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    rows, d = x.shape
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None  # noqa: E731
    out = np.empty((rows, d), np.float32)
    hb = np.empty((rows, d), np.float32) if with_bf16 else None
    rc = L.mw_mx_layernorm(device, fp(x), fp(w), fp(b), rows, d, fp(out), fp(hb))
    if rc != 0:
        raise RuntimeError(f"mw_mx_layernorm failed ({rc}): {last_error()}")
    return out, hb
