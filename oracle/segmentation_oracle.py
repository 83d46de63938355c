"""CPU restatement of LocalVocal's segmentation driver — TEST INFRASTRUCTURE ONLY.

Only tests/ may import this module (it is the checker for obs-localvocal_amd/csrc/segmentation.hip, never
the product).  It restates, in plain Python, the control flow of

  get_data_from_buf_and_resample   /root/reference/src/whisper-utils/vad-processing.cpp:24-123
  vad_disabled_segmentation        /root/reference/src/whisper-utils/vad-processing.cpp:125-200
  vad_based_segmentation           /root/reference/src/whisper-utils/vad-processing.cpp:202-367
  hybrid_vad_segmentation          /root/reference/src/whisper-utils/vad-processing.cpp:369-454
  run_inference_and_callbacks      /root/reference/src/whisper-utils/whisper-processing.cpp:326-359 (buffer handling)
  VadIterator::{reset_states,predict,process}  /root/reference/src/whisper-utils/silero-vad-onnx.cpp:95-109,142-284

for a 16 kHz mono input (the resampler is then the identity), with the Silero ONNX probability replaced
by the same energy gate the build uses (ONNX Runtime and the Silero model cannot run in this image).

PARITY UNPINNED: the reference holds no golden vectors for its segmentation; this file is
a second, independent reading of the same source text, so agreement between it and the C++ driver shows
that two restatements agree, not that either equals a LocalVocal build.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Tuple

import numpy as np

SR = 16000
VAD_STATE_WAS_ON, VAD_STATE_WAS_OFF, VAD_STATE_IS_OFF, VAD_STATE_PARTIAL = 0, 1, 2, 3
VAD_MODE_ACTIVE, VAD_MODE_HYBRID, VAD_MODE_DISABLED = 0, 1, 2


def energy_gate_prob(window: np.ndarray, gate_db: float = -40.0, slope_db: float = 3.0) -> float:
    e = float(np.dot(window.astype(np.float64), window.astype(np.float64)))
    db = 10.0 * math.log10(e / max(len(window), 1) + 1e-12)
    return float(np.float32(1.0 / (1.0 + math.exp(-(db - gate_db) / slope_db))))


class VadIteratorOracle:
    """silero-vad-onnx.cpp:326-353 (constructor arithmetic) + :95-109 + :142-284."""

    def __init__(self, threshold=0.5, sample_rate=16000, window_ms=32, min_silence_ms=100, speech_pad_ms=100,
                 min_speech_ms=100, max_speech_s=float("inf"), gate_db=-40.0, slope_db=3.0, prob_source=None):
        # prob_source: an object with process(wav, reset) -> probabilities of wav's consecutive windows (the Silero
        # restatement, oracle/silero_vad_oracle.py: recurrent state carried between calls unless reset); None: energy gate
        self.prob_source = prob_source
        self._was_reset = True
        self.threshold = float(np.float32(threshold))
        sr_per_ms = sample_rate // 1000
        self.W = window_ms * sr_per_ms
        self.min_speech_samples = sr_per_ms * min_speech_ms
        self.speech_pad_samples = sr_per_ms * speech_pad_ms
        self.max_speech_samples = sample_rate * max_speech_s - self.W - 2.0 * self.speech_pad_samples
        self.min_silence_samples = sr_per_ms * min_silence_ms
        self.min_silence_samples_at_max_speech = sr_per_ms * 98
        self.gate = (gate_db, slope_db)
        self.triggered = False
        self.temp_end = 0
        self.current_sample = 0
        self.prev_end = 0
        self.next_start = 0
        self.speeches: List[Tuple[int, int]] = []
        self.cur_start = -1

    def _reset(self, reset_state: bool):
        if reset_state:
            self.triggered = False
            self._was_reset = True
        self.temp_end = 0
        self.current_sample = 0
        self.prev_end = self.next_start = 0
        self.speeches = []
        self.cur_start = -1

    def _predict(self, p: float):
        self.current_sample += self.W
        if p >= self.threshold:                                     # :150-169
            if self.temp_end != 0:
                self.temp_end = 0
                if self.next_start < self.prev_end:
                    self.next_start = self.current_sample - self.W
            if not self.triggered:
                self.triggered = True
                self.cur_start = self.current_sample - self.W
            return
        # :171-198 (unsigned subtraction in the reference: start == -1 wraps; only reachable while triggered)
        if self.triggered and float((self.current_sample - self.cur_start) & 0xFFFFFFFF) > self.max_speech_samples:
            if self.prev_end > 0:
                self.speeches.append((self.cur_start, self.prev_end))
                self.cur_start = -1
                if self.next_start < self.prev_end:
                    self.triggered = False
                else:
                    self.cur_start = self.next_start
            else:
                self.speeches.append((self.cur_start, self.current_sample))
                self.cur_start = -1
                self.triggered = False
            self.prev_end = self.next_start = 0
            self.temp_end = 0
            return
        if p >= self.threshold - 0.15:                              # :199-220
            return
        if self.triggered:                                          # :223-259
            if self.temp_end == 0:
                self.temp_end = self.current_sample
            if self.current_sample - self.temp_end > self.min_silence_samples_at_max_speech:
                self.prev_end = self.temp_end
            if self.current_sample - self.temp_end >= self.min_silence_samples:
                end = self.temp_end
                if end - self.cur_start > self.min_speech_samples:
                    self.speeches.append((self.cur_start, end))
                    self.cur_start = -1
                    self.prev_end = self.next_start = 0
                    self.temp_end = 0
                    self.triggered = False
                else:
                    self.cur_end_pending = end  # current_speech.end is set but the speech stays open

    def process(self, wav: np.ndarray, reset_state: bool = True):   # :262-284
        self._reset(reset_state)
        n = len(wav)
        nw = n // self.W
        if self.prob_source is not None:
            probs = self.prob_source.process(np.asarray(wav[:nw * self.W], np.float32), self._was_reset)
            if nw > 0:
                self._was_reset = False
        else:
            probs = [energy_gate_prob(wav[j * self.W:(j + 1) * self.W], *self.gate) for j in range(nw)]
        self.probs = np.asarray(probs, np.float32)
        for j in range(nw):
            self._predict(float(probs[j]))
        if self.cur_start >= 0:
            self.speeches.append((self.cur_start, n))
            self.cur_start = -1
            self.prev_end = self.next_start = 0
            self.temp_end = 0
            self.triggered = False
        return list(self.speeches)


@dataclass
class VadState:
    vad_on: bool = False
    start: int = 0
    end: int = 0
    last_partial_end: int = 0

    def as_tuple(self):
        return (self.vad_on, self.start, self.end, self.last_partial_end)


@dataclass
class Request:
    pcm: np.ndarray
    t0: int
    t1: int
    vad_state: int


@dataclass
class SegmenterOracle:
    """State of one filter: the byte rings of the reference become numpy arrays of floats."""
    vad_mode: int = VAD_MODE_ACTIVE
    partial_transcription: bool = False
    partial_latency: int = 1100
    segment_duration: int = 7000
    vad: VadIteratorOracle = field(default_factory=VadIteratorOracle)
    sample_rate: int = SR
    input: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float32))
    infos: List[Tuple[int, int]] = field(default_factory=list)         # (frames, timestamp_offset_ns)
    resampled: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float32))
    whisper: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float32))
    state: VadState = field(default_factory=VadState)
    requests: List[Request] = field(default_factory=list)

    # transcription-filter.cpp:102-116 / localvocal-offline-test.cpp:543-558
    def push_packet(self, pcm: np.ndarray, ts_ns: int):
        self.input = np.concatenate([self.input, pcm.astype(np.float32)])
        self.infos.append((len(pcm), ts_ns))

    # vad-processing.cpp:24-123
    def _get_data(self):
        if len(self.input) == 0:
            return None
        max_frames = self.sample_rate * 10
        n = 0
        start_ns = 0
        info = (0, 0)
        while self.infos:
            info = self.infos.pop(0)
            n += info[0]
            if start_ns == 0:
                start_ns = info[1]
            if n > max_frames:
                n -= info[0]
                self.infos.insert(0, info)
                break
        end_ns = info[1] + info[0] * 1000000000 // self.sample_rate
        if start_ns > end_ns:
            start_ns = end_ns - n * 1000000000 // self.sample_rate
        chunk, self.input = self.input[:n], self.input[n:]
        self.resampled = np.concatenate([self.resampled, chunk])          # 16 kHz mono: identity resample
        return start_ns, end_ns

    # whisper-processing.cpp:326-348
    def _run_inference(self, t0: int, t1: int, vad_state: int):
        margin = SR // 100
        pcm = np.concatenate([np.zeros(margin, np.float32), self.whisper, np.zeros(margin, np.float32)])
        if vad_state != VAD_STATE_PARTIAL:
            self.whisper = np.zeros(0, np.float32)
        self.requests.append(Request(pcm, t0, t1, vad_state))

    def step(self) -> VadState:
        if self.vad_mode == VAD_MODE_HYBRID:
            self.state = self._hybrid(self.state)
        elif self.vad_mode == VAD_MODE_ACTIVE:
            self.state = self._active(self.state)
        else:
            self.state = self._disabled(self.state)
        return self.state

    # vad-processing.cpp:125-200
    def _disabled(self, last: VadState) -> VadState:
        got = self._get_data()
        if got is None:
            if len(self.whisper) > 0:
                self._run_inference(last.start, last.end, VAD_STATE_WAS_OFF)
            return last
        _, end_ns = got
        self.whisper = np.concatenate([self.whisper, self.resampled])
        self.resampled = np.zeros(0, np.float32)
        is_partial = len(self.whisper) < self.segment_duration * SR // 1000
        end_ms = end_ns // 1000000
        if is_partial:
            last = VadState(last.vad_on, last.start, last.end, last.last_partial_end)
            unprocessed = (end_ms - last.last_partial_end) & 0xFFFFFFFFFFFFFFFF
            if unprocessed > self.partial_latency:
                if self.partial_transcription:
                    self._run_inference(last.start, end_ms, VAD_STATE_PARTIAL)
                last.last_partial_end = end_ms
            return VadState(False, last.start, end_ms, last.last_partial_end)
        self._run_inference(last.start, end_ms, VAD_STATE_WAS_OFF)
        return VadState(False, end_ms, end_ms, end_ms)

    # vad-processing.cpp:202-367
    def _active(self, last: VadState) -> VadState:
        got = self._get_data()
        if got is None:
            return last
        start_ns, end_ns = got
        W = self.vad.W
        if len(self.resampled) < W * 8:
            return last
        n_windows = len(self.resampled) // W
        vad_input, self.resampled = self.resampled[:n_windows * W], self.resampled[n_windows * W:]
        stamps = self.vad.process(vad_input, not last.vad_on)
        start_ms, end_ms = start_ns // 1000000, end_ns // 1000000
        cur = VadState(False, start_ms, end_ms, last.last_partial_end)
        if not stamps:
            if last.vad_on:
                self._run_inference(last.start, last.end, VAD_STATE_WAS_ON)
                cur.last_partial_end = 0
            return cur
        last = VadState(*last.as_tuple())
        for i, (s, e) in enumerate(stamps):
            start_frame = stamps[i - 1][1] if i > 0 else max(0, s - SR // 10)
            end_frame = e
            self.whisper = np.concatenate([self.whisper, vad_input[start_frame:end_frame]])
            if e < len(vad_input):
                segment_end_ts = start_ms + end_frame * 1000 // SR
                self._run_inference(last.start, segment_end_ts, VAD_STATE_WAS_ON if last.vad_on else VAD_STATE_WAS_OFF)
                cur.vad_on = False
                cur.start = cur.end
                cur.end = 0
                cur.last_partial_end = 0
                last = VadState(*cur.as_tuple())
                continue
            cur.vad_on = True
            if last.vad_on:
                cur.start = last.start
            else:
                cur.start = start_ms + start_frame * 1000 // SR
            cur.end = start_ms + end_frame * 1000 // SR
            last = VadState(*cur.as_tuple())
            if not self.partial_transcription:
                continue
            length = ((cur.end if cur.end > 0 else cur.start) -
                      (cur.last_partial_end if cur.last_partial_end > 0 else cur.start)) & 0xFFFFFFFFFFFFFFFF
            if length > self.partial_latency:
                cur.last_partial_end = cur.end
                self._run_inference(cur.start, cur.end, VAD_STATE_PARTIAL)
        return cur

    # vad-processing.cpp:369-454
    def _hybrid(self, last: VadState) -> VadState:
        got = self._get_data()
        if got is None:
            return last
        _, end_ns = got
        last = VadState(*last.as_tuple())
        last.end = end_ns // 1000000
        self.whisper = np.concatenate([self.whisper, self.resampled])
        self.resampled = np.zeros(0, np.float32)
        if ((last.end - last.start) & 0xFFFFFFFFFFFFFFFF) >= self.segment_duration:
            self._run_inference(last.start, last.end, VAD_STATE_WAS_ON)
            last.start = end_ns // 1000000
            last.last_partial_end = 0
            return last
        if self.partial_transcription:
            length = ((last.end if last.end > 0 else last.start) -
                      (last.last_partial_end if last.last_partial_end > 0 else last.start)) & 0xFFFFFFFFFFFFFFFF
            if length > self.partial_latency:
                last.last_partial_end = last.end
                stamps = self.vad.process(self.whisper.copy(), True)
                if stamps:
                    self._run_inference(last.start, last.end, VAD_STATE_PARTIAL)
                else:
                    keep = SR // 4
                    if len(self.whisper) > keep:   # the reference underflows below `keep`; both restatements guard it
                        self.whisper = self.whisper[len(self.whisper) - keep:]
        return last
