"""Development aid: time the Silero VAD pass (csrc/silero_vad.hip) on buffers of the sizes vad_based_segmentation sends
(new audio since the last step: 0.1 - 10 s), one stream and 64 streams, on synthetic weights (the deviation from the CPU
restatement is what tests/test_gpu_silero.py checks and prints)."""
import importlib
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import silero_synth as so  # noqa: E402

pkg = importlib.import_module("obs-localvocal_amd")


def main():
    w = so.synth_weights(11)
    path = os.path.join(tempfile.mkdtemp(), "silero_synth.onnx")
    so.write_onnx(path, w)
    vad = pkg.SileroVad(path)
    rng = np.random.default_rng(0)
    x = (0.05 * rng.standard_normal(16000 * 10)).astype(np.float32)
    x[16000 * 3:16000 * 5] *= 0.01
    vad.process(x)
    for seconds in (0.1, 0.5, 1.0, 3.0, 10.0):
        n = int(seconds * 16000)
        for _ in range(3):
            vad.process(x[:n])
        t0 = time.time()
        reps = 20
        for _ in range(reps):
            vad.process(x[:n])
        dt = (time.time() - t0) / reps
        print(f"{seconds:5.1f} s of audio ({n // 512:3d} windows): {dt * 1e3:.3f} ms per call from host PCM = {seconds / dt:,.0f} x real time")
    vad.close()
    # 64 streams, one second of new audio each (what 64 sources hand in per second): one call each vs one batched pass
    vads = [pkg.SileroVad(path) for _ in range(64)]
    wavs = [x[i * 1000:i * 1000 + 16000] for i in range(64)]
    for mode in ("one call per stream", "one batched pass"):
        for rep in range(4):
            if rep == 1:
                t0 = time.time()
            if mode.startswith("one call"):
                for v, w in zip(vads, wavs):
                    v.process(w)
            else:
                pkg.silero_process_batch(vads, wavs, [True] * 64)
        print(f"64 streams x 1 s, {mode}: {(time.time() - t0) / 3 * 1e3:.2f} ms")
    for v in vads:
        v.close()


if __name__ == "__main__":
    main()
