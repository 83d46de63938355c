#!/usr/bin/env python3
"""Diagnostic behind DESIGN.md section 4.1 "mel beside other contexts": which elements of the log-mel differ from the solo
result when other contexts encode beside it, for a library build named by MI355_WHISPER_LIB (csrc/Makefile VARIANT=...).
Same set-up as tests/test_gpu_configs.py::test_mel_is_bit_exact_while_other_contexts_encode; prints (frame, bin, got, want)
of the differing elements and a histogram over the frame's place in its 16-frame workgroup and over the bins."""
import collections
import os
import sys
import tempfile
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import ggml_io  # noqa: E402
from tools.offline_test import load_package  # noqa: E402


def main():
    pkg = load_package()
    d = tempfile.mkdtemp()
    path = os.path.join(d, "m.bin")
    ggml_io.make_model_file(path, "tiny.en", 1234, 1, n_audio_layer=2, n_text_layer=2, n_audio_state=512, n_audio_head=8,
                            n_text_state=512, n_text_head=8)
    n = 8
    clips = [ggml_io.synth_pcm(400 + i, 6.0 + 0.5 * (i % 3)) for i in range(n)]
    first = pkg.Engine(path, precision=pkg.PRECISION_BF16)
    engs = [first] + [pkg.Engine(path, precision=pkg.PRECISION_BF16, share=first) for _ in range(n - 1)]
    for i in range(n):
        engs[i].mel(clips[i])
        engs[i].encode(0)
    solo = [engs[i].mel(clips[i]).copy() for i in range(4)]
    stop = [False]
    diffs = []

    def noisy(i):
        while not stop[0]:
            engs[i].encode(0)

    def checker(i):
        for r in range(60):
            got = engs[i].mel(clips[i])
            if not np.array_equal(got, solo[i]):
                bins, frames = np.nonzero(got != solo[i])      # getter layout: [bin][frame]
                diffs.append((i, r, [(int(f), int(b), float(got[b, f]), float(solo[i][b, f])) for b, f in zip(bins, frames)]))

    tn = [threading.Thread(target=noisy, args=(i,)) for i in range(4, n)]
    tc = [threading.Thread(target=checker, args=(i,)) for i in range(4)]
    for t in tn + tc:
        t.start()
    for t in tc:
        t.join()
    stop[0] = True
    for t in tn:
        t.join()
    print(f"{os.environ.get('MI355_WHISPER_LIB', 'default library')}: {len(diffs)} of 240 mel calls differ")
    by_place, by_bin, n_el = collections.Counter(), collections.Counter(), 0
    for i, r, els in diffs:
        fr = sorted({f for f, _, _, _ in els})
        print(f"  stream {i} call {r}: {len(els)} elements in frames {fr[:12]}{'...' if len(fr) > 12 else ''}")
        for f, b, g, w in els[:6]:
            print(f"     frame {f} (block {f // 16}, place {f % 16}: wave {f % 16 // 4}, pass {f % 4}) bin {b}: {g:.7f} != {w:.7f}")
        for f, b, _, _ in els:
            by_place[f % 16] += 1
            by_bin[b] += 1
            n_el += 1
    print("  elements by place in the 16-frame workgroup:", dict(sorted(by_place.items())))
    print("  elements by mel bin:", dict(sorted(by_bin.items())))
    for e in engs[1:] + engs[:1]:
        e.close()


if __name__ == "__main__":
    main()
