#!/usr/bin/env python3
"""Development tool: encoder passes of one 30 s window for `rocprofv3 --kernel-trace` (run: rocprofv3 --kernel-trace
--output-format csv -d DIR -- python3 tools/encoder_profile.py), and, with --summarise CSV, the per-(kernel, grid)
duration table of that trace (GEMM shapes share a kernel name and differ in the grid)."""
import collections
import csv
import ctypes as C
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def summarise(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        n = re.sub(r"\(.*", "", r["Kernel_Name"].replace("mw::(anonymous namespace)::", "").replace("void ", ""))
        key = (n[:60], r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?"), r.get("Grid_Size_Z", "?"), r.get("Workgroup_Size_X", "?"))
        agg[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    tot = sum(sum(v) for v in agg.values())
    print(f"{'kernel':60s} {'grid(threads)':>22s} {'calls':>6s} {'avg us':>9s} {'min us':>9s} {'share':>6s}")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print(f"{k[0]:60s} {k[1] + 'x' + k[2] + 'x' + k[3]:>22s} {len(v):6d} {sum(v) / len(v) / 1e3:9.2f} {min(v) / 1e3:9.2f} {100.0 * sum(v) / tot:6.2f}")


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
        return summarise(sys.argv[2])
    import importlib.util
    import numpy as np
    spec = importlib.util.spec_from_file_location("obs_localvocal_amd", os.path.join(ROOT, "obs-localvocal_amd", "__init__.py"))
    pkg = importlib.util.module_from_spec(spec)
    sys.modules["obs_localvocal_amd"] = pkg
    spec.loader.exec_module(pkg)
    from tools import ggml_io, synth_large
    model = sys.argv[1] if len(sys.argv) > 1 else "large-v3"
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    windows = int(sys.argv[3]) if len(sys.argv) > 3 else 1   # > 1: that many contexts on one upload encode at the same time
    pname = sys.argv[4] if len(sys.argv) > 4 else "bf16"     # bf16 | fp8 | fp8mx (the encoder's Q|K, fc1, fc2 in MX-fp8)
    prec = {"bf16": pkg.PRECISION_BF16, "fp8": pkg.PRECISION_FP8, "fp8mx": pkg.PRECISION_FP8_MX}[pname]
    d = os.environ.get("MI355_WHISPER_BENCH_DIR", "/tmp/mi355_whisper_bench")
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, f"ggml-{model}-synth-s1234.bin")
    if not os.path.exists(path):
        synth_large.write_large(path, model, seed=1234, quiet_special=True)
    eng = pkg.Engine(path, precision=prec)
    if windows <= 1:
        eng.mel(ggml_io.synth_pcm(100, 30.0))
        ms = C.c_float(0)
        assert eng.L.mw_bench_encode(eng.ctx, iters, C.byref(ms)) == 0
        print(f"encoder + cross-K/V of one 30 s window ({model}, {pname}): {ms.value:.3f} ms per pass ({iters} passes)")
        if prec != pkg.PRECISION_BF16 and os.environ.get("ENC_COMPARE"):
            # what the precision costs: this encoder's output against the bf16 engine's on the same file and audio
            out = eng.encode(0)
            ref_eng = pkg.Engine(path, precision=pkg.PRECISION_BF16)
            ref_eng.mel(ggml_io.synth_pcm(100, 30.0))
            ref = ref_eng.encode(0)
            ref_eng.close()
            print(f"   encoder output vs the bf16 engine: max |d| / max |bf16| = {np.abs(out - ref).max() / np.abs(ref).max():.4f}, "
                  f"rms |d| / rms = {np.sqrt(np.mean((out - ref) ** 2)) / np.sqrt(np.mean(ref ** 2)):.4f}, finite: {bool(np.isfinite(out).all())}")
    else:
        # the windows of `windows` streams through the hub: their encode calls meet in one pass with M = windows x 1500
        import threading
        import time
        engs = [eng] + [pkg.Engine(path, precision=prec, share=eng) for _ in range(windows - 1)]
        for i, e in enumerate(engs):
            e.mel(ggml_io.synth_pcm(100 + i, 30.0))
        gate = threading.Barrier(windows)

        def work(e):
            for _ in range(iters + 1):
                gate.wait()
                assert e.L.mw_encode(e.ctx, 0, 0) == 0

        s0 = eng.hub_stats()
        th = [threading.Thread(target=work, args=(e,)) for e in engs]
        t0 = time.time()
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt = time.time() - t0
        s1 = eng.hub_stats()
        print(f"{windows} windows at a time ({model}, {pname}): {1000.0 * dt / (iters + 1) / windows:.3f} ms per window, "
              f"{s1[4] - s0[4]} windows in {s1[3] - s0[3]} encoder passes")
        for e in engs[1:]:
            e.close()
    eng.close()


if __name__ == "__main__":
    main()
