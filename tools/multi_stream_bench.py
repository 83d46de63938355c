#!/usr/bin/env python3
"""BASELINE.json configs[3] / SURVEY.md §8(d).4: Whisper large-v3-turbo, S concurrent synthetic streams on one GPU
(or sharded `i mod N` over the ranks of a torch.distributed launch), greedy, 96 forced decode steps per 30 s window.

One stream = one context = one worker thread, as in LocalVocal (one whisper_loop thread per filter instance,
/root/reference/src/whisper-utils/whisper-utils.cpp:49-50).  All contexts of a GPU share ONE upload of the weights
(mw_init_shared); each has its own HIP stream, KV caches and captured decode graphs, so the latency-bound decode steps of
different streams overlap on the device.  `--concurrency` bounds how many streams are inside whisper_full at once
(the others wait, like sources whose segment has not ended yet).

Prints one JSON line per concurrency level: aggregate audio-seconds per wall-second and per-stream RTF.  Not the
bench.py headline (that is configs[2]); no oracle involved.
"""
from __future__ import annotations

import argparse
import importlib.util
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402


def load_package():
    if "obs_localvocal_amd" in sys.modules:
        return sys.modules["obs_localvocal_amd"]
    spec = importlib.util.spec_from_file_location("obs_localvocal_amd", os.path.join(ROOT, "obs-localvocal_amd", "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["obs_localvocal_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="large-v3-turbo")
    ap.add_argument("--streams", type=int, default=64, help="streams of the whole job (partitioned i mod world size)")
    ap.add_argument("--windows", type=int, default=2, help="timed 30 s windows per stream")
    ap.add_argument("--decode-steps", type=int, default=96)
    ap.add_argument("--concurrency", default="1,4,16,64", help="comma list: streams inside whisper_full at once, per GPU")
    ap.add_argument("--strategy", default="greedy", choices=["greedy", "beam"])
    ap.add_argument("--model-dir", default=os.environ.get("MI355_WHISPER_BENCH_DIR", "/tmp/mi355_whisper_bench"))
    ap.add_argument("--out", default="")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    pkg = load_package()
    from tools import ggml_io, stream_shard, synth_large

    os.makedirs(a.model_dir, exist_ok=True)
    model_path = os.path.join(a.model_dir, f"ggml-{a.model}-synth-s1234.bin")
    if local_rank == 0 and not os.path.exists(model_path):
        synth_large.write_large(model_path, a.model, seed=1234, quiet_special=True)
    if dist is not None:
        dist.barrier()

    torch.cuda.set_device(local_rank)
    mine = stream_shard.assign_streams(a.streams, world, rank)
    t0 = time.time()
    first = pkg.Engine(model_path, precision=pkg.PRECISION_BF16, device=local_rank)
    engs = [first] + [pkg.Engine(model_path, precision=pkg.PRECISION_BF16, device=local_rank, share=first) for _ in mine[1:]]
    t_load = time.time() - t0
    strategy = pkg.SAMPLING_GREEDY if a.strategy == "greedy" else pkg.SAMPLING_BEAM_SEARCH
    pcms, params = [], []
    for e, sid in zip(engs, mine):
        e.set_force_decode_steps(a.decode_steps)
        params.append(e.default_params(strategy, temperature_inc=0.0, duration_ms=30000, language="en"))
        pcms.append(torch.from_numpy(ggml_io.synth_pcm(100 + sid, 30.0)).cuda(local_rank))
    torch.cuda.synchronize()

    def run_level(conc: int, windows: int):
        gate = threading.Semaphore(conc)
        errs = []

        def work(i):
            try:
                for _ in range(windows):
                    with gate:
                        rc = engs[i].full_device(params[i], pcms[i].data_ptr(), pcms[i].numel())
                        if rc != 0:
                            raise RuntimeError(f"whisper_full failed: {rc} {pkg.last_error()}")
            except Exception as ex:  # noqa: BLE001
                errs.append(repr(ex))

        th = [threading.Thread(target=work, args=(i,)) for i in range(len(engs))]
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        torch.cuda.synchronize()
        el = time.perf_counter() - t
        if errs:
            raise RuntimeError(errs[0])
        if dist is not None:
            el = stream_shard.max_over_ranks(dist, el, device=f"cuda:{local_rank}")
        return el

    run_level(max(1, min(4, len(engs))), 1)  # warm-up: every context captures its decode graphs
    lines = []
    for conc in [int(x) for x in a.concurrency.split(",") if x]:
        conc = max(1, min(conc, len(engs)))
        el = run_level(conc, a.windows)
        audio = 30.0 * a.windows * a.streams
        out = {"metric": "audio-sec transcribed/sec", "value": round(audio / el, 1), "unit": "audio-sec/sec", "n_gpus": world,
               "config": {"workload": f"whisper {a.model} bf16, {a.strategy}, {a.streams} streams ({len(engs)} per GPU, one context + "
                                      f"worker thread each, shared weights), {a.windows} x 30 s windows per stream, {a.decode_steps} "
                                      f"forced decode steps", "concurrency_per_gpu": conc},
               "elapsed_s": round(el, 3), "window_latency_ms": round(1000.0 * el * conc / (a.windows * len(engs)), 1),
               "tokens_per_window": len(engs[0].token_ids()), "contexts_setup_s": round(t_load, 2), "data": "synthetic"}
        lines.append(out)
        if rank == 0:
            print(json.dumps(out), flush=True)
    if rank == 0 and a.out:
        with open(a.out, "w") as fh:
            for ln in lines:
                fh.write(json.dumps(ln) + "\n")
    for e in engs[1:] + engs[:1]:
        e.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
