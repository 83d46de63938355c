#!/usr/bin/env python3
"""bench.py — audio-seconds transcribed per wall-second, Whisper large-v3 on MI355X.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): Whisper large-v3 dims
(d=1280, 32+32 layers, 128 mel bins, vocab 51866), bf16, beam search (beam_size 5) with timestamps,
ONE stream per GPU, one 30 s 16 kHz window per step: log-mel -> encoder -> cross-K/V -> prompt prefill
-> 128 beam-5 decode steps with the logits rules + samplers on the device -> segments.
Weights are seeded synthetic tensors in the real GGML container (no checkpoints offline), parsed and
uploaded by the engine's own loader; PCM is synthetic and already resident in HBM when the timed region
starts.  The decode length is forced to 128 sampled tokens per window (mw_set_force_decode_steps:
random weights would otherwise stop after a handful of tokens), and temperature fallback is off
(temperature_inc = 0) so every step does the same work.

`--gpus N`: one rank per GPU, stream i -> GPU i, no data-path collective; value = all ranks' audio
seconds / max-over-ranks time; scaling "weak".  Under torch.distributed.run (RANK / WORLD_SIZE in the environment) this
process IS one rank; run plainly with --gpus N > 1 it starts the N ranks itself (python -m torch.distributed.run as a
child process, before this process has touched a GPU) and exits with their code.
`--workload streams`: BASELINE configs[3] instead — large-v3-turbo, --streams (64) concurrent synthetic streams
partitioned i mod N over the ranks, greedy, 96 forced decode steps per 30 s window, one context + worker thread per
stream on one upload of the weights per GPU; a step = every stream transcribes one window.

One JSON line on rank 0 (see the contract in the task statement) with two extra objects:
  roofline     the dominant kernel class of the decode step by share of kernel time (cross-attention; further
               classes under roofline_others): algorithmic bytes per launch / average launch duration measured
               live with HIP events on the engine's stream in a launch-by-launch pass right after the timed
               region (the timed region itself replays captured graphs), against 8 TB/s HBM
  cpu_baseline the CPU oracle (a port, NOT whisper.cpp) timed on the box's host cores on a bounded sample
"""
from __future__ import annotations

import argparse
import ctypes as C
import importlib.util
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def load_package():
    spec = importlib.util.spec_from_file_location("obs_localvocal_amd", os.path.join(ROOT, "obs-localvocal_amd", "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["obs_localvocal_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


GEMV_CLASSES = {"qkv": 0, "attn_out": 1, "cross_q": 2, "fc1": 3, "fc2": 4, "logits": 5, "cross_attn": 7}
KERNEL_NAMES = {"qkv": "proj_ln_tile_kernel<rows,EPI_QKV,16>", "cross_q": "proj_ln_tile_kernel<rows,EPI_SCALE,8>",
                "fc1": "proj_ln_tile_kernel<rows,EPI_GELU_BF16,32>", "attn_out": "proj_res_tile_kernel<rows,5,1,8>",
                "fc2": "proj_res_tile_kernel<rows,5,4,8>", "logits": "logits_tile_kernel<rows,10>",
                "cross_attn": "cross_attn_lean_kernel<rows>"}


def gemv_bytes(cls: str, d: int, n_vocab: int, rows: int, wbytes: int, n_audio_ctx: int = 1500, fp8: bool = False) -> int:
    """Algorithmic HBM bytes of ONE launch of a decode-step kernel class.  fp8: the matrix is read as
    e4m3 bytes + one fp32 scale per row; activations, caches and the cross K / V stay bf16."""
    if cls == "cross_attn":  # one layer's cross K and V of the stream (shared by the rows) + q in + rows out
        return 2 * n_audio_ctx * d * wbytes + rows * d * 4 * 2
    N, K = {"qkv": (3 * d, d), "attn_out": (d, d), "cross_q": (d, d), "fc1": (4 * d, d), "fc2": (d, 4 * d),
            "logits": (n_vocab, d)}[cls]
    resid = cls in ("attn_out", "fc2")
    x_b = 2 if resid and wbytes == 2 else 4   # rows between attention / MLP kernels and the residual projections are bf16
    o_b = 2 if cls == "fc1" and wbytes == 2 else 4
    return N * K * (1 if fp8 else wbytes) + N * (8 if fp8 else 4) + rows * K * x_b + rows * N * o_b * (2 if resid else 1)


def cpu_baseline(model_path: str, pcm: np.ndarray, beam: int, decode_steps: int, sample_steps: int, threads: int, reps: int = 3):
    """Time the CPU oracle (port) on a bounded sample of the same window: mel + full encoder + `sample_steps` beam decode
    steps; one warm-up pass, then the median of `reps` passes per component; the per-step time extrapolates to
    `decode_steps`."""
    from oracle import pyoracle
    pyoracle.lib().ow_set_threads(threads)
    t0 = time.time()
    orc = pyoracle.Oracle(model_path)
    t_load = time.time() - t0
    p = orc.default_params(1, temperature_inc=0.0, duration_ms=30000, language="en", force_decode_steps=sample_steps,
                           beam_size=beam)
    t_mel, t_enc, t_full = [], [], []
    for rep in range(reps + 1):
        t0 = time.time()
        orc.mel(pcm)
        t1 = time.time()
        orc.encode(0)
        t2 = time.time()
        rc = orc.full(p, pcm)
        t3 = time.time()
        assert rc == 0, rc
        if rep > 0:  # rep 0 is the warm-up
            t_mel.append(t1 - t0)
            t_enc.append(t2 - t1)
            t_full.append(t3 - t2)
    m_mel, m_enc, m_full = (float(np.median(x)) for x in (t_mel, t_enc, t_full))
    per_step = max(m_full - m_mel - m_enc, 1e-9) / sample_steps
    est = m_mel + m_enc + per_step * decode_steps
    orc.close()
    return dict(value=round(30.0 / est, 4), unit="audio-sec/sec", cores=threads, kind="port",
                sample=("CPU oracle (fp32 restatement, OpenMP; not whisper.cpp): one 30 s window, 1 warm-up + median of %d passes: "
                        "mel %.2fs + full 32-layer encoder %.2fs, %d of %d beam-%d decode steps measured (%.3fs/step incl. prefill) "
                        "and extrapolated to %d; model load %.0fs excluded" %
                        (reps, m_mel, m_enc, sample_steps, decode_steps, beam, per_step, decode_steps, t_load)))


def run_streams(a, rank, local_rank, world, dist, steps=None, warmup=None, pkg=None):
    """BASELINE configs[3]: large-v3-turbo, a.streams greedy streams sharded i mod N, one context + worker thread per stream
    on one weight upload per GPU (LocalVocal: one whisper_loop thread per filter, whisper-utils.cpp:49-50).  A step = every
    stream transcribes one 30 s window; value = all streams' audio seconds / max-over-ranks time.  Returns the result
    object on rank 0 (None elsewhere)."""
    import threading
    pkg = pkg or load_package()
    steps = a.steps if steps is None else steps
    warmup = a.warmup if warmup is None else warmup
    from tools import ggml_io, stream_shard, synth_large
    model = "large-v3-turbo" if a.model == "large-v3" else a.model
    decode_steps = 96 if a.decode_steps == 128 else a.decode_steps
    os.makedirs(a.model_dir, exist_ok=True)
    model_path = os.path.join(a.model_dir, f"ggml-{model}-synth-s1234.bin")
    if local_rank == 0 and not os.path.exists(model_path):
        synth_large.write_large(model_path, model, seed=1234, quiet_special=True)
    if dist is not None:
        dist.barrier()
    torch.cuda.set_device(local_rank)
    mine = stream_shard.assign_streams(a.streams, world, rank)
    if not mine:
        raise SystemExit("fewer streams than ranks")
    t0 = time.time()
    prec_name = a.precision if a.precision in ("fp8", "fp8mx") and getattr(a, "workload", "single") == "streams" else "bf16"
    prec = {"bf16": pkg.PRECISION_BF16, "fp8": pkg.PRECISION_FP8, "fp8mx": pkg.PRECISION_FP8_MX}[prec_name]
    first = pkg.Engine(model_path, precision=prec, device=local_rank)
    engs = [first] + [pkg.Engine(model_path, precision=prec, device=local_rank, share=first) for _ in mine[1:]]
    t_load = time.time() - t0
    pcms, params = [], []
    for e, sid in zip(engs, mine):
        e.set_force_decode_steps(decode_steps)
        params.append(e.default_params(pkg.SAMPLING_GREEDY, temperature_inc=0.0, duration_ms=30000, language="en"))
        pcms.append(torch.from_numpy(ggml_io.synth_pcm(100 + sid, 30.0)).cuda(local_rank))
    torch.cuda.synchronize()
    conc = max(1, min(a.concurrency if a.concurrency > 0 else len(engs), len(engs)))

    def run(windows):
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
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for x in th:
            x.start()
        for x in th:
            x.join()
        for e in engs:
            e.L.mw_synchronize(e.ctx)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t
        if errs:
            raise RuntimeError(errs[0])
        return el

    run(max(1, warmup))
    elapsed = run(steps)
    if dist is not None:
        elapsed = stream_shard.max_over_ranks(dist, elapsed, device=f"cuda:{local_rank}")
    out = None
    if rank == 0:
        audio = 30.0 * steps * a.streams
        hub = engs[0].hub_stats()
        out = {
            "metric": "audio-sec transcribed/sec", "value": round(audio / elapsed, 2), "unit": "audio-sec/sec", "n_gpus": world,
            "steps": steps, "warmup": warmup, "ms_per_step": round(1000.0 * elapsed / steps, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": prec_name, "data": "synthetic",
            "config": {"workload": f"whisper {model} {prec_name}, greedy, {a.streams} streams sharded i mod {world} ({len(engs)} on rank 0, one "
                                   f"context + worker thread each, one weight upload per GPU), 30 s window per stream per step, "
                                   f"{decode_steps} forced decode steps, synthetic GGML weights (seed 1234)",
                       "streams": a.streams, "concurrency_per_gpu": conc, "decode_steps": decode_steps,
                       "tokens_per_window": len(engs[0].token_ids()), "parallelism": f"streams i mod {world} (no collectives)"},
            "rtf_per_stream": round(elapsed / steps / 30.0, 6), "contexts_setup_s": round(t_load, 2),
            "rows_per_decode_step": round(hub[1] / max(hub[0], 1), 2) if hub else None,
            "windows_per_encoder_pass": round(hub[4] / max(hub[3], 1), 2) if hub and len(hub) > 4 else None}
    for e in engs[1:] + engs[:1]:
        e.close()
    if dist is not None:
        dist.barrier()
    return out


def launch_ranks(n: int) -> int:
    """Start n ranks of this script under torch.distributed.run (one per GPU) and return their exit code.
    Called before anything in this process has initialised a GPU."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--workload", default="single", choices=["single", "streams"],
                    help="single: BASELINE configs[2], one large-v3 beam-5 stream per GPU (the headline); "
                         "streams: configs[3], large-v3-turbo, --streams greedy streams sharded i mod N")
    ap.add_argument("--streams", type=int, default=64, help="--workload streams: streams of the whole job")
    ap.add_argument("--concurrency", type=int, default=0,
                    help="streams workload: streams inside whisper_full at once per GPU (0: every stream of the rank, as LocalVocal's "
                         "worker threads run)")
    ap.add_argument("--no-streams", action="store_true",
                    help="single workload: leave out the `streams` sub-object (BASELINE configs[3], 64 turbo streams i mod N, strong "
                         "scaling) that is measured after the headline")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous / reduction path only, gloo backend, no GPU work (CPU test of the N > 1 path)")
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--model", default="large-v3")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "fp8", "fp8mx"],
                    help="fp8: bf16 engine with the decoder's projection / logits matrices streamed as e4m3 (BASELINE configs[4]); "
                         "fp8mx: that and the encoder's Q|K / fc1 / fc2 products in MX-fp8 on the fp8 matrix cores")
    ap.add_argument("--beam", type=int, default=5)
    ap.add_argument("--decode-steps", type=int, default=128)
    ap.add_argument("--roofline-class", default="cross_attn", choices=sorted(GEMV_CLASSES),
                    help="kernel class of the roofline object (default: the class with the largest share of kernel time "
                         "in profiles/r01_kernel_stats_graph_wg.csv)")
    ap.add_argument("--roofline-others", default="fc2,fc1,qkv", help="further classes reported under roofline_others")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-steps", type=int, default=12)
    ap.add_argument("--model-dir", default=os.environ.get("MI355_WHISPER_BENCH_DIR", "/tmp/mi355_whisper_bench"))
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.dry_run:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    n_gpus = max(world, 1)
    if a.gpus != n_gpus and rank == 0:
        print(f"[bench] note: --gpus {a.gpus} but WORLD_SIZE={world}; reporting n_gpus={n_gpus}", file=sys.stderr)
    if a.dry_run:
        # the N > 1 plumbing without a GPU: partition, barrier, timed region, MAX over ranks, one JSON line on rank 0
        from tools import stream_shard
        mine = stream_shard.assign_streams(a.streams if a.workload == "streams" else n_gpus, n_gpus, rank)
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        time.sleep(0.01 * (1 + rank))
        elapsed = time.perf_counter() - t0
        if dist is not None:
            dist.barrier()
            elapsed = stream_shard.max_over_ranks(dist, elapsed)
            counts = [None] * n_gpus
            dist.all_gather_object(counts, len(mine))
        else:
            counts = [len(mine)]
        if rank == 0:
            print(json.dumps({"metric": "audio-sec transcribed/sec", "value": None, "unit": "audio-sec/sec", "n_gpus": n_gpus,
                              "steps": a.steps, "warmup": a.warmup, "dry_run": True, "streams_per_rank": counts,
                              "elapsed_max_s": round(elapsed, 4), "scaling": "weak",
                              # what a real run measures after the headline: configs[3] sharded i mod N over the same ranks
                              "streams": {"metric": "audio-sec transcribed/sec", "value": None, "scaling": "strong", "n_gpus": n_gpus,
                                          "streams_per_rank": [len(stream_shard.assign_streams(a.streams, n_gpus, r)) for r in range(n_gpus)]}}),
                  flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return
    if a.workload == "streams":
        out = run_streams(a, rank, local_rank, world, dist)
        if rank == 0:
            print(json.dumps(out), flush=True)
        if dist is not None:
            dist.destroy_process_group()
        return

    pkg = load_package()
    from tools import ggml_io, synth_large

    # ---- model file: written once per node by local rank 0 ----
    os.makedirs(a.model_dir, exist_ok=True)
    model_path = os.path.join(a.model_dir, f"ggml-{a.model}-synth-s1234.bin")
    if local_rank == 0 and not os.path.exists(model_path):
        t0 = time.time()
        synth_large.write_large(model_path, a.model, seed=1234, quiet_special=True)
        print(f"[bench] wrote synthetic {a.model} weights in {time.time() - t0:.1f}s -> {model_path}", file=sys.stderr)
    if dist is not None:
        dist.barrier()
    else:
        while not os.path.exists(model_path):
            time.sleep(0.5)

    torch.cuda.set_device(local_rank)
    t0 = time.time()
    eng = pkg.Engine(model_path, precision={"bf16": pkg.PRECISION_BF16, "fp32": pkg.PRECISION_FP32, "fp8": pkg.PRECISION_FP8, "fp8mx": pkg.PRECISION_FP8_MX}[a.precision],
                     device=local_rank)
    t_load = time.time() - t0
    eng.set_force_decode_steps(a.decode_steps)
    params = eng.default_params(pkg.SAMPLING_BEAM_SEARCH, temperature_inc=0.0, duration_ms=30000, language="en",
                                beam_size=a.beam)
    pcm_host = ggml_io.synth_pcm(100 + rank, 30.0)
    pcm = torch.from_numpy(pcm_host).cuda(local_rank)
    torch.cuda.synchronize()

    def step():
        rc = eng.full_device(params, pcm.data_ptr(), pcm.numel())
        if rc != 0:
            raise RuntimeError(f"whisper_full failed: {rc} {pkg.last_error()}")

    for _ in range(a.warmup):
        step()
    n_tok = len(eng.token_ids())
    eng.reset_timings()

    def sync_all():
        torch.cuda.synchronize()
        eng.L.mw_synchronize(eng.ctx)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    cls = a.roofline_class
    sync_all()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    tm = eng.timings()
    if dist is not None:
        from tools import stream_shard
        elapsed = stream_shard.max_over_ranks(dist, elapsed, device=f"cuda:{local_rank}")

    # the reference API path hands over HOST PCM (whisper_full, whisper-processing.cpp:218): timed next to the device-PCM figure
    host_ms = None
    if rank == 0:
        eng.full(params, pcm_host)
        sync_all() if dist is None else None
        th0 = time.perf_counter()
        for _ in range(2):
            if eng.full(params, pcm_host) != 0:
                raise RuntimeError("whisper_full (host PCM) failed")
        eng.L.mw_synchronize(eng.ctx)
        host_ms = 1000.0 * (time.perf_counter() - th0) / 2

    # roofline leg: the same step, launch by launch (graphs off while a kernel class is bracketed by HIP events on
    # the engine's stream), right after the timed region, same process, same weights
    prof = {}
    if rank == 0:
        for k in [cls] + [x for x in a.roofline_others.split(",") if x and x != cls]:
            n_launch, tot_ms = C.c_int(0), C.c_float(0)
            eng.L.mw_profile_begin(eng.ctx, GEMV_CLASSES[k], 200000)
            for _ in range(max(1, min(a.steps, 2))):
                step()
            eng.L.mw_profile_end(eng.ctx, C.byref(n_launch), C.byref(tot_ms))
            prof[k] = (n_launch.value, tot_ms.value, max(1, min(a.steps, 2)))
        # second view of the same classes, INSIDE graph replay: every launch of the class is issued twice
        # (mw_set_duplicate_class; results unchanged) and the extra time per window is divided by the launches
        def timed_windows(nw):
            sync_all()
            t = time.perf_counter()
            for _ in range(nw):
                step()
            sync_all()
            return (time.perf_counter() - t) / nw
        nw = max(2, min(a.steps, 4))
        step()
        t_base = timed_windows(nw)
        in_graph = {}
        for k in prof:
            eng.L.mw_set_duplicate_class(eng.ctx, GEMV_CLASSES[k])
            step()  # captures the graphs of this variant
            t_dup = timed_windows(nw)
            eng.L.mw_set_duplicate_class(eng.ctx, -1)
            launches_per_window = prof[k][0] / prof[k][2]
            in_graph[k] = 1e6 * (t_dup - t_base) / launches_per_window if launches_per_window else None
    if rank == 0:
        audio_s = 30.0 * a.steps * n_gpus
        value = audio_s / elapsed
        ms_per_step = 1000.0 * elapsed / a.steps
        wbytes = 4 if a.precision == "fp32" else 2
        rows = a.beam
        # HBM bytes per launch from the PMC passes (separate rocprofv3 --pmc runs of this command, summarised by
        # tools/pmc_summary.py with the gfx950 FETCH_SIZE correction); null when the summary is not in the tree
        pmc, pmc_file = {}, None
        import glob
        for cand in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
            try:
                with open(cand) as fh:
                    pmc = json.load(fh).get("kernels", {})
                pmc_file = os.path.relpath(cand, ROOT)
                break
            except (OSError, ValueError):
                pmc = {}

        def roof(k):
            # Two live measurements of the class, both taken above in this process:
            #  (1) inside graph replay — the product's form: every launch of the class issued twice
            #      (mw_set_duplicate_class; results unchanged), extra time per window / launches.  This is what the rocprofv3
            #      kernel statistics under profiles/ agree with, and what `achieved` / `frac` are computed from;
            #  (2) eager, each launch bracketed by HIP events on the engine's stream (graphs off): includes the gap of
            #      back-to-back eager launches; reported beside it (`*_eager_events`).  Used for `achieved` only when (1) failed.
            n_l, ms, _nw = prof[k]
            b_launch = gemv_bytes(k, eng.n_text_state, eng.n_vocab, rows, wbytes, fp8=a.precision in ("fp8", "fp8mx"))
            eager_us = 1000.0 * ms / max(n_l, 1)
            graph_us = in_graph.get(k) if in_graph.get(k) and in_graph[k] > 0 else None
            used_us = graph_us or eager_us
            achieved = b_launch / (used_us * 1e-6) / 1e9 if used_us > 0 else 0.0
            return {"bound": "hbm", "kernel": KERNEL_NAMES[k], "achieved": round(achieved, 1), "peak": 8000.0, "unit": "GB/s",
                    "frac": round(achieved / 8000.0, 4),
                    "traffic": (pmc[k]["hbm_bytes_per_launch"] if k in pmc and a.model == "large-v3" and a.beam == 5 and
                                a.precision == "bf16" else None),
                    # NOT measured in this run: the committed summary of separate rocprofv3 --pmc passes of this command
                    "traffic_source": f"static: {pmc_file} (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)",
                    "bytes_per_launch": b_launch,
                    "avg_launch_us": round(used_us, 3), "launches_timed": n_l,
                    "method": ("marginal cost inside graph replay (class issued twice)" if graph_us else "eager, HIP events per launch"),
                    "avg_launch_us_eager_events": round(eager_us, 3),
                    "achieved_eager_events": round(b_launch / (eager_us * 1e-6) / 1e9, 1) if eager_us > 0 else None}
        out = {
            "metric": "audio-sec transcribed/sec", "value": round(value, 2), "unit": "audio-sec/sec",
            "n_gpus": n_gpus, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": a.precision, "data": "synthetic",
            "config": {"workload": f"whisper {a.model} {a.precision}, beam={a.beam} with timestamps, 1 stream/GPU, "
                                   f"30 s window/step, {a.decode_steps} forced decode steps, synthetic GGML weights (seed 1234)",
                       "streams_per_gpu": 1, "window_s": 30, "beam_size": a.beam, "decode_steps": a.decode_steps,
                       "tokens_per_window": n_tok, "parallelism": f"streams x{n_gpus} (no collectives)"},
            "rtf_per_stream": round(ms_per_step / 30000.0, 6),
            "ms_per_step_host_pcm": (round(host_ms, 3) if host_ms else None),  # whisper_full with the PCM in host memory (PCIe-inclusive)
            "breakdown_ms_per_step": {"mel": round(tm.mel_ms / a.steps, 3), "encoder+crossKV(host-side launch)": round(tm.encode_ms / a.steps, 3),
                                      "decode": round(tm.decode_ms / a.steps, 3), "rules+sample(sync)": round(tm.sample_ms / a.steps, 3),
                                      "decode_calls": tm.n_decode // max(a.steps, 1)},
            "model_load_s": round(t_load, 2),
            "roofline": roof(cls),
            "roofline_others": {k: roof(k) for k in prof if k != cls},
        }
        if not a.no_cpu_baseline and n_gpus == 1:
            threads = min(16, os.cpu_count() or 1)
            try:
                out["cpu_baseline"] = cpu_baseline(model_path, pcm_host, a.beam, a.decode_steps, a.cpu_sample_steps, threads)
            except Exception as e:  # the baseline is a reported side figure; never fail the bench on it
                out["cpu_baseline"] = {"value": None, "unit": "audio-sec/sec", "cores": threads, "kind": "port",
                                       "sample": f"failed: {e}"}
    eng.close()
    # The number north_star's ">= 7x aggregate at 8 GPUs" is about: BASELINE configs[3] (64 turbo streams, i mod N, STRONG scaling)
    # on the same ranks, next to the weak-scaling headline (which is trivially ~N x: one independent stream per GPU).
    streams = None
    if not a.no_streams:
        try:
            streams = run_streams(a, rank, local_rank, world, dist, steps=max(2, min(a.steps, 3)), warmup=1, pkg=pkg)
        except Exception as e:  # a side figure: never fail the headline on it
            if dist is not None:
                raise
            streams = {"value": None, "error": repr(e)}
    if rank == 0:
        out["streams"] = streams
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
