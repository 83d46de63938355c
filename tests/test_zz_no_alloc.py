"""GPU: the allocation-free log-mel entry point (mw_pcm_to_mel_noalloc, include/mi355_whisper.h) against the existing
entry points it is a variant of (mw_pcm_to_mel + mw_get_mel), through the C ABI.

The entry point keeps its intermediate rows and the global-maximum word in a static __device__ array shared by every
call on the GPU, so the cases run back to back in ONE process, the second with a smaller input than the first (stale
contents of the first call must not reach the second), on two models (80 and 128 mel bins) that share the array.
Bar: bit-exact (the same two kernels produce both results).
"""
import ctypes as C

import numpy as np
import pytest
import torch

from tools import ggml_io

pytestmark = pytest.mark.gpu


def _noalloc_mel(pkg, eng, pcm, poison=7.25):
    x = torch.from_numpy(np.ascontiguousarray(pcm, dtype=np.float32)).cuda()
    n_len_want = (len(pcm) + 480000) // 160
    out = torch.full((n_len_want * eng.n_mels + 17,), poison, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    n_len = C.c_int(-1)
    rc = eng.L.mw_pcm_to_mel_noalloc(eng.ctx, C.c_void_p(x.data_ptr()), len(pcm), C.c_void_p(out.data_ptr()),
                                     out.numel(), C.byref(n_len))
    assert rc == 0, pkg.last_error()
    assert n_len.value == n_len_want
    host = out.cpu().numpy()
    assert (host[n_len_want * eng.n_mels:] == poison).all()          # nothing written beyond n_len * n_mel
    return host[:n_len_want * eng.n_mels].reshape(n_len_want, eng.n_mels)


def test_no_alloc_mel_equals_existing_entry_point_twice_second_smaller(pkg, model_file):
    e80 = pkg.Engine(model_file("micro.en", ftype=1), precision=pkg.PRECISION_FP32)
    e128 = pkg.Engine(model_file("micro", ftype=1, n_mels=128), precision=pkg.PRECISION_BF16)
    try:
        assert e80.n_mels == 80 and e128.n_mels == 128
        # first a full 30 s window (the maximum), then smaller inputs; loud first, quiet second, so a stale maximum word
        # or stale rows of the first call would change the clamp / the rows of the second
        calls = [(e128, 1.0 * ggml_io.synth_pcm(1, 30.0)), (e128, 0.01 * ggml_io.synth_pcm(2, 4.3)),
                 (e80, 0.5 * ggml_io.synth_pcm(3, 11.0)), (e80, 0.002 * ggml_io.synth_pcm(4, 0.9)),
                 (e128, ggml_io.synth_pcm(5, 30.0)[:480640 - 480000 + 479360])]
        for eng, pcm in calls:
            pcm = np.ascontiguousarray(pcm, dtype=np.float32)
            got = _noalloc_mel(pkg, eng, pcm)
            # the call allocates nothing on the device: free memory is the same before and after a second, identical call
            # (inputs and outputs of _noalloc_mel are torch's, allocated from its caching pool before the call)
            x = torch.from_numpy(pcm).cuda()
            out = torch.empty(got.size + 17, dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()
            free0 = torch.cuda.mem_get_info()[0]
            n_len = C.c_int(-1)
            assert eng.L.mw_pcm_to_mel_noalloc(eng.ctx, C.c_void_p(x.data_ptr()), len(pcm), C.c_void_p(out.data_ptr()), out.numel(),
                                               C.byref(n_len)) == 0
            assert torch.cuda.mem_get_info()[0] == free0
            want = eng.mel(pcm)                       # existing entry points: [n_mel][n_len]
            assert got.shape == want.T.shape
            assert np.array_equal(got, want.T), float(np.abs(got - want.T).max())
        # beyond the maximum: rejected with a clear error, nothing launched
        big = torch.zeros(480641, dtype=torch.float32, device="cuda")
        out = torch.zeros(8, dtype=torch.float32, device="cuda")
        n_len = C.c_int(0)
        rc = e80.L.mw_pcm_to_mel_noalloc(e80.ctx, C.c_void_p(big.data_ptr()), 480641, C.c_void_p(out.data_ptr()), 8, C.byref(n_len))
        assert rc != 0 and "480640" in pkg.last_error()
        # output buffer too small: rejected as well
        rc = e80.L.mw_pcm_to_mel_noalloc(e80.ctx, C.c_void_p(big.data_ptr()), 16000, C.c_void_p(out.data_ptr()), 8, C.byref(n_len))
        assert rc != 0 and "output buffer" in pkg.last_error()
    finally:
        e80.close()
        e128.close()
