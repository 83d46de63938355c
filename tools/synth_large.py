"""Fast writer for LARGE synthetic ggml models (bench only: large-v3 is 1.55 G parameters / 3.1 GB).

Same container format and tensor statistics as tools/ggml_io.synthetic_tensors, but generated tensor by
tensor with numpy's PCG64 float32 stream and written straight to disk, so the whole file takes ~20 s
instead of minutes.  Not bit-reproducible across numpy versions (tests use ggml_io's RandomState path).
"""
from __future__ import annotations

import os
import struct

import numpy as np

from tools import ggml_io


def write_large(path: str, preset: str = "large-v3", seed: int = 1234, quiet_special: bool = True,
                fp8_decoder: bool = False, mx_encoder: bool = False) -> ggml_io.HParams:
    """fp8_decoder: the tensors MW_PRECISION_FP8 quantises are written already rounded to e4m3 x 2^s(row), so that an fp32
    reader of the file (the oracle) and the engine's fp8 mode hold exactly the same weights.  mx_encoder: likewise the
    encoder matrices MW_PRECISION_FP8_MX multiplies in MX-fp8 (blocks of 32 along k)."""
    hp = ggml_io.PRESETS[preset]
    rng = np.random.default_rng(seed)
    tmp = path + ".tmp.%d" % os.getpid()
    with open(tmp, "wb") as f:
        f.write(struct.pack("<I", ggml_io.GGML_MAGIC))
        f.write(struct.pack("<11i", *hp.as_list()))
        filt = ggml_io.mel_filterbank(hp.n_mels)
        f.write(struct.pack("<ii", *filt.shape))
        f.write(np.ascontiguousarray(filt, dtype="<f4").tobytes())
        vocab = ggml_io.synthetic_vocab(ggml_io.n_vocab_in_file(hp.n_vocab))
        f.write(struct.pack("<i", len(vocab)))
        for w in vocab:
            f.write(struct.pack("<I", len(w)))
            f.write(w)
        for name, shape, is_mat in ggml_io.tensor_shapes(hp):
            n = int(np.prod(shape))
            if name.endswith("ln.weight") or name.endswith("ln_post.weight"):
                a = 1.0 + 0.05 * rng.standard_normal(n, dtype=np.float32)
            elif name.endswith(".bias"):
                a = 0.02 * rng.standard_normal(n, dtype=np.float32)
            elif name.endswith("positional_embedding"):
                a = 0.05 * rng.standard_normal(n, dtype=np.float32)
            elif name == "decoder.token_embedding.weight":
                a = (0.05 * rng.standard_normal(n, dtype=np.float32)).reshape(shape)
                if quiet_special:
                    a[50256:] *= 0.01
                a = a.reshape(-1)
            else:
                fan_in = int(np.prod(shape[1:]))
                a = rng.standard_normal(n, dtype=np.float32) * np.float32(max(0.02, 1.0 / np.sqrt(fan_in)))
            if fp8_decoder and ggml_io.is_fp8_decoder_tensor(name):
                a = ggml_io.e4m3_quantize_dequantize_rows(a.reshape(shape[0], -1)).reshape(-1)
            if mx_encoder and ggml_io.is_mx_encoder_tensor(name):
                a = ggml_io.mxfp8_quantize_dequantize(a.reshape(shape[0], -1)).reshape(-1)
            use_f16 = hp.ftype == 1 and is_mat
            ne = list(reversed(shape))
            nb = name.encode()
            f.write(struct.pack("<iii", len(ne), len(nb), ggml_io.GGML_TYPE_F16 if use_f16 else ggml_io.GGML_TYPE_F32))
            f.write(struct.pack("<%di" % len(ne), *ne))
            f.write(nb)
            f.write(a.astype("<f2" if use_f16 else "<f4").tobytes())
    os.replace(tmp, path)
    return hp


if __name__ == "__main__":
    import sys
    import time
    t = time.time()
    write_large(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "large-v3")
    print("wrote", sys.argv[1], "in %.1f s" % (time.time() - t))
