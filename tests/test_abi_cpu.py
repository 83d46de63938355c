"""CPU: the C-ABI library loads and exports every symbol include/*.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for h in ("whisper.h", "mi355_whisper.h", "localvocal_hotpath.h"):
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        for m in re.finditer(r"\b((?:whisper|mw|lv)_[a-z0-9_]+)\s*\(", src):
            names.add(m.group(1))
    # function-pointer typedefs are types, not exports
    return {n for n in names if not (n.startswith("whisper_") and n.endswith("_callback"))}


def test_library_exports_every_declared_symbol(pkg):
    if not os.path.exists(pkg.LIB_PATH):
        pkg.build()
    lib = ctypes.CDLL(pkg.LIB_PATH)
    decl = declared_symbols()
    assert len(decl) > 60
    missing = [n for n in sorted(decl) if not hasattr(lib, n)]
    assert not missing, missing
    assert set(pkg.EXPORTED_SYMBOLS) == decl


def test_struct_layouts_match_header(pkg):
    # sizes as a C compiler lays the include/whisper.h structs out on x86-64 (checked by compiling a probe)
    import subprocess, tempfile
    probe = r'''
#include <stdio.h>
#include "whisper.h"
#include "mi355_whisper.h"
#include "localvocal_hotpath.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu\n", sizeof(struct whisper_full_params), sizeof(whisper_token_data),
         sizeof(struct whisper_context_params), sizeof(struct mw_timings), sizeof(lv_detection));
  return 0; }'''
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "p.c")
        open(src, "w").write(probe)
        exe = os.path.join(d, "p")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        sizes = [int(x) for x in subprocess.check_output([exe]).split()]
    assert sizes == [ctypes.sizeof(pkg.FullParams), ctypes.sizeof(pkg.TokenData), ctypes.sizeof(pkg.ContextParams),
                     ctypes.sizeof(pkg.Timings), ctypes.sizeof(pkg.LvDetection)]


def test_defaults_and_pure_host_entry_points(pkg):
    L = pkg.lib()
    p = L.whisper_full_default_params(pkg.SAMPLING_BEAM_SEARCH)
    assert p.beam_search.beam_size == 5 and p.greedy.best_of == -1 and p.no_context and p.suppress_blank
    assert abs(p.entropy_thold - 2.4) < 1e-6 and abs(p.logprob_thold + 1.0) < 1e-6 and p.language == b"en"
    g = L.whisper_full_default_params(pkg.SAMPLING_GREEDY)
    assert g.greedy.best_of == 5 and g.beam_search.beam_size == -1
    assert L.whisper_lang_str(0) == b"en" and L.whisper_lang_str(99) == b"yue" and L.whisper_lang_id(b"de") == 2
    assert L.whisper_lang_str(-7) == b"en"  # clamped: LocalVocal never checks the id
    buf = ctypes.create_string_buffer(32)
    L.lv_to_timestamp(3723004, buf, 32)
    assert buf.value == b"62:03.004"
    # no GPU here: model loading must fail loudly (nullptr), never fall back to a CPU path
    import torch
    if not torch.cuda.is_available():
        assert not L.mw_init_from_file(b"/nonexistent.bin", 0, 0)


def test_missing_gpu_is_an_error_not_a_fallback(pkg, model_file):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        pkg.Engine(model_file("micro.en"))


def test_update_whisper_model_decisions_that_need_no_gpu(pkg):
    # whisper-model-utils.cpp:38-47,75-79: empty name, empty external path, unknown / not-downloaded catalogue names
    f = pkg.Filter()
    try:
        assert f.update_model() == 1                               # "Whisper model path is empty"
        f.set_setting("whisper_model_path", "!!!external!!!")
        assert f.update_model() == 1                               # "External model file path is empty"
        f.set_setting("whisper_model_path", "No Such Model")
        assert f.update_model() == 1                               # "Model '...' does not exist"
        f.register_model("Listed", None)
        f.set_setting("whisper_model_path", "Listed")
        assert f.update_model() == 1                               # listed, not on disk (download dialog in the reference)
        st = f.model_state()
        assert st == dict(model_path="", file_loaded="", loaded_new=False, worker_running=False, dtw=False, context=False)
    finally:
        f.close()
