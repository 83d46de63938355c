"""CPU: the model catalogue next to the load path — load_models_info / find_model_bin_file
(/root/reference/src/model-utils/model-infos.cpp:114-260, model-downloader.cpp:9-89) over a models_directory.json in the
reference's schema, SHA-256 against hashlib and the FIPS 180-4 vectors, and the digest check the reference never makes.
The fixture is written by the test (same keys as data/models/models_directory.json: one entry carries the reference
catalogue's name / folder / digest of ggml-large-v3-q5_0, the others point at files the test creates)."""
import hashlib
import json
import os

import pytest


def _catalogue(tmp_path, blobs):
    models = [
        {"friendly_name": "Whisper Large v3 q5 (1Gb)", "local_folder_name": "ggml-large-v3-q5_0", "type": "MODEL_TYPE_TRANSCRIPTION",
         "files": [{"url": "https://huggingface.co/ggerganov/whisper.cpp/resolve/main/ggml-large-v3-q5_0.bin",
                    "sha256": "d75795ecff3f83b5faa89d1900604ad8c780abd5739fae406de19f23ecd98ad1"}]},
        {"friendly_name": "In data", "local_folder_name": "in-data", "type": "MODEL_TYPE_TRANSCRIPTION",
         "files": [{"url": "https://example.invalid/a.bin", "sha256": hashlib.sha256(blobs["a"]).hexdigest().upper()}],
         "extra": {"language": "en", "description": "unicode é中 \"quoted\" \\ back", "source": "test"}},
        {"friendly_name": "In config", "local_folder_name": "in-config", "type": "MODEL_TYPE_TRANSCRIPTION",
         "files": [{"url": "https://example.invalid/b.bin", "sha256": "00" * 32}]},
        {"friendly_name": "No digest", "local_folder_name": "no-digest", "type": "MODEL_TYPE_TRANSCRIPTION",
         "files": [{"url": "https://example.invalid/c.bin"}]},
        {"friendly_name": "A translation model", "local_folder_name": "m2m", "type": "MODEL_TYPE_TRANSLATION", "files": []},
        {"local_folder_name": "nameless", "type": "MODEL_TYPE_TRANSCRIPTION", "files": []},   # skipped: no friendly_name
    ]
    data, config = tmp_path / "data_models", tmp_path / "config_models"
    for folder, root, key in (("in-data", data, "a"), ("in-config", config, "b"), ("no-digest", data, "c")):
        (root / folder).mkdir(parents=True)
        (root / folder / "readme.txt").write_text("not the model")
        (root / folder / ("ggml-%s.bin" % key)).write_bytes(blobs[key])
    path = tmp_path / "models_directory.json"
    path.write_text(json.dumps({"models": models}, indent=2, ensure_ascii=True))
    return str(path), str(data), str(config)


def test_sha256_known_answers_and_against_hashlib(pkg, tmp_path):
    cases = [b"", b"abc", b"abcdbcdecdefdefgefghfghighijhijkijkljklmklmnlmnomnopnopq", os.urandom(55), os.urandom(56), os.urandom(64),
             os.urandom(3 * (1 << 20) + 17)]
    assert hashlib.sha256(b"abc").hexdigest() == "ba7816bf8f01cfea414140de5dae2223b00361a396177a9cb410ff61f20015ad"
    for i, blob in enumerate(cases):
        p = tmp_path / ("blob%d" % i)
        p.write_bytes(blob)
        assert pkg.model_file_sha256(str(p)) == hashlib.sha256(blob).hexdigest()
    with pytest.raises(RuntimeError):
        pkg.model_file_sha256(str(tmp_path / "missing"))


def test_catalogue_registers_models_and_update_whisper_model_uses_it(pkg, tmp_path):
    blobs = {k: os.urandom(4096 + i) for i, k in enumerate("abc")}
    path, data, config = _catalogue(tmp_path, blobs)
    f = pkg.Filter()
    try:
        assert f.load_catalogue(path, data, config) == 4          # transcription models with a friendly name
        # digests: upper-case hex in the catalogue compares equal; a wrong digest is reported; nothing to compare -> 2
        assert f.verify_model("In data") == 0
        assert f.verify_model("In config") == 1
        assert f.verify_model("No digest") == 2
        assert f.verify_model("Whisper Large v3 q5 (1Gb)") == 2    # listed, not on disk
        assert f.verify_model("A translation model") == 2 and f.verify_model("nameless") == 2
        # update_whisper_model looks names up in it: listed but not downloaded / unknown -> filter left stopped (no GPU needed)
        f.set_setting("whisper_model_path", "Whisper Large v3 q5 (1Gb)")
        assert f.update_model() == 1
        f.set_setting("whisper_model_path", "A translation model")
        assert f.update_model() == 1
        # a catalogue without a data folder falls back to the config folder only
        g = pkg.Filter()
        try:
            assert g.load_catalogue(path, None, config) == 4
            assert g.verify_model("In config") == 1 and g.verify_model("In data") == 2
        finally:
            g.close()
        # malformed input is an error code, not an exception across the ABI
        bad = tmp_path / "bad.json"
        bad.write_text('{"models": [ {"friendly_name": "x", } ]}')
        assert f.load_catalogue(str(bad), data, config) < 0
        bad.write_text('{"no_models": []}')
        assert f.load_catalogue(str(bad), data, config) < 0
        assert f.load_catalogue(str(tmp_path / "missing.json"), data, config) < 0
    finally:
        f.close()
