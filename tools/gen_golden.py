"""Generate tests/golden/*.npz — run ONCE in the build container (needs `transformers`, CPU only).

The reference holds no golden vectors for the Whisper path and whisper.cpp itself is
not in the container, so the oracle is pinned against an INDEPENDENT implementation of the same
architecture: HuggingFace transformers' Whisper (random seeded weights, `gelu_new` = the tanh GELU
ggml uses).  Everything stored is data (inputs are re-derived from seeds by tools/ggml_io.py):

  mel_*.npz      WhisperFeatureExtractor log-mel of seeded PCM          (pins oracle A.2)
  enc_*.npz      encoder output rows + statistics                        (pins oracle A.3)
  dec_*.npz      teacher-forced decoder logits (selected columns)        (pins oracle A.4)
  rules_*.npz    WhisperTimeStampLogitsProcessor suppression masks       (pins oracle A.5 steps 7-11)

Usage: python tools/gen_golden.py  (writes into tests/golden/)
"""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HF_HUB_OFFLINE", "1")

import torch  # noqa: E402
from transformers import WhisperConfig, WhisperFeatureExtractor, WhisperForConditionalGeneration  # noqa: E402
from transformers.generation.logits_process import WhisperTimeStampLogitsProcessor  # noqa: E402

from tools import ggml_io  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def hf_model(hp: ggml_io.HParams, tensors) -> WhisperForConditionalGeneration:
    cfg = WhisperConfig(
        vocab_size=hp.n_vocab, num_mel_bins=hp.n_mels, d_model=hp.n_audio_state,
        encoder_layers=hp.n_audio_layer, encoder_attention_heads=hp.n_audio_head,
        decoder_layers=hp.n_text_layer, decoder_attention_heads=hp.n_text_head,
        encoder_ffn_dim=4 * hp.n_audio_state, decoder_ffn_dim=4 * hp.n_text_state,
        max_source_positions=hp.n_audio_ctx, max_target_positions=hp.n_text_ctx,
        activation_function="gelu_new", dropout=0.0, attention_dropout=0.0, activation_dropout=0.0,
        pad_token_id=50256, bos_token_id=50257, eos_token_id=50256, decoder_start_token_id=50257,
        attn_implementation="eager")
    m = WhisperForConditionalGeneration(cfg).eval()
    sd = {}

    def put(dst, src):
        sd[dst] = torch.from_numpy(np.ascontiguousarray(tensors[src]))

    put("model.encoder.conv1.weight", "encoder.conv1.weight")
    sd["model.encoder.conv1.bias"] = torch.from_numpy(tensors["encoder.conv1.bias"].reshape(-1).copy())
    put("model.encoder.conv2.weight", "encoder.conv2.weight")
    sd["model.encoder.conv2.bias"] = torch.from_numpy(tensors["encoder.conv2.bias"].reshape(-1).copy())
    put("model.encoder.embed_positions.weight", "encoder.positional_embedding")
    amap = {"query": "q_proj", "key": "k_proj", "value": "v_proj", "out": "out_proj"}
    for i in range(hp.n_audio_layer):
        s, d = f"encoder.blocks.{i}.", f"model.encoder.layers.{i}."
        for wb in ("weight", "bias"):
            put(d + "self_attn_layer_norm." + wb, s + "attn_ln." + wb)
            put(d + "final_layer_norm." + wb, s + "mlp_ln." + wb)
            put(d + "fc1." + wb, s + "mlp.0." + wb)
            put(d + "fc2." + wb, s + "mlp.2." + wb)
            for a, b in amap.items():
                if a == "key" and wb == "bias":
                    continue
                put(d + f"self_attn.{b}." + wb, s + f"attn.{a}." + wb)
    for wb in ("weight", "bias"):
        put("model.encoder.layer_norm." + wb, "encoder.ln_post." + wb)
        put("model.decoder.layer_norm." + wb, "decoder.ln." + wb)
    put("model.decoder.embed_positions.weight", "decoder.positional_embedding")
    put("model.decoder.embed_tokens.weight", "decoder.token_embedding.weight")
    sd["proj_out.weight"] = sd["model.decoder.embed_tokens.weight"]
    for i in range(hp.n_text_layer):
        s, d = f"decoder.blocks.{i}.", f"model.decoder.layers.{i}."
        for wb in ("weight", "bias"):
            put(d + "self_attn_layer_norm." + wb, s + "attn_ln." + wb)
            put(d + "encoder_attn_layer_norm." + wb, s + "cross_attn_ln." + wb)
            put(d + "final_layer_norm." + wb, s + "mlp_ln." + wb)
            put(d + "fc1." + wb, s + "mlp.0." + wb)
            put(d + "fc2." + wb, s + "mlp.2." + wb)
            for a, b in amap.items():
                if a == "key" and wb == "bias":
                    continue
                put(d + f"self_attn.{b}." + wb, s + f"attn.{a}." + wb)
                put(d + f"encoder_attn.{b}." + wb, s + f"cross_attn.{a}." + wb)
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("k_proj.bias" in k for k in missing), missing
    return m


# columns of the logits row kept in fixtures (full rows would be 200 KB each)
def logit_columns(n_vocab: int) -> np.ndarray:
    rs = np.random.RandomState(99)
    cols = np.unique(np.concatenate([rs.randint(0, n_vocab, 500), np.arange(50250, min(50400, n_vocab)),
                                     np.arange(n_vocab - 64, n_vocab)]))
    return cols.astype(np.int64)


def main():
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)

    # ---- mel: HF feature extractor on seeded PCM (80 and 128 bins; 30 s, 3 s, 0.9 s clips) ----
    for n_mels in (80, 128):
        fe = WhisperFeatureExtractor(feature_size=n_mels)
        filt_hf = np.asarray(fe.mel_filters, dtype=np.float64).T  # [n_mels, 201]
        filt_own = ggml_io.mel_filterbank(n_mels)
        assert np.abs(filt_hf - filt_own).max() < 1e-6, np.abs(filt_hf - filt_own).max()
        for seed, secs in ((0, 30.0), (1, 3.0), (2, 0.9)):
            pcm = ggml_io.synth_pcm(seed, secs)
            feats = fe(pcm, sampling_rate=16000, return_tensors="np")["input_features"][0]  # [n_mels, 3000]
            n_valid = int(len(pcm) // 160)  # frames whose window never touches HF's right reflect pad
            np.savez_compressed(os.path.join(GOLD, f"mel_{n_mels}_seed{seed}.npz"),
                                seed=seed, seconds=secs, n_valid=n_valid,
                                mel=feats[:, ::7].astype(np.float32),  # every 7th frame keeps the file small
                                stride=7)
            print("mel", n_mels, seed, secs, feats.shape)

    # ---- encoder / decoder: micro models with HF forward ----
    for preset in ("micro.en", "micro"):
        hp = ggml_io.PRESETS[preset]
        tensors = ggml_io.f16_round(ggml_io.synthetic_tensors(hp, seed=1234), hp)
        model = hf_model(hp, tensors)
        fe = WhisperFeatureExtractor(feature_size=hp.n_mels)
        pcm = ggml_io.synth_pcm(0, 30.0)
        feats = fe(pcm, sampling_rate=16000, return_tensors="pt")["input_features"]  # [1, n_mels, 3000]
        with torch.no_grad():
            enc = model.model.encoder(feats).last_hidden_state  # [1, 1500, d]
            rs = np.random.RandomState(7)
            sot = 50257 if hp.n_vocab < 51865 else 50258
            toks = [sot] + [int(t) for t in rs.randint(0, 50000, size=11)]
            dec_in = torch.tensor([toks], dtype=torch.long)
            out = model(input_features=feats, decoder_input_ids=dec_in)
            logits = out.logits[0].numpy()  # [12, n_vocab]
        enc = enc[0].numpy()
        cols = logit_columns(hp.n_vocab)
        np.savez_compressed(os.path.join(GOLD, f"encdec_{preset}.npz"),
                            mel_in=feats[0].numpy()[:, ::50].astype(np.float32),
                            enc_rows=enc[::25].astype(np.float32), enc_mean=enc.mean(), enc_std=enc.std(),
                            tokens=np.asarray(toks, np.int32), cols=cols,
                            logits=logits[:, cols].astype(np.float32),
                            argmax=logits.argmax(-1).astype(np.int32),
                            logits_max=logits.max(-1).astype(np.float32))
        print(preset, "enc", enc.shape, float(enc.mean()), float(enc.std()), "logits", logits.shape)

    # ---- timestamp rules: HF processor masks on seeded random logits ----
    for n_vocab, beg, eot, not_ in ((51864, 50363, 50256, 50362), (51866, 50365, 50257, 50364)):
        class G:  # minimal generation-config stand-in
            pass
        g = G()
        g.eos_token_id = eot
        g.no_timestamps_token_id = not_
        g.max_initial_timestamp_index = 50
        g.forced_decoder_ids = None
        begin_index = 1
        proc = WhisperTimeStampLogitsProcessor(g, begin_index=begin_index)
        hists = [
            [],                                   # initial
            [beg + 10, 500, 600],                 # ts, text, text
            [beg + 10, 500, beg + 40],            # closing timestamp: must be followed by a timestamp
            [beg + 10, 500, beg + 40, beg + 40],  # pair done: text next
            [700, 800],                           # no timestamps yet
            [beg + 5],                            # single leading timestamp
        ]
        masks, ts_forced = [], []
        for hi, h in enumerate(hists):
            rs = np.random.RandomState(1000 + hi)
            logits = rs.standard_normal(n_vocab).astype(np.float32) * 3.0
            if hi == 4:
                logits[beg:] += 4.0   # make the timestamp mass dominate -> text suppressed
            ids = torch.tensor([[eot + 1] + h], dtype=torch.long)
            sc = proc(ids, torch.from_numpy(logits.copy())[None])[0].numpy()
            masks.append(np.packbits(np.isneginf(sc)))
            ts_forced.append(bool(np.isneginf(sc[:beg]).all()))
        np.savez_compressed(os.path.join(GOLD, f"rules_{n_vocab}.npz"), n_vocab=n_vocab, beg=beg, eot=eot,
                            hists=np.asarray([h + [-1] * (4 - len(h)) for h in hists], np.int32),
                            masks=np.stack(masks), ts_forced=np.asarray(ts_forced))
        print("rules", n_vocab, ts_forced)


if __name__ == "__main__":
    main()
