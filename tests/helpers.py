"""Shared test helpers: build the same seeded synthetic model for the oracle and the HIP path."""
import contextlib
import dataclasses
import functools

import numpy as np
import torch

from whisper_rs_amd import synth
from whisper_rs_amd.consts import MODEL_DIMS, TokenIds, NON_SPEECH_TOKENS

torch.set_num_threads(8)


@functools.lru_cache(maxsize=8)
def weights(dims_name, seed=0, emb_gain=1.0, force_eot_after=None):
    return synth.make_weights(MODEL_DIMS[dims_name], seed=seed, emb_gain=emb_gain, force_eot_after=force_eot_after)


def oracle_model(dims_name, seed=0, **kw):
    from oracle import model as OM
    return OM.Whisper(weights(dims_name, seed, **kw), MODEL_DIMS[dims_name])


_gpu_models = {}


def gpu_model(W, dims_name, seed=0, **kw):
    key = (dims_name, seed, tuple(sorted(kw.items())))
    if key not in _gpu_models:
        _gpu_models[key] = W.Whisper(MODEL_DIMS[dims_name], weights(dims_name, seed, **kw))
    return _gpu_models[key]


@contextlib.contextmanager
def switched_models(monkeypatch, **env):
    """The library reads its experiment switches (WRS_DEC_PRING, WRS_DEC_FUSED, WRS_SELF_MFMA_MIN_AUDIO, WRS_DEC_BIG_GEMM_ROWS ...)
    when a MODEL is created: inside this context every gpu_model() is built afresh under the given environment, and dropped
    again on the way out so later tests get default-path models."""
    saved = dict(_gpu_models)
    _gpu_models.clear()
    for k, v in env.items():
        monkeypatch.setenv(k, str(v))
    try:
        yield
    finally:
        for m in _gpu_models.values():
            m.close()
        _gpu_models.clear()
        _gpu_models.update(saved)
        for k in env:
            monkeypatch.delenv(k, raising=False)


def oracle_ids(n_vocab=51864):
    from oracle import decode as OD
    return OD.TokenIds(**dataclasses.asdict(TokenIds.for_vocab(n_vocab)))


def oracle_decode_options(**kw):
    from oracle import decode as OD
    kw.setdefault("non_speech_tokens", NON_SPEECH_TOKENS)
    return OD.DecodeOptions(**kw)


@functools.lru_cache(maxsize=8)
def mel_windows(n, seed=1234):
    from oracle import audio as OA
    a = synth.make_audio(n, seed=seed)
    f = OA.mel_filter()
    return np.stack([OA.log_mel_spectrogram(torch.from_numpy(a[i]), f).numpy() for i in range(n)])


@functools.lru_cache(maxsize=1)
def synthetic_tokenizer_json(n_base=50256, seed=3):
    """A full-size byte-level BPE tokenizer.json built without any corpus or download: the 256 byte symbols plus
    pseudo-random merges up to n_base ids, then <|endoftext|> — the same id layout as the reference's gpt2.json
    (specials start at 50257, timestamps at 50363), so model outputs decode to *some* text on machines that do not
    have the reference's vocabulary (the GPU box)."""
    import json
    import random
    direct = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAD)) + list(range(0xAE, 0x100))
    chars, nxt = {}, 256
    for b in range(256):
        if b in direct:
            chars[b] = chr(b)
        else:
            chars[b] = chr(nxt)
            nxt += 1
    toks = [chars[b] for b in range(256)]
    have = set(toks)
    rng = random.Random(seed)
    merges = []
    while len(toks) < n_base:
        l, r = toks[rng.randrange(len(toks))], toks[rng.randrange(len(toks))]
        if len(l) + len(r) <= 12 and l + r not in have:
            have.add(l + r)
            toks.append(l + r)
            merges.append(f"{l} {r}")
    vocab = {t: i for i, t in enumerate(toks)}
    vocab["<|endoftext|>"] = n_base
    doc = {"version": "1.0", "truncation": None, "padding": None,
           "added_tokens": [{"id": n_base, "content": "<|endoftext|>", "single_word": False, "lstrip": False, "rstrip": False,
                             "normalized": True, "special": True}],
           "normalizer": None,
           "pre_tokenizer": {"type": "ByteLevel", "add_prefix_space": False, "trim_offsets": True, "use_regex": True},
           "post_processor": {"type": "ByteLevel", "add_prefix_space": True, "trim_offsets": False, "use_regex": True},
           "decoder": {"type": "ByteLevel", "add_prefix_space": True, "trim_offsets": True, "use_regex": True},
           "model": {"type": "BPE", "dropout": None, "unk_token": None, "continuing_subword_prefix": "", "end_of_word_suffix": "",
                     "fuse_unk": False, "vocab": vocab, "merges": merges}}
    return json.dumps(doc, ensure_ascii=False).encode("utf-8")
