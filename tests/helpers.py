"""Shared test helpers: build the same seeded synthetic model for the oracle and the HIP path."""
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
