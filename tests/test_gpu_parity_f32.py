""""Token-id outputs bit-exact" as a demonstrated statement (BASELINE.json north_star).

The product path multiplies in fp16, and on random-weight models two beam candidates are occasionally closer than that
rounding noise (tests/parity.py prices every such split in the oracle's own scores).  Here the SAME decode loop, filters,
top-k, beam search, ancestry table and ranking run with the model in f32 parity mode (csrc/parity_f32.hip: fp32 weights,
activations, K/V caches and products on v_mfma_f32_16x16x4_f32) and must reproduce the fp32 CPU oracle's token ids AND
beam source indices at every step of a FREE-RUNNING decode, to the last step, in every golden case and in the
live-oracle cases below: no split is accepted (exact=True), candidate log-probs agree to 2e-4.  Whatever differs in the
fp16 mode is therefore fp16 rounding and nothing else.  Reference semantics: decode/extractor/beam_search.rs:22-206,
decode/mod.rs:220-418."""
import json
import os

import numpy as np
import pytest
import torch

import parity as P
from helpers import weights, oracle_decode_options, mel_windows
from test_gpu_decode import _oracle_run
from whisper_rs_amd import debug as DBG
from whisper_rs_amd.consts import MODEL_DIMS, TokenIds

pytestmark = pytest.mark.gpu
G_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
F32_CAND_TOL = 2e-4   # fp32 device vs fp32 oracle log-probs (different summation orders)

_models = {}


def _f32_model(dims_name, **kw):
    key = (dims_name, tuple(sorted(kw.items())))
    if key not in _models:
        _models[key] = DBG.parity_f32_model(MODEL_DIMS[dims_name], weights(dims_name, 0, **kw))
    return _models[key]


@pytest.mark.parametrize("name", ["nano_beam5", "micro_beam5_x2", "micro_greedy_x2", "micro_forced_eot", "micro_prompt",
                                  "micro_beam3_patience2_notimestamps"])
def test_f32_mode_reproduces_every_golden_case_exactly(gpu, name):
    c = json.load(open(os.path.join(G_DIR, "oracle_decode_cases.json")))[name]
    o = c["options"]
    mode = gpu.TokenExtractMode.Greedy() if o.get("beam_size", 5) is None else gpu.TokenExtractMode.BeamSearch(o.get("beam_size", 5), o.get("patience", 1.0))
    task = gpu.DecodeTask(_f32_model(c["dims"], **c["weights_kw"]), TokenIds.english(),
                          gpu.DecodeOptions(sample_len=o["sample_len"], token_extract_mode=mode, timestamps=o.get("timestamps", True), max_audio=c["n_audio"]))
    if c["prompt"] is not None:
        task.set_prompt(c["prompt"])
    group = 1 if o.get("beam_size", 5) is None else o.get("beam_size", 5)
    view = P.OracleView.from_golden(c, MODEL_DIMS[c["dims"]].n_vocab, c["initial_tokens"])
    splits = P.run_with_resync("f32_golden_" + name, task, mel_windows(c["n_audio"]), view, group, TokenIds.english().eot,
                               exact=True, cand_tol=F32_CAND_TOL)
    assert splits == []
    task.close()


@pytest.mark.parametrize("dims_name,n_audio,sample_len,beam", [("micro", 3, 40, 5), ("micro", 2, 64, None), ("nano", 2, 48, 5), ("tiny.en", 1, 32, 5)])
def test_f32_mode_free_running_vs_live_oracle(gpu, dims_name, n_audio, sample_len, beam):
    mel = mel_windows(n_audio)
    view, _ = _oracle_run(dims_name, mel, oracle_decode_options(sample_len=sample_len, beam_size=beam))
    mode = gpu.TokenExtractMode.Greedy() if beam is None else gpu.TokenExtractMode.BeamSearch(beam, 1.0)
    task = gpu.DecodeTask(_f32_model(dims_name), TokenIds.english(), gpu.DecodeOptions(sample_len=sample_len, max_audio=n_audio, token_extract_mode=mode))
    splits = P.run_with_resync(f"f32_{dims_name}_x{n_audio}_beam{beam}", task, mel, view, beam or 1, TokenIds.english().eot,
                               exact=True, cand_tol=F32_CAND_TOL)
    assert splits == []
    task.close()


def test_f32_mode_encoder_matches_oracle_tightly(gpu):
    from helpers import oracle_model
    mel = mel_windows(1)
    want = oracle_model("micro").encoder.forward(torch.from_numpy(mel)).numpy()
    got = _f32_model("micro").encoder_forward(mel)
    err = np.abs(got - want).max()
    print(f"f32 mode encoder: max abs err {err:.2e}")
    assert err < 2e-4
