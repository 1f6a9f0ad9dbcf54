"""GPU parity against the COMMITTED golden vectors (tests/golden/, made by scripts/make_golden.py from the CPU
oracle).  Complements the live-oracle comparisons in the other test_gpu_* files: here nothing of the oracle
runs, only its stored outputs are used."""
import json
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

import parity as P
from helpers import gpu_model, mel_windows
from whisper_rs_amd.consts import MODEL_DIMS
from whisper_rs_amd import synth
from whisper_rs_amd.consts import TokenIds

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_log_mel_vs_golden(gpu):
    g = json.load(open(os.path.join(G, "oracle_log_mel.json")))
    mel = gpu.audio.log_mel_spectrogram(synth.make_audio(2, seed=1234))
    rows, cols = np.array(g["rows"])[:, None], np.array(g["cols"])[None, :]
    for i in range(2):
        assert np.abs(mel[i][rows, cols] - np.array(g["samples"][i])).max() < 1e-3
        assert abs(float(mel[i].astype(np.float64).sum()) - g["sum"][i]) < 5.0  # 240k elements, ~2e-5 mean error
        assert abs(mel[i].max() - g["max"][i]) < 1e-3 and abs(mel[i].min() - g["min"][i]) < 1e-3


def test_encoder_vs_golden(gpu):
    g = json.load(open(os.path.join(G, "oracle_encoder_micro.json")))
    enc = gpu_model(gpu, "micro").encoder_forward(mel_windows(1))[0]
    rows, cols = np.array(g["rows"])[:, None], np.array(g["cols"])[None, :]
    assert np.abs(enc[rows, cols] - np.array(g["samples"])).max() < 1e-2  # fp16 activations, unit-scale outputs


@pytest.mark.parametrize("name", ["nano_beam5", "micro_beam5_x2", "micro_greedy_x2", "micro_forced_eot", "micro_prompt",
                                  "micro_beam3_patience2_notimestamps"])
def test_decode_vs_golden(gpu, name):
    c = json.load(open(os.path.join(G, "oracle_decode_cases.json")))[name]
    o = c["options"]
    mode = gpu.TokenExtractMode.Greedy() if o.get("beam_size", 5) is None else gpu.TokenExtractMode.BeamSearch(o.get("beam_size", 5), o.get("patience", 1.0))
    gm = gpu_model(gpu, c["dims"], **c["weights_kw"])
    task = gpu.DecodeTask(gm, TokenIds.english(), gpu.DecodeOptions(sample_len=o["sample_len"], token_extract_mode=mode,
                                                                    timestamps=o.get("timestamps", True), max_audio=c["n_audio"]))
    if c["prompt"] is not None:
        task.set_prompt(c["prompt"])
    group = 1 if o.get("beam_size", 5) is None else o.get("beam_size", 5)
    view = P.OracleView.from_golden(c, MODEL_DIMS[c["dims"]].n_vocab, c["initial_tokens"])
    splits = P.run_with_resync("golden_" + name, task, mel_windows(c["n_audio"]), view, group, TokenIds.english().eot)
    print(f"{name}: steps {task.n_steps}, explained splits {len(splits)}")
    task.close()
