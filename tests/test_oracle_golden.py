"""CPU: pin the oracle.  (1) against the reference's own fixtures; (2) against the committed golden vectors
(regression guard: the GPU tests compare the HIP path with the same oracle / vectors)."""
import dataclasses
import json
import os

import numpy as np
import pytest
import torch

from helpers import oracle_model, oracle_ids, oracle_decode_options, mel_windows
from whisper_rs_amd import synth
from whisper_rs_amd.consts import MODEL_DIMS
from oracle import audio as OA, decode as OD, transcribe as OT

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_mel_filter_matches_reference_fixture():
    """the reference's own known-answer test (audio.rs:266-279): mel_filter()[:8,:8] vs test/data/mel-filter-8x8.csv,
    allclose(rtol 1e-5, atol 1e-8)."""
    gold = np.loadtxt(os.path.join(G, "mel-filter-8x8.csv"), delimiter=",", dtype=np.float32)
    f = OA.mel_filter().numpy()
    assert f.shape == (80, 201) and f.dtype == np.float32
    np.testing.assert_allclose(f[:8, :8], gold, rtol=1e-5, atol=1e-8)
    assert int((f != 0).sum()) == 391  # banded: SURVEY.md K4


def test_log_mel_invariants_of_reference_fixture():
    """test/data/mel-spectrogram.csv cannot be reproduced (its input WAV is not shipped), but its invariants pin the
    post-processing of audio.rs:254-258: dynamic range exactly 8/4 = 2.0 and shape 80 x n/160."""
    st = json.load(open(os.path.join(G, "reference_mel_spectrogram_stats.json")))
    assert st["shape"][0] == 80
    assert abs((st["max"] - st["min"]) - 2.0) < 1e-6
    a = synth.make_audio(1, seed=5, n_samples=st["shape"][1] * 160)[0]
    a[:4000] = 0.0  # a silent stretch so the floor (max - 8) is reached, as in real speech
    m = OA.log_mel_spectrogram(torch.from_numpy(a), OA.mel_filter()).numpy()
    assert list(m.shape) == st["shape"]
    assert abs((m.max() - m.min()) - 2.0) < 1e-5


def test_pad_or_trim():
    x = torch.arange(12.0).reshape(3, 4)
    assert torch.equal(OA.pad_or_trim(x, 2), x[:, :2])
    assert torch.equal(OA.pad_or_trim(x, 6)[:, 4:], torch.zeros(3, 2))
    assert torch.equal(OA.pad_or_trim(x, 4), x)


def test_oracle_log_mel_golden():
    g = json.load(open(os.path.join(G, "oracle_log_mel.json")))
    a = synth.make_audio(2, seed=1234)
    rows, cols = np.array(g["rows"])[:, None], np.array(g["cols"])[None, :]
    for i in range(2):
        m = OA.log_mel_spectrogram(torch.from_numpy(a[i]), OA.mel_filter()).numpy()
        np.testing.assert_allclose(m[rows, cols], np.array(g["samples"][i]), atol=2e-5)
        assert abs(float(m.astype(np.float64).sum()) - g["sum"][i]) < 0.5


def test_oracle_encoder_golden():
    g = json.load(open(os.path.join(G, "oracle_encoder_micro.json")))
    enc = oracle_model("micro").encoder.forward(torch.from_numpy(mel_windows(1))).numpy()[0]
    rows, cols = np.array(g["rows"])[:, None], np.array(g["cols"])[None, :]
    np.testing.assert_allclose(enc[rows, cols], np.array(g["samples"]), atol=2e-4)


def _run_case(c):
    m = oracle_model(c["dims"], **c["weights_kw"])
    t = OD.DecodeTask(m, oracle_ids(MODEL_DIMS[c["dims"]].n_vocab), oracle_decode_options(**c["options"]))
    if c["prompt"] is not None:
        t.set_prompt(torch.tensor(c["prompt"], dtype=torch.int64))
    return t, t.run_batch(torch.from_numpy(mel_windows(c["n_audio"])))


@pytest.mark.parametrize("name", ["nano_beam5", "micro_beam5_x2", "micro_greedy_x2", "micro_forced_eot", "micro_prompt",
                                  "micro_beam3_patience2_notimestamps"])
def test_oracle_decode_golden(name):
    c = json.load(open(os.path.join(G, "oracle_decode_cases.json")))[name]
    t, outs = _run_case(c)
    assert [o.tokens.tolist() for o in outs] == c["tokens"]
    np.testing.assert_allclose([o.sum_logprob for o in outs], c["sum_logprob"], atol=2e-3)
    np.testing.assert_allclose([o.no_speech_prob for o in outs], c["no_speech_prob"], rtol=1e-3, atol=1e-7)
    assert len(t.extractor.trace) == c["n_steps"]
    assert [x[0].tolist() for x in t.extractor.trace] == c["trace_tokens"]
    assert [x[1].tolist() for x in t.extractor.trace] == c["trace_sources"]


def test_forced_eot_case_really_finishes_early():
    c = json.load(open(os.path.join(G, "oracle_decode_cases.json")))["micro_forced_eot"]
    assert c["n_steps"] < c["options"]["sample_len"]
    assert all(len(t) < c["options"]["sample_len"] for t in c["tokens"])


@pytest.mark.parametrize("cond", [False, True])
def test_oracle_transcribe_golden(cond):
    g = json.load(open(os.path.join(G, "oracle_transcribe_nano.json")))["condition_on_prev_text"][str(cond)]
    audio = synth.make_audio(1, seed=77, n_samples=16000 * 50)[0]
    r = OT.TranscribeTask(oracle_model("nano"), oracle_ids(), oracle_decode_options(sample_len=10), None, cond).run(torch.from_numpy(audio))
    assert r.tokens.tolist() == g["tokens"]
    assert [w[0] for w in r.windows] == g["seeks"]
    assert [dataclasses.asdict(s) for s in r.segments] == g["segments"]
    assert len(g["seeks"]) >= 2
