"""CPU: properties of the reference's host algorithms as restated by the oracle (beam selection, prompt
assembly, ranker, timestamp->segment slicing).  The C++ host sequencer is checked against these on the GPU."""
import math
import numpy as np
import pytest
import torch

from helpers import oracle_ids
from oracle import decode as OD, transcribe as OT


def _cands(lp_rows):
    lp = torch.tensor([[v for v, _ in r] for r in lp_rows], dtype=torch.float32)
    tk = torch.tensor([[t for _, t in r] for r in lp_rows], dtype=torch.int64)
    return lp, tk


def test_beam_select_step0_dedups_identical_beams_and_keeps_last_source():
    # all 3 beams identical (first step): 4 unique candidates; sources = last beam (beam_search.rs:77 overwrite)
    rows = [[(-0.1, 7), (-0.5, 8), (-0.9, 9), (-2.0, 10)]] * 3
    slp = torch.zeros(3)
    nt, src, fin = OD.beam_select([[1]] * 3, slp, *_cands(rows), 3, eot=99)
    assert [t[-1] for t in nt] == [7, 8, 9] and src == [2, 2, 2] and fin == [{}]
    np.testing.assert_allclose(slp.numpy(), [-0.1, -0.5, -0.9], atol=1e-7)


def test_beam_select_moves_eot_to_finished_and_refills():
    rows = [[(-0.1, 99), (-0.2, 5), (-0.3, 6)], [(-0.15, 7), (-0.25, 99), (-0.9, 8)]]
    slp = torch.tensor([-1.0, -1.0])
    nt, src, fin = OD.beam_select([[1, 2], [1, 3]], slp, *_cands(rows), 2, eot=99)
    assert [t for t in nt] == [[1, 3, 7], [1, 2, 5]] and src == [1, 0]
    # only EOT candidates ranked above the last kept live beam are recorded (the loop breaks at beam_size, :100-102)
    assert list(fin[0].keys()) == [(1, 2, 99)]
    assert abs(fin[0][(1, 2, 99)] + 1.1) < 1e-6


def test_beam_select_scores_are_f64_sums_stored_as_f32():
    big = 123456.789
    slp = torch.tensor([big], dtype=torch.float32)
    OD.beam_select([[1]], slp, *_cands([[(-0.001, 5), (-3.0, 6)]]), 1, eot=99)
    assert float(slp[0]) == float(np.float32(float(np.float32(big)) + float(np.float32(-0.001))))


def test_decode_prompt():
    ids = oracle_ids()
    p = OD.DecodePrompt(None, ids)
    assert p.initial_tokens.tolist() == [ids.sot] and p.sample_begin == 1 and p.sot_idx == 0
    p = OD.DecodePrompt(torch.tensor([5, 6, 7]), ids)
    assert p.initial_tokens.tolist() == [ids.startofprev, 5, 6, 7, ids.sot] and p.sample_begin == 5 and p.sot_idx == 4


def test_ranker_length_rules():
    toks = [[torch.zeros(4), torch.zeros(2)]]
    assert OD.MaximumLikelihoodRanker(None).rank(toks, [[-4.0, -3.0]]) == [0]         # -1.0 vs -1.5
    assert OD.MaximumLikelihoodRanker(1.0).rank(toks, [[-4.0, -3.0]]) == [1]          # /1.5 vs /(7/6)
    assert OD.MaximumLikelihoodRanker(None).rank(toks, [[-4.0, -2.0]]) == [0]         # tie -> first


def test_round_half_away():
    assert OD._round_half_away(2.5) == 3 and OD._round_half_away(0.5 * 5) == 3 and OD._round_half_away(1.0 / 0.02) == 50


class _FakeDecode:
    """stands in for DecodeTask.run: returns scripted token tensors per window."""
    def __init__(self, outs):
        self.outs, self.prompts = list(outs), []

    def set_prompt(self, p):
        self.prompts.append(p.tolist())

    def run(self, mel):
        t = torch.tensor(self.outs.pop(0), dtype=torch.int64)
        return [OD.DecodeOutput(audio_features=None, tokens=t, avg_logprob=-1.0, no_speech_prob=0.1)]


def test_transcribe_segment_slicing(monkeypatch):
    ids = oracle_ids()
    ts = ids.ts_begin
    t = OT.TranscribeTask.__new__(OT.TranscribeTask)
    from whisper_rs_amd.consts import MODEL_DIMS
    class M: dims = MODEL_DIMS["nano"]
    t.model, t.ids, t.prompt_tokens, t.condition_on_prev_text = M, ids, None, True
    # window 1: <0.00> a b <2.00><2.00> c <4.00>   -> one closed segment, seek advances by 2.00 s = 200 frames
    # window 2: <0.00> d (no closing stamp)        -> single segment of full length, seek += 3000
    t.decode_task = _FakeDecode([[ts, 11, 12, ts + 100, ts + 100, 13, ts + 200], [ts, 14]])
    monkeypatch.setattr(OT.A, "log_mel_spectrogram", lambda a, f: torch.zeros(80, 3100))
    out = t.run(torch.zeros(16000))
    assert [w[0] for w in out.windows] == [0, 200]
    s0, s1 = out.segments
    assert (s0.seek, s0.start_time, s0.end_time, s0.start_token, s0.end_token) == (0, 0.0, 2.0, 1, 4)
    assert (s1.seek, s1.start_token, s1.end_token) == (200, 0, 2) and abs(s1.start_time - 2.0) < 1e-9 and abs(s1.end_time - 32.0) < 1e-9
    assert out.tokens.tolist() == [ts, 11, 12, ts + 100, ts + 100, ts, 14]
    assert t.decode_task.prompts == [[], [ts, 11, 12, ts + 100, ts + 100]]  # conditioning on previous text (:155-158)


def test_oracle_sot_sequence_tail_and_languages():
    """oracle side of BASELINE config 4 (beyond the reference: OpenAI's [sot, lang, task] prompt and per-window language
    tokens): prompt layout, language override and detection are self-consistent on a multilingual-vocabulary model."""
    import torch
    from helpers import oracle_model, oracle_ids, oracle_decode_options, mel_windows
    from whisper_rs_amd.consts import MODEL_DIMS, TokenIds
    from oracle import decode as OD
    ids = TokenIds.multilingual()
    oids = oracle_ids(MODEL_DIMS["micro.ml"].n_vocab)
    first = ids.sot + 1
    p = OD.DecodePrompt(torch.tensor([7, 8]), oids, (first + 3, ids.transcribe))
    assert p.initial_tokens.tolist() == [ids.startofprev, 7, 8, ids.sot, first + 3, ids.transcribe]
    assert p.sample_begin == 6 and p.sot_idx == 3
    om = oracle_model("micro.ml")
    mel = torch.from_numpy(mel_windows(2))
    task = OD.DecodeTask(om, oids, oracle_decode_options(sample_len=5, beam_size=2, sot_sequence_tail=(first, ids.transcribe)))
    lang, probs = task.detect_language(mel, first, 99)
    assert probs.shape == (2, 99) and torch.allclose(probs.sum(-1), torch.ones(2), atol=1e-5)
    assert all(first <= int(x) < first + 99 for x in lang) and torch.equal(lang - first, probs.argmax(-1))
    base = task.run_batch(mel)
    task.languages = [first + 10, first]      # window 0 switches language, window 1 keeps the default of the tail
    other = task.run_batch(mel)
    assert torch.equal(base[1].tokens, other[1].tokens) and abs(base[1].sum_logprob - other[1].sum_logprob) < 1e-5
    assert not (torch.equal(base[0].tokens, other[0].tokens) and base[0].sum_logprob == other[0].sum_logprob)
