"""GPU parity at the configurations BASELINE.json names (SURVEY.md section 8(d) "configs restated").

  config 2  tiny.en, greedy, 32 windows                      -> free-running comparison with the oracle
  config 3  base.en, beam 5 + ApplyTimestampRules, 16 windows -> teacher-forced comparison with the oracle
  config 4  small multilingual dims (V = 51865): language detection + [sot, lang, task] prompts, beam 5
  config 5  large-v2, beam 5: the oracle cannot run this size in seconds, so the full-size checks are the
            size-independent properties the domain offers: run-to-run determinism, batch invariance (a window
            decodes the same alone and inside a batch), and the invariants ApplyTimestampRules / SuppressTokens /
            SuppressBlank impose on ANY output sequence (timestamp_tokens.rs:33-97, suppress_*.rs).

The oracle's language detection and sot-sequence handling follow OpenAI whisper (the reference declares
Task::LanguageId but does not implement it): parity unpinned by the reference for config 4."""
import os
import wave

import numpy as np
import pytest
import torch

import parity as P
from helpers import gpu_model, oracle_model, oracle_ids, oracle_decode_options, mel_windows
from test_gpu_decode import _oracle_run, _free_run, _teacher_forced_check, LP_TOL
from whisper_rs_amd import synth
from whisper_rs_amd.consts import MODEL_DIMS, TokenIds, NON_SPEECH_TOKENS
from oracle import decode as OD, transcribe as OT

pytestmark = pytest.mark.gpu


def test_config1_tiny_en_greedy_wav_file(gpu, tmp_path):
    """BASELINE config 1: the reference example's flow (examples/transcribe.rs:19-71) with tiny.en dims and greedy
    selection on ONE 30-s 16 kHz WAV file: file -> load_audio -> TranscribeTask::run (log-mel, encoder, decode loop,
    segments), against the oracle's TranscribeTask on the very samples the loader returned."""
    audio = synth.make_audio(1, seed=4242)[0]
    path = tmp_path / "window.wav"
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
        w.writeframes(np.clip(np.round(audio * 32767), -32768, 32767).astype("<i2").tobytes())
    samples = gpu.audio.load_audio(path)                       # audio::load_audio (audio.rs:20-173)
    assert samples.shape == (480000,) and samples.dtype == np.float32
    np.testing.assert_allclose(samples, np.clip(np.round(audio * 32767), -32768, 32767) / 32768.0, atol=1e-7)
    sample_len = 64
    ot = OT.TranscribeTask(oracle_model("tiny.en"), oracle_ids(), oracle_decode_options(sample_len=sample_len, beam_size=None), None, True)
    want = ot.run(torch.from_numpy(samples))
    gt = gpu.TranscribeTask(gpu_model(gpu, "tiny.en"), gpu.TranscribeOptions(
        decode=gpu.DecodeOptions(sample_len=sample_len, token_extract_mode=gpu.TokenExtractMode.Greedy()),
        prompt=gpu.TranscribePrompt.None_(True)))               # examples/transcribe.rs:58-60
    got = gt.run(samples)
    gt.close()
    seeks_equal = [w_[0] for w_ in got.windows] == [w_[0] for w_ in want.windows]
    if not (seeks_equal and np.array_equal(got.tokens, want.tokens.numpy())):
        # A free-running greedy decode: ONE near-tie argmax (fp16 products against the fp32 oracle) changes every later token and,
        # through the timestamps, the seek sequence (seen in round 3: a 1-ulp change of the GELU's reciprocal moved window 0's
        # last timestamp from 1980 to 658).  Both loops start at window 0 with no prompt, so the window-level machinery shows the
        # first divergence there: P.run_with_resync raises unless it is a near-tie priced in the ORACLE's own scores, and requires
        # the device to follow the oracle to the last step once re-synchronised.
        # Each side decodes ITS OWN log-mel, as the two loops did (they agree to 1e-3, test_gpu_mel.py): the candidate errors
        # measured in this run — which bound the accepted margin — then include what the spectrogram difference does to the scores.
        from oracle import audio as OA
        omel = OA.log_mel_spectrogram(torch.from_numpy(samples), OA.mel_filter()).numpy()[None]
        gmel = gpu.audio.log_mel_spectrogram(samples[None])
        assert np.abs(gmel - omel).max() < 1e-3
        view, _ = _oracle_run("tiny.en", omel, oracle_decode_options(sample_len=sample_len, beam_size=None))
        task = gpu.DecodeTask(gpu_model(gpu, "tiny.en"), TokenIds.english(), gpu.DecodeOptions(sample_len=sample_len, token_extract_mode=gpu.TokenExtractMode.Greedy()))
        splits = _free_run("config1_tiny_en_greedy", task, gmel, view, 1)
        task.close()
        assert splits, "transcribe output differs from the oracle but the decode of window 0 shows no near-tie"
        # the loop's own bookkeeping on the device's windows (transcribe.rs:151-234; bit-exact against the oracle on scripted
        # windows in test_host_logic_cpu.py): seeks strictly increase inside the recording, every segment belongs to a window
        seeks = [w_[0] for w_ in got.windows]
        assert seeks[0] == 0 and all(b > a for a, b in zip(seeks, seeks[1:])) and seeks[-1] < 3000
        # (segment times are window offset + timestamp * 0.02 s; the reference's timestamp rules do not order a pair, timestamp_tokens.rs:47-78)
        assert all(sg.seek in seeks and min(sg.start_time, sg.end_time) >= sg.seek * 0.01 - 1e-9 for sg in got.segments)
        print(f"config 1: explained near-tie split(s) {[(r['step'], round(r['margin'], 5)) for r in splits]}")
    else:
        assert len(got.segments) == len(want.segments)
        for g, o in zip(got.segments, want.segments):
            assert (g.seek, g.start_token, g.end_token) == (o.seek, o.start_token, o.end_token)
            assert abs(g.start_time - o.start_time) < 1e-9 and abs(g.end_time - o.end_time) < 1e-9
    print(f"config 1: {len(got.tokens)} tokens, {len(got.segments)} segments, windows {[w_[0] for w_ in got.windows]}")


def test_config2_tiny_en_greedy_batch32(gpu):
    n = 32
    mel = mel_windows(n)
    view, _ = _oracle_run("tiny.en", mel, oracle_decode_options(sample_len=64, beam_size=None))
    gm = gpu_model(gpu, "tiny.en")
    task = gpu.DecodeTask(gm, TokenIds.english(), gpu.DecodeOptions(sample_len=64, max_audio=n, token_extract_mode=gpu.TokenExtractMode.Greedy()))
    splits = _free_run("config2_tiny_en_greedy_x32", task, mel, view, 1)
    print("config 2: steps", task.n_steps, "near-tie splits", len(splits))
    task.close()


def test_config3_base_en_beam5_timestamps_batch16(gpu):
    exact, steps = _teacher_forced_check(gpu, "base.en", 16, 64, 0.95)
    print(f"config 3: {exact}/{steps} selections reproduced exactly")


def test_config3_base_en_beam5_free_running(gpu):
    """the same configuration free-running on 4 windows (every split explained and re-synchronised)."""
    mel = mel_windows(4)
    view, _ = _oracle_run("base.en", mel, oracle_decode_options(sample_len=64, beam_size=5))
    task = gpu.DecodeTask(gpu_model(gpu, "base.en"), TokenIds.english(), gpu.DecodeOptions(sample_len=64, max_audio=4))
    _free_run("config3_base_en_beam5_x4", task, mel, view, 5)
    task.close()


def test_config4_multilingual_language_detection_and_decode(gpu):
    dims_name = "micro.ml"
    ids = TokenIds.multilingual()
    first_lang, n_lang = ids.sot + 1, 99
    n = 3
    mel = mel_windows(n)
    om = oracle_model(dims_name)
    oids = oracle_ids(MODEL_DIMS[dims_name].n_vocab)
    oopts = oracle_decode_options(sample_len=64, beam_size=5, sot_sequence_tail=(first_lang, ids.transcribe))
    ot = OD.DecodeTask(om, oids, oopts)
    want_lang, want_probs = ot.detect_language(torch.from_numpy(mel), first_lang, n_lang)
    gm = gpu_model(gpu, dims_name)
    task = gpu.DecodeTask(gm, ids, gpu.DecodeOptions(sample_len=64, max_audio=n, sot_sequence_tail=[first_lang, ids.transcribe]))
    got_lang, got_probs = task.detect_language(mel, first_lang, n_lang)
    np.testing.assert_allclose(got_probs.sum(axis=1), 1.0, atol=1e-5)
    np.testing.assert_allclose(got_probs, want_probs.numpy(), atol=2e-3)
    top2 = np.sort(want_probs.numpy(), axis=1)[:, -2:]
    for a in range(n):  # ids must agree unless the oracle's two best languages are closer than the fp16 noise
        if top2[a, 1] - top2[a, 0] > 4e-3:
            assert got_lang[a] == int(want_lang[a])
    # decode with per-window language tokens
    langs = [int(x) for x in want_lang]
    view, _ = _oracle_run(dims_name, mel, oopts, languages=langs)
    task.set_languages(langs)
    _free_run("config4_micro_ml_langs", task, mel, view, 5)
    # windows whose encoder output is still resident decode the same through run_encoded as through the mel entry
    g1 = task.run(mel)
    g2 = task.run_encoded(n)
    for a, b in zip(g1, g2):
        np.testing.assert_array_equal(a.tokens, b.tokens)
    # with no per-window override the tail's own language token is used
    task.set_languages(None)
    view, _ = _oracle_run(dims_name, mel, oopts, languages=None)
    _free_run("config4_micro_ml_default_lang", task, mel, view, 5)
    task.close()


def test_config4_small_dims_language_detection_and_decode(gpu):
    """the real `small` shape (D 768, 12 heads, 12 layers, V 51865) on 2 windows: detection probabilities vs the oracle,
    then 64 teacher-forced beam-5 steps with the [sot, lang, transcribe] prompt."""
    dims_name = "small"
    ids = TokenIds.multilingual()
    first_lang = ids.sot + 1
    mel = mel_windows(2)
    ot = OD.DecodeTask(oracle_model(dims_name), oracle_ids(MODEL_DIMS[dims_name].n_vocab), oracle_decode_options(sample_len=4, beam_size=5))
    want_lang, want_probs = ot.detect_language(torch.from_numpy(mel), first_lang, 99)
    task = gpu.DecodeTask(gpu_model(gpu, dims_name), ids, gpu.DecodeOptions(sample_len=4, max_audio=2))
    got_lang, got_probs = task.detect_language(mel, first_lang, 99)
    np.testing.assert_allclose(got_probs, want_probs.numpy(), atol=3e-3)
    task.close()
    tail = (first_lang + 3, ids.transcribe)
    exact, steps = _teacher_forced_check(gpu, dims_name, 2, 64, 0.95, ids=ids, okw=dict(sot_sequence_tail=tail), gkw=dict(sot_sequence_tail=list(tail)))
    print(f"config 4 (small dims): {exact}/{steps} selections reproduced exactly")


# ------------------------------------------------------------------------------------------------ config 5
@pytest.fixture(scope="module")
def large_v2(gpu):
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import make_weights_device
    dims = MODEL_DIMS["large-v2"]
    dev = torch.device("cuda", 0)
    model = gpu.Whisper(dims)
    for name, t in make_weights_device(dims, dev, seed=0).items():
        model.set_tensor(name, t)
    model.finalize()
    torch.cuda.empty_cache()
    yield model
    model.close()


def _check_sequence_rules(tokens, ids, mit_index=50):
    """what the filter chain guarantees for every sampled sequence, whatever the model (decode/filter/*.rs)."""
    t = [int(x) for x in tokens]
    tsb = ids.ts_begin
    assert len(t) >= 1
    assert tsb <= t[0] <= tsb + mit_index, f"first token {t[0]} is not an initial timestamp <= {mit_index}"  # timestamp_tokens.rs:68-78
    banned = set(NON_SPEECH_TOKENS) | {ids.no_timestamps}
    for q, x in enumerate(t):
        assert x not in banned, f"suppressed token {x} sampled at {q}"                      # suppress_tokens.rs:21-23, timestamp_tokens.rs:37-41
        last_ts = q >= 1 and t[q - 1] >= tsb
        second_ts = q < 2 or t[q - 2] >= tsb
        if last_ts and second_ts:
            assert x < tsb, f"timestamp {x} after a timestamp pair at {q}"                 # timestamp_tokens.rs:56-59
        elif last_ts:
            assert x >= ids.eot, f"text token {x} after an unpaired timestamp at {q}"      # timestamp_tokens.rs:60-63


def test_config5_large_v2_full_size_properties(gpu, large_v2):
    dims = MODEL_DIMS["large-v2"]
    ids = TokenIds.for_vocab(dims.n_vocab)
    n = 8
    audio = synth.make_audio(n, seed=4321)
    task = gpu.DecodeTask(large_v2, ids, gpu.DecodeOptions(sample_len=224, max_audio=n))
    a = task.run_audio(audio)
    assert task.n_steps == 224 or all(len(o.tokens) < 224 for o in a)
    # determinism (cache / ancestry / graph state fully reset between runs)
    b = task.run_audio(audio)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x.tokens, y.tokens)
        assert x.sum_logprob == y.sum_logprob and x.no_speech_prob == y.no_speech_prob
    # filter-chain invariants on full-length sequences
    for o in a:
        _check_sequence_rules(o.tokens, ids)
        assert abs(o.avg_logprob - o.sum_logprob / (len(o.tokens) + 1)) < 1e-9       # decode/mod.rs:394-397
        assert 0.0 <= o.no_speech_prob <= 1.0
    task.close()
    # batch invariance: window 3 alone == window 3 inside the batch (independent units: the sharding premise)
    short = gpu.DecodeTask(large_v2, ids, gpu.DecodeOptions(sample_len=24, max_audio=n))
    full = short.run_audio(audio)
    solo = short.run_audio(audio[3:4])
    if not np.array_equal(full[3].tokens, solo[0].tokens):  # different GEMM split plans round differently: only a near-tie may flip
        assert abs(full[3].sum_logprob - solo[0].sum_logprob) < LP_TOL
    else:
        assert abs(full[3].sum_logprob - solo[0].sum_logprob) < LP_TOL
    assert abs(full[3].no_speech_prob - solo[0].no_speech_prob) < 1e-3
    short.close()
