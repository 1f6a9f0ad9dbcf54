"""GPU parity: TranscribeTask::run (reference src/transcribe.rs:103-243) vs the oracle loop."""
import numpy as np
import pytest
import torch

from helpers import gpu_model, oracle_model, oracle_ids, oracle_decode_options
from whisper_rs_amd import synth
from whisper_rs_amd.consts import TokenIds
from oracle import transcribe as OT

pytestmark = pytest.mark.gpu


def _segments_equal(gs, os_):
    assert len(gs) == len(os_)
    for g, o in zip(gs, os_):
        assert (g.seek, g.start_token, g.end_token) == (o.seek, o.start_token, o.end_token)
        assert abs(g.start_time - o.start_time) < 1e-9 and abs(g.end_time - o.end_time) < 1e-9


@pytest.mark.parametrize("condition", [False, True])
def test_transcribe_long_form(gpu, condition):
    audio = synth.make_audio(1, seed=77, n_samples=16000 * 50)[0]  # 50 s -> at least 2 windows
    om = oracle_model("nano")
    ot = OT.TranscribeTask(om, oracle_ids(), oracle_decode_options(sample_len=10), None, condition)
    want = ot.run(torch.from_numpy(audio))
    gm = gpu_model(gpu, "nano")
    gt = gpu.TranscribeTask(gm, gpu.TranscribeOptions(decode=gpu.DecodeOptions(sample_len=10),
                                                      prompt=gpu.TranscribePrompt.None_(condition)))
    got = gt.run(audio)
    assert [w[0] for w in got.windows] == [w[0] for w in want.windows], "seek sequence differs"
    np.testing.assert_array_equal(got.tokens, want.tokens.numpy())
    _segments_equal(got.segments, want.segments)
    gt.close()


def test_transcribe_chunked_equals_independent_windows(gpu):
    """fixed-stride sharding unit: chunked batch decode == windows decoded one by one."""
    audio = synth.make_audio(1, seed=5, n_samples=16000 * 95)[0]
    gm = gpu_model(gpu, "nano")
    opts = gpu.TranscribeOptions(decode=gpu.DecodeOptions(sample_len=8, max_audio=4), prompt=gpu.TranscribePrompt.None_(False))
    gt = gpu.TranscribeTask(gm, opts)
    a = gt.run_chunked(audio, 4)
    b = gt.run_chunked(audio, 1)
    np.testing.assert_array_equal(a.tokens, b.tokens)
    assert len(a.windows) == 4 and [w[0] for w in a.windows] == [0, 3000, 6000, 9000]
    _segments_equal(a.segments, b.segments)
    gt.close()


def test_text_outputs_with_tokenizer(gpu):
    """TranscribeOutput.text / segment text / DecodeOutput.text (transcribe.rs:200,227,236; decode/mod.rs:404) are the
    attached tokenizer's decode of exactly the token slices the reference decodes; Text prompts are encoded by it."""
    from helpers import synthetic_tokenizer_json
    tok = gpu.Tokenizer.new_from_bytes(synthetic_tokenizer_json())
    import dataclasses
    assert dataclasses.replace(tok.ids, space=220) == TokenIds.english()  # same id layout; only the id of ' ' differs
    audio = synth.make_audio(1, seed=77, n_samples=16000 * 50)[0]
    gm = gpu_model(gpu, "nano")
    dopt = gpu.DecodeOptions(sample_len=10)
    plain = gpu.TranscribeTask(gm, gpu.TranscribeOptions(decode=dopt, prompt=gpu.TranscribePrompt.None_(True))).run(audio)
    assert plain.text is None and all(s.text is None for s in plain.segments)
    withtok = gpu.TranscribeTask(gm, gpu.TranscribeOptions(decode=dopt, prompt=gpu.TranscribePrompt.None_(True), tokenizer=tok)).run(audio)
    np.testing.assert_array_equal(withtok.tokens, plain.tokens)
    assert withtok.text == tok.decode(withtok.tokens)
    assert len(withtok.segments) > 0
    for s in withtok.segments:
        assert isinstance(s.text, str)
        if s.start_token > 0:  # consecutive-timestamp slice: tokens[start_token-1 : end_token] (transcribe.rs:186-200)
            assert s.text == tok.decode(withtok.tokens[s.start_token - 1:s.end_token])
    # TranscribePrompt::Text == Pretokenized(encode(text))
    t1 = gpu.TranscribeTask(gm, gpu.TranscribeOptions(decode=dopt, prompt=gpu.TranscribePrompt.Text(" hello there"), tokenizer=tok)).run(audio)
    t2 = gpu.TranscribeTask(gm, gpu.TranscribeOptions(decode=dopt, prompt=gpu.TranscribePrompt.Pretokenized(tok.encode(" hello there")))).run(audio)
    np.testing.assert_array_equal(t1.tokens, t2.tokens)
    assert t1.text.startswith(" hello there")
    # DecodeTask with a tokenizer
    from helpers import mel_windows
    task = gpu.DecodeTask(gm, tok.ids, dopt)
    task.set_tokenizer(tok)
    out = task.run(mel_windows(1))[0]
    assert out.text == tok.decode(out.tokens)
    task.close()


def test_window_pool_on_whole_file_spectrogram_equals_chunked_transcribe(gpu):
    """one long recording, the reference's whole-file log-mel normalisation (transcribe.rs:111-112), fixed-stride windows
    sharded over a pool's replicas (wrs_window_pool_run_mel) == TranscribeTask::run_chunked on the same recording."""
    import helpers
    from whisper_rs_amd.consts import MODEL_DIMS
    audio = synth.make_audio(1, seed=5, n_samples=16000 * 95)[0]
    audio[16000 * 40:16000 * 70] *= 0.01  # a quiet stretch: per-window and whole-file normalisation now differ
    dopt = gpu.DecodeOptions(sample_len=8, max_audio=2)
    gt = gpu.TranscribeTask(gpu_model(gpu, "nano"), gpu.TranscribeOptions(decode=dopt, prompt=gpu.TranscribePrompt.None_(False)))
    want = gt.run_chunked(audio, 2)
    gt.close()
    mel = gpu.audio.log_mel_spectrogram(audio)  # [80, 9500], global max
    replicas = [gpu.Whisper(MODEL_DIMS["nano"], helpers.weights("nano")) for _ in range(2)]
    pool = gpu.WindowPool(replicas, TokenIds.english(), dopt)
    outs = pool.run_mel(mel)
    assert len(outs) == 4 == len(want.windows)
    np.testing.assert_array_equal(np.concatenate([o.tokens for o in outs]), want.tokens)
    # the per-window entry normalises each window as its own recording: on this file that is a different spectrogram
    per_window = pool.run_audio(np.stack([np.pad(audio[i * 480000:(i + 1) * 480000], (0, max(0, 480000 - len(audio[i * 480000:(i + 1) * 480000])))) for i in range(4)]))
    assert len(per_window) == 4
    pool.close()
    for r in replicas:
        r.close()
