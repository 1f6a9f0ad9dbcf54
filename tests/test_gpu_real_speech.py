"""GPU parity on the reference's one real-speech artefact: test/data/mel-spectrogram.csv (committed verbatim as
tests/golden/mel-spectrogram.csv), the expected log-mel of a real speech clip (80 x 1100 frames = 11 s) from the
reference's own `test_mel_spectrogram` (src/audio.rs:281-301).  Its WAV is not shipped, so the fixture cannot pin
log_mel_spectrogram itself; it is used here as INPUT: real-speech statistics (silences at the floor, formant structure)
instead of the noise + two-sine windows every other test decodes.

  pad_or_trim (audio.rs:213-228)  ->  AudioEncoder  ->  DecodeTask (beam 5, timestamps)  ->  TranscribeTask loop from the
  spectrogram (seek advances by the decoded timestamps, transcribe.rs:151-234), HIP vs the CPU oracle.
Weights are synthetic (none ship with the reference): parity of the arithmetic, not of a transcript."""
import os

import numpy as np
import pytest
import torch

import parity as P
from helpers import gpu_model, oracle_model, oracle_ids, oracle_decode_options
from test_gpu_decode import _oracle_run, _free_run
from whisper_rs_amd.consts import TokenIds, N_FRAMES
from oracle import audio as OA, transcribe as OT

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def speech_mel():
    m = np.loadtxt(os.path.join(G, "mel-spectrogram.csv"), delimiter=",", dtype=np.float32)
    assert m.shape == (80, 1100)
    return m


def test_pad_or_trim_and_encoder_on_real_speech(gpu, speech_mel):
    padded = gpu.audio.pad_or_trim(speech_mel, N_FRAMES)                      # wrs_pad_or_trim
    want_pad = OA.pad_or_trim(torch.from_numpy(speech_mel), N_FRAMES).numpy()
    np.testing.assert_array_equal(padded, want_pad)
    assert padded.shape == (80, 3000) and np.all(padded[:, 1100:] == 0.0)
    np.testing.assert_array_equal(gpu.audio.pad_or_trim(speech_mel, 700), speech_mel[:, :700])
    for dims_name in ("tiny.en", "base.en"):
        want = oracle_model(dims_name).encoder.forward(torch.from_numpy(padded[None])).numpy()
        got = gpu_model(gpu, dims_name).encoder_forward(padded[None])
        err = np.abs(got - want)
        print(f"encoder on real speech, {dims_name}: max abs err {err.max():.3e}, mean {err.mean():.3e}")
        assert err.max() < 4e-2 and err.mean() < 3e-3  # fp16 activations on LayerNorm-ed (unit-scale) outputs


@pytest.mark.parametrize("dims_name", ["tiny.en", "base.en"])
def test_decode_on_real_speech(gpu, speech_mel, dims_name):
    mel = gpu.audio.pad_or_trim(speech_mel, N_FRAMES)[None]
    view, _ = _oracle_run(dims_name, mel, oracle_decode_options(sample_len=64, beam_size=5))
    task = gpu.DecodeTask(gpu_model(gpu, dims_name), TokenIds.english(), gpu.DecodeOptions(sample_len=64))
    _free_run(f"real_speech_{dims_name}_beam5", task, mel, view, 5)
    task.close()


@pytest.mark.parametrize("condition", [True, False])
def test_transcribe_loop_on_real_speech(gpu, speech_mel, condition):
    """TranscribeTask::run_inner from the spectrogram on (transcribe.rs:113-243): windows are cut at the seek the decoded
    timestamps give, the previous tokens become the prompt (condition_on_prev_text)."""
    dims_name = "tiny.en"
    ot = OT.TranscribeTask(oracle_model(dims_name), oracle_ids(), oracle_decode_options(sample_len=24, beam_size=5), None, condition)
    with torch.no_grad():
        want = ot.run_mel(torch.from_numpy(speech_mel))
    gt = gpu.TranscribeTask(gpu_model(gpu, dims_name), gpu.TranscribeOptions(decode=gpu.DecodeOptions(sample_len=24),
                                                                            prompt=gpu.TranscribePrompt.None_(condition)))
    got = gt.run_mel(speech_mel)
    gt.close()
    print(f"transcribe on real speech (condition={condition}): windows {[w[0] for w in got.windows]} tokens {len(got.tokens)} "
          f"segments {len(got.segments)} truncated {got.truncated}")
    assert len(want.windows) >= 2, "fixture should need at least two windows"
    assert [w[0] for w in got.windows] == [w[0] for w in want.windows], "seek sequence differs"
    np.testing.assert_array_equal(got.tokens, want.tokens.numpy())
    assert len(got.segments) == len(want.segments)
    for g, o in zip(got.segments, want.segments):
        assert (g.seek, g.start_token, g.end_token) == (o.seek, o.start_token, o.end_token)
        assert abs(g.start_time - o.start_time) < 1e-9 and abs(g.end_time - o.end_time) < 1e-9
