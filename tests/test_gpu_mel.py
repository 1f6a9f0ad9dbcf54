"""GPU parity: log-mel front-end (reference src/audio.rs:175-259) vs the CPU oracle, through the C-ABI.
Floating point: tolerance stated per assert (the reference's own test uses rtol 0.1 / atol 0.005,
audio.rs:295-297; we hold 1e-3 abs on a [-1, 1.5] range)."""
import numpy as np
import pytest
import torch

from whisper_rs_amd import synth
from oracle import audio as OA

pytestmark = pytest.mark.gpu


def test_mel_filter_matches_reference_fixture(gpu, golden_dir):
    f = gpu.audio.mel_filter()
    gold = np.loadtxt(f"{golden_dir}/mel-filter-8x8.csv", delimiter=",", dtype=np.float32)
    assert f.shape == (80, 201)
    np.testing.assert_allclose(f[:8, :8], gold, rtol=1e-5, atol=1e-8)  # reference audio.rs:273-274
    np.testing.assert_allclose(f, OA.mel_filter().numpy(), rtol=1e-6, atol=1e-9)
    assert int((f != 0).sum()) == 391


@pytest.mark.parametrize("n", [480000, 176000, 16000 * 3 + 77, 201 + 160])
def test_log_mel_vs_oracle(gpu, n):
    a = synth.make_audio(1, seed=7, n_samples=n)[0]
    got = gpu.audio.log_mel_spectrogram(a)
    want = OA.log_mel_spectrogram(torch.from_numpy(a), OA.mel_filter()).numpy()
    assert got.shape == want.shape == (80, n // 160)
    err = np.abs(got - want)
    assert err.max() < 1e-3, f"max abs err {err.max()}"
    assert err.mean() < 2e-5, f"mean abs err {err.mean()}"
    assert abs((got.max() - got.min()) - (want.max() - want.min())) < 1e-3


def test_log_mel_batch_is_per_recording(gpu):
    a = synth.make_audio(3, seed=11)
    a[1] *= 0.01  # different global max per row
    got = gpu.audio.log_mel_spectrogram(a)
    for i in range(3):
        want = OA.log_mel_spectrogram(torch.from_numpy(a[i]), OA.mel_filter()).numpy()
        assert np.abs(got[i] - want).max() < 1e-3


def test_log_mel_silence_and_custom_filter(gpu):
    z = np.zeros(16000, np.float32)
    got = gpu.audio.log_mel_spectrogram(z)
    want = OA.log_mel_spectrogram(torch.from_numpy(z), OA.mel_filter()).numpy()
    np.testing.assert_allclose(got, want, atol=1e-6)  # all bins clamp to 1e-10 -> (-10 + 4) / 4
    filt = np.abs(np.random.default_rng(0).standard_normal((80, 201))).astype(np.float32) * 0.01
    a = synth.make_audio(1, seed=3, n_samples=32000)[0]
    got = gpu.audio.log_mel_spectrogram(a, filt)
    want = OA.log_mel_spectrogram(torch.from_numpy(a), torch.from_numpy(filt)).numpy()
    assert np.abs(got - want).max() < 1e-3


def test_log_mel_rejects_short_input(gpu):
    with pytest.raises(gpu.WrsError):
        gpu.audio.log_mel_spectrogram(np.zeros(100, np.float32))


def test_pad_or_trim(gpu):
    x = np.arange(12, dtype=np.float32).reshape(3, 4)
    np.testing.assert_array_equal(gpu.audio.pad_or_trim(x, 2), x[:, :2])
    np.testing.assert_array_equal(gpu.audio.pad_or_trim(x, 6), OA.pad_or_trim(torch.from_numpy(x), 6).numpy())
    np.testing.assert_array_equal(gpu.audio.pad_or_trim(x, 4), x)
