"""CPU: WAV ingest (reference src/audio.rs:20-173, `load_audio`) through the C-ABI.  16 kHz files must come back
sample-exact (integer PCM scaled by 2^(bits-1), channels averaged); resampled files are checked against the ideal
band-limited result, because the reference's resampler (rubato FftFixedIn) is a third-party dependency that is not
available here — parity unpinned for that branch."""
import struct
import wave

import numpy as np
import pytest

import whisper_rs_amd as W


def _write_wav(path, data, rate, sampwidth):
    """data: int array [frames, channels]"""
    with wave.open(str(path), "wb") as w:
        w.setnchannels(data.shape[1])
        w.setsampwidth(sampwidth)
        w.setframerate(rate)
        if sampwidth == 1:
            raw = data.astype(np.uint8).tobytes()
        elif sampwidth == 2:
            raw = data.astype("<i2").tobytes()
        elif sampwidth == 3:
            raw = b"".join(struct.pack("<i", int(v))[:3] for v in data.reshape(-1))
        else:
            raw = data.astype("<i4").tobytes()
        w.writeframes(raw)


@pytest.mark.parametrize("sampwidth,scale", [(2, 32768.0), (3, 8388608.0), (4, 2147483648.0)])
def test_pcm_16k_is_exact_and_downmixed(tmp_path, sampwidth, scale):
    rng = np.random.default_rng(sampwidth)
    lim = int(scale) - 1
    data = rng.integers(-lim, lim, size=(4000, 2), dtype=np.int64)
    p = tmp_path / "a.wav"
    _write_wav(p, data, 16000, sampwidth)
    got = W.audio.load_audio(p)
    want = ((data[:, 0].astype(np.float64) / scale).astype(np.float32) + (data[:, 1].astype(np.float64) / scale).astype(np.float32)) / np.float32(2)
    assert got.dtype == np.float32 and got.shape == (4000,)
    np.testing.assert_array_equal(got, want.astype(np.float32))


def test_pcm8_and_mono_passthrough(tmp_path):
    data = np.arange(256, dtype=np.int64).reshape(-1, 1)
    p = tmp_path / "b.wav"
    _write_wav(p, data, 16000, 1)
    np.testing.assert_array_equal(W.audio.load_audio(p), ((data[:, 0] - 128) / 128.0).astype(np.float32))


def test_float_wav_with_extra_chunks(tmp_path):
    x = np.linspace(-1, 1, 1001, dtype=np.float32)
    body = x.tobytes()
    fmt = struct.pack("<HHIIHH", 3, 1, 16000, 16000 * 4, 4, 32)
    junk = b"LIST" + struct.pack("<I", 5) + b"abcde" + b"\x00"  # odd-sized chunk + pad byte
    riff = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + junk + b"data" + struct.pack("<I", len(body)) + body
    p = tmp_path / "c.wav"
    p.write_bytes(b"RIFF" + struct.pack("<I", len(riff)) + riff)
    np.testing.assert_array_equal(W.audio.load_audio(p), x)


@pytest.mark.parametrize("rate", [8000, 22050, 44100, 48000])
def test_resampled_sines_match_the_ideal(tmp_path, rate):
    dur = 0.5
    t_in = np.arange(int(rate * dur)) / rate
    freqs = [220.0, 1000.0, 3100.0]
    x = sum(0.25 * np.sin(2 * np.pi * f * t_in + 0.3 * i) for i, f in enumerate(freqs))
    p = tmp_path / "d.wav"
    _write_wav(p, np.round(x * 32767).astype(np.int64).reshape(-1, 1), rate, 2)
    got = W.audio.load_audio(p)
    n_out = -(-len(t_in) * 16000 // rate)
    assert got.shape == (n_out,)
    t_out = np.arange(n_out) / 16000.0
    want = sum(0.25 * np.sin(2 * np.pi * f * t_out + 0.3 * i) for i, f in enumerate(freqs))
    mid = slice(400, n_out - 400)  # away from the file edges (the filter sees zeros beyond them)
    assert np.abs(got[mid] - want[mid]).max() < 2e-3  # 16-bit quantisation + filter ripple
    # content above the new Nyquist is removed (anti-aliasing), checked where the input rate allows such content
    if rate > 2 * 9000:
        hi = 0.5 * np.sin(2 * np.pi * 9000.0 * t_in)
        y = W.audio.resample(hi.astype(np.float32), rate, 16000)
        assert np.abs(y[400:-400]).max() < 1e-3


def test_errors(tmp_path):
    with pytest.raises(W.WrsError, match="failed to open audio file"):
        W.audio.load_audio(tmp_path / "missing.wav")
    p = tmp_path / "e.flac"
    p.write_bytes(b"fLaC" + b"\x00" * 64)
    with pytest.raises(W.WrsError, match="failed to determine audio format"):
        W.audio.load_audio(p)
    assert W.audio.resample(np.zeros(0, np.float32), 44100).shape == (0,)
