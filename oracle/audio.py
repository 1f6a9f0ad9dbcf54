"""ORACLE (test infrastructure) — audio front-end.  See oracle/__init__.py.

Restates reference src/audio.rs:12-18 (consts), :175-188 (mel_filter),
:213-228 (pad_or_trim), :230-259 (log_mel_spectrogram).
"""
import numpy as np
import torch

SAMPLE_RATE = 16000
N_FFT = 400
N_MELS = 80
HOP_LENGTH = 160
CHUNK_LENGTH = 30
N_SAMPLES = CHUNK_LENGTH * SAMPLE_RATE
N_FRAMES = N_SAMPLES // HOP_LENGTH


def mel_filter(sr=SAMPLE_RATE, n_fft=N_FFT, n_mels=N_MELS) -> torch.Tensor:
    """audio.rs:175-188: `mel_filter::mel::<f32>(16000, 400, Some(80), None, None,
    false, NormalizationFactor::One)`.

    The arithmetic lives in the third-party crate mel_filter 0.1.1 (Cargo.lock:567),
    absent from /root/reference.  Its published algorithm is the librosa
    `filters.mel(htk=False, norm='slaney')` port: Slaney mel scale (linear below
    1 kHz, log above), triangular weights on the rfft bin centres, each triangle
    scaled by 2/(f_hi - f_lo).  Pinned by the reference's known-answer fixture
    test/data/mel-filter-8x8.csv (tests/test_oracle_golden.py).
    """
    fmin, fmax = 0.0, sr / 2.0

    def hz_to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        f_sp = 200.0 / 3
        mels = f / f_sp
        min_log_hz = 1000.0
        min_log_mel = min_log_hz / f_sp
        logstep = np.log(6.4) / 27.0
        return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)

    def mel_to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        f_sp = 200.0 / 3
        freqs = f_sp * m
        min_log_hz = 1000.0
        min_log_mel = min_log_hz / f_sp
        logstep = np.log(6.4) / 27.0
        return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)

    fftfreqs = np.linspace(0.0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    weights = np.zeros((n_mels, 1 + n_fft // 2), dtype=np.float64)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0.0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, None]
    return torch.from_numpy(weights.astype(np.float32))


def pad_or_trim(x: torch.Tensor, length: int) -> torch.Tensor:
    """audio.rs:213-228: slice the last dim to `length`, or right-pad it with constant 0."""
    n = x.shape[-1]
    if n > length:
        return x[..., :length]
    return torch.nn.functional.pad(x, (0, length - n), mode="constant", value=0.0)


def log_mel_spectrogram(audio: torch.Tensor, filt: torch.Tensor) -> torch.Tensor:
    """audio.rs:230-259.  audio: [N] f32 -> [80, N // 160] f32."""
    window = torch.hann_window(N_FFT, dtype=torch.float32)  # :233 (periodic)
    stft = torch.stft(audio, N_FFT, HOP_LENGTH, None, window, center=True, pad_mode="reflect",
                      normalized=False, onesided=True, return_complex=True)  # :236-246
    magnitudes = stft[:, :-1].abs().pow(2)  # :247
    mel_spec = filt.matmul(magnitudes)  # :252
    log_spec = mel_spec.clamp_min(1e-10).log10()  # :254
    log_spec = torch.maximum(log_spec, log_spec.max() - 8.0)  # :256 (global max over the whole input)
    return (log_spec + 4.0) / 4.0  # :258
