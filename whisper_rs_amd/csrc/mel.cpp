// log-mel front-end on device: reference src/audio.rs:175-188 (mel_filter), :230-259
// (log_mel_spectrogram).  STFT = reflect pad + [frames x 400] x [400 x 402] exact-f32 MFMA GEMM
// against a Hann-windowed DFT matrix (frames are overlapping rows of the padded signal: lda = hop),
// power in the GEMM epilogue, mel projection as a second f32 GEMM with log10 + per-recording
// max (atomic) in its epilogue, then one floor/scale/transpose pass.
#include "model.h"
#include <cmath>

namespace wrs {

// Slaney-scale, area-normalised filterbank == mel_filter 0.1.1 `mel(16000, 400, 80, None, None,
// false, NormalizationFactor::One)` (reference audio.rs:175-188); pinned by the reference's
// test/data/mel-filter-8x8.csv.
void mel_filter_host(float* out) {
    const int n_mels = WRS_N_MELS, n_freqs = WRS_N_FFT / 2 + 1;
    const double sr = WRS_SAMPLE_RATE, f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp;
    const double logstep = std::log(6.4) / 27.0;
    auto hz_to_mel = [&](double f) { return f >= min_log_hz ? min_log_mel + std::log(f / min_log_hz) / logstep : f / f_sp; };
    auto mel_to_hz = [&](double m) { return m >= min_log_mel ? min_log_hz * std::exp(logstep * (m - min_log_mel)) : f_sp * m; };
    std::vector<double> mel_f(n_mels + 2), fft(n_freqs);
    const double m_lo = hz_to_mel(0.0), m_hi = hz_to_mel(sr / 2.0);
    for (int i = 0; i < n_mels + 2; ++i) mel_f[i] = mel_to_hz(m_lo + (m_hi - m_lo) * i / (n_mels + 1));
    for (int k = 0; k < n_freqs; ++k) fft[k] = (sr / 2.0) * k / (n_freqs - 1);
    for (int i = 0; i < n_mels; ++i) {
        const double enorm = 2.0 / (mel_f[i + 2] - mel_f[i]);
        for (int k = 0; k < n_freqs; ++k) {
            const double lower = (fft[k] - mel_f[i]) / (mel_f[i + 1] - mel_f[i]);
            const double upper = (mel_f[i + 2] - fft[k]) / (mel_f[i + 2] - mel_f[i + 1]);
            out[i * n_freqs + k] = (float)(std::max(0.0, std::min(lower, upper)) * enorm);
        }
    }
}

static const int kPowerLd = 208;  // 201 bins padded to a multiple of 16 (f32 GEMM K granularity)

static void ensure_tables(MelWorkspace& ws, const float* filter_host_or_null) {
    if (!ws.tables_ready) {
        // DFT matrix rows: 2b = hann[n]*cos(2 pi b n / 400), 2b+1 = -hann[n]*sin(...); rows 402..447 zero
        std::vector<float> tw((size_t)448 * 400, 0.f);
        const double two_pi = 6.283185307179586476925286766559;
        for (int b = 0; b <= 200; ++b)
            for (int n = 0; n < 400; ++n) {
                const double win = 0.5 - 0.5 * std::cos(two_pi * n / 400.0);  // periodic Hann (audio.rs:233)
                const double ang = two_pi * (double)((b * n) % 400) / 400.0;
                tw[(size_t)(2 * b) * 400 + n] = (float)(win * std::cos(ang));
                tw[(size_t)(2 * b + 1) * 400 + n] = (float)(-win * std::sin(ang));
            }
        ws.twiddle.alloc(tw.size() * 4);
        HIP_CHECK(hipMemcpyAsync(ws.twiddle.p, tw.data(), tw.size() * 4, hipMemcpyHostToDevice, utility_stream()));
        HIP_CHECK(hipStreamSynchronize(utility_stream()));
        ws.filter.alloc((size_t)128 * kPowerLd * 4);
        ws.tables_ready = true;
        ws.filter_host.clear();
    }
    std::vector<float> f(80 * 201);
    if (filter_host_or_null) memcpy(f.data(), filter_host_or_null, f.size() * 4);
    else mel_filter_host(f.data());
    if (f != ws.filter_host) {
        std::vector<float> padded((size_t)128 * kPowerLd, 0.f);
        for (int i = 0; i < 80; ++i)
            for (int k = 0; k < 201; ++k) padded[(size_t)i * kPowerLd + k] = f[i * 201 + k];
        HIP_CHECK(hipMemcpyAsync(ws.filter.p, padded.data(), padded.size() * 4, hipMemcpyHostToDevice, utility_stream()));
        HIP_CHECK(hipStreamSynchronize(utility_stream()));
        ws.filter_host = f;
    }
}

void log_mel_device(MelWorkspace& ws, const float* audio_d, int64_t n_batch, int64_t n, const float* filter_host_or_null,
                    float* out_d, hipStream_t s) {
    WRS_REQUIRE(n_batch >= 1 && n >= 201, "log_mel_spectrogram: need >= 201 samples for reflect padding (got %ld)", (long)n);
    const int64_t F = n / WRS_HOP_LENGTH;  // 1 + n/160 frames, last one dropped (audio.rs:247)
    WRS_REQUIRE(F >= 1 && n_batch * F < (1ll << 31), "log_mel_spectrogram: bad frame count");
    ensure_tables(ws, filter_host_or_null);
    const int64_t stride = round_up(n + 400, 4);
    ws.padded.ensure((size_t)n_batch * stride * 4 + 4096);
    ws.power.ensure((size_t)n_batch * F * kPowerLd * 4);
    ws.logmel.ensure((size_t)n_batch * F * 80 * 4);
    ws.maxbuf.ensure((size_t)n_batch * 4);
    launch_reflect_pad(audio_d, n_batch, n, ws.padded.as<float>(), stride, s);
    HIP_CHECK(hipMemsetAsync(ws.maxbuf.p, 0x80, (size_t)n_batch * 4, s));  // order-key of a very negative float
    launch_gemm_f32(ws.padded.as<float>(), WRS_HOP_LENGTH, F, stride, ws.twiddle.as<float>(), 400, n_batch * F, 402, 400, 0,
                    ws.power.as<float>(), kPowerLd, nullptr, s);
    launch_gemm_f32(ws.power.as<float>(), kPowerLd, F, F * kPowerLd, ws.filter.as<float>(), kPowerLd, n_batch * F, 80, kPowerLd, 1,
                    ws.logmel.as<float>(), 80, ws.maxbuf.as<float>(), s);
    launch_mel_finalize(ws.logmel.as<float>(), ws.maxbuf.as<float>(), n_batch, F, out_d, s);
}

}  // namespace wrs
