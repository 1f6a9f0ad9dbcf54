// log-mel front-end on device: reference src/audio.rs:175-188 (mel_filter), :230-259 (log_mel_spectrogram).
// Host side: filterbank, tables; the device work is one fused kernel + an elementwise scale pass (mel_fused.hip).
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

static void ensure_tables(MelWorkspace& ws, const float* filter_host_or_null) {
    if (!ws.tables_ready) {
        std::vector<float> tw(mel_twiddle_packed_floats());
        mel_pack_twiddle(tw.data());
        ws.twiddle.alloc(tw.size() * 4);
        HIP_CHECK(hipMemcpyAsync(ws.twiddle.p, tw.data(), tw.size() * 4, hipMemcpyHostToDevice, utility_stream()));
        HIP_CHECK(hipStreamSynchronize(utility_stream()));
        ws.filter.alloc((size_t)80 * 201 * 4);
        ws.krange.alloc((size_t)80 * 2 * 4);
        ws.tables_ready = true;
        ws.filter_host.clear();
    }
    std::vector<float> f(80 * 201);
    if (filter_host_or_null) memcpy(f.data(), filter_host_or_null, f.size() * 4);
    else mel_filter_host(f.data());
    if (f != ws.filter_host) {
        std::vector<int> kr(160);
        for (int i = 0; i < 80; ++i) {  // band of non-zeros of every filter row (5 bins on average for the Slaney bank; any filter works)
            int k0 = 201, k1 = 0;
            for (int k = 0; k < 201; ++k)
                if (f[i * 201 + k] != 0.f) { k0 = std::min(k0, k); k1 = k + 1; }
            if (k1 == 0) k0 = 0;
            kr[2 * i] = k0; kr[2 * i + 1] = k1;
        }
        HIP_CHECK(hipMemcpyAsync(ws.filter.p, f.data(), f.size() * 4, hipMemcpyHostToDevice, utility_stream()));
        HIP_CHECK(hipMemcpyAsync(ws.krange.p, kr.data(), kr.size() * 4, hipMemcpyHostToDevice, utility_stream()));
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
    ws.maxbuf.ensure((size_t)n_batch * 4);
    HIP_CHECK(hipMemsetAsync(ws.maxbuf.p, 0x80, (size_t)n_batch * 4, s));  // order-key of a very negative float
    launch_mel_fused(audio_d, n_batch, n, ws.twiddle.as<float>(), ws.filter.as<float>(), ws.krange.as<int>(), out_d, ws.maxbuf.as<int>(), s);
}

}  // namespace wrs
