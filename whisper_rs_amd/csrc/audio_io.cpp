// Audio ingest for RIFF/WAVE files: reference `audio::load_audio` (src/audio.rs:20-173) restricted to the container
// BASELINE config 1 names (one 16 kHz WAV).  Host-only; SURVEY.md section 8(f) rank 3.
//   decode    PCM 8/16/24/32-bit integer and 32/64-bit IEEE float, WAVE_FORMAT_EXTENSIBLE included; integer samples are
//             scaled to [-1, 1) by 2^(bits-1) like symphonia's SampleBuffer<f32> conversion (audio.rs:116-118)
//   downmix   mean over channels (audio.rs:121-127)
//   resample  to 16 kHz when the file's rate differs (audio.rs:72-88,129-170).  The reference uses rubato 0.12
//             `FftFixedIn` (Cargo.lock; not vendored, cannot be built here): PARITY UNPINNED for resampled files.  This
//             build uses a Kaiser-windowed-sinc polyphase resampler (16 zero crossings, beta 8.6: > 80 dB stop band) and
//             returns ceil(n * 16000 / rate) samples (rubato pads the tail to whole 1024-sample chunks instead).
// Other containers/codecs (flac, mp3, ogg: symphonia in the reference) stay with host libraries: out of scope.
#include "common.h"
#include <algorithm>
#include <cmath>
#include <fstream>
#include <numeric>

namespace wrs {

static double bessel_i0(double x) {
    double sum = 1.0, term = 1.0;
    const double q = x * x / 4.0;
    for (int k = 1; k < 64; ++k) {
        term *= q / ((double)k * (double)k);
        sum += term;
        if (term < 1e-17 * sum) break;
    }
    return sum;
}

// y[n] = L * sum_k h[n*M - k*L] x[k],  h = Kaiser-windowed sinc at the L-times-upsampled rate, cut-off 1/(2 max(L, M))
std::vector<float> resample_f32(const float* x, int64_t n, int rate_in, int rate_out) {
    WRS_REQUIRE(rate_in > 0 && rate_out > 0 && n >= 0, "resample: bad rates %d -> %d", rate_in, rate_out);
    if (rate_in == rate_out || n == 0) return std::vector<float>(x, x + n);
    const int64_t g = std::gcd((int64_t)rate_in, (int64_t)rate_out);
    const int64_t L = rate_out / g, M = rate_in / g, big = std::max(L, M);
    WRS_REQUIRE(big <= 200000, "resample: ratio %d/%d needs a %ld-phase filter", rate_out, rate_in, (long)big);
    const int Z = 16;
    const double beta = 8.6, rolloff = 0.945;
    const int64_t half = (int64_t)Z * big;
    std::vector<double> h((size_t)half + 1);  // symmetric: h[|t|]
    const double i0b = bessel_i0(beta);
    for (int64_t t = 0; t <= half; ++t) {
        const double u = (double)t / (double)big;           // in units of the slower stream's samples
        const double arg = M_PI * u * rolloff;
        const double sinc = t == 0 ? 1.0 : std::sin(arg) / arg;
        const double r = (double)t / (double)half;
        const double win = bessel_i0(beta * std::sqrt(std::max(0.0, 1.0 - r * r))) / i0b;
        h[(size_t)t] = rolloff * sinc * win / (double)big;   // unity pass-band gain after the factor L below
    }
    const int64_t n_out = (n * L + M - 1) / M;
    std::vector<float> y((size_t)n_out);
    for (int64_t o = 0; o < n_out; ++o) {
        const int64_t c = o * M;  // position on the upsampled grid
        int64_t k_lo = (c - half + L - 1) / L, k_hi = (c + half) / L;
        if (c - half < 0) k_lo = 0;
        k_lo = std::max<int64_t>(k_lo, 0);
        k_hi = std::min<int64_t>(k_hi, n - 1);
        double acc = 0.0;
        for (int64_t k = k_lo; k <= k_hi; ++k) {
            const int64_t t = c - k * L;
            acc += h[(size_t)(t < 0 ? -t : t)] * (double)x[k];
        }
        y[(size_t)o] = (float)(acc * (double)L);
    }
    return y;
}

namespace {
struct Reader {
    const std::vector<uint8_t>& b;
    size_t pos = 0;
    bool has(size_t n) const { return pos + n <= b.size(); }
    uint32_t u32() { uint32_t v = b[pos] | (b[pos + 1] << 8) | (b[pos + 2] << 16) | ((uint32_t)b[pos + 3] << 24); pos += 4; return v; }
    uint16_t u16() { uint16_t v = (uint16_t)(b[pos] | (b[pos + 1] << 8)); pos += 2; return v; }
};
}  // namespace

// decoded, downmixed mono samples + the file's sample rate
std::vector<float> read_wav_mono(const std::vector<uint8_t>& bytes, int* rate_out) {
    Reader r{bytes};
    WRS_REQUIRE(r.has(12) && !memcmp(bytes.data(), "RIFF", 4) && !memcmp(bytes.data() + 8, "WAVE", 4),
                "failed to determine audio format: not a RIFF/WAVE file (other containers are out of scope)");
    r.pos = 12;
    int fmt = 0, channels = 0, rate = 0, bits = 0, block = 0;
    bool have_fmt = false;
    while (r.has(8)) {
        const char* id = (const char*)bytes.data() + r.pos;
        r.pos += 4;
        size_t sz = r.u32();
        if (!memcmp(id, "fmt ", 4)) {
            WRS_REQUIRE(sz >= 16 && r.has(16), "wav: short fmt chunk");
            const size_t start = r.pos;
            fmt = r.u16(); channels = r.u16(); rate = (int)r.u32(); (void)r.u32(); block = r.u16(); bits = r.u16();
            if (fmt == 0xFFFE && sz >= 26 && r.has(10)) {  // WAVE_FORMAT_EXTENSIBLE: sub-format GUID starts with the real tag
                (void)r.u16(); (void)r.u16(); (void)r.u32();
                fmt = r.u16();
            }
            have_fmt = true;
            r.pos = start + sz + (sz & 1);
        } else if (!memcmp(id, "data", 4)) {
            WRS_REQUIRE(have_fmt, "wav: data chunk before fmt chunk");
            WRS_REQUIRE(channels >= 1 && channels <= 64, "unknown channel count %d", channels);
            WRS_REQUIRE(rate > 0, "track sample rate is unknown");
            WRS_REQUIRE((fmt == 1 && (bits == 8 || bits == 16 || bits == 24 || bits == 32)) || (fmt == 3 && (bits == 32 || bits == 64)),
                        "failed to create decoder for track: wav format tag %d with %d bits is not supported", fmt, bits);
            sz = std::min(sz, bytes.size() - r.pos);  // streaming writers leave 0 / 0xFFFFFFFF here
            const int bps = bits / 8;
            const size_t frame = (size_t)bps * channels;
            WRS_REQUIRE(block == 0 || (size_t)block == frame, "wav: block align %d does not match %d channels x %d bits", block, channels, bits);
            const size_t frames = sz / frame;
            std::vector<float> out(frames);
            const uint8_t* p = bytes.data() + r.pos;
            for (size_t f = 0; f < frames; ++f) {
                float acc = 0.f;
                for (int c = 0; c < channels; ++c, p += bps) {
                    float v;
                    if (fmt == 3) {
                        if (bits == 32) { float t; memcpy(&t, p, 4); v = t; }
                        else { double t; memcpy(&t, p, 8); v = (float)t; }
                    } else if (bits == 8) v = ((float)p[0] - 128.f) / 128.f;
                    else if (bits == 16) v = (float)(int16_t)(p[0] | (p[1] << 8)) / 32768.f;
                    else if (bits == 24) v = (float)(((int32_t)((uint32_t)p[0] << 8 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 24)) >> 8) / 8388608.f;
                    else v = (float)((double)(int32_t)(p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24)) / 2147483648.0);
                    acc += v;
                }
                out[f] = channels == 1 ? acc : acc / (float)channels;  // audio.rs:121-127
            }
            *rate_out = rate;
            return out;
        } else {
            r.pos += sz + (sz & 1);
        }
    }
    WRS_THROW(WRS_EINVAL, "failed to get default audio track: wav file has no data chunk");
}

std::vector<float> load_audio_file(const char* path) {
    std::ifstream f(path, std::ios::binary);
    WRS_REQUIRE(f.good(), "failed to open audio file %s", path);
    std::vector<uint8_t> bytes((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    int rate = 0;
    std::vector<float> mono = read_wav_mono(bytes, &rate);
    if (rate == WRS_SAMPLE_RATE) return mono;  // audio.rs:72-74
    return resample_f32(mono.data(), (int64_t)mono.size(), rate, WRS_SAMPLE_RATE);
}

}  // namespace wrs
