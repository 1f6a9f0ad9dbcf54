// WAV ingest + resampling (reference src/audio.rs:20-173); see audio_io.cpp.
#pragma once
#include "common.h"

namespace wrs {
std::vector<float> resample_f32(const float* x, int64_t n, int rate_in, int rate_out);
std::vector<float> read_wav_mono(const std::vector<uint8_t>& bytes, int* rate_out);
std::vector<float> load_audio_file(const char* path);
}  // namespace wrs
