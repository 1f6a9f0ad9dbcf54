// TranscribeTask (reference src/transcribe.rs:13-101).
#pragma once
#include "decode.h"

namespace wrs {

struct TranscribeTask {
    Model* m = nullptr;
    DecodeTask dt;
    std::vector<int64_t> prompt;  // TranscribePrompt::Pretokenized
    bool has_prompt = false;
    bool condition = true;
    MelWorkspace mel_ws;
    DevBuf audio_dev, mel_dev;

    void set_tokenizer(const Tokenizer* tok) { dt.tokenizer = tok; }
    void set_prompt_text(const char* utf8);  // TranscribePrompt::Text (transcribe.rs:130-134)
    void configure(Model* model, const wrs_token_ids& ids, const wrs_transcribe_options& o);
    int64_t prepare_mel(const float* audio, int64_t n);  // whole-file log-mel into mel_dev; returns frames
    void run(const float* audio, int64_t n, wrs_transcribe_output* out);
    void run_mel(const float* mel, int64_t n_frames, wrs_transcribe_output* out);  // mel [80, n_frames] f32 host
    void run_from_mel(int64_t n_frames, wrs_transcribe_output* out);               // spectrogram already in mel_dev
    void run_chunked(const float* audio, int64_t n, int batch, wrs_transcribe_output* out);
};

void simulate_transcribe(int ts_begin, int64_t n_audio_ctxs, int64_t n_frames, const int64_t* prompt, int64_t n_prompt,
                         const int64_t* window_tokens, const int32_t* window_lens, int32_t n_windows, const Tokenizer* tok,
                         wrs_transcribe_output* out);

}  // namespace wrs
