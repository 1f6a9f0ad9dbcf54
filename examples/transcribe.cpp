// Counterpart of reference examples/transcribe.rs on top of the C-ABI: same dims (base.en), same
// option values (:44-60), same flow: load weights -> tokenizer -> TranscribeTask -> load_audio -> run -> print.
//   transcribe [audio.wav [tokenizer.json [weights.safetensors]]]
// There are no checkpoints/audio in the build environment, so every input is optional: without a weights file the
// model gets a deterministic pseudo-random fill, without a tokenizer file the output is token ids only (no text),
// without a WAV the audio is 35 s of synthetic tone+noise.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "wrs.h"

#define CK(x) do { int _c = (x); if (_c) { fprintf(stderr, "%s failed (%d): %s\n", #x, _c, wrs_last_error()); return 1; } } while (0)

static uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s; }
static float gauss(uint32_t& s) {
    float u1 = (lcg(s) >> 8) * (1.0f / 16777216.0f) + 1e-7f, u2 = (lcg(s) >> 8) * (1.0f / 16777216.0f);
    return std::sqrt(-2.f * std::log(u1)) * std::cos(6.2831853f * u2);
}

static int set(wrs_model* m, const std::string& name, std::vector<int64_t> shape, float scale, float offset, uint32_t& seed) {
    int64_t n = 1;
    for (auto d : shape) n *= d;
    std::vector<float> v(n);
    for (auto& x : v) x = offset + scale * gauss(seed);
    return wrs_model_set_tensor(m, name.c_str(), WRS_F32, shape.data(), (int)shape.size(), v.data(), WRS_MEM_HOST);
}

int main(int argc, char** argv) {
    if (wrs_device_count() < 1) { fprintf(stderr, "no gfx950 device: this program has no CPU path\n"); return 2; }
    wrs_model_dims dims = {51864, 1500, 512, 8, 6, 448, 512, 8, 6};  // examples/transcribe.rs:25-35
    wrs_model* model = nullptr;
    CK(wrs_model_create(&dims, 0, &model));
    uint32_t seed = 1;
    if (argc > 3) {  // vars.load("weights.ot") (examples/transcribe.rs:37), safetensors container
        int32_t n = 0;
        CK(wrs_model_load_safetensors(model, argv[3], &n));
        CK(wrs_model_finalize(model));
        fprintf(stderr, "loaded model weights (%d tensors from %s)\n", n, argv[3]);
    } else {
    const int64_t D = dims.n_audio_states;
    auto block = [&](const std::string& p, bool cross) -> int {
        int rc = 0;
        auto attn = [&](const std::string& a) {
            rc |= set(model, a + ".query.weight", {D, D}, 1.f / std::sqrt((float)D), 0, seed);
            rc |= set(model, a + ".query.bias", {D}, 0.02f, 0, seed);
            rc |= set(model, a + ".key.weight", {D, D}, 1.f / std::sqrt((float)D), 0, seed);
            rc |= set(model, a + ".value.weight", {D, D}, 1.f / std::sqrt((float)D), 0, seed);
            rc |= set(model, a + ".value.bias", {D}, 0.02f, 0, seed);
            rc |= set(model, a + ".out.weight", {D, D}, 1.f / std::sqrt((float)D), 0, seed);
            rc |= set(model, a + ".out.bias", {D}, 0.02f, 0, seed);
        };
        attn(p + ".attn");
        rc |= set(model, p + ".attn_ln.weight", {D}, 0.1f, 1.f, seed);
        rc |= set(model, p + ".attn_ln.bias", {D}, 0.1f, 0, seed);
        if (cross) {
            attn(p + ".cross_attn");
            rc |= set(model, p + ".cross_attn_ln.weight", {D}, 0.1f, 1.f, seed);
            rc |= set(model, p + ".cross_attn_ln.bias", {D}, 0.1f, 0, seed);
        }
        rc |= set(model, p + ".mlp.0.weight", {4 * D, D}, 1.f / std::sqrt((float)D), 0, seed);
        rc |= set(model, p + ".mlp.0.bias", {4 * D}, 0.02f, 0, seed);
        rc |= set(model, p + ".mlp.2.weight", {D, 4 * D}, 0.5f / std::sqrt((float)D), 0, seed);
        rc |= set(model, p + ".mlp.2.bias", {D}, 0.02f, 0, seed);
        rc |= set(model, p + ".mlp_ln.weight", {D}, 0.1f, 1.f, seed);
        rc |= set(model, p + ".mlp_ln.bias", {D}, 0.1f, 0, seed);
        return rc;
    };
    CK(set(model, "encoder.conv1.weight", {D, 80, 3}, 1.f / std::sqrt(240.f), 0, seed));
    CK(set(model, "encoder.conv1.bias", {D}, 0.02f, 0, seed));
    CK(set(model, "encoder.conv2.weight", {D, D, 3}, 1.f / std::sqrt(3.f * D), 0, seed));
    CK(set(model, "encoder.conv2.bias", {D}, 0.02f, 0, seed));
    for (int i = 0; i < dims.n_audio_layers; ++i) CK(block("encoder.blocks." + std::to_string(i), false));
    CK(set(model, "encoder.ln_post.weight", {D}, 0.1f, 1.f, seed));
    CK(set(model, "encoder.ln_post.bias", {D}, 0.1f, 0, seed));
    CK(set(model, "decoder.token_embedding.weight", {dims.n_vocab, D}, 0.02f, 0, seed));
    CK(set(model, "decoder.positional_embedding", {dims.n_text_ctxs, D}, 0.01f, 0, seed));
    for (int i = 0; i < dims.n_text_layers; ++i) CK(block("decoder.blocks." + std::to_string(i), true));
    CK(set(model, "decoder.ln.weight", {D}, 0.1f, 1.f, seed));
    CK(set(model, "decoder.ln.bias", {D}, 0.1f, 0, seed));
    CK(wrs_model_finalize(model));
    fprintf(stderr, "loaded model weights (synthetic)\n");
    }

    wrs_token_ids ids = {50256, 50257, 50357, 50358, 50359, 50360, 50361, 50362, 50363, 220};
    wrs_tokenizer* tokenizer = nullptr;  // Tokenizer::new_from_file(Task::Transcribe, "tokenizer/gpt2.json") (:40)
    if (argc > 2) {
        CK(wrs_tokenizer_create_from_file(argv[2], &tokenizer));
        CK(wrs_tokenizer_token_ids(tokenizer, &ids));
    }
    static const int32_t non_speech[] = {1, 2, 7, 8, 9, 10, 14, 25, 26, 27, 28, 29, 31, 58, 59, 60, 61, 62, 63, 90, 91, 92, 93, 357, 366, 438, 532, 685, 705, 796, 930, 1058, 1220, 1267, 1279, 1303, 1343, 1377, 1391, 1635, 1782, 1875, 2162, 2361, 2488, 3467, 4008, 4211, 4600, 4808, 5299, 5855, 6329, 7203, 9609, 9959, 10563, 10786, 11420, 11709, 11907, 13163, 13697, 13700, 14808, 15306, 16410, 16791, 17992, 19203, 19510, 20724, 22305, 22935, 27007, 30109, 30420, 33409, 34949, 40283, 40493, 40549, 47282, 49146};
    wrs_transcribe_options opt;
    memset(&opt, 0, sizeof(opt));
    opt.decode.sample_len = 0;                               // None
    opt.decode.extract_mode = WRS_EXTRACT_BEAM_SEARCH;       // beam_size 5, patience 1.0
    opt.decode.group_size = 5;
    opt.decode.patience = 1.0f;
    opt.decode.has_len_penalty = 0;                          // len_penalty: None
    opt.decode.has_max_initial_timestamp = 1;                // Some(1.0)
    opt.decode.max_initial_timestamp = 1.0;
    opt.decode.timestamps = 1;
    opt.decode.suppress_blank = 1;
    opt.decode.suppress_non_speech = 1;
    opt.decode.non_speech_tokens = non_speech;
    opt.decode.n_non_speech_tokens = (int)(sizeof(non_speech) / sizeof(non_speech[0]));
    opt.decode.position_offset_quirk = 1;
    opt.decode.max_audio = 1;
    opt.condition_on_prev_text = 1;                          // TranscribePrompt::None { condition_on_prev_text: true }
    wrs_transcribe_task* task = nullptr;
    CK(wrs_transcribe_task_create(model, &ids, &opt, &task));
    if (tokenizer) CK(wrs_transcribe_task_set_tokenizer(task, tokenizer));
    fprintf(stderr, "initialized transcription task\n");

    std::vector<float> audio;
    if (argc > 1) {  // whisper::audio::load_audio(path) (:65)
        float* samples = nullptr;
        int64_t n = 0;
        CK(wrs_load_audio(argv[1], &samples, &n));
        audio.assign(samples, samples + n);
        wrs_audio_free(samples);
    } else {
        audio.resize(16000 * 35);
        for (size_t i = 0; i < audio.size(); ++i)
            audio[i] = 0.1f * gauss(seed) + 0.2f * std::sin(6.2831853f * 220.f * i / 16000.f);
    }
    fprintf(stderr, "loaded audio (%zu samples)\n", audio.size());
    wrs_transcribe_output out;
    CK(wrs_transcribe_task_run(task, audio.data(), (int64_t)audio.size(), &out));
    printf("transcribed audio: %ld tokens, %ld segments\n", (long)out.n_tokens, (long)out.n_segments);
    if (out.text) printf("text: %s\n", out.text);
    for (int64_t i = 0; i < out.n_segments; ++i)
        printf("  segment seek=%ld [%.2f, %.2f] tokens %ld..%ld%s%s\n", (long)out.segments[i].seek, out.segments[i].start_time,
               out.segments[i].end_time, (long)out.segments[i].start_token, (long)out.segments[i].end_token,
               out.segments[i].text ? " text: " : "", out.segments[i].text ? out.segments[i].text : "");
    wrs_transcribe_output_free(&out);
    wrs_transcribe_task_destroy(task);
    wrs_tokenizer_destroy(tokenizer);
    wrs_model_destroy(model);
    return 0;
}
