// Long-form transcription loop: reference src/transcribe.rs:110-243 (TranscribeTask::run_inner).
// Host logic only: whole-file log-mel on device, then per window mel slice -> DecodeTask -> timestamp
// bookkeeping.  `run_chunked` is the fixed-stride variant (the reference's `seek += N_FRAMES`
// branch, transcribe.rs:231, with conditioning off): windows become independent units that are
// decoded `batch` at a time — the unit that shards across GPUs.
#include "transcribe.h"
#include <algorithm>

namespace wrs {

static const double kQuantum = (double)WRS_HOP_LENGTH / (double)WRS_SAMPLE_RATE;  // transcribe.rs:117

void TranscribeTask::configure(Model* model, const wrs_token_ids& ids, const wrs_transcribe_options& o) {
    m = model;
    wrs_decode_options d = o.decode;
    if (d.max_audio < 1) d.max_audio = 1;
    dt.configure(model, ids, d);
    prompt.clear();
    has_prompt = o.prompt_tokens != nullptr && o.n_prompt_tokens > 0;
    if (has_prompt) prompt.assign(o.prompt_tokens, o.prompt_tokens + o.n_prompt_tokens);
    // any explicit prompt implies conditioning (transcribe.rs:142-147)
    condition = (o.prompt_tokens != nullptr) ? true : (o.condition_on_prev_text != 0);
}

void TranscribeTask::set_prompt_text(const char* utf8) {
    WRS_REQUIRE(dt.tokenizer, "transcribe: a Text prompt needs a tokenizer (wrs_transcribe_task_set_tokenizer)");
    WRS_REQUIRE(utf8, "transcribe: null prompt text");
    const std::vector<int32_t> t = dt.tokenizer->encode(utf8);
    prompt.assign(t.begin(), t.end());
    has_prompt = true;
    condition = true;  // transcribe.rs:142-147
}

struct OutBuilder {
    const Tokenizer* tok = nullptr;
    std::vector<int64_t> tokens;
    std::vector<wrs_segment> segments;
    std::vector<int64_t> wseek;
    std::vector<double> wavg, wns;
    bool truncated = false;
    void emit(wrs_transcribe_output* out) {
        memset(out, 0, sizeof(*out));
        out->truncated = truncated ? 1 : 0;
        out->n_tokens = (int64_t)tokens.size();
        out->tokens = (int64_t*)malloc(sizeof(int64_t) * std::max<size_t>(1, tokens.size()));
        memcpy(out->tokens, tokens.data(), tokens.size() * sizeof(int64_t));
        out->n_segments = (int64_t)segments.size();
        out->segments = (wrs_segment*)malloc(sizeof(wrs_segment) * std::max<size_t>(1, segments.size()));
        memcpy(out->segments, segments.data(), segments.size() * sizeof(wrs_segment));
        out->n_windows = (int64_t)wseek.size();
        out->window_seek = (int64_t*)malloc(sizeof(int64_t) * std::max<size_t>(1, wseek.size()));
        out->window_avg_logprob = (double*)malloc(sizeof(double) * std::max<size_t>(1, wseek.size()));
        out->window_no_speech_prob = (double*)malloc(sizeof(double) * std::max<size_t>(1, wseek.size()));
        memcpy(out->window_seek, wseek.data(), wseek.size() * sizeof(int64_t));
        memcpy(out->window_avg_logprob, wavg.data(), wavg.size() * sizeof(double));
        memcpy(out->window_no_speech_prob, wns.data(), wns.size() * sizeof(double));
        out->text = tok ? dup_cstr(tok->decode(tokens.data(), (int64_t)tokens.size())) : nullptr;  // transcribe.rs:236
    }
};

// timestamp bookkeeping of one decoded window (transcribe.rs:167-233).  Returns the seek advance in frames
// the reference would apply; appends segments and the tokens the reference keeps.
static int64_t absorb_window(OutBuilder& ob, const int64_t* seg, int n, int64_t seek, int tsb, int64_t input_stride,
                             double time_precision, bool fixed_stride) {
    std::vector<int> consecutive;  // :168-176
    for (int q = 0; q + 1 < n; ++q)
        if (seg[q] >= tsb && seg[q + 1] >= tsb) consecutive.push_back(q + 1);
    const double ts_offset = (double)seek * kQuantum;
    const int64_t token_offset = (int64_t)ob.tokens.size();
    if (!consecutive.empty()) {  // :181-210
        int last_slice = 0;
        for (int cur : consecutive) {
            const int64_t start_pos = seg[last_slice] - tsb, end_pos = seg[cur - 1] - tsb;
            ob.segments.push_back({seek, ts_offset + (double)start_pos * time_precision, ts_offset + (double)end_pos * time_precision,
                                   token_offset + last_slice + 1, token_offset + cur,
                                   ob.tok ? dup_cstr(ob.tok->decode(seg + last_slice, cur - last_slice)) : nullptr});  // :200
            last_slice = cur;
        }
        const int64_t last_ts_pos = seg[last_slice - 1] - tsb;
        if (fixed_stride) {
            ob.tokens.insert(ob.tokens.end(), seg, seg + n);
            return WRS_N_FRAMES;
        }
        const int keep = std::min(n, last_slice + 1);
        ob.tokens.insert(ob.tokens.end(), seg, seg + keep);  // :207-210
        return last_ts_pos * input_stride;                   // :206
    }
    // :211-233 (the reference indexes timestamps[-1] before checking emptiness and panics when there is
    // no timestamp token; guarded here)
    double duration = (double)WRS_CHUNK_LENGTH;
    int64_t last_ts = -1;
    for (int q = 0; q < n; ++q)
        if (seg[q] >= tsb) last_ts = seg[q];
    if (last_ts >= 0 && last_ts != tsb) duration = (double)(last_ts - tsb) * time_precision;
    ob.segments.push_back({seek, ts_offset, ts_offset + duration, 0, (int64_t)n,
                           ob.tok ? dup_cstr(ob.tok->decode(seg, n)) : nullptr});  // :227 (result.text)
    ob.tokens.insert(ob.tokens.end(), seg, seg + n);
    return WRS_N_FRAMES;
}

// Bookkeeping of the long-form loop alone (test hook, no device): the windows' decoded token lists are given by the caller
// in the order the loop would request them; seek / segments / accumulated tokens evolve exactly as in TranscribeTask::run.
void simulate_transcribe(int ts_begin, int64_t n_audio_ctxs, int64_t n_frames, const int64_t* prompt, int64_t n_prompt,
                         const int64_t* window_tokens, const int32_t* window_lens, int32_t n_windows, const Tokenizer* tok,
                         wrs_transcribe_output* out) {
    WRS_REQUIRE(out && n_audio_ctxs > 0 && n_frames > 0 && (window_tokens || n_windows == 0) && n_windows >= 0, "simulate_transcribe: bad argument");
    const int64_t input_stride = WRS_N_FRAMES / n_audio_ctxs;
    const double time_precision = (double)input_stride * kQuantum;
    OutBuilder ob;
    ob.tok = tok;
    if (prompt && n_prompt > 0) ob.tokens.assign(prompt, prompt + n_prompt);
    int64_t seek = 0, off = 0;
    for (int w = 0; w < n_windows && seek < n_frames; ++w) {
        ob.wseek.push_back(seek);
        ob.wavg.push_back(0.0);
        ob.wns.push_back(0.0);
        const int64_t adv = absorb_window(ob, window_tokens + off, window_lens[w], seek, ts_begin, input_stride, time_precision, false);
        off += window_lens[w];
        if (adv == 0) { ob.truncated = true; break; }
        seek += adv;
    }
    ob.emit(out);
}

int64_t TranscribeTask::prepare_mel(const float* audio, int64_t n) {
    WRS_REQUIRE(audio && n >= 201, "transcribe: need at least 201 samples");
    m->use();
    const int64_t F = n / WRS_HOP_LENGTH;
    audio_dev.ensure((size_t)n * 4);
    mel_dev.ensure((size_t)80 * F * 4);
    HIP_CHECK(hipMemcpyAsync(audio_dev.p, audio, (size_t)n * 4, hipMemcpyHostToDevice, m->stream));
    log_mel_device(mel_ws, audio_dev.as<float>(), 1, n, nullptr, mel_dev.as<float>(), m->stream);  // :111-112
    return F;
}

void TranscribeTask::run(const float* audio, int64_t n, wrs_transcribe_output* out) {
    WRS_REQUIRE(out, "transcribe: null output");
    const int64_t F = prepare_mel(audio, n);
    run_from_mel(F, out);
}

// the loop of TranscribeTask::run_inner after its log-mel (transcribe.rs:113-243) on a caller-supplied spectrogram
void TranscribeTask::run_mel(const float* mel, int64_t n_frames, wrs_transcribe_output* out) {
    WRS_REQUIRE(out && mel && n_frames >= 1, "transcribe: bad mel argument");
    m->use();
    mel_dev.ensure((size_t)80 * n_frames * 4);
    HIP_CHECK(hipMemcpyAsync(mel_dev.p, mel, (size_t)80 * n_frames * 4, hipMemcpyHostToDevice, m->stream));
    HIP_CHECK(hipStreamSynchronize(m->stream));  // the caller's buffer is free on return
    run_from_mel(n_frames, out);
}

void TranscribeTask::run_from_mel(int64_t F, wrs_transcribe_output* out) {
    const int64_t input_stride = WRS_N_FRAMES / m->T;                   // :120
    const double time_precision = (double)input_stride * kQuantum;      // :122
    OutBuilder ob;
    ob.tok = dt.tokenizer;
    ob.tokens = prompt;  // :126-140
    int64_t seek = 0;
    const int tsb = dt.ids.ts_begin;
    m->ensure_encoder_ws(1);
    while (seek < F) {  // :151
        m->load_mel_window(mel_dev.as<float>(), F, seek, 0);  // :152 (pad_or_trim of mel[:, seek:])
        if (condition) dt.set_prompt(ob.tokens.data(), (int)ob.tokens.size());  // :155-158
        wrs_decode_result res;
        dt.run_loaded(1, &res);  // :161-162
        const wrs_decode_output& o = res.outputs[0];
        ob.wseek.push_back(seek);
        ob.wavg.push_back(o.avg_logprob);
        ob.wns.push_back(o.no_speech_prob);
        const int64_t adv = absorb_window(ob, o.tokens, o.n_tokens, seek, tsb, input_stride, time_precision, false);
        free_decode_result(&res);
        if (adv == 0 || ob.wseek.size() >= 10000) {  // the reference would loop forever on a <|0.00|> pair
            ob.truncated = true;                     // reported: the rest of the audio was NOT transcribed
            break;
        }
        seek += adv;
    }
    ob.emit(out);
}

void TranscribeTask::run_chunked(const float* audio, int64_t n, int batch, wrs_transcribe_output* out) {
    WRS_REQUIRE(out, "transcribe: null output");
    WRS_REQUIRE(batch >= 1 && batch <= dt.max_audio, "transcribe: batch %d exceeds the task capacity %d", batch, dt.max_audio);
    const int64_t F = prepare_mel(audio, n);
    const int64_t input_stride = WRS_N_FRAMES / m->T;
    const double time_precision = (double)input_stride * kQuantum;
    OutBuilder ob;
    ob.tok = dt.tokenizer;
    ob.tokens = prompt;
    const int tsb = dt.ids.ts_begin;
    dt.set_prompt(nullptr, 0);
    const int64_t n_win = cdiv(F, WRS_N_FRAMES);
    m->ensure_encoder_ws(batch);
    for (int64_t w0 = 0; w0 < n_win; w0 += batch) {
        const int nb = (int)std::min<int64_t>(batch, n_win - w0);
        for (int j = 0; j < nb; ++j)
            m->load_mel_window(mel_dev.as<float>(), F, (w0 + j) * WRS_N_FRAMES, j);
        wrs_decode_result res;
        dt.run_loaded(nb, &res);
        for (int j = 0; j < nb; ++j) {
            const wrs_decode_output& o = res.outputs[j];
            const int64_t seek = (w0 + j) * WRS_N_FRAMES;
            ob.wseek.push_back(seek);
            ob.wavg.push_back(o.avg_logprob);
            ob.wns.push_back(o.no_speech_prob);
            absorb_window(ob, o.tokens, o.n_tokens, seek, tsb, input_stride, time_precision, true);
        }
        free_decode_result(&res);
    }
    ob.emit(out);
}

}  // namespace wrs
