// extern "C" surface of libwrs_mi355x.so — see include/wrs.h for the reference interface each
// entry replaces.  No exceptions cross this boundary; no torch types appear in it.
#include "handles.h"
#include "audio_io.h"
#include <map>
#include <mutex>
#include <fstream>
#include <sstream>


namespace wrs {
const std::string& last_error();
}
using namespace wrs;

extern "C" {

const char* wrs_last_error(void) { return wrs::last_error().c_str(); }
const char* wrs_version(void) { return "whisper-rs_amd 0.1.0 (gfx950)"; }

int wrs_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    int ok = 0;
    for (int i = 0; i < n; ++i) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, i) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) ok++;
    }
    return ok;
}

// ------------------------------------------------------------------ model
int wrs_model_create(const wrs_model_dims* dims, int device_id, wrs_model** out) {
    return guarded([&] {
        WRS_REQUIRE(dims && out, "wrs_model_create: null argument");
        WRS_REQUIRE(dims->n_vocab > 0 && dims->n_audio_states > 0 && dims->n_audio_heads > 0 && dims->n_audio_layers > 0 &&
                        dims->n_text_states > 0 && dims->n_text_heads > 0 && dims->n_text_layers > 0 && dims->n_audio_ctxs > 0 &&
                        dims->n_text_ctxs > 0, "wrs_model_create: non-positive dimension");
        require_device(device_id);
        std::unique_ptr<wrs_model> h(new wrs_model());
        h->m.dims = *dims;
        h->m.device = device_id;
        HIP_CHECK(hipSetDevice(device_id));
        // highest stream priority: the decode steps are a chain of short dependent kernels; work queued on a task's side
        // stream (overlapped mode, lowest priority) fills the CUs they leave idle instead of delaying them
        int prio_least = 0, prio_greatest = 0;
        HIP_CHECK(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
        HIP_CHECK(hipStreamCreateWithPriority(&h->m.stream, hipStreamNonBlocking, prio_greatest));
        h->m.timer = new StageTimer();
        h->m.read_switches();
        *out = h.release();
    });
}
int wrs_model_set_tensor(wrs_model* m, const char* name, int dtype, const int64_t* shape, int ndim, const void* data, int mem_kind) {
    return guarded([&] {
        WRS_REQUIRE(m && name, "wrs_model_set_tensor: null argument");
        m->m.set_tensor(name, dtype, shape, ndim, data, mem_kind);
    });
}
int wrs_model_load_safetensors(wrs_model* m, const char* path, int32_t* n_loaded) {
    return guarded([&] {
        WRS_REQUIRE(m && path, "wrs_model_load_safetensors: null argument");
        const int n = load_safetensors(m->m, path);
        if (n_loaded) *n_loaded = n;
    });
}
int wrs_model_finalize(wrs_model* m) {
    return guarded([&] { WRS_REQUIRE(m, "null model"); m->m.finalize(); });
}
int wrs_model_device(const wrs_model* m) { return m ? m->m.device : -1; }
int wrs_model_get_dims(const wrs_model* m, wrs_model_dims* out) {
    return guarded([&] { WRS_REQUIRE(m && out, "null argument"); *out = m->m.dims; });
}
void wrs_model_destroy(wrs_model* m) {
    if (!m) return;
    (void)hipSetDevice(m->m.device);
    delete m;
}

int wrs_encoder_forward(wrs_model* h, const float* mel, int n_audio, float* out) {
    return guarded([&] {
        WRS_REQUIRE(h && mel && out && n_audio >= 1, "wrs_encoder_forward: bad argument");
        Model& m = h->m;
        WRS_REQUIRE(m.finalized, "model not finalized");
        m.use();
        m.load_mel(mel, n_audio, WRS_MEM_HOST, WRS_N_FRAMES, 0);
        m.encoder_from_mel_t(n_audio);
        HIP_CHECK(hipMemcpyAsync(out, m.enc_ws.xa32.p, (size_t)n_audio * m.T * m.D * 4, hipMemcpyDeviceToHost, m.stream));
        HIP_CHECK(hipStreamSynchronize(m.stream));
    });
}

int wrs_decoder_forward(wrs_model* h, const int64_t* tokens, int n_batch, int n_tokens, const float* xa, int64_t offset, float* logits) {
    return guarded([&] {
        WRS_REQUIRE(h && tokens && logits && n_batch >= 1 && n_tokens >= 1 && offset >= 0, "wrs_decoder_forward: bad argument");
        Model& m = h->m;
        WRS_REQUIRE(m.finalized, "model not finalized");
        m.use();
        if (!m.stage_state || m.stage_state->a_cap < n_batch) {
            WRS_REQUIRE(!m.stage_state || m.stage_state->len == 0, "wrs_decoder_forward: batch grew while the cache is not empty");
            m.stage_state.reset(new DecState());
            m.init_state(*m.stage_state, n_batch, 1);
        }
        DecState& st = *m.stage_state;
        if (xa) {  // Cached::forward recomputes Linear(xa) on every call (model.rs:79)
            const size_t n = (size_t)n_batch * m.T * m.D;
            m.ensure_encoder_ws(n_batch);
            ++m.enc_ws.epoch;
            HIP_CHECK(hipMemcpyAsync(m.enc_ws.xa32.p, xa, n * 4, hipMemcpyHostToDevice, m.stream));
            launch_f32_to_f16(m.enc_ws.xa32.as<float>(), m.enc_ws.xa16.as<half_t>(), (int64_t)n, m.stream);
            m.load_cross_kv(st, m.enc_ws.xa16.as<half_t>(), n_batch);
        }
        WRS_REQUIRE(st.n_audio >= n_batch, "wrs_decoder_forward: xa is NULL but no cross-attention K/V are cached for %d rows", n_batch);
        if (st.len == 0) {
            launch_anc_init(st.anc.as<int>(), (int64_t)st.r_cap * st.ctx, st.ctx, st.r_cap, 1, m.stream);
            st.parity = 0;
        }
        WRS_REQUIRE(st.len + n_tokens <= m.n_ctx && offset + n_tokens <= m.n_ctx, "wrs_decoder_forward: %d cached + %d new tokens (offset %ld) exceed n_text_ctxs %d",
                    st.len, n_tokens, (long)offset, m.n_ctx);
        if (n_tokens > 1 && st.len == 0 && !m.parity_f32) {
            // all tokens in one masked pass, as the reference does (model.rs:191-201,490-522): batched causal prefill
            std::vector<int> toks((size_t)n_batch * n_tokens);
            for (size_t i = 0; i < toks.size(); ++i) {
                WRS_REQUIRE(tokens[i] >= 0 && tokens[i] < m.V, "wrs_decoder_forward: token %ld out of range", (long)tokens[i]);
                toks[i] = (int)tokens[i];
            }
            WRS_REQUIRE(offset == 0, "wrs_decoder_forward: a multi-token call on an empty cache starts at offset 0 (got %ld)", (long)offset);
            st.pending_src.clear();
            m.decoder_prefill(st, n_batch, n_tokens, 0, 1, toks.data(), true);
            HIP_CHECK(hipMemcpy2DAsync(logits, (size_t)m.V * 4, st.prefill.logits_all.p, (size_t)st.logits_ld * 4, (size_t)m.V * 4,
                                       (size_t)n_batch * n_tokens, hipMemcpyDeviceToHost, m.stream));
            HIP_CHECK(hipStreamSynchronize(m.stream));
            st.len = n_tokens;
            return;
        }
        for (int t = 0; t < n_tokens; ++t) {
            StepParams* sp = st.sp_h();
            st.parity ^= 1;
            sp->cache_pos = st.len; sp->pos_idx = (int)offset + t; sp->parity = st.parity; sp->first_step = 0;
            sp->want_no_speech = 0; sp->n_rows = n_batch;
            for (int b = 0; b < n_batch; ++b) {
                const int64_t tok = tokens[(size_t)b * n_tokens + t];
                WRS_REQUIRE(tok >= 0 && tok < m.V, "wrs_decoder_forward: token %ld out of range", (long)tok);
                st.tokens_h()[b] = (int)tok;
                st.src_h()[b] = (t == 0 && !st.pending_src.empty()) ? st.pending_src[b] : b;
                st.tsmode_h()[b] = 0;
            }
            HIP_CHECK(hipMemcpyAsync(st.sp_dev.p, st.sp_host.p, st.sp_bytes(), hipMemcpyHostToDevice, m.stream));
            m.decoder_step(st, n_batch, 1, 1, true, true);
            HIP_CHECK(hipMemcpy2DAsync(logits + (size_t)t * m.V, (size_t)n_tokens * m.V * 4, st.logits.p, (size_t)st.logits_ld * 4, (size_t)m.V * 4,
                                       n_batch, hipMemcpyDeviceToHost, m.stream));
            HIP_CHECK(hipStreamSynchronize(m.stream));
            if (t == 0) st.pending_src.clear();
            st.len += 1;
        }
    });
}

int wrs_model_update_cache(wrs_model* h, const int64_t* src, int n_batch) {
    return guarded([&] {
        WRS_REQUIRE(h && src && n_batch >= 1, "wrs_model_update_cache: bad argument");
        Model& m = h->m;
        if (!m.stage_state || m.stage_state->len == 0) return;  // Cached::update_cache on an empty cache is a no-op (model.rs:60)
        DecState& st = *m.stage_state;
        WRS_REQUIRE(n_batch <= st.r_cap, "wrs_model_update_cache: %d rows exceed the cache (%d)", n_batch, st.r_cap);
        std::vector<int> next(n_batch);
        for (int r = 0; r < n_batch; ++r) {
            WRS_REQUIRE(src[r] >= 0 && src[r] < st.r_cap, "wrs_model_update_cache: source index %ld out of range", (long)src[r]);
            next[r] = st.pending_src.empty() ? (int)src[r] : st.pending_src[src[r]];
        }
        st.pending_src = next;
    });
}

int wrs_model_clear_cache(wrs_model* h) {
    return guarded([&] {
        WRS_REQUIRE(h, "null model");
        if (h->m.stage_state) { h->m.stage_state->len = 0; h->m.stage_state->pending_src.clear(); }
    });
}

// ------------------------------------------------------------------ audio
int wrs_mel_filter(float* out) {
    return guarded([&] { WRS_REQUIRE(out, "null output"); mel_filter_host(out); });
}

int wrs_pad_or_trim(const float* x, int64_t rows, int64_t n, int64_t len, float* out) {
    return guarded([&] {
        WRS_REQUIRE(x && out && rows >= 0 && n >= 0 && len >= 0, "wrs_pad_or_trim: bad argument");
        const int64_t c = std::min(n, len);
        for (int64_t r = 0; r < rows; ++r) {
            memcpy(out + r * len, x + r * n, (size_t)c * 4);
            for (int64_t i = c; i < len; ++i) out[r * len + i] = 0.f;
        }
    });
}

struct MelCtx { MelWorkspace ws; DevBuf in, out; hipStream_t s = nullptr; };
static std::mutex g_mel_mu;
static std::map<int, std::unique_ptr<MelCtx>> g_mel;

int wrs_log_mel_spectrogram(int device_id, const float* audio, int64_t n_batch, int64_t n_samples, const float* filter, float* out, int mem_kind) {
    return guarded([&] {
        WRS_REQUIRE(audio && out && n_batch >= 1, "wrs_log_mel_spectrogram: bad argument");
        require_device(device_id);
        HIP_CHECK(hipSetDevice(device_id));
        std::lock_guard<std::mutex> lk(g_mel_mu);
        auto& ctx = g_mel[device_id];
        if (!ctx) {
            ctx.reset(new MelCtx());
            HIP_CHECK(hipStreamCreateWithFlags(&ctx->s, hipStreamNonBlocking));
        }
        const int64_t F = n_samples / WRS_HOP_LENGTH;
        const float* a_d = audio;
        float* o_d = out;
        if (mem_kind == WRS_MEM_HOST) {
            ctx->in.ensure((size_t)n_batch * n_samples * 4);
            ctx->out.ensure((size_t)n_batch * 80 * std::max<int64_t>(F, 1) * 4);
            HIP_CHECK(hipMemcpyAsync(ctx->in.p, audio, (size_t)n_batch * n_samples * 4, hipMemcpyHostToDevice, ctx->s));
            a_d = ctx->in.as<float>();
            o_d = ctx->out.as<float>();
        }
        log_mel_device(ctx->ws, a_d, n_batch, n_samples, filter, o_d, ctx->s);
        if (mem_kind == WRS_MEM_HOST)
            HIP_CHECK(hipMemcpyAsync(out, o_d, (size_t)n_batch * 80 * F * 4, hipMemcpyDeviceToHost, ctx->s));
        HIP_CHECK(hipStreamSynchronize(ctx->s));
    });
}

// ------------------------------------------------------------------ decode
int wrs_decode_task_create(wrs_model* m, const wrs_token_ids* ids, const wrs_decode_options* opt, wrs_decode_task** out) {
    return guarded([&] {
        WRS_REQUIRE(m && ids && opt && out, "wrs_decode_task_create: null argument");
        std::unique_ptr<wrs_decode_task> h(new wrs_decode_task());
        h->t.configure(&m->m, *ids, *opt);
        *out = h.release();
    });
}
int wrs_decode_task_set_prompt(wrs_decode_task* t, const int64_t* prompt, int32_t n) {
    return guarded([&] { WRS_REQUIRE(t, "null task"); t->t.set_prompt(prompt, n); });
}
int wrs_decode_task_run(wrs_decode_task* t, const float* mel, int32_t n_audio, int mem_kind, wrs_decode_result* result) {
    return guarded([&] {
        WRS_REQUIRE(t && t->t.pending.empty(), "batches queued with wrs_decode_task_encode_async must be decoded (wrs_decode_task_run_pending) first");
        WRS_REQUIRE(t && mel && result, "wrs_decode_task_run: null argument");
        WRS_REQUIRE(n_audio >= 1 && n_audio <= t->t.max_audio, "wrs_decode_task_run: %d windows exceed capacity %d", n_audio, t->t.max_audio);
        Model& m = *t->t.m;
        m.use();
        m.load_mel(mel, n_audio, mem_kind, WRS_N_FRAMES, 0);  // unsqueeze(0) / batch (decode/mod.rs:288)
        t->t.run_loaded(n_audio, result);
    });
}
int wrs_decode_task_run_features(wrs_decode_task* t, const float* feats, int32_t n_audio, int mem_kind, wrs_decode_result* result) {
    return guarded([&] {
        WRS_REQUIRE(t && t->t.pending.empty(), "batches queued with wrs_decode_task_encode_async must be decoded (wrs_decode_task_run_pending) first");
        WRS_REQUIRE(t && feats && result, "wrs_decode_task_run_features: null argument");
        WRS_REQUIRE(n_audio >= 1 && n_audio <= t->t.max_audio, "wrs_decode_task_run_features: %d windows exceed capacity %d", n_audio, t->t.max_audio);
        Model& m = *t->t.m;
        m.use();
        m.ensure_encoder_ws(n_audio);
        const size_t n = (size_t)n_audio * m.T * m.D;
        HIP_CHECK(hipMemcpyAsync(m.enc_ws.xa32.p, feats, n * 4, mem_kind == WRS_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, m.stream));
        launch_f32_to_f16(m.enc_ws.xa32.as<float>(), m.enc_ws.xa16.as<half_t>(), (int64_t)n, m.stream);
        m.enc_ws.n_encoded = n_audio;
        ++m.enc_ws.epoch;
        t->t.run_encoded(n_audio, result);
    });
}
int wrs_decode_task_read_audio_features(wrs_decode_task* t, int32_t audio_index, float* out, int64_t n_floats) {
    return guarded([&] {
        WRS_REQUIRE(t && out, "wrs_decode_task_read_audio_features: null argument");
        Model& m = *t->t.m;
        WRS_REQUIRE(audio_index >= 0 && audio_index < t->t.features_n && t->t.features_epoch == m.enc_ws.epoch,
                    "wrs_decode_task_read_audio_features: window %d is not in the encoder workspace any more (%d windows of the last decode are)",
                    audio_index, t->t.features_n);
        WRS_REQUIRE(n_floats == (int64_t)m.T * m.D, "wrs_decode_task_read_audio_features: %ld floats, the features are [%d, %d]", (long)n_floats, m.T, m.D);
        m.use();
        HIP_CHECK(hipMemcpyAsync(out, m.enc_ws.xa32.as<float>() + (size_t)audio_index * m.T * m.D, (size_t)n_floats * 4, hipMemcpyDeviceToHost, m.stream));
        HIP_CHECK(hipStreamSynchronize(m.stream));
    });
}
int wrs_decode_task_run_audio(wrs_decode_task* t, const float* audio, int32_t n_audio, int mem_kind, wrs_decode_result* result) {
    return guarded([&] {
        WRS_REQUIRE(t && t->t.pending.empty(), "batches queued with wrs_decode_task_encode_async must be decoded (wrs_decode_task_run_pending) first");
        WRS_REQUIRE(t && audio && result, "wrs_decode_task_run_audio: null argument");
        WRS_REQUIRE(n_audio >= 1 && n_audio <= t->t.max_audio, "wrs_decode_task_run_audio: %d windows exceed capacity %d", n_audio, t->t.max_audio);
        DecodeTask& d = t->t;
        Model& m = *d.m;
        m.use();
        const float* a_d = audio;
        if (mem_kind == WRS_MEM_HOST) {
            d.audio_dev.ensure((size_t)n_audio * WRS_N_SAMPLES * 4);
            HIP_CHECK(hipMemcpyAsync(d.audio_dev.p, audio, (size_t)n_audio * WRS_N_SAMPLES * 4, hipMemcpyHostToDevice, m.stream));
            a_d = d.audio_dev.as<float>();
        }
        d.mel_dev.ensure((size_t)n_audio * 80 * WRS_N_FRAMES * 4);
        {
            Timed tm(m.timer, "mel", m.stream, (double)n_audio * (WRS_N_SAMPLES * 4.0 + 80.0 * WRS_N_FRAMES * 4.0));
            log_mel_device(d.mel_ws, a_d, n_audio, WRS_N_SAMPLES, nullptr, d.mel_dev.as<float>(), m.stream);
        }
        m.load_mel(d.mel_dev.as<float>(), n_audio, WRS_MEM_DEVICE, WRS_N_FRAMES, 0);
        d.run_loaded(n_audio, result);
    });
}
int wrs_decode_task_encode_async(wrs_decode_task* t, const float* audio, int32_t n_audio, int mem_kind) {
    return guarded([&] { WRS_REQUIRE(t && audio, "wrs_decode_task_encode_async: null argument"); t->t.encode_async(audio, n_audio, mem_kind); });
}
int wrs_decode_task_run_pending(wrs_decode_task* t, wrs_decode_result* result) {
    return guarded([&] { WRS_REQUIRE(t && result, "wrs_decode_task_run_pending: null argument"); t->t.run_pending(result); });
}
int wrs_decode_task_run_encoded(wrs_decode_task* t, int32_t n_audio, wrs_decode_result* result) {
    return guarded([&] {
        WRS_REQUIRE(t && result, "wrs_decode_task_run_encoded: null argument");
        WRS_REQUIRE(n_audio >= 1 && n_audio <= t->t.m->enc_ws.n_encoded, "wrs_decode_task_run_encoded: %d windows requested, %d encoded", n_audio, t->t.m->enc_ws.n_encoded);
        t->t.m->use();
        t->t.run_encoded(n_audio, result);
    });
}
int wrs_decode_task_detect_language(wrs_decode_task* t, const float* mel, int32_t n_audio, int mem_kind, int32_t first_language_id,
                                    int32_t n_languages, int32_t* out_language_ids, float* out_probs) {
    return guarded([&] {
        WRS_REQUIRE(t && t->t.pending.empty(), "batches queued with wrs_decode_task_encode_async must be decoded (wrs_decode_task_run_pending) first");
        WRS_REQUIRE(t && mel && out_language_ids, "wrs_decode_task_detect_language: null argument");
        WRS_REQUIRE(n_audio >= 1 && n_audio <= t->t.max_audio, "wrs_decode_task_detect_language: %d windows exceed capacity %d", n_audio, t->t.max_audio);
        Model& m = *t->t.m;
        m.use();
        m.load_mel(mel, n_audio, mem_kind, WRS_N_FRAMES, 0);
        m.encoder_from_mel_t(n_audio);
        t->t.detect_language(n_audio, first_language_id, n_languages, out_language_ids, out_probs);
    });
}
int wrs_decode_task_set_languages(wrs_decode_task* t, const int32_t* language_ids, int32_t n) {
    return guarded([&] { WRS_REQUIRE(t, "null task"); t->t.set_languages(language_ids, n); });
}
int wrs_decode_task_trace(wrs_decode_task* t, int32_t* n_steps, int32_t* n_rows, const int32_t** tokens, const int32_t** sources, const float** slp) {
    return guarded([&] {
        WRS_REQUIRE(t && n_steps && n_rows, "wrs_decode_task_trace: null argument");
        *n_steps = t->t.n_steps;
        *n_rows = t->t.tr_rows;
        if (tokens) *tokens = t->t.tr_tokens.data();
        if (sources) *sources = t->t.tr_src.data();
        if (slp) *slp = t->t.tr_slp.data();
    });
}
void wrs_decode_result_free(wrs_decode_result* r) { free_decode_result(r); }
void wrs_decode_task_destroy(wrs_decode_task* t) {
    if (!t) return;
    (void)hipSetDevice(t->t.m->device);
    delete t;
}

int wrs_filter_topk(wrs_decode_task* t, const float* logits, const int64_t* tokens, int32_t n_rows, int32_t n_tokens, int32_t sample_begin,
                    int32_t k, float* out_lp, int32_t* out_ids, float* out_filtered) {
    return guarded([&] {
        WRS_REQUIRE(t && logits && tokens && out_lp && out_ids, "wrs_filter_topk: null argument");
        DecodeTask& d = t->t;
        Model& m = *d.m;
        DecState& st = d.st;
        WRS_REQUIRE(n_rows >= 1 && n_rows <= st.r_cap, "wrs_filter_topk: %d rows exceed capacity %d", n_rows, st.r_cap);
        WRS_REQUIRE(k >= 1 && k <= 8 && n_tokens >= sample_begin && sample_begin >= 0, "wrs_filter_topk: bad k / token counts");
        m.use();
        HIP_CHECK(hipMemcpy2DAsync(st.logits.p, (size_t)st.logits_ld * 4, logits, (size_t)m.V * 4, (size_t)m.V * 4, n_rows, hipMemcpyHostToDevice, m.stream));
        StepParams* sp = st.sp_h();
        memset(sp, 0, sizeof(*sp));
        sp->first_step = (n_tokens == sample_begin) ? 1 : 0;
        sp->n_rows = n_rows;
        for (int r = 0; r < n_rows; ++r) {
            std::vector<int64_t> row(tokens + (size_t)r * n_tokens, tokens + (size_t)(r + 1) * n_tokens);
            const int n = n_tokens - sample_begin;
            const bool last = n >= 1 && row[n_tokens - 1] >= d.ids.ts_begin;
            const bool second = n < 2 || row[n_tokens - 2] >= d.ids.ts_begin;
            st.tsmode_h()[r] = d.timestamps ? (last ? (second ? 1 : 2) : 0) : 0;
        }
        HIP_CHECK(hipMemcpyAsync(st.sp_dev.p, st.sp_host.p, st.sp_bytes(), hipMemcpyHostToDevice, m.stream));
        DevBuf filt;
        if (out_filtered) filt.alloc((size_t)n_rows * m.V * 4);
        launch_filter_topk(st.logits.as<float>(), n_rows, d.filter_cfg(), st.sp_d(), st.tsmode_d(), k, st.topk_lp_d(), st.topk_id_d(),
                           st.no_speech_d(), out_filtered ? filt.as<float>() : nullptr, m.stream);
        if (out_filtered) HIP_CHECK(hipMemcpyAsync(out_filtered, filt.p, (size_t)n_rows * m.V * 4, hipMemcpyDeviceToHost, m.stream));
        HIP_CHECK(hipStreamSynchronize(m.stream));
        memcpy(out_lp, st.topk_lp(), (size_t)n_rows * k * 4);
        memcpy(out_ids, st.topk_id(), (size_t)n_rows * k * 4);
    });
}

int wrs_decode_task_set_tokenizer(wrs_decode_task* t, const wrs_tokenizer* tok) {
    return guarded([&] { WRS_REQUIRE(t, "null task"); t->t.tokenizer = tok ? &tok->t : nullptr; });
}

// ------------------------------------------------------------------ audio ingest (host only)
static float* dup_samples(const std::vector<float>& v) {
    float* p = (float*)malloc(std::max<size_t>(1, v.size()) * sizeof(float));
    memcpy(p, v.data(), v.size() * sizeof(float));
    return p;
}
int wrs_load_audio(const char* path, float** out, int64_t* n_samples) {
    return guarded([&] {
        WRS_REQUIRE(path && out && n_samples, "wrs_load_audio: null argument");
        const std::vector<float> a = load_audio_file(path);
        *out = dup_samples(a);
        *n_samples = (int64_t)a.size();
    });
}
int wrs_resample(const float* in, int64_t n, int32_t rate_in, int32_t rate_out, float** out, int64_t* n_out) {
    return guarded([&] {
        WRS_REQUIRE((in || n == 0) && out && n_out, "wrs_resample: null argument");
        const std::vector<float> a = resample_f32(in, n, rate_in, rate_out);
        *out = dup_samples(a);
        *n_out = (int64_t)a.size();
    });
}
void wrs_audio_free(float* p) { free(p); }

// ------------------------------------------------------------------ tokenizer (host only)
int wrs_tokenizer_create_from_bytes(const void* bytes, size_t n_bytes, wrs_tokenizer** out) {
    return guarded([&] {
        WRS_REQUIRE(bytes && out, "wrs_tokenizer_create_from_bytes: null argument");
        std::unique_ptr<wrs_tokenizer> h(new wrs_tokenizer());
        h->t.load_json((const char*)bytes, n_bytes, false);
        *out = h.release();
    });
}
int wrs_tokenizer_create_from_file(const char* path, wrs_tokenizer** out) {
    return guarded([&] {
        WRS_REQUIRE(path && out, "wrs_tokenizer_create_from_file: null argument");
        std::ifstream f(path, std::ios::binary);
        WRS_REQUIRE(f.good(), "failed to read tokenizer file %s", path);
        std::stringstream ss;
        ss << f.rdbuf();
        const std::string bytes = ss.str();
        std::unique_ptr<wrs_tokenizer> h(new wrs_tokenizer());
        h->t.load_json(bytes.data(), bytes.size(), false);
        *out = h.release();
    });
}
void wrs_tokenizer_destroy(wrs_tokenizer* t) { delete t; }
int wrs_tokenizer_token_ids(const wrs_tokenizer* t, wrs_token_ids* out) {
    return guarded([&] { WRS_REQUIRE(t && out, "null argument"); *out = t->t.ids; });
}
static void copy_ids(const std::vector<int32_t>& v, int32_t* out, int32_t cap, int32_t* n) {
    WRS_REQUIRE(n && cap >= 0 && (out || cap == 0), "bad output buffer");
    *n = (int32_t)v.size();
    for (int32_t i = 0; i < cap && i < *n; ++i) out[i] = v[i];
}
int wrs_tokenizer_encode(const wrs_tokenizer* t, const char* utf8, size_t n_bytes, int32_t* out, int32_t cap, int32_t* n) {
    return guarded([&] { WRS_REQUIRE(t && (utf8 || n_bytes == 0), "null argument"); copy_ids(t->t.encode(std::string(utf8 ? utf8 : "", n_bytes)), out, cap, n); });
}
int wrs_tokenizer_decode(const wrs_tokenizer* t, const int64_t* ids, int64_t n, char** out_text, size_t* out_len) {
    return guarded([&] {
        WRS_REQUIRE(t && out_text && n >= 0 && (ids || n == 0), "null argument");
        const std::string s = t->t.decode(ids, n);
        *out_text = dup_cstr(s);
        if (out_len) *out_len = s.size();
    });
}
void wrs_string_free(char* s) { free(s); }
int wrs_tokenizer_non_speech_tokens(const wrs_tokenizer* t, int32_t* out, int32_t cap, int32_t* n) {
    return guarded([&] { WRS_REQUIRE(t, "null argument"); copy_ids(t->t.non_speech_tokens(), out, cap, n); });
}
int32_t wrs_tokenizer_token_to_id(const wrs_tokenizer* t, const char* token) { return (t && token) ? t->t.token_to_id(token) : -1; }
int32_t wrs_tokenizer_language_token(const wrs_tokenizer* t, const char* code) { return (t && code) ? t->t.language_token(code) : -1; }
int32_t wrs_tokenizer_vocab_size(const wrs_tokenizer* t) { return t ? (int32_t)t->t.id_to_tok.size() : 0; }

// ------------------------------------------------------------------ transcribe
int wrs_transcribe_task_create(wrs_model* m, const wrs_token_ids* ids, const wrs_transcribe_options* opt, wrs_transcribe_task** out) {
    return guarded([&] {
        WRS_REQUIRE(m && ids && opt && out, "wrs_transcribe_task_create: null argument");
        std::unique_ptr<wrs_transcribe_task> h(new wrs_transcribe_task());
        h->t.configure(&m->m, *ids, *opt);
        *out = h.release();
    });
}
int wrs_transcribe_task_set_tokenizer(wrs_transcribe_task* t, const wrs_tokenizer* tok) {
    return guarded([&] { WRS_REQUIRE(t, "null task"); t->t.set_tokenizer(tok ? &tok->t : nullptr); });
}
int wrs_transcribe_task_set_prompt_text(wrs_transcribe_task* t, const char* utf8) {
    return guarded([&] { WRS_REQUIRE(t, "null task"); t->t.set_prompt_text(utf8); });
}
int wrs_transcribe_task_run(wrs_transcribe_task* t, const float* audio, int64_t n, wrs_transcribe_output* out) {
    return guarded([&] { WRS_REQUIRE(t, "null task"); t->t.run(audio, n, out); });
}
int wrs_transcribe_task_run_mel(wrs_transcribe_task* t, const float* mel, int64_t n_frames, wrs_transcribe_output* out) {
    return guarded([&] { WRS_REQUIRE(t, "null task"); t->t.run_mel(mel, n_frames, out); });
}
int wrs_transcribe_task_run_chunked(wrs_transcribe_task* t, const float* audio, int64_t n, int32_t batch, wrs_transcribe_output* out) {
    return guarded([&] { WRS_REQUIRE(t, "null task"); t->t.run_chunked(audio, n, batch, out); });
}
void wrs_transcribe_output_free(wrs_transcribe_output* o) {
    if (!o) return;
    if (o->segments)
        for (int64_t i = 0; i < o->n_segments; ++i) free(o->segments[i].text);
    free(o->text);
    free(o->tokens); free(o->segments); free(o->window_seek); free(o->window_avg_logprob); free(o->window_no_speech_prob);
    memset(o, 0, sizeof(*o));
}
void wrs_transcribe_task_destroy(wrs_transcribe_task* t) {
    if (!t) return;
    (void)hipSetDevice(t->t.m->device);
    delete t;
}

// ------------------------------------------------------------------ measurement
int wrs_decode_task_last_run_times(wrs_decode_task* t, double* encoder_ms, double* prefill_ms, double* steps_ms, double* host_ms) {
    return guarded([&] {
        WRS_REQUIRE(t, "null task");
        if (encoder_ms) *encoder_ms = t->t.t_encoder_ms;
        if (prefill_ms) *prefill_ms = t->t.t_prefill_ms;
        if (steps_ms) *steps_ms = t->t.t_steps_ms;
        if (host_ms) *host_ms = t->t.t_host_ms;
    });
}
// live, sampled HIP-event timing of the cross-attention decode kernel inside the replayed step graph
int wrs_decode_task_set_cross_timing(wrs_decode_task* t, int enabled, int layer_stride) {
    return guarded([&] {
        WRS_REQUIRE(t && layer_stride >= 1, "bad argument");
        Model& m = *t->t.m;
        HIP_CHECK(hipStreamSynchronize(m.stream));
        m.cross_ev_enabled = enabled != 0;
        m.cross_ev_stride = layer_stride;
        m.cross_ev_step = 0;
        m.cross_ev_ms = 0; m.cross_ev_launches = 0; m.cross_ev_bytes = 0;
    });
}
int wrs_decode_task_get_cross_timing(wrs_decode_task* t, double* ms, int64_t* launches, double* bytes) {
    return guarded([&] {
        WRS_REQUIRE(t && ms && launches && bytes, "bad argument");
        Model& m = *t->t.m;
        *ms = m.cross_ev_ms; *launches = m.cross_ev_launches; *bytes = m.cross_ev_bytes;
    });
}

int wrs_decode_task_set_profiling(wrs_decode_task* t, int enabled) {
    return guarded([&] {
        WRS_REQUIRE(t, "null task");
        StageTimer* tm = t->t.m->timer;
        HIP_CHECK(hipStreamSynchronize(t->t.m->stream));
        tm->reset();
        tm->enabled = enabled != 0;
    });
}
int wrs_decode_task_stage_times(wrs_decode_task* t, wrs_stage_time* out, int32_t cap, int32_t* n) {
    return guarded([&] {
        WRS_REQUIRE(t && n, "null argument");
        StageTimer* tm = t->t.m->timer;
        HIP_CHECK(hipStreamSynchronize(t->t.m->stream));
        tm->collect();
        *n = (int32_t)tm->entries.size();
        for (int i = 0; i < *n && i < cap && out; ++i) {
            out[i].name = tm->entries[i].name;
            out[i].ms = tm->entries[i].ms;
            out[i].launches = tm->entries[i].launches;
            out[i].bytes = tm->entries[i].bytes;
            out[i].flops = tm->entries[i].flops;
        }
    });
}

}  // extern "C"
