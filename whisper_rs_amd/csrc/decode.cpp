// Host sequencer of one window batch.  Restates reference src/decode/mod.rs (DecodeTask),
// decode/extractor/{beam_search,greedy}.rs and decode/sequencer/max_likelihood.rs on top of the
// device step (Model::decoder_step) and the fused filter/top-k kernel.  The device produces
// (beam+1) (logprob, id) pairs per row per step in pinned memory; everything the reference does
// with HashMaps on the host stays on the host, made deterministic:
//   candidates keep first-insertion order, duplicates overwrite (score, source) in place
//   (beam_search.rs:77), stable sort by score descending (:86), newly finished sequences are
//   admitted in that order (:131-137), final candidates are listed in admission order.
#include "decode.h"
#include <algorithm>
#include <chrono>
#include <cmath>

namespace wrs {

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static double round_half_away(double x) { return x >= 0 ? std::floor(x + 0.5) : -std::floor(-x + 0.5); }

DecodeTask::~DecodeTask() {
    if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
    if (graph) (void)hipGraphDestroy(graph);
    if (side) { (void)hipStreamSynchronize(side); (void)hipStreamDestroy(side); }
    for (auto& e : side_done) if (e) (void)hipEventDestroy(e);
}

bool DecodeTask::side_idle() {
    if (!side) return true;
    const hipError_t e = hipStreamQuery(side);
    if (e == hipSuccess) return true;
    if (e != hipErrorNotReady) HIP_CHECK(e);
    (void)hipGetLastError();
    return false;
}

// Queue log-mel, AudioEncoder and the cross-attention K/V projection of a window batch on the side stream, into the
// cross-K/V buffer the decode steps are not reading.  The MFMA-bound encoder of batch i+1 then overlaps the HBM-bound
// decode steps of batch i.  The encoder workspace is single: queued batches run one after the other on the side stream.
void DecodeTask::encode_async(const float* audio, int n_audio, int mem_kind, bool is_mel) {
    WRS_REQUIRE(audio && n_audio >= 1 && n_audio <= max_audio, "encode_async: %d windows exceed the task capacity %d", n_audio, max_audio);
    WRS_REQUIRE(pending.size() < 2, "encode_async: two batches are already queued (call run_pending first)");
    WRS_REQUIRE(!m->parity_f32, "encode_async: the f32 parity mode has no overlapped path");
    m->use();
    features_n = 0;  // the encoder workspace is about to be overwritten
    if (!side) {
        int prio_least = 0, prio_greatest = 0;
        HIP_CHECK(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
        HIP_CHECK(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, prio_least));
        for (auto& e : side_done) HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    // the buffer that is neither queued nor (if nothing is queued) the one the last decode read: decode is synchronous
    // on the host, so no decode step is in flight here and, with nothing queued, the buffer in use is free as well — the
    // second buffer is only allocated once a batch is queued while another one is pending
    int buf = pending.empty() ? st.cross_sel : (pending.back().buf ^ 1);
    if (buf == 1 && st.cross_kp2.bytes == 0) {
        st.cross_kp2.alloc(st.cross_kp.bytes);
        st.cross_vp2.alloc(st.cross_vp.bytes);
    }
    hipStream_t main_stream = m->stream;
    m->stream = side;  // every launch helper of the model enqueues on m->stream
    try {
        if (is_mel) {  // windows of a spectrogram the caller normalised itself (whole-file max): [n_audio, 80, 3000]
            m->load_mel(audio, n_audio, mem_kind, WRS_N_FRAMES, 0);
        } else {
            const float* a_d = audio;
            if (mem_kind == WRS_MEM_HOST) {
                audio_dev.ensure((size_t)n_audio * WRS_N_SAMPLES * 4);
                HIP_CHECK(hipMemcpyAsync(audio_dev.p, audio, (size_t)n_audio * WRS_N_SAMPLES * 4, hipMemcpyHostToDevice, side));
                a_d = audio_dev.as<float>();
            }
            mel_dev.ensure((size_t)n_audio * 80 * WRS_N_FRAMES * 4);
            log_mel_device(mel_ws, a_d, n_audio, WRS_N_SAMPLES, nullptr, mel_dev.as<float>(), side);
            m->load_mel(mel_dev.as<float>(), n_audio, WRS_MEM_DEVICE, WRS_N_FRAMES, 0);
        }
        m->encoder_from_mel_t(n_audio);
        st.cross_wsel = buf;
        m->load_cross_kv(st, m->enc_ws.xa16.as<half_t>(), n_audio);
        st.cross_wsel = st.cross_sel;
        HIP_CHECK(hipEventRecord(side_done[buf], side));
    } catch (...) {
        m->stream = main_stream;
        st.cross_wsel = st.cross_sel;
        throw;
    }
    m->stream = main_stream;
    pending.push_back({buf, n_audio});
}

void DecodeTask::run_pending(wrs_decode_result* out) {
    WRS_REQUIRE(out, "run_pending: null result");
    WRS_REQUIRE(!pending.empty(), "run_pending: nothing queued (call encode_async first)");
    m->use();
    const Pending pd = pending.front();
    pending.erase(pending.begin());
    const double t0 = now_ms();
    HIP_CHECK(hipEventSynchronize(side_done[pd.buf]));  // host-visible part of the encoder time (what was not hidden)
    st.cross_sel = st.cross_wsel = pd.buf;
    st.n_audio = pd.n_audio;
    run_decode(pd.n_audio, out);
    t_encoder_ms = now_ms() - t0 - (t_prefill_ms + t_steps_ms + t_host_ms);
}

void DecodeTask::configure(Model* model, const wrs_token_ids& tid, const wrs_decode_options& o) {
    WRS_REQUIRE(model && model->finalized, "DecodeTask: model is not finalized");
    m = model;
    ids = tid;
    const int V = m->V;
    WRS_REQUIRE(ids.eot >= 0 && ids.eot < V && ids.sot >= 0 && ids.sot < V && ids.ts_begin > 0 && ids.ts_begin <= V &&
                    ids.no_speech >= 0 && ids.no_speech < V && ids.no_timestamps >= 0 && ids.no_timestamps < V &&
                    ids.startofprev >= 0 && ids.startofprev < V && ids.space >= 0 && ids.space < V,
                "DecodeTask: special token ids out of range for n_vocab %d", V);
    sample_len = o.sample_len > 0 ? o.sample_len : m->n_ctx / 2;  // decode/mod.rs:129-131
    mode = o.extract_mode;
    WRS_REQUIRE(mode == WRS_EXTRACT_GREEDY || mode == WRS_EXTRACT_BEAM_SEARCH, "DecodeTask: unknown extract mode %d", mode);
    group = o.group_size;
    WRS_REQUIRE(group >= 1 && group <= 7, "DecodeTask: group/beam size %d not in 1..7", group);
    patience = o.patience;
    if (mode == WRS_EXTRACT_BEAM_SEARCH) WRS_REQUIRE(patience > 0, "DecodeTask: patience must be > 0");
    has_len_penalty = o.has_len_penalty != 0;
    len_penalty = o.len_penalty;
    timestamps = o.timestamps != 0;
    suppress_blank = o.suppress_blank != 0;
    position_quirk = o.position_offset_quirk != 0;
    max_audio = o.max_audio > 0 ? o.max_audio : 1;
    mit = -1;
    if (timestamps && o.has_max_initial_timestamp) {  // decode/mod.rs:176-183
        const double precision = (double)WRS_CHUNK_LENGTH / (double)m->T;
        mit = (int)round_half_away(o.max_initial_timestamp / precision);
    }
    suppress_ids.clear();
    if ((o.suppress_tokens && o.n_suppress_tokens > 0) || o.suppress_non_speech) {  // decode/mod.rs:166-174
        for (int i = 0; i < o.n_suppress_tokens; ++i) suppress_ids.push_back(o.suppress_tokens[i]);
        if (o.suppress_non_speech) {
            WRS_REQUIRE(o.non_speech_tokens || o.n_non_speech_tokens == 0, "DecodeTask: suppress_non_speech without a token list");
            for (int i = 0; i < o.n_non_speech_tokens; ++i) suppress_ids.push_back(o.non_speech_tokens[i]);
        }
    }
    for (int id : suppress_ids) WRS_REQUIRE(id >= 0 && id < V, "DecodeTask: suppress id %d out of range", id);
    m->use();
    suppress_dev.alloc(std::max<size_t>(suppress_ids.size(), 1) * 4);
    if (!suppress_ids.empty()) {
        HIP_CHECK(hipMemcpyAsync(suppress_dev.p, suppress_ids.data(), suppress_ids.size() * 4, hipMemcpyHostToDevice, utility_stream()));
        HIP_CHECK(hipStreamSynchronize(utility_stream()));
    }
    sot_tail.clear();
    audio_lang.clear();
    for (int i = 0; i < o.n_sot_sequence_tail; ++i) {
        WRS_REQUIRE(o.sot_sequence_tail && o.sot_sequence_tail[i] >= 0 && o.sot_sequence_tail[i] < V, "DecodeTask: sot sequence token out of range");
        sot_tail.push_back(o.sot_sequence_tail[i]);
    }
    set_prompt(nullptr, 0);
    // K/V cache positions by need (prompt + sample_len), grown by ensure_ctx when a longer prompt is set: at large-v2 and 1 535
    // rows all 448 positions would take 113 GB
    m->init_state(st, max_audio, group, ctx_for(sample_begin + sample_len));
    const char* env = exp_env("WRS_NO_GRAPH");
    use_graph = !(env && env[0] == '1');
}

int DecodeTask::ctx_for(int positions) const { return (int)std::min<int64_t>(m->n_ctx, round_up(std::max(positions, 1), 32)); }
// the self-attention cache must hold `positions` tokens per row: grow it (contents lost: between runs only) and drop the step graph
void DecodeTask::ensure_ctx(int positions) {
    if (std::min(positions, m->n_ctx) <= st.ctx) return;
    HIP_CHECK(hipStreamSynchronize(m->stream));
    m->resize_ctx(st, m->n_ctx);  // a prompt outgrew the initial size (long-form conditioning keeps growing it): all positions, once
    if (graph_exec) { (void)hipGraphExecDestroy(graph_exec); graph_exec = nullptr; }
    graph_rows = -1;
}

void DecodeTask::set_prompt(const int64_t* p, int n) {  // decode/mod.rs:46-92,206-218
    initial_tokens.clear();
    if (p && n > 0) {
        initial_tokens.push_back(ids.startofprev);
        for (int i = 0; i < n; ++i) {
            WRS_REQUIRE(p[i] >= 0 && p[i] < m->V, "set_prompt: token %ld out of range", (long)p[i]);
            initial_tokens.push_back(p[i]);
        }
    }
    initial_tokens.push_back(ids.sot);  // sequence_sot() == [sot] (tokenize.rs:209-213)
    for (int64_t t : sot_tail) initial_tokens.push_back(t);  // multilingual: [sot, lang, task]
    sample_begin = (int)initial_tokens.size();
    sot_idx = 0;
    for (size_t i = 0; i < initial_tokens.size(); ++i)
        if (initial_tokens[i] == ids.sot) { sot_idx = (int)i; break; }
}

void DecodeTask::set_languages(const int32_t* lang, int n) {
    if (n <= 0 || !lang) { audio_lang.clear(); return; }
    WRS_REQUIRE(!sot_tail.empty(), "set_languages: the task has no language slot (n_sot_sequence_tail == 0)");
    WRS_REQUIRE(n <= max_audio, "set_languages: %d languages exceed the task capacity %d", n, max_audio);
    for (int i = 0; i < n; ++i) WRS_REQUIRE(lang[i] >= 0 && lang[i] < m->V, "set_languages: token %d out of range", lang[i]);
    audio_lang.assign(lang, lang + n);
}

// OpenAI whisper decoding.py detect_language: logits of the decoder on [sot] at position 0, restricted to the
// language tokens; argmax + softmax.  The encoder output of the windows is in m->enc_ws.xa16.
void DecodeTask::detect_language(int n_audio, int first_id, int n_lang, int32_t* out_ids, float* out_probs) {
    WRS_REQUIRE(n_audio >= 1 && n_audio <= max_audio, "detect_language: %d windows exceed the task capacity %d", n_audio, max_audio);
    WRS_REQUIRE(out_ids && first_id >= 0 && n_lang >= 1 && first_id + n_lang <= m->V, "detect_language: bad language id range");
    m->use();
    hipStream_t s = m->stream;
    m->load_cross_kv(st, m->enc_ws.xa16.as<half_t>(), n_audio);
    launch_anc_init(st.anc.as<int>(), (int64_t)st.r_cap * st.ctx, st.ctx, st.r_cap, group, s);
    StepParams* sp = st.sp_h();
    sp->cache_pos = 0; sp->pos_idx = 0; sp->parity = 0; sp->first_step = 1; sp->want_no_speech = 0; sp->n_rows = n_audio;
    for (int a = 0; a < n_audio; ++a) { st.tokens_h()[a] = ids.sot; st.src_h()[a] = a; st.tsmode_h()[a] = 0; }
    upload_step(n_audio);
    m->decoder_step(st, n_audio, group, 1, false, true);
    DevBuf d_ids, d_probs;
    d_ids.alloc((size_t)n_audio * 4);
    d_probs.alloc((size_t)n_audio * n_lang * 4);
    launch_lang_detect(st.logits.as<float>(), n_audio, st.logits_ld, first_id, n_lang, d_ids.as<int>(), d_probs.as<float>(), s);
    HIP_CHECK(hipMemcpyAsync(out_ids, d_ids.p, (size_t)n_audio * 4, hipMemcpyDeviceToHost, s));
    if (out_probs) HIP_CHECK(hipMemcpyAsync(out_probs, d_probs.p, (size_t)n_audio * n_lang * 4, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
}

FilterCfg DecodeTask::filter_cfg() const {
    FilterCfg c;
    c.n_vocab = m->V;
    c.ld = st.logits_ld;
    c.ts_begin = ids.ts_begin; c.eot = ids.eot; c.no_timestamps = ids.no_timestamps; c.no_speech = ids.no_speech;
    c.space = ids.space;
    c.suppress_blank = suppress_blank ? 1 : 0;
    c.timestamps = timestamps ? 1 : 0;
    c.max_initial_ts_index = mit;
    c.suppress_ids = suppress_dev.as<int>();
    c.n_suppress = (int)suppress_ids.size();
    c.ws = const_cast<DecState&>(st).filter_ws.as<float>();
    c.ws_rows = std::min(st.r_cap, FILTER_SPLIT_MAX_ROWS);
    return c;
}

void DecodeTask::upload_step(int) {
    HIP_CHECK(hipMemcpyAsync(st.sp_dev.p, st.sp_host.p, st.sp_bytes(), hipMemcpyHostToDevice, m->stream));
}

static int ts_mode_of(const SeqStore& seqs, int row, int sample_begin, int tsb) {
    // timestamp_tokens.rs:47-66
    const int n = seqs.len(row) - sample_begin;
    const bool last = n >= 1 && seqs.back(row) >= tsb;
    const bool second = n < 2 || seqs.back(seqs.parent(row)) >= tsb;
    return last ? (second ? 1 : 2) : 0;
}

void DecodeTask::init_rows(int n_audio) {
    const int G = group, R = n_audio * G;
    seqs.clear();
    seqs.reserve((size_t)n_audio * initial_tokens.size() + (size_t)(R + n_audio) * (size_t)(std::max(sample_len, 1) + 1));
    rows.resize(R);
    std::vector<int64_t> prompt;
    int shared = -1;
    for (int a = 0; a < n_audio; ++a) {
        int root;
        if (audio_lang.empty()) {
            if (shared < 0) shared = seqs.chain(initial_tokens);
            root = shared;
        } else {
            prompt = initial_tokens;
            prompt[sot_idx + 1] = audio_lang[a];
            root = seqs.chain(prompt);
        }
        for (int j = 0; j < G; ++j) rows[a * G + j] = root;
    }
}

void DecodeTask::step_decode(int R) {
    const int k = (mode == WRS_EXTRACT_BEAM_SEARCH) ? group + 1 : 1;
    const FilterCfg cfg = filter_cfg();
    auto body = [&]() {
        upload_step(R);
        m->decoder_step(st, R, 1, group, true, true);
        launch_filter_topk(st.logits.as<float>(), R, cfg, st.sp_d(), st.tsmode_d(), k, st.topk_lp_d(), st.topk_id_d(),
                           st.no_speech_d(), nullptr, m->stream);
    };
    const bool profiling = m->timer && m->timer->enabled;
    // roofline sampling: every n-th step is launched eagerly with event pairs around the cross-attention kernel; in
    // overlapped mode only while no encoder work is queued on the side stream, so a sample is the kernel's own duration
    const bool sample_now = m->cross_ev_enabled && (m->cross_ev_step++ % m->cross_ev_every == 0) && side_idle();
    if (use_graph && !profiling && !sample_now) {
        if (!graph_exec || graph_rows != R || graph_sel != st.cross_sel) {
            if (graph_exec) { (void)hipGraphExecDestroy(graph_exec); graph_exec = nullptr; }
            if (graph) { (void)hipGraphDestroy(graph); graph = nullptr; }
            HIP_CHECK(hipStreamBeginCapture(m->stream, hipStreamCaptureModeThreadLocal));
            try {
                body();
            } catch (...) {
                hipGraph_t g = nullptr;
                (void)hipStreamEndCapture(m->stream, &g);
                if (g) (void)hipGraphDestroy(g);
                throw;
            }
            HIP_CHECK(hipStreamEndCapture(m->stream, &graph));
            HIP_CHECK(hipGraphInstantiate(&graph_exec, graph, nullptr, nullptr, 0));
            graph_rows = R;
            graph_sel = st.cross_sel;
        }
        HIP_CHECK(hipGraphLaunch(graph_exec, m->stream));
        HIP_CHECK(hipStreamSynchronize(m->stream));
    } else {
        m->cross_ev_now = sample_now;
        body();
        HIP_CHECK(hipStreamSynchronize(m->stream));
        m->collect_cross_events(R / group);
        m->cross_ev_now = false;
    }
}

void DecodeTask::record_candidates(int R, bool step0) {
    if (!keep_candidates) return;
    const int k = (mode == WRS_EXTRACT_BEAM_SEARCH) ? group + 1 : 1;
    for (int r = 0; r < R; ++r) {
        const int trow = step0 ? r / group : r;  // step 0: one prefill row per audio, shared by its beams
        for (int c = 0; c < k; ++c) {
            tr_cand_lp.push_back(st.topk_lp()[trow * k + c]);
            tr_cand_id.push_back(st.topk_id()[trow * k + c]);
        }
    }
}

// test aid: replace the outcome of step `step` by the given decisions (the finished lists keep what update_* admitted)
void DecodeTask::apply_forced(int step, const std::vector<int>& old_rows) {
    const int R = (int)old_rows.size();
    for (int r = 0; r < R; ++r) {
        const int src = forced_src[(size_t)step * R + r];
        WRS_REQUIRE(src >= 0 && src < R, "forced decode: source %d out of range", src);
        rows[r] = seqs.append(old_rows[src], forced_tok[(size_t)step * R + r]);
        sources[r] = src;
        sum_logprobs[r] = forced_slp[(size_t)step * R + r];
    }
}

void DecodeTask::record_trace(int R) {
    for (int r = 0; r < R; ++r) {
        tr_tokens.push_back((int32_t)seqs.back(rows[r]));
        tr_src.push_back(sources[r]);
        tr_slp.push_back(sum_logprobs[r]);
    }
}

// BeamSearchTokenExtractor::update (beam_search.rs:22-149).  step0: every beam of an audio reads the
// audio's single prefill row (identical logits for identical rows; the reference computes them 5x).
bool DecodeTask::update_beam(int n_audio, bool step0, const float* lp, const int* id) {
    const int G = group, k = G + 1;
    struct Cand { int beam; int64_t token; double score; int source; };
    // a step appends one node per row to the prefix tree (no sequence is copied, nothing is allocated on the step path)
    std::vector<int>& new_rows = scratch_rows;
    new_rows.resize((size_t)n_audio * G);
    std::vector<float>& new_slp = scratch_slp;
    std::vector<int>& new_src = scratch_src;
    new_slp.assign((size_t)n_audio * G, 0.f);
    new_src.assign((size_t)n_audio * G, 0);
    const int max_candidates = (int)round_half_away((double)(patience * (float)G));  // :122
    bool completed = true;
    std::vector<Cand> cands;
    cands.reserve((size_t)G * k);
    for (int i = 0; i < n_audio; ++i) {
        cands.clear();
        for (int j = 0; j < G; ++j) {
            const int idx = i * G + j;
            const int trow = step0 ? i : idx;
            for (int c = 0; c < k; ++c) {
                const int tok = id[trow * k + c];
                if (tok < 0) continue;
                const double score = (double)sum_logprobs[idx] + (double)lp[trow * k + c];  // :71
                bool dup = false;
                for (auto& e : cands) {
                    if (e.token == tok && (e.beam == j || seqs.equal(rows[i * G + e.beam], rows[idx]))) {
                        e.score = score;  // HashMap insert overwrites the value, key position unchanged (:77)
                        e.source = idx;
                        dup = true;
                        break;
                    }
                }
                if (!dup) cands.push_back({j, tok, score, idx});
            }
        }
        // :86 stable descending sort (at most G * (G + 1) candidates: insertion sort, no temporary buffer)
        for (size_t a = 1; a < cands.size(); ++a) {
            const Cand c = cands[a];
            size_t b = a;
            for (; b > 0 && cands[b - 1].score < c.score; --b) cands[b] = cands[b - 1];
            cands[b] = c;
        }
        std::vector<FinishedSeq> newly;
        int saved = 0;
        for (const auto& c : cands) {  // :87-104
            if (c.token == ids.eot) {
                FinishedSeq f;
                seqs.materialize(rows[i * G + c.beam], f.tokens);
                f.tokens.push_back(c.token);
                f.score = c.score;
                newly.push_back(std::move(f));
            } else {
                const int dst = i * G + saved;
                new_rows[dst] = seqs.append(rows[i * G + c.beam], c.token);
                new_slp[dst] = (float)c.score;  // :91-95 stored into the f32 tensor
                new_src[dst] = c.source;
                if (++saved == G) break;
            }
        }
        WRS_REQUIRE(saved == G, "beam search: only %d live candidates for audio %d (logits degenerate?)", saved, i);
        auto& prev = finished[i];
        for (auto& f : newly) {  // :124-138
            if ((int)prev.size() >= max_candidates) break;
            bool found = false;
            for (auto& e : prev)
                if (e.tokens == f.tokens) { e.score = f.score; found = true; break; }
            if (!found) prev.push_back(std::move(f));
        }
        if ((int)prev.size() < max_candidates) completed = false;  // :141-146
    }
    rows.swap(new_rows);
    sum_logprobs.swap(new_slp);
    sources.swap(new_src);
    return completed;
}

// GreedyTokenExtractor::update (greedy.rs:17-51), temperature 0.  completed = every row's last
// token is EOT (OpenAI); the literal `tokens.eq(eot).all()` can never hold and the
// outputs are identical either way because finished rows keep emitting EOT with no logprob added.
bool DecodeTask::update_greedy(int n_audio, bool step0, const float* lp, const int* id) {
    const int G = group;
    bool completed = true;
    for (int r = 0; r < n_audio * G; ++r) {
        const int trow = step0 ? r / G : r;
        int64_t tok = id[trow];
        WRS_REQUIRE(tok >= 0, "greedy: no finite logit in row %d", r);
        const bool last_eot = seqs.back(rows[r]) == ids.eot;
        if (!last_eot) sum_logprobs[r] = sum_logprobs[r] + lp[trow];  // :38 (f32 add)
        else tok = ids.eot;                                           // :40-44
        rows[r] = seqs.append(rows[r], tok);
        sources[r] = r;
        if (tok != ids.eot) completed = false;
    }
    return completed;
}

// The decoder's first forward pass (decode/mod.rs:234-258): the S = sample_begin prompt tokens of every audio, on one row
// per audio.  Leaves the step-0 candidates (top-k of the filtered logits of the last prompt position) in the pinned
// buffer and the no-speech probabilities (softmax of the raw logits at sot_idx, :250-255) in no_speech_out[n_audio].
// S >= 2: one batched causal pass (Model::decoder_prefill); S == 1, the f32 parity mode or WRS_PREFILL_SEQUENTIAL=1:
// one decoder step per position.
void DecodeTask::prefill(int n_audio, double* no_speech_out) {
    const int G = group, S = sample_begin;
    const int k = (mode == WRS_EXTRACT_BEAM_SEARCH) ? G + 1 : 1;
    const FilterCfg cfg = filter_cfg();
    hipStream_t s = m->stream;
    static const bool sequential_env = exp_env("WRS_PREFILL_SEQUENTIAL") && exp_env("WRS_PREFILL_SEQUENTIAL")[0] == '1';
    auto filter_at = [&](bool want_ns) {
        StepParams* sp = st.sp_h();
        sp->cache_pos = S - 1; sp->pos_idx = S - 1; sp->parity = 0; sp->first_step = 1;
        sp->want_no_speech = want_ns ? 1 : 0; sp->n_rows = n_audio;
        for (int a = 0; a < n_audio; ++a) st.tsmode_h()[a] = 0;
        upload_step(n_audio);
    };
    if (S >= 2 && !m->parity_f32 && !sequential_env) {
        std::vector<int> toks((size_t)n_audio * S);
        for (int a = 0; a < n_audio; ++a)
            for (int p = 0; p < S; ++p) toks[(size_t)a * S + p] = (int)initial_token(a, p);
        m->decoder_prefill(st, n_audio, S, 0, G, toks.data(), false);
        if (sot_idx != S - 1) {  // no-speech probability from the logits at the sot position
            filter_at(true);
            m->prefill_logits(st, n_audio, S, sot_idx);
            launch_filter_topk(st.logits.as<float>(), n_audio, cfg, st.sp_d(), st.tsmode_d(), k, st.topk_lp_d(), st.topk_id_d(), st.no_speech_d(), nullptr, s);
            HIP_CHECK(hipStreamSynchronize(s));  // also: `toks` and the pinned step block are free again
            for (int a = 0; a < n_audio; ++a) no_speech_out[a] = (double)st.no_speech_h()[a];
        }
        filter_at(sot_idx == S - 1);
        m->prefill_logits(st, n_audio, S, S - 1);
        launch_filter_topk(st.logits.as<float>(), n_audio, cfg, st.sp_d(), st.tsmode_d(), k, st.topk_lp_d(), st.topk_id_d(), st.no_speech_d(), nullptr, s);
        HIP_CHECK(hipStreamSynchronize(s));
        if (sot_idx == S - 1)
            for (int a = 0; a < n_audio; ++a) no_speech_out[a] = (double)st.no_speech_h()[a];
        return;
    }
    for (int p = 0; p < S; ++p) {
        StepParams* sp = st.sp_h();
        sp->cache_pos = p; sp->pos_idx = p; sp->parity = 0; sp->first_step = 1;
        sp->want_no_speech = (p == sot_idx) ? 1 : 0; sp->n_rows = n_audio;
        for (int a = 0; a < n_audio; ++a) { st.tokens_h()[a] = (int)initial_token(a, p); st.src_h()[a] = a; st.tsmode_h()[a] = 0; }
        upload_step(n_audio);
        const bool want_logits = (p == S - 1) || p == sot_idx;
        m->decoder_step(st, n_audio, G, 1, false, want_logits);
        if (want_logits)
            launch_filter_topk(st.logits.as<float>(), n_audio, cfg, st.sp_d(), st.tsmode_d(), k, st.topk_lp_d(), st.topk_id_d(), st.no_speech_d(), nullptr, s);
        // the pinned step-parameter block is rewritten for the next position: its async H2D copy must have run
        HIP_CHECK(hipStreamSynchronize(s));
        if (p == sot_idx)
            for (int a = 0; a < n_audio; ++a) no_speech_out[a] = (double)st.no_speech_h()[a];  // decode/mod.rs:250-255
    }
}

void DecodeTask::run_loaded(int n_audio, wrs_decode_result* out) {
    WRS_REQUIRE(n_audio >= 1 && n_audio <= max_audio, "run: %d windows exceed the task capacity %d", n_audio, max_audio);
    m->use();
    const double t0 = now_ms();
    m->encoder_from_mel_t(n_audio);  // decode/mod.rs:296
    run_encoded(n_audio, out);
    t_encoder_ms += 0;  // encoder is asynchronous; its time is attributed inside run_encoded (first sync)
    (void)t0;
}

void DecodeTask::run_encoded(int n_audio, wrs_decode_result* out) {
    WRS_REQUIRE(out, "run: null result");
    WRS_REQUIRE(n_audio >= 1 && n_audio <= max_audio, "run: %d windows exceed the task capacity %d", n_audio, max_audio);
    m->use();
    memset(out, 0, sizeof(*out));
    const int G = group, R = n_audio * G, S = sample_begin;
    WRS_REQUIRE(S <= m->n_ctx, "prompt of %d tokens exceeds n_text_ctxs %d (the reference panics slicing positional_embedding)", S, m->n_ctx);
    hipStream_t s = m->stream;

    WRS_REQUIRE(pending.empty(), "run: batches queued with encode_async must be decoded (run_pending) first");
    double tA = now_ms();
    st.cross_wsel = st.cross_sel;
    m->load_cross_kv(st, m->enc_ws.xa16.as<half_t>(), n_audio);
    HIP_CHECK(hipStreamSynchronize(s));  // attribute encoder + cross-K/V time (everything queued so far)
    run_decode(n_audio, out);
    t_encoder_ms = now_ms() - tA - (t_prefill_ms + t_steps_ms + t_host_ms);
}

void DecodeTask::run_decode(int n_audio, wrs_decode_result* out) {
    memset(out, 0, sizeof(*out));
    const int G = group, R = n_audio * G, S = sample_begin;
    WRS_REQUIRE(S <= m->n_ctx, "prompt of %d tokens exceeds n_text_ctxs %d (the reference panics slicing positional_embedding)", S, m->n_ctx);
    hipStream_t s = m->stream;
    ensure_ctx(S + sample_len);
    launch_anc_init(st.anc.as<int>(), (int64_t)st.r_cap * st.ctx, st.ctx, st.r_cap, G, s);
    t_prefill_ms = t_steps_ms = t_host_ms = 0;
    double tA = now_ms();

    // host state (decode/mod.rs:297-325)
    if (!audio_lang.empty()) WRS_REQUIRE((int)audio_lang.size() >= n_audio, "run: %d windows but only %d languages set", n_audio, (int)audio_lang.size());
    init_rows(n_audio);
    scratch_rows.resize(R);
    tr_tokens.reserve((size_t)R * sample_len); tr_src.reserve((size_t)R * sample_len); tr_slp.reserve((size_t)R * sample_len);
    sum_logprobs.assign(R, 0.f);
    sources.assign(R, 0);
    finished.assign(n_audio, {});
    no_speech.assign(n_audio, NAN);
    tr_tokens.clear(); tr_src.clear(); tr_slp.clear();
    tr_cand_lp.clear(); tr_cand_id.clear();
    tr_rows = R;
    n_steps = 0;

    const int k = (mode == WRS_EXTRACT_BEAM_SEARCH) ? G + 1 : 1;
    const FilterCfg cfg = filter_cfg();

    // ---- forward pass i = 0 (decode/mod.rs:234-245): the S prompt tokens on ONE row per audio (all beams of an audio are
    // identical until the first selection)
    prefill(n_audio, no_speech.data());
    if (m->timer && m->timer->enabled) { HIP_CHECK(hipStreamSynchronize(s)); m->timer->collect(); }
    t_prefill_ms = now_ms() - tA;

    tA = now_ms();
    record_candidates(R, true);
    std::vector<int> old_rows;
    if (forced_n > 0) old_rows = rows;
    bool completed = (mode == WRS_EXTRACT_BEAM_SEARCH) ? update_beam(n_audio, true, st.topk_lp(), st.topk_id()) : update_greedy(n_audio, true, st.topk_lp(), st.topk_id());
    if (forced_n > 0) apply_forced(0, old_rows);
    record_trace(R);
    n_steps = 1;
    t_host_ms += now_ms() - tA;

    for (int i = 1; i < sample_len; ++i) {  // decode/mod.rs:234-277
        if (completed || seqs.len(rows[0]) > m->n_ctx) break;
        tA = now_ms();
        StepParams* sp = st.sp_h();
        sp->cache_pos = S + i - 1;
        sp->pos_idx = position_quirk ? i : S + i - 1;
        WRS_REQUIRE(sp->pos_idx < m->n_ctx && sp->cache_pos < st.ctx, "decode: position %d out of range (cache holds %d)", sp->cache_pos, st.ctx);
        sp->parity = i & 1; sp->first_step = 0; sp->want_no_speech = 0; sp->n_rows = R;
        for (int r = 0; r < R; ++r) {
            st.tokens_h()[r] = (int)seqs.back(rows[r]);
            st.src_h()[r] = sources[r];
            st.tsmode_h()[r] = timestamps ? ts_mode_of(seqs, rows[r], S, ids.ts_begin) : 0;
        }
        const double t1 = now_ms();
        step_decode(R);
        const double t2 = now_ms();
        if (m->timer && m->timer->enabled) m->timer->collect();
        record_candidates(R, false);
        if (i < forced_n) old_rows = rows;
        completed = (mode == WRS_EXTRACT_BEAM_SEARCH) ? update_beam(n_audio, false, st.topk_lp(), st.topk_id()) : update_greedy(n_audio, false, st.topk_lp(), st.topk_id());
        if (i < forced_n) apply_forced(i, old_rows);
        record_trace(R);
        n_steps = i + 1;
        t_steps_ms += t2 - t1;
        t_host_ms += (t1 - tA) + (now_ms() - t2);
        tA = now_ms();
    }
    finalize(n_audio, out);
    out->n_steps = n_steps;
}

// Host sequencer alone (test hook, no device): the candidate lists a device step would have produced are given by the
// caller — lp/id [n_steps][rows][k], k = beam + 1 (beam search) or 1 (greedy); step 0 reads row a*G of audio a like the
// real loop (all beams of an audio are one prefill row) — and update / finalize / rank run exactly as in run_decode.
void DecodeTask::simulate(const wrs_token_ids& tid, const wrs_decode_options& o, const int64_t* initial, int n_initial, int n_audio,
                          int n_steps, const float* lp, const int32_t* id, wrs_decode_result* out) {
    WRS_REQUIRE(initial && n_initial >= 1 && n_audio >= 1 && n_steps >= 1 && lp && id && out, "simulate: bad argument");
    m = nullptr;
    ids = tid;
    mode = o.extract_mode;
    group = o.group_size;
    WRS_REQUIRE(group >= 1 && group <= 7, "simulate: group/beam size %d not in 1..7", group);
    patience = o.patience;
    has_len_penalty = o.has_len_penalty != 0;
    len_penalty = o.len_penalty;
    initial_tokens.assign(initial, initial + n_initial);
    sample_begin = n_initial;
    const int G = group, R = n_audio * G, k = (mode == WRS_EXTRACT_BEAM_SEARCH) ? G + 1 : 1;
    memset(out, 0, sizeof(*out));
    audio_lang.clear();
    sample_len = std::max(1, n_steps);
    init_rows(n_audio);
    sum_logprobs.assign(R, 0.f);
    sources.assign(R, 0);
    finished.assign(n_audio, {});
    no_speech.assign(n_audio, NAN);
    tr_tokens.clear(); tr_src.clear(); tr_slp.clear();
    tr_rows = R;
    n_steps = std::max(1, n_steps);
    std::vector<float> lp0((size_t)R * k);
    std::vector<int> id0((size_t)R * k);
    int taken = 0;
    for (int s = 0; s < n_steps; ++s) {
        const float* l = lp + (size_t)s * R * k;
        const int32_t* d = id + (size_t)s * R * k;
        if (s == 0) {  // compact to one row per audio, as the prefill leaves it in the pinned candidate buffer
            for (int a = 0; a < n_audio; ++a)
                for (int c = 0; c < k; ++c) { lp0[(size_t)a * k + c] = l[(size_t)a * G * k + c]; id0[(size_t)a * k + c] = d[(size_t)a * G * k + c]; }
            l = lp0.data(); d = id0.data();
        }
        const bool completed = (mode == WRS_EXTRACT_BEAM_SEARCH) ? update_beam(n_audio, s == 0, l, d) : update_greedy(n_audio, s == 0, l, d);
        record_trace(R);
        taken = s + 1;
        if (completed) break;
    }
    this->n_steps = taken;
    finalize(n_audio, out);
    out->n_steps = taken;
}

// Teacher-forced run (test hook).  Step s uses forced tokens/sources [s][R] as the outcome of step s; the
// candidates the device produced for step s (before that outcome is applied) are returned in out_lp/out_id
// [n_steps][R][k].  Step 0 candidates come from the per-audio prefill row, replicated over its beams.
void DecodeTask::run_forced(int n_audio, int n_steps, const int32_t* ftok, const int32_t* fsrc, float* out_lp,
                            int32_t* out_id, double* out_no_speech, bool have_features) {
    WRS_REQUIRE(n_audio >= 1 && n_audio <= max_audio && n_steps >= 1 && ftok && fsrc && out_lp && out_id, "run_forced: bad argument");
    m->use();
    if (!have_features) m->encoder_from_mel_t(n_audio);
    const int G = group, R = n_audio * G, S = sample_begin;
    const int k = (mode == WRS_EXTRACT_BEAM_SEARCH) ? G + 1 : 1;
    WRS_REQUIRE(S + n_steps - 1 <= m->n_ctx, "run_forced: %d prompt + %d steps exceed n_text_ctxs", S, n_steps);
    hipStream_t s = m->stream;
    ensure_ctx(S + n_steps);
    m->load_cross_kv(st, m->enc_ws.xa16.as<half_t>(), n_audio);
    launch_anc_init(st.anc.as<int>(), (int64_t)st.r_cap * st.ctx, st.ctx, st.r_cap, G, s);
    const FilterCfg cfg = filter_cfg();
    {
        const int keep = sample_len;
        sample_len = std::max(sample_len, n_steps);  // node reservation of init_rows
        init_rows(n_audio);
        sample_len = keep;
    }
    {
        std::vector<double> ns(n_audio, NAN);
        prefill(n_audio, ns.data());
        if (out_no_speech)
            for (int a = 0; a < n_audio; ++a) out_no_speech[a] = ns[a];
    }
    for (int r = 0; r < R; ++r)
        for (int c = 0; c < k; ++c) {
            out_lp[(size_t)r * k + c] = st.topk_lp()[(r / G) * k + c];
            out_id[(size_t)r * k + c] = st.topk_id()[(r / G) * k + c];
        }
    auto apply = [&](int step) {
        std::vector<int> nr(R);
        for (int r = 0; r < R; ++r) {
            const int src = fsrc[(size_t)step * R + r];
            WRS_REQUIRE(src >= 0 && src < R, "run_forced: source %d out of range", src);
            nr[r] = seqs.append(rows[src], ftok[(size_t)step * R + r]);
        }
        rows.swap(nr);
        sources.assign(fsrc + (size_t)step * R, fsrc + (size_t)(step + 1) * R);
    };
    sources.assign(R, 0);
    apply(0);
    for (int i = 1; i < n_steps; ++i) {
        StepParams* sp = st.sp_h();
        sp->cache_pos = S + i - 1;
        sp->pos_idx = position_quirk ? i : S + i - 1;
        sp->parity = i & 1; sp->first_step = 0; sp->want_no_speech = 0; sp->n_rows = R;
        for (int r = 0; r < R; ++r) {
            WRS_REQUIRE(seqs.back(rows[r]) >= 0 && seqs.back(rows[r]) < m->V, "run_forced: token out of range");
            st.tokens_h()[r] = (int)seqs.back(rows[r]);
            st.src_h()[r] = sources[r];
            st.tsmode_h()[r] = timestamps ? ts_mode_of(seqs, rows[r], S, ids.ts_begin) : 0;
        }
        step_decode(R);
        memcpy(out_lp + (size_t)i * R * k, st.topk_lp(), (size_t)R * k * 4);
        memcpy(out_id + (size_t)i * R * k, st.topk_id(), (size_t)R * k * 4);
        apply(i);
    }
}

// finalize (beam_search.rs:151-206 / greedy.rs:54-58), slice [sample_begin, first EOT) (decode/mod.rs:356-370),
// MaximumLikelihoodRanker::rank (max_likelihood.rs:13-34), avg_logprob (decode/mod.rs:389-397).
void DecodeTask::finalize(int n_audio, wrs_decode_result* out) {
    features_n = (m && pending.empty()) ? n_audio : 0;
    features_epoch = m ? m->enc_ws.epoch : 0;
    const int G = group;
    std::vector<std::vector<FinishedSeq>> cands(n_audio);
    for (int i = 0; i < n_audio; ++i) {
        if (mode == WRS_EXTRACT_BEAM_SEARCH) {
            auto& fin = finished[i];
            if ((int)fin.size() < G) {
                std::vector<int> order(G);
                for (int j = 0; j < G; ++j) order[j] = j;
                std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return sum_logprobs[i * G + a] < sum_logprobs[i * G + b]; });
                for (int t = G - 1; t >= 0; --t) {
                    const int j = order[t];
                    FinishedSeq f;
                    seqs.materialize(rows[i * G + j], f.tokens);
                    f.tokens.push_back(ids.eot);
                    f.score = (double)sum_logprobs[i * G + j];
                    bool found = false;
                    for (auto& e : fin)
                        if (e.tokens == f.tokens) { e.score = f.score; found = true; break; }
                    if (!found) fin.push_back(std::move(f));
                    if ((int)fin.size() >= G) break;
                }
            }
            cands[i] = fin;
        } else {
            for (int j = 0; j < G; ++j) {
                FinishedSeq f;
                seqs.materialize(rows[i * G + j], f.tokens);
                f.tokens.push_back(ids.eot);
                f.score = (double)sum_logprobs[i * G + j];
                cands[i].push_back(std::move(f));
            }
        }
    }
    out->outputs = (wrs_decode_output*)calloc(n_audio, sizeof(wrs_decode_output));
    out->n_outputs = n_audio;
    for (int i = 0; i < n_audio; ++i) {
        int best = 0;
        double best_score = -INFINITY;
        bool have = false;
        std::vector<std::pair<int, int>> spans(cands[i].size());
        for (size_t c = 0; c < cands[i].size(); ++c) {
            const auto& t = cands[i][c].tokens;
            int end = (int)t.size();
            for (int q = 0; q < (int)t.size(); ++q)
                if (t[q] == ids.eot) { end = q; break; }
            const int begin = std::min(sample_begin, end);
            spans[c] = {begin, end};
            const double length = (double)(end - begin);
            const double penalty = has_len_penalty ? std::pow((5.0 + length) / 6.0, len_penalty) : length;
            const double score = cands[i][c].score / penalty;
            if (!have || score > best_score) { best = (int)c; best_score = score; have = true; }  // argmax: first maximum
        }
        const auto& t = cands[i][best].tokens;
        const int begin = spans[best].first, end = spans[best].second;
        wrs_decode_output& o = out->outputs[i];
        o.n_tokens = end - begin;
        o.tokens = (int64_t*)malloc(sizeof(int64_t) * std::max(1, o.n_tokens));
        for (int q = begin; q < end; ++q) o.tokens[q - begin] = t[q];
        o.sum_logprob = cands[i][best].score;
        o.avg_logprob = cands[i][best].score / ((double)o.n_tokens + 1.0);
        o.no_speech_prob = no_speech[i];
        // overlapped mode: a batch encoded after this one already owns the encoder workspace, its features are gone
        o.audio_features_device = (m && pending.empty()) ? m->enc_ws.xa32.as<float>() + (size_t)i * m->T * m->D : nullptr;
        o.text = tokenizer ? dup_cstr(tokenizer->decode(o.tokens, o.n_tokens)) : nullptr;  // decode/mod.rs:404
    }
}

char* dup_cstr(const std::string& s) {
    char* p = (char*)malloc(s.size() + 1);
    memcpy(p, s.data(), s.size());
    p[s.size()] = 0;
    return p;
}

void free_decode_result(wrs_decode_result* r) {
    if (!r || !r->outputs) return;
    for (int i = 0; i < r->n_outputs; ++i) { free(r->outputs[i].tokens); free(r->outputs[i].text); }
    free(r->outputs);
    r->outputs = nullptr;
    r->n_outputs = 0;
}

}  // namespace wrs
