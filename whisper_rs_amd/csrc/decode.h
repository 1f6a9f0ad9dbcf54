// DecodeTask: one <=30-s window batch: encoder -> sampling loop -> rank (reference src/decode/mod.rs).
#pragma once
#include "model.h"
#include "tokenizer.h"

namespace wrs {

struct FinishedSeq {
    std::vector<int64_t> tokens;
    double score;
};

// Token histories of the decoder rows as a prefix tree.  Beam search only ever extends or forks sequences (a new row is
// its parent's sequence plus one token: beam_search.rs:57-62,88-96), so a row is a node id and a step appends R nodes
// instead of copying R whole sequences (1280 rows x up to 227 tokens per step at the benchmark's batch).  The reference
// keys its candidate map by the full token vector (beam_search.rs:60-77); `equal` is that comparison, exact: the 64-bit
// prefix hash only rejects early, a match is confirmed token by token down to the first shared node.
struct SeqStore {
    struct Node { int32_t parent; int32_t len; int64_t token; uint64_t hash; };
    std::vector<Node> nodes;
    void clear() { nodes.clear(); }
    void reserve(size_t n) { nodes.reserve(n); }
    int append(int parent, int64_t token) {
        const uint64_t ph = parent >= 0 ? nodes[parent].hash : 0x9e3779b97f4a7c15ull;
        uint64_t z = ph ^ ((uint64_t)token + 0x9e3779b97f4a7c15ull + (ph << 6) + (ph >> 2));
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        z ^= z >> 31;
        nodes.push_back({(int32_t)parent, parent >= 0 ? nodes[parent].len + 1 : 1, token, z});
        return (int)nodes.size() - 1;
    }
    int chain(const std::vector<int64_t>& tokens) {
        int id = -1;
        for (int64_t t : tokens) id = append(id, t);
        return id;
    }
    int64_t back(int id) const { return nodes[id].token; }
    int len(int id) const { return id >= 0 ? nodes[id].len : 0; }
    int parent(int id) const { return nodes[id].parent; }
    bool equal(int a, int b) const {
        if (a == b) return true;
        if (a < 0 || b < 0 || nodes[a].len != nodes[b].len || nodes[a].hash != nodes[b].hash) return false;
        while (a != b) {
            if (nodes[a].token != nodes[b].token) return false;
            a = nodes[a].parent;
            b = nodes[b].parent;
        }
        return true;
    }
    void materialize(int id, std::vector<int64_t>& out) const {
        out.resize((size_t)len(id));
        for (int q = len(id) - 1; q >= 0; --q, id = nodes[id].parent) out[q] = nodes[id].token;
    }
};

struct DecodeTask {
    Model* m = nullptr;
    wrs_token_ids ids{};
    const Tokenizer* tokenizer = nullptr;  // optional: fills DecodeOutput.text (decode/mod.rs:404)
    // options (reference DecodeOptions, decode/mod.rs:95-106)
    int sample_len = 0;
    int mode = WRS_EXTRACT_BEAM_SEARCH;
    int group = 5;
    float patience = 1.f;
    bool has_len_penalty = false;
    double len_penalty = 0;
    bool timestamps = true, suppress_blank = true;
    int mit = -1;  // max_initial_timestamp_index, -1 = None
    bool position_quirk = true;
    int max_audio = 1;
    std::vector<int> suppress_ids;
    DevBuf suppress_dev;

    // prompt (reference DecodePrompt, decode/mod.rs:46-92)
    std::vector<int64_t> initial_tokens;
    int sample_begin = 1, sot_idx = 0;
    std::vector<int64_t> sot_tail;    // sequence_sot() after the sot token (empty = the reference)
    std::vector<int64_t> audio_lang;  // per-window replacement of sot_tail[0] (language token), or empty
    std::vector<int64_t> last_prompt; // the prompt given to set_prompt (kept so the sot sequence can be rebuilt)
    int64_t initial_token(int audio, int p) const {
        return (!audio_lang.empty() && p == sot_idx + 1 && audio < (int)audio_lang.size()) ? audio_lang[audio] : initial_tokens[p];
    }

    DecState st;
    MelWorkspace mel_ws;
    DevBuf audio_dev, mel_dev;

    // host sequencer state
    SeqStore seqs;                           // token histories incl. prompt
    std::vector<int> rows;                   // [R] node of row r in `seqs`
    std::vector<int> scratch_rows;           // next step's rows (swapped with `rows`)
    void init_rows(int n_audio);             // every beam of audio a starts at the audio's prompt (decode/mod.rs:297-313)
    std::vector<float> scratch_slp;
    std::vector<int> scratch_src;
    std::vector<float> sum_logprobs;         // [R] (f32 like the reference's tensor)
    std::vector<int> sources;                // [R]
    std::vector<std::vector<FinishedSeq>> finished;  // per audio, admission order
    std::vector<double> no_speech;
    int n_steps = 0;
    // wall-clock split of the last run (ms): encoder+cross-KV (incl. wait), prefill, device decode steps, host sequencer
    double t_encoder_ms = 0, t_prefill_ms = 0, t_steps_ms = 0, t_host_ms = 0;

    // trace (test aid)
    std::vector<int32_t> tr_tokens, tr_src;
    std::vector<float> tr_slp;
    int tr_rows = 0;
    // test aids (include/wrs_debug.h): per-step device candidates of the last run, and a forced prefix of beam decisions
    // (re-synchronisation with the oracle after a near-tie: steps < forced_n take the given tokens / sources / scores)
    bool keep_candidates = false;
    std::vector<float> tr_cand_lp;
    std::vector<int32_t> tr_cand_id;
    int forced_n = 0;
    const int32_t* forced_tok = nullptr;
    const int32_t* forced_src = nullptr;
    const float* forced_slp = nullptr;
    void record_candidates(int R, bool step0);
    void apply_forced(int step, const std::vector<int>& old_rows);

    // graph replay of the decode step
    bool use_graph = true;
    hipGraphExec_t graph_exec = nullptr;
    hipGraph_t graph = nullptr;
    int ctx_for(int positions) const;  // cache positions to allocate for `positions` tokens per row
    void ensure_ctx(int positions);
    int graph_rows = -1, graph_sel = -1;  // the captured step reads the cross K/V buffer `graph_sel`

    // overlapped mode: mel + encoder + cross-K/V projection of up to two window batches queued on a side stream
    hipStream_t side = nullptr;
    hipEvent_t side_done[2] = {nullptr, nullptr};
    struct Pending { int buf, n_audio; };
    std::vector<Pending> pending;  // FIFO, oldest first
    bool side_idle();              // every queued encode has finished on the device

    ~DecodeTask();
    void configure(Model* model, const wrs_token_ids& ids, const wrs_decode_options& o);
    void set_prompt(const int64_t* p, int n);
    FilterCfg filter_cfg() const;

    // window batch already loaded in m->enc_ws.mel_t
    void run_loaded(int n_audio, wrs_decode_result* out);
    // encoder output already in m->enc_ws.xa16/xa32
    void run_encoded(int n_audio, wrs_decode_result* out);
    // cross K/V of the batch already in st's buffer `st.cross_sel` (queued work waited for by the caller)
    void run_decode(int n_audio, wrs_decode_result* out);
    // overlapped mode (beyond the reference, whose loop is sequential): queue mel/encoder/cross-K/V of a batch of raw
    // audio windows on the side stream and return; run_pending decodes the oldest queued batch on the main stream
    void encode_async(const float* audio, int n_audio, int mem_kind, bool is_mel = false);  // is_mel: [n_audio, 80, 3000] log-mel windows instead of raw audio
    void run_pending(wrs_decode_result* out);
    // test hook: follow externally given beam decisions (tokens/sources per step) and return the device's raw
    // per-step top-k candidates; the window batch must already be in m->enc_ws.mel_t
    // language identification on the encoder output in m->enc_ws.xa16 (OpenAI detect_language)
    void detect_language(int n_audio, int first_id, int n_lang, int32_t* out_ids, float* out_probs);
    void set_languages(const int32_t* lang, int n);
    void run_forced(int n_audio, int n_steps, const int32_t* ftok, const int32_t* fsrc, float* out_lp, int32_t* out_id,
                    double* out_no_speech, bool have_features = false);

    void prefill(int n_audio, double* no_speech_out);  // first forward pass: prompt tokens, step-0 candidates, no-speech
    void upload_step(int rows_n);
    void step_decode(int rows_n);  // decoder_step + filter_topk for rows_n = n_audio*group rows (graph-replayed)
    bool update_beam(int n_audio, bool step0, const float* lp, const int* id);
    bool update_greedy(int n_audio, bool step0, const float* lp, const int* id);
    void simulate(const wrs_token_ids& tid, const wrs_decode_options& o, const int64_t* initial, int n_initial, int n_audio, int n_steps,
                  const float* lp, const int32_t* id, wrs_decode_result* out);
    void finalize(int n_audio, wrs_decode_result* out);
    uint64_t features_epoch = 0;
    int features_n = 0;  // windows of the last decoded batch whose encoder output still sits in m->enc_ws.xa32 (0: overwritten)
    void record_trace(int R);
};

void free_decode_result(wrs_decode_result* r);
char* dup_cstr(const std::string& s);  // malloc-ed copy (released by the *_free entry points)

}  // namespace wrs
