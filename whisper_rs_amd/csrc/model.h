// Host-side model, encoder workspace and decoder state (device memory owned here).
#pragma once
#include <map>
#include <memory>
#include "kernels.h"

namespace wrs {

struct BlockW {
    float *attn_ln_g = nullptr, *attn_ln_b = nullptr;
    half_t* qkv_w = nullptr;  // packed [3D, D]  (query | key | value)
    float* qkv_b = nullptr;   // [3D], key part zero (reference model.rs:138-141: key has no bias)
    half_t* out_w = nullptr;
    float* out_b = nullptr;
    // decoder only
    float *cross_ln_g = nullptr, *cross_ln_b = nullptr;
    half_t* cq_w = nullptr;
    float* cq_b = nullptr;
    half_t* ckv_w = nullptr;  // packed [2D, D] (key | value)
    float* ckv_b = nullptr;
    half_t* cout_w = nullptr;
    float* cout_b = nullptr;
    float *mlp_ln_g = nullptr, *mlp_ln_b = nullptr;
    half_t* mlp0_w = nullptr;
    float* mlp0_b = nullptr;
    half_t* mlp2_w = nullptr;
    float* mlp2_b = nullptr;
    // f32 parity mode only (parity_f32.hip): row-major fp32 copies of the matrices
    float *qkv_w32 = nullptr, *out_w32 = nullptr, *cq_w32 = nullptr, *ckv_w32 = nullptr, *cout_w32 = nullptr, *mlp0_w32 = nullptr,
          *mlp2_w32 = nullptr;
};

// f32 parity mode workspaces
struct EncoderWorkspaceF32 {
    int cap = 0;
    DevBuf mel_t, c1, xn, qkv, att, h;
};
struct DecStateF32 {
    DevBuf cross;    // [L][a_cap][T][2D]: K (scaled) | V
    DevBuf kc, vc;   // [L][r_cap][n_ctx][D]
    DevBuf xn, qkv, att, h;
};

struct EncoderWorkspace {
    int cap = 0;  // windows
    int n_encoded = 0;  // windows whose encoder output currently sits in xa16/xa32
    uint64_t epoch = 0;  // bumped by every write of xa16/xa32 (a DecodeTask remembers the epoch its outputs' features belong to)
    DevBuf mel_t, c1, x, xn, qkv, kp, vp, att, h, xa16, xa32;
    // host spectrograms (Model::load_mel): two persistent device staging buffers used in turn, the event after the kernel that
    // last read each, and the event of the last upload
    DevBuf mel_stage[2];
    hipEvent_t stage_read[2] = {nullptr, nullptr};
    hipEvent_t stage_copied = nullptr;
    int stage_next = 0;
};

// workspace of the batched prompt prefill (Model::decoder_prefill): rows are (audio, prompt position)
struct PrefillWorkspace {
    int64_t cap_rows = 0;
    int cap_audio = 0, cap_s = 0;
    DevBuf tokens, x, xn, qkv, kp, vp, att, h, logits_all;
};

// decoder K/V caches + per-step scratch for up to `a_cap` audios x `group` rows
struct DecState {
    int a_cap = 0, group = 1, r_cap = 0, m_pad = 0, act_rows = 0, logits_ld = 0;
    int ctx = 0;      // positions per row of kc / vc / anc / self_keys (<= the model's n_text_ctxs; Model::resize_ctx)
    int n_audio = 0;  // audios whose cross K/V are loaded
    DevBuf cross_kp, cross_vp;  // [L][a_cap][H][Tp*64]
    // second cross-attention K/V buffer (allocated on first use by the overlapped mode: the next window batch's encoder
    // and K/V projection run on a side stream while this batch's decode steps read the other buffer)
    DevBuf cross_kp2, cross_vp2;
    int cross_sel = 0;   // buffer the decode steps read
    int cross_wsel = 0;  // buffer load_cross_kv writes
    half_t* ckp(int sel) const { return (sel ? cross_kp2 : cross_kp).as<half_t>(); }
    half_t* cvp(int sel) const { return (sel ? cross_vp2 : cross_vp).as<half_t>(); }
    DevBuf kc, vc;              // [L][r_cap][ctx][D]
    DevBuf anc;                 // [2][r_cap][ctx]
    DevBuf filter_ws;           // partial results of the split filter / top-k (kernels.h filter_ws_floats), few-row steps
    DevBuf self_keys;           // [a_cap][G * n_ctx] distinct (position, row) pairs of every audio's beams + [a_cap] counts (attn_self_mfma.hip)
    int self_keys_stride = 0;
    DevBuf x, xn, att, h, slab, logits, ckv_tmp;
    DevBuf slab2, x2;           // <= 16-row path with fused producers (gemm_dec_fused): second slab and residual-stream buffers (ping-pong)
    DevBuf q16;                 // [act_rows, D] fp16 row-major: the step's query projection (self- and cross-attention), >= 480-row path
    std::unique_ptr<DecStateF32> f32;  // f32 parity mode only
    PrefillWorkspace prefill;   // allocated on first use
    DevBuf sp_dev;              // StepParams + tokens[r_cap] + src[r_cap] + ts_mode[r_cap]
    PinnedBuf sp_host;
    PinnedBuf topk;             // lp [r_cap*8] f32 | id [r_cap*8] i32 | no_speech [r_cap] f32
    int slab_ksplit_cap = 16;
    int len = 0;                // cached self-attention positions (stage API)
    int parity = 0;
    std::vector<int> pending_src;  // stage API: update_cache() to apply at the next forward
    int m_pad_for(int rows) const { return skinny_m_pad(rows); }  // slab row stride used by the skinny GEMM for `rows`
    StepParams* sp_h() { return sp_host.as<StepParams>(); }
    int* tokens_h() { return reinterpret_cast<int*>(sp_host.as<char>() + sizeof(StepParams)); }
    int* src_h() { return tokens_h() + r_cap; }
    int* tsmode_h() { return src_h() + r_cap; }
    const StepParams* sp_d() const { return sp_dev.as<StepParams>(); }
    const int* tokens_d() const { return reinterpret_cast<const int*>(sp_dev.as<char>() + sizeof(StepParams)); }
    const int* src_d() const { return tokens_d() + r_cap; }
    const int* tsmode_d() const { return src_d() + r_cap; }
    size_t sp_bytes() const { return sizeof(StepParams) + (size_t)3 * r_cap * sizeof(int); }
    float* topk_lp() { return topk.as<float>(); }
    int* topk_id() { return reinterpret_cast<int*>(topk.as<float>() + (size_t)r_cap * 8); }
    float* no_speech_h() { return topk.as<float>() + (size_t)r_cap * 16; }
    float* topk_lp_d() { return topk.dev_as<float>(); }
    int* topk_id_d() { return reinterpret_cast<int*>(topk.dev_as<float>() + (size_t)r_cap * 8); }
    float* no_speech_d() { return topk.dev_as<float>() + (size_t)r_cap * 16; }
};

struct StageTimer;

struct Model {
    wrs_model_dims dims{};
    int device = 0;
    int D = 0, H = 0, L_enc = 0, L_dec = 0, V = 0, T = 0, n_ctx = 0;
    int64_t v_pad = 0;
    bool finalized = false;
    hipStream_t stream = nullptr;

    std::map<std::string, DevBuf> staging;  // fp32 tensors as loaded
    std::vector<DevBuf> owned;              // packed / converted tensors

    // encoder
    half_t* conv1_w = nullptr; float* conv1_b = nullptr;
    half_t* conv2_w = nullptr; float* conv2_b = nullptr;
    float* enc_pos = nullptr;  // sinusoids [T, D] fp32 (reference model.rs:102-112, computed not loaded)
    std::vector<BlockW> enc;
    float *ln_post_g = nullptr, *ln_post_b = nullptr;
    // decoder
    half_t* emb16 = nullptr;   // [V, D] row-major (embedding lookup)
    half_t* emb_p = nullptr;   // packed [v_pad, D] (tied output projection, model.rs:521)
    float* dec_pos = nullptr;  // [n_ctx, D]
    std::vector<BlockW> dec;
    float *ln_g = nullptr, *ln_b = nullptr;

    // f32 parity mode (parity_f32.hip; wrs_debug_model_set_parity_f32 before finalize): fp32 weights,
    // activations, caches and products, to demonstrate bit-exact token ids against the fp32 oracle.  Test aid, slow.
    bool parity_f32 = false;
    float *conv1_w32 = nullptr, *conv2_w32 = nullptr, *emb32 = nullptr;
    EncoderWorkspaceF32 enc_ws32;
    void finalize_f32();
    void ensure_encoder_ws_f32(int n);
    void load_mel_f32(const float* mel_dev, int n, int64_t n_frames, int64_t mel_bs, int64_t seek, int slot = 0);
    void encoder_f32(int n);
    void init_state_f32(DecState& st);
    void load_cross_kv_f32(DecState& st, int n_audio);
    void decoder_step_f32(DecState& st, int rows, int phys_mul, int q_per_audio, bool update_anc, bool want_logits);

    EncoderWorkspace enc_ws;
    std::unique_ptr<DecState> stage_state;  // used by the stage API (wrs_decoder_forward)
    StageTimer* timer = nullptr;            // optional per-kernel timing sink
    int debug_enc_stop = 0;                 // debug: stop the encoder after N launches
    int dec_big_gemm_rows = 480;            // decode steps with at least this many rows use the tiled GEMM (env WRS_DEC_BIG_GEMM_ROWS)
    bool dec_fused_tiny = false;            // <= 16 rows, WRS_DEC_FUSED=1: finish kernels fused into the next GEMM's prologue (8 launches per layer instead of 12, but slower: DESIGN.md section 4)
    int self_mfma_min_audio = 64;           // beam-de-duplicating MFMA self-attention from this many audios per step (WRS_SELF_MFMA_MIN_AUDIO; 0 = never)
    bool fused_kv_pack = true;              // K / V projections written straight into the packed attention formats by the GEMM epilogue (test build: WRS_FUSED_PACK=0 restores launch_pack_kv)
    bool dec_slab_path = true;              // split-K fp32 slabs + finish kernels; WRS_DEC_PRING=1 selects the persistent-ring form (slower, see model.cpp)
    int step_paths_seen = 0;                // diagnosis: OR of the forms decoder_step has taken on this model (1 tiled GEMMs, 2 persistent ring, 4 fused prologue, 8 MFMA self-attention, 16 QKV hand-over)
    void read_switches();                   // experiment switches from the environment, ONCE per model (wrs_model_create)
    // live timing of the dominant decode kernel (cross-attention) with HIP events on this stream: every
    // `cross_ev_every`-th decode step is launched eagerly (not from the graph) with event pairs around the
    // kernel in every `cross_ev_stride`-th layer; all other steps replay the captured graph untouched
    bool cross_ev_enabled = false;
    bool cross_ev_now = false;
    int cross_ev_stride = 8;
    int cross_ev_every = 16;
    int64_t cross_ev_step = 0;
    std::vector<hipEvent_t> cross_ev;       // 2 per sampled layer
    double cross_ev_ms = 0;
    int64_t cross_ev_launches = 0;
    double cross_ev_bytes = 0;
    void collect_cross_events(int n_audio);

    ~Model();
    void set_tensor(const std::string& name, int dtype, const int64_t* shape, int ndim, const void* data, int mem_kind);
    void finalize();
    void use() const;  // hipSetDevice
    void check_device(const char* what) const;  // throws unless the calling thread's current device is this model's

    // encoder over `n` windows; input already in enc_ws.mel_t ([n,3002,80] fp16). Output enc_ws.xa16 / xa32.
    void ensure_encoder_ws(int n);
    void encoder_from_mel_t(int n);
    // upload mel [n,80,3000] f32 (host or device) into enc_ws.mel_t
    void load_mel(const float* mel, int n, int mem_kind, int64_t n_frames, int64_t seek);
    // window `slot` of the encoder input = frames seek..seek+3000 of a device-resident spectrogram [80, n_frames]
    void load_mel_window(const float* mel_dev, int64_t n_frames, int64_t seek, int slot);

    // decoder
    void init_state(DecState& st, int a_cap, int group, int ctx = 0);  // ctx: cache positions per row (0 = n_text_ctxs)
    void resize_ctx(DecState& st, int ctx);
    // cross-attention K/V for `n_audio` audios from xa16 [n_audio*T, D] fp16 (device)
    void load_cross_kv(DecState& st, const half_t* xa16, int n_audio);
    // one token per row. rows = logical rows this step; phys_mul = physical cache-row stride per logical row;
    // q_per_audio = rows sharing one audio's cross K/V.  Step scalars / tokens / src must already be in st.sp_dev.
    void decoder_step(DecState& st, int rows, int phys_mul, int q_per_audio, bool update_anc, bool want_logits);
    void decoder_step_big(DecState& st, int rows, int phys_mul, int q_per_audio, bool update_anc, bool want_logits);
    void decoder_step_tiny(DecState& st, int rows, int phys_mul, int q_per_audio, bool update_anc, bool want_logits);  // <= 16 rows
    // tied output projection of st.xn (packed activations, `rows` rows) into st.logits (model.rs:521)
    void decoder_logits(DecState& st, int rows);
    // The decoder's first forward pass: all S prompt tokens of every audio at once (reference decode/mod.rs:236-245,
    // TextDecoder::forward_ext with the causal mask, model.rs:191-201,490-522).  tokens_host [n_audio][S]; positions
    // pos0 .. pos0+S-1 of physical row a * phys_mul of the self-attention K/V cache are written.  Afterwards
    // prefill_logits(st, n_audio, S, p) puts the logits of prompt position p of every audio into st.logits rows 0..n_audio-1;
    // logits_all (optional, device [n_audio*S, logits_ld] fp32) receives every position's logits (stage API).
    void decoder_prefill(DecState& st, int n_audio, int S, int pos0, int phys_mul, const int* tokens_host, bool all_logits);
    void prefill_logits(DecState& st, int n_audio, int S, int p);
};

// log-mel on device (reference audio.rs:230-259).  audio_d [n_batch, n] f32 device -> out_d [n_batch, 80, n/160] f32 device
struct MelWorkspace {
    DevBuf maxbuf, twiddle, filter, krange;
    bool tables_ready = false;
    std::vector<float> filter_host;
};
void mel_filter_host(float* out /*80*201*/);
void log_mel_device(MelWorkspace& ws, const float* audio_d, int64_t n_batch, int64_t n, const float* filter_host_or_null,
                    float* out_d, hipStream_t s);

// optional timing of named stages with HIP events on the model stream
struct StageTimer {
    struct Entry { const char* name; double ms = 0; int64_t launches = 0; double bytes = 0; double flops = 0; };
    std::vector<Entry> entries;
    struct Pending { int idx; hipEvent_t a, b; };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> pool;
    bool enabled = false;
    ~StageTimer();
    int slot(const char* name);
    hipEvent_t get_event();
    void begin(int idx, hipStream_t s, hipEvent_t& a);
    void end(int idx, hipStream_t s, hipEvent_t a, double bytes, double flops);
    void collect();  // after a stream sync
    void reset();
};

struct Timed {
    StageTimer* t = nullptr;
    int idx = 0;
    hipStream_t s;
    hipEvent_t a;
    double bytes, flops;
    Timed(StageTimer* tm, const char* name, hipStream_t st, double by = 0, double fl = 0) : s(st), bytes(by), flops(fl) {
        if (tm && tm->enabled) {
            t = tm;
            idx = t->slot(name);
            t->begin(idx, s, a);
        }
    }
    ~Timed() {
        if (t) t->end(idx, s, a, bytes, flops);
    }
};

// shape the architecture expects for a VarStore path; false for names the model does not have
void build_block_f32(Model& m, BlockW& bw, const std::string& pre, int64_t d, bool cross, const std::map<std::string, DevBuf>& staging);
bool expected_shape(const Model& m, const std::string& name, std::vector<int64_t>& shape);
// weights_io.cpp: reads every tensor of the architecture found in a safetensors file; returns how many were set
int load_safetensors(Model& m, const char* path);

}  // namespace wrs
