// Model: weight intake/packing, AudioEncoder forward, TextDecoder single-token step.
// Restates the *structure* of reference src/model.rs on top of the fused gfx950 kernels:
//   AudioEncoder::forward        model.rs:417-441  -> Model::encoder_from_mel_t
//   ResidualAttentionBlock       model.rs:301-319  -> per-layer kernel sequences below
//   TextDecoder::forward_ext     model.rs:490-522  -> Model::decoder_step (one token per row)
//   Cached<Linear> K/V caches    model.rs:37-98    -> DecState (pre-allocated ring + ancestry table)
#include "model.h"
#include <cmath>

namespace wrs {

// ------------------------------------------------------------------------------------------ timer
StageTimer::~StageTimer() {
    for (auto& p : pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto e : pool) (void)hipEventDestroy(e);
}
int StageTimer::slot(const char* name) {
    for (size_t i = 0; i < entries.size(); ++i)
        if (entries[i].name == name || strcmp(entries[i].name, name) == 0) return (int)i;
    Entry e; e.name = name;
    entries.push_back(e);
    return (int)entries.size() - 1;
}
hipEvent_t StageTimer::get_event() {
    if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e;
    HIP_CHECK(hipEventCreate(&e));
    return e;
}
void StageTimer::begin(int, hipStream_t s, hipEvent_t& a) {
    a = get_event();
    HIP_CHECK(hipEventRecord(a, s));
}
void StageTimer::end(int idx, hipStream_t s, hipEvent_t a, double bytes, double flops) {
    hipEvent_t b = get_event();
    HIP_CHECK(hipEventRecord(b, s));
    pending.push_back({idx, a, b});
    entries[idx].launches += 1;
    entries[idx].bytes += bytes;
    entries[idx].flops += flops;
}
void StageTimer::collect() {
    for (auto& p : pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) entries[p.idx].ms += ms;
        pool.push_back(p.a);
        pool.push_back(p.b);
    }
    pending.clear();
}
void StageTimer::reset() {
    collect();
    entries.clear();
}

// ------------------------------------------------------------------------------------------ model
Model::~Model() {
    for (auto e : cross_ev) (void)hipEventDestroy(e);
    for (auto e : enc_ws.stage_read) if (e) (void)hipEventDestroy(e);
    if (enc_ws.stage_copied) (void)hipEventDestroy(enc_ws.stage_copied);
    if (stream) (void)hipStreamDestroy(stream);
    delete timer;
}

void Model::collect_cross_events(int n_audio) {
    if (!cross_ev_now) return;
    for (size_t i = 0; i + 1 < cross_ev.size(); i += 2) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, cross_ev[i], cross_ev[i + 1]) == hipSuccess) {
            cross_ev_ms += ms;
            cross_ev_launches += 1;
            cross_ev_bytes += (double)n_audio * 2.0 * T * D * 2.0;  // K and V of every audio, fp16, once per launch
        }
    }
}

void Model::use() const { HIP_CHECK(hipSetDevice(device)); }

// every launch of a model's work must come from a thread whose current HIP device is the model's (a window pool drives
// several GPUs from several threads; a kernel launched while another device is current would run on the wrong GPU, on
// pointers that do not exist there)
void Model::check_device(const char* what) const {
    int cur = -1;
    HIP_CHECK(hipGetDevice(&cur));
    if (cur != device) WRS_THROW(WRS_ESTATE, "%s: the calling thread's HIP device is %d but the model lives on device %d", what, cur, device);
}

bool expected_shape(const Model& m, const std::string& name, std::vector<int64_t>& shape) {
    const int64_t D = m.dims.n_audio_states, Dt = m.dims.n_text_states;
    auto ends = [&](const char* suf) {
        const size_t n = strlen(suf);
        return name.size() >= n && name.compare(name.size() - n, n, suf) == 0;
    };
    const bool is_enc = name.rfind("encoder.", 0) == 0;
    const int64_t d = is_enc ? D : Dt;
    if (name == "encoder.conv1.weight") { shape = {D, WRS_N_MELS, 3}; return true; }
    if (name == "encoder.conv2.weight") { shape = {D, D, 3}; return true; }
    if (name == "encoder.conv1.bias" || name == "encoder.conv2.bias") { shape = {D}; return true; }
    if (name == "decoder.token_embedding.weight") { shape = {m.dims.n_vocab, Dt}; return true; }
    if (name == "decoder.positional_embedding") { shape = {m.dims.n_text_ctxs, Dt}; return true; }
    if (name.find(".blocks.") == std::string::npos) {
        if (name == "encoder.ln_post.weight" || name == "encoder.ln_post.bias" || name == "decoder.ln.weight" ||
            name == "decoder.ln.bias") { shape = {d}; return true; }
        return false;
    }
    if (ends(".mlp.0.weight")) { shape = {4 * d, d}; return true; }
    if (ends(".mlp.0.bias")) { shape = {4 * d}; return true; }
    if (ends(".mlp.2.weight")) { shape = {d, 4 * d}; return true; }
    if (ends(".query.weight") || ends(".key.weight") || ends(".value.weight") || ends(".out.weight")) { shape = {d, d}; return true; }
    if (ends(".bias") || ends("_ln.weight")) { shape = {d}; return true; }
    return false;
}

void Model::set_tensor(const std::string& name, int dtype, const int64_t* shape, int ndim, const void* data, int mem_kind) {
    WRS_REQUIRE(!finalized, "set_tensor(%s): model already finalized", name.c_str());
    WRS_REQUIRE(data && shape && ndim >= 1 && ndim <= 3, "set_tensor(%s): bad arguments", name.c_str());
    WRS_REQUIRE(dtype == WRS_F32 || dtype == WRS_F16, "set_tensor(%s): dtype must be WRS_F32 or WRS_F16", name.c_str());
    std::vector<int64_t> want;
    WRS_REQUIRE(expected_shape(*this, name, want), "set_tensor: unknown tensor name '%s'", name.c_str());
    WRS_REQUIRE((int)want.size() == ndim, "set_tensor(%s): rank %d, expected %d", name.c_str(), ndim, (int)want.size());
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) {
        WRS_REQUIRE(shape[i] == want[i], "set_tensor(%s): dim %d is %ld, expected %ld", name.c_str(), i, (long)shape[i], (long)want[i]);
        n *= shape[i];
    }
    use();
    DevBuf buf;
    buf.alloc((size_t)n * 4);
    const hipMemcpyKind kind = mem_kind == WRS_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    if (dtype == WRS_F32) {
        HIP_CHECK(hipMemcpyAsync(buf.p, data, (size_t)n * 4, kind, stream));
    } else {
        DevBuf tmp;
        tmp.alloc((size_t)n * 2);
        HIP_CHECK(hipMemcpyAsync(tmp.p, data, (size_t)n * 2, kind, stream));
        launch_f16_to_f32(tmp.as<half_t>(), buf.as<float>(), n, stream);
        HIP_CHECK(hipStreamSynchronize(stream));
    }
    HIP_CHECK(hipStreamSynchronize(stream));
    staging[name] = std::move(buf);
}

static const float* need(Model& m, const std::string& name) {
    auto it = m.staging.find(name);
    if (it == m.staging.end()) WRS_THROW(WRS_ESTATE, "finalize: tensor '%s' was never set", name.c_str());
    return it->second.as<float>();
}

static half_t* pack_linear(Model& m, const float* w, int64_t n, int64_t k) {
    const int64_t n_pad = round_up(n, 128), k_pad = round_up(k, 64);
    DevBuf b;
    b.alloc(packed_weight_halfs(n_pad, k_pad) * 2);
    launch_pack_weight_f32(w, n, k, k, b.as<half_t>(), n_pad, k_pad, m.stream);
    half_t* p = b.as<half_t>();
    m.owned.push_back(std::move(b));
    return p;
}

static float* keep_f32(Model& m, const float* src, int64_t n) {
    DevBuf b;
    b.alloc((size_t)n * 4);
    HIP_CHECK(hipMemcpyAsync(b.p, src, (size_t)n * 4, hipMemcpyDeviceToDevice, m.stream));
    float* p = b.as<float>();
    m.owned.push_back(std::move(b));
    return p;
}

// concatenate row-blocks of [d, k] matrices into one staging buffer, then pack
static half_t* pack_concat(Model& m, std::vector<const float*> parts, int64_t d, int64_t k) {
    DevBuf cat;
    cat.alloc((size_t)parts.size() * d * k * 4);
    for (size_t i = 0; i < parts.size(); ++i)
        HIP_CHECK(hipMemcpyAsync(cat.as<float>() + i * d * k, parts[i], (size_t)d * k * 4, hipMemcpyDeviceToDevice, m.stream));
    half_t* p = pack_linear(m, cat.as<float>(), (int64_t)parts.size() * d, k);
    HIP_CHECK(hipStreamSynchronize(m.stream));
    return p;
}

static float* concat_bias(Model& m, std::vector<const float*> parts, int64_t d) {
    DevBuf b;
    b.alloc(parts.size() * d * 4);  // zero-initialised
    for (size_t i = 0; i < parts.size(); ++i)
        if (parts[i]) HIP_CHECK(hipMemcpyAsync(b.as<float>() + i * d, parts[i], (size_t)d * 4, hipMemcpyDeviceToDevice, m.stream));
    float* p = b.as<float>();
    m.owned.push_back(std::move(b));
    return p;
}

static void build_block(Model& m, BlockW& bw, const std::string& pre, int64_t d, bool cross) {
    auto t = [&](const char* s) { return need(m, pre + s); };
    bw.attn_ln_g = keep_f32(m, t(".attn_ln.weight"), d);
    bw.attn_ln_b = keep_f32(m, t(".attn_ln.bias"), d);
    bw.qkv_w = pack_concat(m, {t(".attn.query.weight"), t(".attn.key.weight"), t(".attn.value.weight")}, d, d);
    bw.qkv_b = concat_bias(m, {t(".attn.query.bias"), nullptr, t(".attn.value.bias")}, d);
    bw.out_w = pack_linear(m, t(".attn.out.weight"), d, d);
    bw.out_b = keep_f32(m, t(".attn.out.bias"), d);
    if (cross) {
        bw.cross_ln_g = keep_f32(m, t(".cross_attn_ln.weight"), d);
        bw.cross_ln_b = keep_f32(m, t(".cross_attn_ln.bias"), d);
        bw.cq_w = pack_linear(m, t(".cross_attn.query.weight"), d, d);
        bw.cq_b = keep_f32(m, t(".cross_attn.query.bias"), d);
        bw.ckv_w = pack_concat(m, {t(".cross_attn.key.weight"), t(".cross_attn.value.weight")}, d, d);
        bw.ckv_b = concat_bias(m, {nullptr, t(".cross_attn.value.bias")}, d);
        bw.cout_w = pack_linear(m, t(".cross_attn.out.weight"), d, d);
        bw.cout_b = keep_f32(m, t(".cross_attn.out.bias"), d);
    }
    bw.mlp_ln_g = keep_f32(m, t(".mlp_ln.weight"), d);
    bw.mlp_ln_b = keep_f32(m, t(".mlp_ln.bias"), d);
    bw.mlp0_w = pack_linear(m, t(".mlp.0.weight"), 4 * d, d);
    bw.mlp0_b = keep_f32(m, t(".mlp.0.bias"), 4 * d);
    bw.mlp2_w = pack_linear(m, t(".mlp.2.weight"), d, 4 * d);
    bw.mlp2_b = keep_f32(m, t(".mlp.2.bias"), d);
    HIP_CHECK(hipStreamSynchronize(m.stream));
    // release this block's staging memory early (large-v2 staging is 6 GB fp32)
    for (auto it = m.staging.begin(); it != m.staging.end();)
        if (it->first.rfind(pre + ".", 0) == 0) it = m.staging.erase(it);
        else ++it;
}

void Model::finalize() {
    WRS_REQUIRE(!finalized, "finalize: already finalized");
    use();
    WRS_REQUIRE(dims.n_audio_states == dims.n_text_states, "n_audio_states != n_text_states is not supported (cross-attention width)");
    D = (int)dims.n_text_states;
    H = (int)dims.n_text_heads;
    WRS_REQUIRE(dims.n_audio_heads == dims.n_text_heads, "n_audio_heads != n_text_heads is not supported");
    WRS_REQUIRE(D == H * 64, "head_dim must be 64 (n_state %d, heads %d)", D, H);
    WRS_REQUIRE(D % 64 == 0 && D <= 2048, "n_state %d unsupported (multiple of 64, <= 2048)", D);
    L_enc = (int)dims.n_audio_layers;
    L_dec = (int)dims.n_text_layers;
    V = (int)dims.n_vocab;
    T = (int)dims.n_audio_ctxs;
    n_ctx = (int)dims.n_text_ctxs;
    WRS_REQUIRE(T * 2 == WRS_N_FRAMES, "n_audio_ctxs must be %d", WRS_N_FRAMES / 2);
    WRS_REQUIRE(n_ctx >= 2 && n_ctx <= 512, "n_text_ctxs %d unsupported (<= 512)", n_ctx);
    v_pad = round_up(V, 128);

    // conv weights in implicit-GEMM order
    {
        const int64_t k1 = round_up(3 * WRS_N_MELS, 64), k2 = round_up(3 * (int64_t)D, 64), n_pad = round_up(D, 128);
        DevBuf b1, b2;
        b1.alloc(packed_weight_halfs(n_pad, k1) * 2);
        b2.alloc(packed_weight_halfs(n_pad, k2) * 2);
        launch_pack_conv_weight_f32(need(*this, "encoder.conv1.weight"), D, WRS_N_MELS, b1.as<half_t>(), n_pad, k1, stream);
        launch_pack_conv_weight_f32(need(*this, "encoder.conv2.weight"), D, D, b2.as<half_t>(), n_pad, k2, stream);
        conv1_w = b1.as<half_t>();
        conv2_w = b2.as<half_t>();
        owned.push_back(std::move(b1));
        owned.push_back(std::move(b2));
        conv1_b = keep_f32(*this, need(*this, "encoder.conv1.bias"), D);
        conv2_b = keep_f32(*this, need(*this, "encoder.conv2.bias"), D);
    }
    // sinusoidal positions, exactly the f32 op sequence of reference model.rs:102-112
    {
        std::vector<float> tab((size_t)T * D);
        const int half = D / 2;
        const float inc = std::log(10000.0f) / (float)(half - 1);
        for (int t = 0; t < T; ++t)
            for (int c = 0; c < half; ++c) {
                const float inv = std::exp(-inc * (float)c);
                const float st = (float)t * inv;
                tab[(size_t)t * D + c] = std::sin(st);
                tab[(size_t)t * D + half + c] = std::cos(st);
            }
        DevBuf b;
        b.alloc(tab.size() * 4);
        HIP_CHECK(hipMemcpyAsync(b.p, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, utility_stream()));
        HIP_CHECK(hipStreamSynchronize(utility_stream()));
        enc_pos = b.as<float>();
        owned.push_back(std::move(b));
    }
    enc.resize(L_enc);
    for (int i = 0; i < L_enc; ++i) {
        if (parity_f32) build_block_f32(*this, enc[i], "encoder.blocks." + std::to_string(i), D, false, staging);
        build_block(*this, enc[i], "encoder.blocks." + std::to_string(i), D, false);
    }
    ln_post_g = keep_f32(*this, need(*this, "encoder.ln_post.weight"), D);
    ln_post_b = keep_f32(*this, need(*this, "encoder.ln_post.bias"), D);

    {
        const float* e = need(*this, "decoder.token_embedding.weight");
        DevBuf b;
        b.alloc((size_t)V * D * 2);
        launch_f32_to_f16(e, b.as<half_t>(), (int64_t)V * D, stream);
        emb16 = b.as<half_t>();
        owned.push_back(std::move(b));
        emb_p = pack_linear(*this, e, V, D);
        dec_pos = keep_f32(*this, need(*this, "decoder.positional_embedding"), (int64_t)n_ctx * D);
    }
    dec.resize(L_dec);
    for (int i = 0; i < L_dec; ++i) {
        if (parity_f32) build_block_f32(*this, dec[i], "decoder.blocks." + std::to_string(i), D, true, staging);
        build_block(*this, dec[i], "decoder.blocks." + std::to_string(i), D, true);
    }
    ln_g = keep_f32(*this, need(*this, "decoder.ln.weight"), D);
    ln_b = keep_f32(*this, need(*this, "decoder.ln.bias"), D);
    if (parity_f32) finalize_f32();
    HIP_CHECK(hipStreamSynchronize(stream));
    staging.clear();
    finalized = true;
}

// ------------------------------------------------------------------------------------------ encoder
void Model::ensure_encoder_ws(int n) {
    EncoderWorkspace& w = enc_ws;
    if (n <= w.cap) return;
    const int64_t M = (int64_t)n * T;
    w.mel_t.alloc(((size_t)n * 3002 * 80 + 256) * 2);
    w.c1.alloc(((size_t)n * 3001 * D + 256) * 2);
    w.x.alloc((size_t)M * D * 4);
    w.xn.alloc((size_t)M * D * 2);
    w.qkv.alloc((size_t)M * 3 * D * 2);
    w.kp.alloc((size_t)n * H * packed_kv_halfs_per_head(T) * 2);
    w.vp.alloc((size_t)n * H * packed_kv_halfs_per_head(T) * 2);
    w.att.alloc((size_t)M * D * 2);
    w.h.alloc((size_t)M * 4 * D * 2);
    w.xa16.alloc((size_t)M * D * 2);
    w.xa32.alloc((size_t)M * D * 4);
    w.cap = n;
}

void Model::load_mel(const float* mel, int n, int mem_kind, int64_t n_frames, int64_t seek) {
    ensure_encoder_ws(n);
    const size_t bytes = (size_t)n * 80 * n_frames * 4;
    const float* src = mel;
    int sb = -1;
    if (mem_kind == WRS_MEM_HOST) {
        // Host spectrogram: through one of two persistent staging buffers (nothing is allocated or freed per call: hipFree is
        // a device-wide synchronisation and this runs once per window batch of a pool worker).  The copy runs on the utility
        // stream and only IT is waited for — the caller's buffer is free when this returns, and a work stream that still holds
        // the previous batch's encoder (overlapped mode) is not drained.
        EncoderWorkspace& w = enc_ws;
        sb = w.stage_next;
        w.stage_next ^= 1;
        hipStream_t us = utility_stream();
        if (w.stage_read[sb]) HIP_CHECK(hipStreamWaitEvent(us, w.stage_read[sb], 0));  // the kernel that last read this buffer
        w.mel_stage[sb].ensure(bytes);
        HIP_CHECK(hipMemcpyAsync(w.mel_stage[sb].p, mel, bytes, hipMemcpyHostToDevice, us));
        if (!w.stage_copied) HIP_CHECK(hipEventCreateWithFlags(&w.stage_copied, hipEventDisableTiming));
        HIP_CHECK(hipEventRecord(w.stage_copied, us));
        HIP_CHECK(hipEventSynchronize(w.stage_copied));
        HIP_CHECK(hipStreamWaitEvent(stream, w.stage_copied, 0));
        src = w.mel_stage[sb].as<float>();
    }
    launch_mel_prep(src, n, n_frames, 80 * n_frames, seek, enc_ws.mel_t.as<half_t>(), stream);
    if (parity_f32) load_mel_f32(src, n, n_frames, 80 * n_frames, seek);
    if (sb >= 0) {
        if (!enc_ws.stage_read[sb]) HIP_CHECK(hipEventCreateWithFlags(&enc_ws.stage_read[sb], hipEventDisableTiming));
        HIP_CHECK(hipEventRecord(enc_ws.stage_read[sb], stream));
    }
}

void Model::load_mel_window(const float* mel_dev, int64_t n_frames, int64_t seek, int slot) {
    WRS_REQUIRE(slot >= 0 && slot < enc_ws.cap, "load_mel_window: slot %d out of range", slot);
    launch_mel_prep(mel_dev, 1, n_frames, 80 * n_frames, seek, enc_ws.mel_t.as<half_t>() + (size_t)slot * 3002 * 80, stream);
    if (parity_f32) load_mel_f32(mel_dev, 1, n_frames, 80 * n_frames, seek, slot);
}

void Model::encoder_from_mel_t(int n) {
    WRS_REQUIRE(finalized, "model not finalized");
    check_device("encoder");
    if (parity_f32) { encoder_f32(n); return; }
    WRS_REQUIRE(n >= 1 && n <= enc_ws.cap, "encoder: bad window count %d", n);
    EncoderWorkspace& w = enc_ws;
    w.n_encoded = n;
    ++w.epoch;
    const int64_t M = (int64_t)n * T;
    const float scale = std::pow((float)(D / H), -0.25f);
    const double dM = (double)M, dD = (double)D;
    int dbg_left = debug_enc_stop;  // debug hook: run only the first N launches (0 = all)
    const bool fused = fused_kv_pack && debug_enc_stop == 0 && D % 256 == 0 && T % 4 == 0 && gemm_pack_kv_supported(M, 3 * (int64_t)D);
#define WRS_DBG_STEP() do { if (debug_enc_stop > 0 && --dbg_left <= 0) return; } while (0)
    {   // conv1 + GELU (model.rs:421) as implicit GEMM, K = 3*80 (padded to 256 with zero weights)
        Timed tm(timer, "enc_conv1", stream, 0, 2.0 * n * 3000 * 240 * dD);
        GemmA A{w.mel_t.as<half_t>(), 80, 3000, 3002 * 80};
        GemmEpi e;
        e.bias = conv1_b; e.act = 1; e.c = w.c1.p; e.c_f32 = 0; e.ldc = D;
        e.c_rows_per_batch = 3000; e.c_batch_rows = 3001; e.c_row_off = 1;
        launch_gemm_f16(A, conv1_w, (int64_t)n * 3000, D, round_up(240, 64), e, stream);
    }
    WRS_DBG_STEP();
    {   // conv2 (stride 2) + GELU + permute + positional embedding (model.rs:422-431)
        Timed tm(timer, "enc_conv2", stream, 0, 2.0 * dM * 3 * dD * dD);
        GemmA A{w.c1.as<half_t>(), 2 * (int64_t)D, T, 3001 * (int64_t)D};
        GemmEpi e;
        e.bias = conv2_b; e.act = 1; e.posemb = enc_pos; e.c = w.x.p; e.c_f32 = 1; e.ldc = D;
        e.c_rows_per_batch = T; e.c_batch_rows = T; e.c_row_off = 0;
        launch_gemm_f16(A, conv2_w, M, D, 3 * (int64_t)D, e, stream);
    }
    WRS_DBG_STEP();
    for (int l = 0; l < L_enc; ++l) {
        const BlockW& b = enc[l];
        {
            Timed tm(timer, "enc_ln", stream, dM * dD * 6);
            launch_layernorm(w.x.as<float>(), b.attn_ln_g, b.attn_ln_b, M, D, w.xn.as<half_t>(), nullptr, stream);
        }
        WRS_DBG_STEP();
        {
            Timed tm(timer, "enc_gemm_qkv", stream, 0, 2.0 * dM * 3 * dD * dD);
            GemmA A{w.xn.as<half_t>(), D};
            GemmEpi e;
            e.bias = b.qkv_b; e.scale = scale; e.scale_n_end = 2 * D; e.c = w.qkv.p; e.ldc = 3 * D;
            if (fused) {  // q -> qkv buffer as before; k, v -> packed K / V from the epilogue
                e.pack_k = w.kp.as<half_t>(); e.pack_v = w.vp.as<half_t>(); e.pack_T = T; e.pack_H = H; e.pack_k_col0 = D; e.pack_v_col0 = 2 * D;
            }
            launch_gemm_f16(A, b.qkv_w, M, 3 * D, D, e, stream);
        }
        WRS_DBG_STEP();
        if (!fused) {
            Timed tm(timer, "enc_pack_kv", stream, dM * dD * 2 * 2 * 2);
            launch_pack_kv(w.qkv.as<half_t>(), 3 * D, D, 2 * D, n, T, H, w.kp.as<half_t>(), w.vp.as<half_t>(), stream);
        }
        WRS_DBG_STEP();
        {
            Timed tm(timer, "enc_attn", stream, 0, 4.0 * n * (double)T * T * dD);
            launch_attn_full(w.qkv.as<half_t>(), 3 * D, w.kp.as<half_t>(), w.vp.as<half_t>(), n, T, T, H, w.att.as<half_t>(), D, stream);
        }
        WRS_DBG_STEP();
        {
            Timed tm(timer, "enc_gemm_out", stream, 0, 2.0 * dM * dD * dD);
            GemmA A{w.att.as<half_t>(), D};
            GemmEpi e;
            e.bias = b.out_b; e.residual = w.x.as<float>(); e.c = w.x.p; e.c_f32 = 1; e.ldc = D;
            launch_gemm_f16(A, b.out_w, M, D, D, e, stream);
        }
        WRS_DBG_STEP();
        {
            Timed tm(timer, "enc_ln", stream, dM * dD * 6);
            launch_layernorm(w.x.as<float>(), b.mlp_ln_g, b.mlp_ln_b, M, D, w.xn.as<half_t>(), nullptr, stream);
        }
        WRS_DBG_STEP();
        {
            Timed tm(timer, "enc_gemm_mlp1", stream, 0, 2.0 * dM * 4 * dD * dD);
            GemmA A{w.xn.as<half_t>(), D};
            GemmEpi e;
            e.bias = b.mlp0_b; e.act = 1; e.c = w.h.p; e.ldc = 4 * D;
            launch_gemm_f16(A, b.mlp0_w, M, 4 * D, D, e, stream);
        }
        WRS_DBG_STEP();
        {
            Timed tm(timer, "enc_gemm_mlp2", stream, 0, 2.0 * dM * 4 * dD * dD);
            GemmA A{w.h.as<half_t>(), 4 * (int64_t)D};
            GemmEpi e;
            e.bias = b.mlp2_b; e.residual = w.x.as<float>(); e.c = w.x.p; e.c_f32 = 1; e.ldc = D;
            launch_gemm_f16(A, b.mlp2_w, M, D, 4 * D, e, stream);
        }
        WRS_DBG_STEP();
    }
    {
        Timed tm(timer, "enc_ln", stream, dM * dD * 10);
        launch_layernorm(w.x.as<float>(), ln_post_g, ln_post_b, M, D, w.xa16.as<half_t>(), w.xa32.as<float>(), stream);
    }
}

// ------------------------------------------------------------------------------------------ decoder
// Experiment switches (unset = the default path).  Read ONCE per model, at creation: a decode task's captured step graph is keyed
// on (rows, cross buffer) only, so a switch that changed under a live model would leave graph and eager steps on different paths.
void Model::read_switches() {
    if (const char* e = exp_env("WRS_DEC_BIG_GEMM_ROWS")) dec_big_gemm_rows = atoi(e);
    const char* e = exp_env("WRS_DEC_PRING");
    dec_slab_path = !(e && e[0] == '1');
    if (const char* f = exp_env("WRS_FUSED_PACK")) fused_kv_pack = f[0] == '1';
    e = exp_env("WRS_DEC_FUSED");
    dec_fused_tiny = e && e[0] == '1';
    e = exp_env("WRS_SELF_MFMA_MIN_AUDIO");
    self_mfma_min_audio = e ? atoi(e) : 64;
}

void Model::init_state(DecState& st, int a_cap, int group, int ctx) {
    WRS_REQUIRE(finalized, "model not finalized");
    WRS_REQUIRE(a_cap >= 1 && group >= 1 && group <= 16, "decoder state: bad capacity (audios %d, group %d)", a_cap, group);
    const int r_cap = a_cap * group;
    const int m_pad = skinny_m_pad(r_cap);
    WRS_REQUIRE(m_pad > 0, "decoder state: %d rows exceed the 2560-row limit of the decode kernels", r_cap);
    use();
    st.a_cap = a_cap; st.group = group; st.r_cap = r_cap; st.m_pad = m_pad;
    const size_t per_head = packed_kv_halfs_per_head(T);
    st.cross_kp.alloc((size_t)L_dec * a_cap * H * per_head * 2);
    st.cross_vp.alloc((size_t)L_dec * a_cap * H * per_head * 2);
    resize_ctx(st, ctx > 0 ? std::min(ctx, n_ctx) : n_ctx);
    st.act_rows = (int)round_up(m_pad, 128);  // packed activations are also read in 128-row tiles by the big GEMM
    st.x.alloc((size_t)m_pad * D * 4);
    st.xn.alloc((size_t)st.act_rows * D * 2);
    st.att.alloc((size_t)st.act_rows * D * 2);
    st.h.alloc((size_t)st.act_rows * 4 * D * 2);
    st.q16.alloc((size_t)st.act_rows * D * 2);
    st.slab2.alloc((size_t)40 * 16 * 4 * D * 4);
    st.x2.alloc((size_t)16 * D * 4);
    st.slab.alloc((size_t)40 * m_pad * 4 * D * 4);  // ksplit (<= K/32 <= 40 for N = 4D; <= 160 for N = D) * m_pad * N floats
    st.logits_ld = (int)round_up(V, 8);  // row pitch of the logits buffer (16-byte aligned rows for vector stores)
    st.logits.alloc((size_t)r_cap * st.logits_ld * 4);
    st.ckv_tmp.alloc((size_t)a_cap * T * 2 * D * 2);
    st.sp_dev.alloc(st.sp_bytes());
    st.sp_host.alloc(st.sp_bytes());
    st.topk.alloc((size_t)r_cap * 17 * 4);
    st.filter_ws.alloc(filter_ws_floats(std::min(r_cap, FILTER_SPLIT_MAX_ROWS)) * 4);
    st.len = 0;
    st.parity = 0;
    st.n_audio = 0;
    st.pending_src.clear();
    if (parity_f32) init_state_f32(st);
}

void Model::load_cross_kv(DecState& st, const half_t* xa16, int n_audio) {
    WRS_REQUIRE(n_audio >= 1 && n_audio <= st.a_cap, "cross K/V: %d audios exceed capacity %d", n_audio, st.a_cap);
    check_device("cross K/V projection");
    if (parity_f32) { load_cross_kv_f32(st, n_audio); return; }  // from enc_ws.xa32, which always accompanies xa16
    const int64_t M = (int64_t)n_audio * T;
    const float scale = std::pow((float)(D / H), -0.25f);
    const size_t per_layer = (size_t)st.a_cap * H * packed_kv_halfs_per_head(T);
    const bool fused = fused_kv_pack && D % 256 == 0 && T % 4 == 0 && gemm_pack_kv_supported(M, 2 * (int64_t)D);
    for (int l = 0; l < L_dec; ++l) {
        const BlockW& b = dec[l];
        {
            Timed tm(timer, "dec_cross_kv_proj", stream, 0, 2.0 * (double)M * 2 * D * (double)D);
            GemmA A{xa16, D};
            GemmEpi e;
            e.bias = b.ckv_b; e.scale = scale; e.scale_n_end = D; e.c = st.ckv_tmp.p; e.ldc = 2 * D;
            if (fused) {
                e.pack_k = st.ckp(st.cross_wsel) + l * per_layer; e.pack_v = st.cvp(st.cross_wsel) + l * per_layer;
                e.pack_T = T; e.pack_H = H; e.pack_k_col0 = 0; e.pack_v_col0 = D;
            }
            launch_gemm_f16(A, b.ckv_w, M, 2 * D, D, e, stream);
        }
        if (!fused) {
            Timed tm(timer, "dec_cross_kv_pack", stream, (double)M * D * 2 * 2 * 2);
            launch_pack_kv(st.ckv_tmp.as<half_t>(), 2 * D, 0, D, n_audio, T, H, st.ckp(st.cross_wsel) + l * per_layer,
                           st.cvp(st.cross_wsel) + l * per_layer, stream);
        }
    }
    st.n_audio = n_audio;
}

// The self-attention K/V cache, the ancestry table and the key lists hold `ctx` positions per row (<= n_text_ctxs): a decode
// task sizes them by what its prompt and sample_len can reach (large-v2, 1 535 rows: 58 GB for 256 positions against 113 GB for
// all 448) and grows them when a longer prompt arrives.  Contents are lost: call between runs only.
void Model::resize_ctx(DecState& st, int ctx) {
    WRS_REQUIRE(ctx >= 1 && ctx <= n_ctx && st.r_cap >= 1, "decoder state: %d cache positions not in 1..%d", ctx, n_ctx);
    use();
    st.kc.release(); st.vc.release();  // before the new allocation: both sizes together may not fit
    st.ctx = ctx;
    st.kc.alloc((size_t)L_dec * st.r_cap * ctx * D * 2);
    st.vc.alloc((size_t)L_dec * st.r_cap * ctx * D * 2);
    st.anc.alloc((size_t)2 * st.r_cap * ctx * 4);
    st.self_keys_stride = (st.r_cap / std::max(st.a_cap, 1)) * ctx + 1;  // [0] = count
    st.self_keys.alloc(((size_t)st.a_cap * st.self_keys_stride + st.a_cap) * 4);
    if (st.f32) {  // f32 parity mode: its fp32 caches follow
        st.f32->kc.alloc((size_t)L_dec * st.r_cap * ctx * D * 4);
        st.f32->vc.alloc((size_t)L_dec * st.r_cap * ctx * D * 4);
    }
    st.len = 0;
}

// split-K factor of a decode GEMM: enough workgroups (N/32 * ksplit >= ~200) to spread the weight stream over
// the 256 CUs, at most 10 k-steps per wave (gemm.hip SK_NKS) and at least 4 k-steps per workgroup.  The total
// activation traffic of a GEMM does not depend on ksplit (every workgroup reads only its k-slice of A), so
// splitting only costs slab bytes (ksplit * M * N * 4), which the consumer kernel reduces.
static int pick_ksplit(int64_t N, int64_t K, int cap) {
    const int ksteps = (int)(K / 32);
    const int wgs = (int)(N / 32);
    int ks = 1;
    while (ks * 2 <= cap && ksteps % (ks * 2) == 0 && ksteps / (ks * 2) >= 4 && (wgs * ks < 200 || ksteps / ks > 40)) ks *= 2;
    return ks;
}

void Model::decoder_step(DecState& st, int rows, int phys_mul, int q_per_audio, bool update_anc, bool want_logits) {
    WRS_REQUIRE(rows >= 1 && rows <= st.r_cap, "decoder_step: bad row count %d", rows);
    check_device("decoder step");
    WRS_REQUIRE((rows - 1) * phys_mul < st.r_cap, "decoder_step: physical row out of range");
    WRS_REQUIRE(rows % q_per_audio == 0 && rows / q_per_audio <= st.n_audio, "decoder_step: rows %d do not match %d audios x %d", rows, st.n_audio, q_per_audio);
    if (parity_f32) { decoder_step_f32(st, rows, phys_mul, q_per_audio, update_anc, want_logits); return; }
    const float scale = std::pow((float)(D / H), -0.25f);
    const StepParams* sp = st.sp_d();
    const int64_t anc_stride = (int64_t)st.r_cap * st.ctx;
    const size_t kv_layer = (size_t)st.r_cap * st.ctx * D;
    const size_t cross_layer = (size_t)st.a_cap * H * packed_kv_halfs_per_head(T);
    const int n_audio = rows / q_per_audio;
    const double wD = (double)D * D * 2;  // bytes of a [D, D] fp16 matrix
    int ks_qkv = gemm_dec_plan(rows, 3 * D, D, nullptr, nullptr), ks_d = gemm_dec_plan(rows, D, D, nullptr, nullptr),
        ks_m1 = gemm_dec_plan(rows, 4 * D, D, nullptr, nullptr), ks_m2 = gemm_dec_plan(rows, D, 4 * (int64_t)D, nullptr, nullptr);
    const int mp = st.m_pad_for(rows);
    float* slab = st.slab.as<float>();
    // Few rows: latency-optimised split-K kernel (slabs).  Many rows (>= big_rows): the step's GEMMs are MFMA/L2-bound
    // like the encoder's, so the 128x128 tiled kernel runs them (packed-A mode) and writes ONE fp32 slab.
    const bool big = rows >= dec_big_gemm_rows;
    auto gemm_slab = [&](const char* name, const half_t* a, const half_t* w, int64_t N, int64_t K, int& ks, double bytes) {
        Timed tm(timer, name, stream, bytes);
        if (big) {
            GemmA A;
            A.a = a; A.packed = 1; A.packed_rows = st.act_rows;
            GemmEpi e;
            e.c = slab; e.c_f32 = 1; e.ldc = N;
            // keep >= ~256 workgroups in flight: split K when the tile grid alone is too small (N = D shapes)
            const int64_t tiles = cdiv(rows, 128) * cdiv(N, 128);
            int split = 1;
            // ... but stay within one workgroup per CU: that grid runs on the loader/consumer ring kernel (gemm.hip), whose
            // k-loop is 1.6x faster, and every extra slice is another fp32 slab to write and reduce
            while (tiles * split * 2 <= 256 && (K / 64) % (split * 2) == 0 && K / 64 / (split * 2) >= 10) split *= 2;
            e.ksplit = split; e.slab_stride = (int64_t)mp * N;
            launch_gemm_f16(A, w, rows, N, K, e, stream);
            ks = split;
        } else {
            launch_gemm_dec(a, w, rows, N, round_up(N, 128), K, slab, stream);
        }
    };

    if (big && !dec_slab_path) {
        step_paths_seen |= 1 | 2;
        decoder_step_big(st, rows, phys_mul, q_per_audio, update_anc, want_logits);
        return;
    }
    if (rows <= 16 && dec_fused_tiny) {
        step_paths_seen |= 4;
        decoder_step_tiny(st, rows, phys_mul, q_per_audio, update_anc, want_logits);
        return;
    }
    if (update_anc) {
        Timed tm(timer, "dec_anc_update", stream);
        launch_anc_update(st.anc.as<int>(), anc_stride, st.ctx, rows, sp, st.src_d(), stream);
    }
    // many audios with several beams each: self-attention over the DISTINCT (position, row) pairs of an audio's beams, on
    // the matrix cores (attn_self_mfma.hip); the pair list follows from the ancestry table, once per step for all layers
    const bool self_mfma = self_mfma_min_audio > 0 && phys_mul == 1 && q_per_audio >= 2 && q_per_audio <= 8 && n_audio >= self_mfma_min_audio &&
                           q_per_audio * st.ctx + 1 <= st.self_keys_stride;
    int* self_klist = st.self_keys.as<int>();
    int* self_kcount = self_klist + (size_t)st.a_cap * st.self_keys_stride;
    // ... and at >= dec_big_gemm_rows rows the QKV GEMM's epilogue hands the attention kernel its operands directly (query as
    // fp16, new key / value rows appended to the cache: no fp32 slab to write and re-read; WRS_DEC_QKV_SLAB=1 keeps the slab)
    static const bool qkv_slab_forced = exp_env("WRS_DEC_QKV_SLAB") && exp_env("WRS_DEC_QKV_SLAB")[0] == '1';
    const bool qkv_direct = self_mfma && big && !qkv_slab_forced;
    step_paths_seen |= (big ? 1 : 0) | (self_mfma ? 8 : 0) | (qkv_direct ? 16 : 0);
    if (self_mfma) {
        Timed tm(timer, "dec_self_keys", stream);
        launch_self_keys(st.anc.as<int>(), anc_stride, st.ctx, n_audio, q_per_audio, sp, self_klist, st.self_keys_stride, self_kcount, qkv_direct, stream);
    }
    {
        Timed tm(timer, "dec_embed_ln", stream);
        launch_embed_ln(emb16, dec_pos, sp, st.tokens_d(), dec[0].attn_ln_g, dec[0].attn_ln_b, rows, D, st.x.as<float>(),
                        st.xn.as<half_t>(), stream);
    }
    for (int l = 0; l < L_dec; ++l) {
        const BlockW& b = dec[l];
        // x + attn(attn_ln(x))   (model.rs:302-305)
        if (qkv_direct) {
            Timed tm(timer, "dec_gemm_qkv", stream, 3 * wD);
            GemmA A;
            A.a = st.xn.as<half_t>(); A.packed = 1; A.packed_rows = st.act_rows;
            GemmEpi e;
            e.bias = b.qkv_b; e.scale = scale; e.scale_n_end = 2 * D; e.c = st.q16.p; e.c_f32 = 0; e.ldc = D;
            e.qkv_d_model = D; e.kcache = st.kc.as<half_t>() + l * kv_layer; e.vcache = st.vc.as<half_t>() + l * kv_layer;
            e.cache_row_stride = (int64_t)st.ctx * D; e.sp = sp;
            launch_gemm_f16(A, b.qkv_w, rows, 3 * D, D, e, stream);
        } else {
            gemm_slab("dec_gemm_qkv", st.xn.as<half_t>(), b.qkv_w, 3 * D, D, ks_qkv, 3 * wD);
        }
        {
            Timed tm(timer, "dec_attn_self", stream);
            if (qkv_direct)
                launch_attn_self_mfma_q16(st.q16.as<half_t>(), st.kc.as<half_t>() + l * kv_layer, st.vc.as<half_t>() + l * kv_layer, st.ctx, rows,
                                          q_per_audio, H, D, sp, self_klist, st.self_keys_stride, self_kcount, st.att.as<half_t>(), stream);
            else if (self_mfma)
                launch_attn_self_mfma(slab, ks_qkv, mp, b.qkv_b, scale, st.kc.as<half_t>() + l * kv_layer, st.vc.as<half_t>() + l * kv_layer, st.ctx,
                                      rows, q_per_audio, H, D, sp, self_klist, st.self_keys_stride, self_kcount, st.att.as<half_t>(), stream);
            else
                launch_attn_self(slab, ks_qkv, mp, b.qkv_b, scale, st.kc.as<half_t>() + l * kv_layer, st.vc.as<half_t>() + l * kv_layer,
                                 st.anc.as<int>(), anc_stride, st.ctx, rows, q_per_audio, phys_mul, H, D, sp, st.att.as<half_t>(), stream);
        }
        gemm_slab("dec_gemm_out", st.att.as<half_t>(), b.out_w, D, D, ks_d, wD);
        {
            Timed tm(timer, "dec_finish_ln", stream);
            launch_finish_residual_ln(st.x.as<float>(), slab, ks_d, mp, b.out_b, b.cross_ln_g, b.cross_ln_b, rows, D,
                                      st.xn.as<half_t>(), stream);
        }
        // + cross_attn(cross_attn_ln(x), xa)   (model.rs:309-310)
        gemm_slab("dec_gemm_cq", st.xn.as<half_t>(), b.cq_w, D, D, ks_d, wD);
        {
            Timed tm(timer, "dec_attn_cross", stream, (double)n_audio * 2.0 * T * D * 2.0);
            const bool sample = cross_ev_now && (l % cross_ev_stride == 0);
            const int slot = l / cross_ev_stride;
            if (sample) {
                while ((int)cross_ev.size() < 2 * (slot + 1)) { hipEvent_t e; HIP_CHECK(hipEventCreate(&e)); cross_ev.push_back(e); }
                HIP_CHECK(hipEventRecord(cross_ev[2 * slot], stream));
            }
            launch_attn_cross(slab, ks_d, mp, b.cq_b, scale, st.ckp(st.cross_sel) + l * cross_layer,
                              st.cvp(st.cross_sel) + l * cross_layer, n_audio, q_per_audio, T, H, st.att.as<half_t>(), stream);
            if (sample) HIP_CHECK(hipEventRecord(cross_ev[2 * slot + 1], stream));
        }
        gemm_slab("dec_gemm_out", st.att.as<half_t>(), b.cout_w, D, D, ks_d, wD);
        {
            Timed tm(timer, "dec_finish_ln", stream);
            launch_finish_residual_ln(st.x.as<float>(), slab, ks_d, mp, b.cout_b, b.mlp_ln_g, b.mlp_ln_b, rows, D,
                                      st.xn.as<half_t>(), stream);
        }
        // + mlp(mlp_ln(x))   (model.rs:315)
        if (big) {  // bias + GELU + packed fp16 store in the GEMM epilogue: no slab, no finish kernel
            Timed tm(timer, "dec_gemm_mlp1", stream, 4 * wD);
            GemmA A;
            A.a = st.xn.as<half_t>(); A.packed = 1; A.packed_rows = st.act_rows;
            GemmEpi e;
            e.bias = b.mlp0_b; e.act = 1; e.c = st.h.p; e.c_f32 = 0; e.ldc = 4 * D; e.c_packed = 1;
            launch_gemm_f16(A, b.mlp0_w, rows, 4 * D, D, e, stream);
        } else {
            gemm_slab("dec_gemm_mlp1", st.xn.as<half_t>(), b.mlp0_w, 4 * D, D, ks_m1, 4 * wD);
            Timed tm(timer, "dec_finish_gelu", stream);
            launch_finish_act(slab, ks_m1, mp, b.mlp0_b, 1, rows, 4 * D, st.h.as<half_t>(), stream);
        }
        gemm_slab("dec_gemm_mlp2", st.h.as<half_t>(), b.mlp2_w, D, 4 * (int64_t)D, ks_m2, 4 * wD);
        {
            Timed tm(timer, "dec_finish_ln", stream);
            const float* g = (l + 1 < L_dec) ? dec[l + 1].attn_ln_g : ln_g;
            const float* bb = (l + 1 < L_dec) ? dec[l + 1].attn_ln_b : ln_b;
            launch_finish_residual_ln(st.x.as<float>(), slab, ks_m2, mp, b.mlp2_b, g, bb, rows, D, st.xn.as<half_t>(), stream);
        }
    }
    if (want_logits) decoder_logits(st, rows);
}

// Decode step for <= 16 rows (one window x beam 5: the reference's own usage mode, TranscribeTask::run).  There the step is
// launch-bound (12 dependent launches per layer), so the three "finish" kernels and the MLP-1 finish of a layer run inside
// the NEXT GEMM's prologue (gemm_dec_fused: every workgroup rebuilds its activation fragments from the previous GEMM's
// split-K slabs; 5 rows of L2-resident data): 8 launches per layer.  Slabs and the residual stream ping-pong between two
// buffers (a fused GEMM reads one while writing the other).
void Model::decoder_step_tiny(DecState& st, int rows, int phys_mul, int q_per_audio, bool update_anc, bool want_logits) {
    const float scale = std::pow((float)(D / H), -0.25f);
    const StepParams* sp = st.sp_d();
    const int64_t anc_stride = (int64_t)st.r_cap * st.ctx;
    const size_t kv_layer = (size_t)st.r_cap * st.ctx * D;
    const size_t cross_layer = (size_t)st.a_cap * H * packed_kv_halfs_per_head(T);
    const int n_audio = rows / q_per_audio;
    const int mp = 16;
    const int64_t np_d = round_up(D, 128), np_3d = round_up(3 * D, 128), np_4d = round_up(4 * (int64_t)D, 128);
    float* slabs[2] = {st.slab.as<float>(), st.slab2.as<float>()};
    float* xs[2] = {st.x.as<float>(), st.x2.as<float>()};
    int sb = 0, xb = 0;  // buffer a GEMM writes next / buffer that holds the residual stream
    half_t* att = st.att.as<half_t>();
    if (update_anc) {
        Timed tm(timer, "dec_anc_update", stream);
        launch_anc_update(st.anc.as<int>(), anc_stride, st.ctx, rows, sp, st.src_d(), stream);
    }
    {
        Timed tm(timer, "dec_embed_ln", stream);
        launch_embed_ln(emb16, dec_pos, sp, st.tokens_d(), dec[0].attn_ln_g, dec[0].attn_ln_b, rows, D, xs[xb], st.xn.as<half_t>(), stream);
    }
    // pending producer: the previous GEMM's slabs + bias, not yet added to the residual stream
    const float* pend_slab = nullptr;
    const float* pend_bias = nullptr;
    int pend_ks = 0;
    auto plain = [&](const char* name, const half_t* a, const half_t* w, int64_t N, int64_t n_pad, int64_t K) -> int {
        Timed tm(timer, name, stream);
        launch_gemm_dec(a, w, rows, N, n_pad, K, slabs[sb], stream);
        pend_slab = slabs[sb];
        sb ^= 1;
        return gemm_dec_plan(rows, N, K, nullptr, nullptr);
    };
    auto fused_ln = [&](const char* name, const float* g, const float* b, const half_t* w, int64_t N, int64_t n_pad) -> int {
        Timed tm(timer, name, stream);
        const int ks = launch_gemm_dec_fused(0, pend_slab, pend_ks, mp, pend_bias, xs[xb], xs[xb ^ 1], g, b, w, rows, N, n_pad, D, slabs[sb], stream);
        xb ^= 1;
        pend_slab = slabs[sb];
        sb ^= 1;
        return ks;
    };
    for (int l = 0; l < L_dec; ++l) {
        const BlockW& b = dec[l];
        int ks_qkv;
        if (l == 0) ks_qkv = plain("dec_gemm_qkv", st.xn.as<half_t>(), b.qkv_w, 3 * D, np_3d, D);
        else ks_qkv = fused_ln("dec_gemm_qkv", b.attn_ln_g, b.attn_ln_b, b.qkv_w, 3 * D, np_3d);  // + the previous layer's MLP-2 finish
        {
            Timed tm(timer, "dec_attn_self", stream);
            launch_attn_self(pend_slab, ks_qkv, mp, b.qkv_b, scale, st.kc.as<half_t>() + l * kv_layer, st.vc.as<half_t>() + l * kv_layer,
                             st.anc.as<int>(), anc_stride, st.ctx, rows, q_per_audio, phys_mul, H, D, sp, att, stream);
        }
        pend_ks = plain("dec_gemm_out", att, b.out_w, D, np_d, D);
        pend_bias = b.out_b;
        const int ks_cq = fused_ln("dec_gemm_cq", b.cross_ln_g, b.cross_ln_b, b.cq_w, D, np_d);
        {
            Timed tm(timer, "dec_attn_cross", stream, (double)n_audio * 2.0 * T * D * 2.0);
            launch_attn_cross(pend_slab, ks_cq, mp, b.cq_b, scale, st.ckp(st.cross_sel) + l * cross_layer, st.cvp(st.cross_sel) + l * cross_layer,
                              n_audio, q_per_audio, T, H, att, stream);
        }
        pend_ks = plain("dec_gemm_out", att, b.cout_w, D, np_d, D);
        pend_bias = b.cout_b;
        const int ks_m1 = fused_ln("dec_gemm_mlp1", b.mlp_ln_g, b.mlp_ln_b, b.mlp0_w, 4 * D, np_4d);
        {   // MLP-2 with the MLP-1 finish (bias + exact GELU) in its prologue
            Timed tm(timer, "dec_gemm_mlp2", stream);
            pend_ks = launch_gemm_dec_fused(1, pend_slab, ks_m1, mp, b.mlp0_b, nullptr, nullptr, nullptr, nullptr, b.mlp2_w, rows, D, np_d,
                                            4 * (int64_t)D, slabs[sb], stream);
            pend_slab = slabs[sb];
            sb ^= 1;
            pend_bias = b.mlp2_b;
        }
    }
    {   // the last MLP-2's finish + final LayerNorm (model.rs:518) stay a kernel: the logits GEMM takes packed activations
        Timed tm(timer, "dec_finish_ln", stream);
        launch_finish_residual_ln(xs[xb], pend_slab, pend_ks, mp, pend_bias, ln_g, ln_b, rows, D, st.xn.as<half_t>(), stream);
    }
    if (want_logits) decoder_logits(st, rows);
}

// Decode step at >= dec_big_gemm_rows rows, alternative form (WRS_DEC_PRING=1): the persistent ring GEMM (gemm_pring.hip)
// with epilogues that write every consumer's operand directly — no split-K slabs, no "finish" kernels.  Correct (kernel-
// and model-level parity tests) but NOT faster than the slab path at 1280 rows — measured per layer 12.6 / 11.1 / 31.5 /
// 34.2 / 34.2 us (out, query, MLP-2, QKV, MLP-1) against 10.1 / 10.1 / 23.9 / 23.4 / 29.7 us for the slab kernels
// (profiles/r02_gemm_pring_micro.log): a persistent workgroup cannot overlap one tile's epilogue with the next tile's
// MFMAs, which two co-resident one-tile workgroups do for free — so it is not the default.  Per layer (11 launches):
//   QKV GEMM (q -> fp16, k / v -> the cache) -> self-attention -> out GEMM (x += ..) -> LN ->
//   query GEMM (-> fp16) -> cross-attention -> out GEMM (x += ..) -> LN -> MLP-1 GEMM (GELU, packed) -> MLP-2 GEMM (x += ..) -> LN
void Model::decoder_step_big(DecState& st, int rows, int phys_mul, int q_per_audio, bool update_anc, bool want_logits) {
    const float scale = std::pow((float)(D / H), -0.25f);
    const StepParams* sp = st.sp_d();
    const int64_t anc_stride = (int64_t)st.r_cap * st.ctx;
    const size_t kv_layer = (size_t)st.r_cap * st.ctx * D;
    const size_t cross_layer = (size_t)st.a_cap * H * packed_kv_halfs_per_head(T);
    const int n_audio = rows / q_per_audio;
    const double wD = (double)D * D * 2;
    float* x = st.x.as<float>();
    half_t* xn = st.xn.as<half_t>();
    half_t* att = st.att.as<half_t>();
    half_t* q16 = st.q16.as<half_t>();
    auto resid = [&](const char* name, const half_t* a, const half_t* w, const float* bias, int64_t K, double bytes) {
        Timed tm(timer, name, stream, bytes);
        PringEpi e;
        e.mode = PRING_RESID; e.bias = bias; e.x = x; e.ldc = D;
        launch_gemm_pring(a, st.act_rows, w, rows, D, K, 64, e, stream);
    };
    auto ln = [&](const float* g, const float* b) {
        Timed tm(timer, "dec_ln", stream);
        launch_layernorm_packed(x, g, b, rows, D, xn, stream);
    };
    if (update_anc) {
        Timed tm(timer, "dec_anc_update", stream);
        launch_anc_update(st.anc.as<int>(), anc_stride, st.ctx, rows, sp, st.src_d(), stream);
    }
    {
        Timed tm(timer, "dec_embed_ln", stream);
        launch_embed_ln(emb16, dec_pos, sp, st.tokens_d(), dec[0].attn_ln_g, dec[0].attn_ln_b, rows, D, x, xn, stream);
    }
    for (int l = 0; l < L_dec; ++l) {
        const BlockW& b = dec[l];
        half_t* kc = st.kc.as<half_t>() + l * kv_layer;
        half_t* vc = st.vc.as<half_t>() + l * kv_layer;
        {   // x + attn(attn_ln(x))   (model.rs:302-305)
            Timed tm(timer, "dec_gemm_qkv", stream, 3 * wD);
            PringEpi e;
            e.mode = PRING_QKV; e.bias = b.qkv_b; e.scale = scale; e.scale_n_end = 2 * D; e.c16 = q16; e.ldc = D;
            e.kcache = kc; e.vcache = vc; e.d_model = D; e.cache_row_stride = (int64_t)st.ctx * D; e.phys_mul = phys_mul; e.sp = sp;
            launch_gemm_pring(xn, st.act_rows, b.qkv_w, rows, 3 * D, D, 128, e, stream);
        }
        {
            Timed tm(timer, "dec_attn_self", stream);
            launch_attn_self_q16(q16, kc, vc, st.anc.as<int>(), anc_stride, st.ctx, rows, q_per_audio, phys_mul, H, D, sp, att, stream);
        }
        resid("dec_gemm_out", att, b.out_w, b.out_b, D, wD);
        ln(b.cross_ln_g, b.cross_ln_b);
        {   // + cross_attn(cross_attn_ln(x), xa)   (model.rs:309-310)
            Timed tm(timer, "dec_gemm_cq", stream, wD);
            PringEpi e;
            e.mode = PRING_F16; e.bias = b.cq_b; e.scale = scale; e.scale_n_end = D; e.c16 = q16; e.ldc = D;
            launch_gemm_pring(xn, st.act_rows, b.cq_w, rows, D, D, 64, e, stream);
        }
        {
            Timed tm(timer, "dec_attn_cross", stream, (double)n_audio * 2.0 * T * D * 2.0);
            const bool sample = cross_ev_now && (l % cross_ev_stride == 0);
            const int slot = l / cross_ev_stride;
            if (sample) {
                while ((int)cross_ev.size() < 2 * (slot + 1)) { hipEvent_t ev; HIP_CHECK(hipEventCreate(&ev)); cross_ev.push_back(ev); }
                HIP_CHECK(hipEventRecord(cross_ev[2 * slot], stream));
            }
            launch_attn_cross_q16(q16, st.ckp(st.cross_sel) + l * cross_layer, st.cvp(st.cross_sel) + l * cross_layer, n_audio, q_per_audio, T, H, att, stream);
            if (sample) HIP_CHECK(hipEventRecord(cross_ev[2 * slot + 1], stream));
        }
        resid("dec_gemm_out", att, b.cout_w, b.cout_b, D, wD);
        ln(b.mlp_ln_g, b.mlp_ln_b);
        {   // + mlp(mlp_ln(x))   (model.rs:315)
            Timed tm(timer, "dec_gemm_mlp1", stream, 4 * wD);
            PringEpi e;
            e.mode = PRING_GELU_PACKED; e.bias = b.mlp0_b; e.c16 = st.h.as<half_t>(); e.ldc = 4 * D;
            launch_gemm_pring(xn, st.act_rows, b.mlp0_w, rows, 4 * D, D, 128, e, stream);
        }
        resid("dec_gemm_mlp2", st.h.as<half_t>(), b.mlp2_w, b.mlp2_b, 4 * (int64_t)D, 4 * wD);
        ln((l + 1 < L_dec) ? dec[l + 1].attn_ln_g : ln_g, (l + 1 < L_dec) ? dec[l + 1].attn_ln_b : ln_b);
    }
    if (want_logits) decoder_logits(st, rows);
}

void Model::decoder_logits(DecState& st, int rows) {
    Timed tm(timer, "dec_gemm_logits", stream, (double)V * D * 2 + (double)rows * V * 4);
    if (rows >= dec_big_gemm_rows) {  // tied output projection on the tiled kernel; logits rows padded to st.logits_ld floats
        GemmA A;
        A.a = st.xn.as<half_t>(); A.packed = 1; A.packed_rows = st.act_rows;
        GemmEpi e;
        e.c = st.logits.p; e.c_f32 = 1; e.ldc = st.logits_ld;
        launch_gemm_f16(A, emb_p, rows, st.logits_ld, D, e, stream);  // columns >= V come from zero weight rows
    } else {
        SkinnyEpi e;
        e.kind = SK_F32; e.c = st.logits.p; e.ldc = st.logits_ld;
        launch_gemm_skinny(st.xn.as<half_t>(), emb_p, rows, V, v_pad, D, 1, e, stream);
    }
}

// ------------------------------------------------------------------------------------------ prompt prefill
// All S prompt tokens of every audio in ONE pass (reference decode/mod.rs:236-245): the rows (audio, position) go through
// the encoder's kernels — row-major fp16 activations, tiled MFMA GEMMs with fused bias / scale / GELU / residual
// epilogues — with a causal self-attention (model.rs:191-201) whose K/V also land in the decode cache, and the
// encoder-style full attention over the audio's packed cross K/V.  (Round 1 fed the prompt one token at a time: ~2.5 ms per
// token on large-v2, which doubled the latency of every window after the first with condition_on_prev_text.)
void Model::decoder_prefill(DecState& st, int n_audio, int S, int pos0, int phys_mul, const int* tokens_host, bool all_logits) {
    WRS_REQUIRE(finalized && !parity_f32, "decoder_prefill: not available in this mode");
    check_device("decoder prefill");
    WRS_REQUIRE(n_audio >= 1 && n_audio <= st.n_audio && S >= 1 && pos0 >= 0 && pos0 + S <= st.ctx, "decoder_prefill: bad shape (%d audios, %d tokens at %d)", n_audio, S, pos0);
    WRS_REQUIRE((int64_t)(n_audio - 1) * phys_mul < st.r_cap, "decoder_prefill: physical row out of range");
    WRS_REQUIRE(pos0 == 0, "decoder_prefill: a non-empty cache is not supported (causal attention over cached + new positions)");
    PrefillWorkspace& w = st.prefill;
    const int64_t M = (int64_t)n_audio * S;
    if (M > w.cap_rows || n_audio > w.cap_audio || S > w.cap_s) {
        const int64_t rows = std::max<int64_t>(M, w.cap_rows);
        const int na = std::max(n_audio, w.cap_audio), ss = std::max(S, w.cap_s);
        const int64_t r = std::max<int64_t>(rows, (int64_t)na * ss);
        w.tokens.alloc((size_t)r * 4);
        w.x.alloc((size_t)r * D * 4);
        w.xn.alloc(((size_t)r * D + 4096) * 2);       // tiled GEMMs may over-read up to one tile of rows: clamped in-kernel, padding for K tails
        w.qkv.alloc((size_t)r * 3 * D * 2);
        w.kp.alloc((size_t)na * H * packed_kv_halfs_per_head(ss) * 2);
        w.vp.alloc((size_t)na * H * packed_kv_halfs_per_head(ss) * 2);
        w.att.alloc(((size_t)r * D + 4096) * 2);
        w.h.alloc(((size_t)r * 4 * D + 4096) * 2);
        w.cap_rows = r; w.cap_audio = na; w.cap_s = ss;
    }
    if (all_logits) w.logits_all.ensure((size_t)M * st.logits_ld * 4);
    const float scale = std::pow((float)(D / H), -0.25f);
    const size_t kv_layer = (size_t)st.r_cap * st.ctx * D;
    const size_t cross_layer = (size_t)st.a_cap * H * packed_kv_halfs_per_head(T);
    HIP_CHECK(hipMemcpyAsync(w.tokens.p, tokens_host, (size_t)M * 4, hipMemcpyHostToDevice, stream));
    float* x = w.x.as<float>();
    {
        Timed tm(timer, "pre_embed_ln", stream);
        launch_prefill_embed_ln(emb16, dec_pos, w.tokens.as<int>(), S, pos0, dec[0].attn_ln_g, dec[0].attn_ln_b, (int)M, D, x, w.xn.as<half_t>(), stream);
    }
    for (int l = 0; l < L_dec; ++l) {
        const BlockW& b = dec[l];
        {   // x + attn(attn_ln(x), mask)   (model.rs:302-305)
            Timed tm(timer, "pre_gemm", stream);
            GemmA A{w.xn.as<half_t>(), D};
            GemmEpi e;
            e.bias = b.qkv_b; e.scale = scale; e.scale_n_end = 2 * D; e.c = w.qkv.p; e.ldc = 3 * D;
            launch_gemm_f16(A, b.qkv_w, M, 3 * D, D, e, stream);
        }
        {
            Timed tm(timer, "pre_attn_self", stream);
            launch_prefill_store_kv(w.qkv.as<half_t>(), n_audio, S, pos0, D, st.ctx, phys_mul, st.kc.as<half_t>() + l * kv_layer,
                                    st.vc.as<half_t>() + l * kv_layer, stream);
            launch_pack_kv(w.qkv.as<half_t>(), 3 * D, D, 2 * D, n_audio, S, H, w.kp.as<half_t>(), w.vp.as<half_t>(), stream);
            launch_attn_full(w.qkv.as<half_t>(), 3 * D, w.kp.as<half_t>(), w.vp.as<half_t>(), n_audio, S, S, H, w.att.as<half_t>(), D, stream, true);
        }
        {
            Timed tm(timer, "pre_gemm", stream);
            GemmA A{w.att.as<half_t>(), D};
            GemmEpi e;
            e.bias = b.out_b; e.residual = x; e.c = x; e.c_f32 = 1; e.ldc = D;
            launch_gemm_f16(A, b.out_w, M, D, D, e, stream);
        }
        launch_layernorm(x, b.cross_ln_g, b.cross_ln_b, M, D, w.xn.as<half_t>(), nullptr, stream);
        {   // + cross_attn(cross_attn_ln(x), xa)   (model.rs:309-310): every prompt position attends to the audio's 1500 keys
            Timed tm(timer, "pre_gemm", stream);
            GemmA A{w.xn.as<half_t>(), D};
            GemmEpi e;
            e.bias = b.cq_b; e.scale = scale; e.scale_n_end = D; e.c = w.qkv.p; e.ldc = D;
            launch_gemm_f16(A, b.cq_w, M, D, D, e, stream);
        }
        {
            Timed tm(timer, "pre_attn_cross", stream);
            launch_attn_full(w.qkv.as<half_t>(), D, st.ckp(st.cross_sel) + l * cross_layer, st.cvp(st.cross_sel) + l * cross_layer, n_audio, S, T, H,
                             w.att.as<half_t>(), D, stream);
        }
        {
            Timed tm(timer, "pre_gemm", stream);
            GemmA A{w.att.as<half_t>(), D};
            GemmEpi e;
            e.bias = b.cout_b; e.residual = x; e.c = x; e.c_f32 = 1; e.ldc = D;
            launch_gemm_f16(A, b.cout_w, M, D, D, e, stream);
        }
        launch_layernorm(x, b.mlp_ln_g, b.mlp_ln_b, M, D, w.xn.as<half_t>(), nullptr, stream);
        {   // + mlp(mlp_ln(x))   (model.rs:315)
            Timed tm(timer, "pre_gemm", stream);
            GemmA A{w.xn.as<half_t>(), D};
            GemmEpi e;
            e.bias = b.mlp0_b; e.act = 1; e.c = w.h.p; e.ldc = 4 * D;
            launch_gemm_f16(A, b.mlp0_w, M, 4 * D, D, e, stream);
        }
        {
            Timed tm(timer, "pre_gemm", stream);
            GemmA A{w.h.as<half_t>(), 4 * (int64_t)D};
            GemmEpi e;
            e.bias = b.mlp2_b; e.residual = x; e.c = x; e.c_f32 = 1; e.ldc = D;
            launch_gemm_f16(A, b.mlp2_w, M, D, 4 * D, e, stream);
        }
        if (l + 1 < L_dec) launch_layernorm(x, dec[l + 1].attn_ln_g, dec[l + 1].attn_ln_b, M, D, w.xn.as<half_t>(), nullptr, stream);
    }
    if (all_logits) {  // stage API: logits of every position (TextDecoder::forward_ext returns [B, S, V])
        launch_layernorm(x, ln_g, ln_b, M, D, w.xn.as<half_t>(), nullptr, stream);
        GemmA A{w.xn.as<half_t>(), D};
        GemmEpi e;
        e.c = w.logits_all.p; e.c_f32 = 1; e.ldc = st.logits_ld;
        launch_gemm_f16(A, emb_p, M, st.logits_ld, D, e, stream);
    }
}

void Model::prefill_logits(DecState& st, int n_audio, int S, int p) {
    WRS_REQUIRE(p >= 0 && p < S && n_audio >= 1 && (int64_t)n_audio * S <= st.prefill.cap_rows, "prefill_logits: bad position %d of %d", p, S);
    launch_gather_ln_packed(st.prefill.x.as<float>(), S, p, ln_g, ln_b, n_audio, D, st.xn.as<half_t>(), stream);
    decoder_logits(st, n_audio);
}

}  // namespace wrs
