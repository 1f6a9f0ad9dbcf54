// f32 PARITY MODE (wrs_debug_model_set_parity_f32, test build only): the same hot path with every activation, weight,
// K/V cache entry and product in fp32 (v_mfma_f32_16x16x4_f32 GEMMs, fp32 attention).  It exists to DEMONSTRATE the
// north-star's "token-id outputs bit-exact": in this mode the free-running beam search must reproduce the fp32 CPU oracle's
// token ids and source indices to the last step in every golden case (tests/test_gpu_parity_f32.py), which shows that
// the near-tie splits of the fp16 product path come from fp16 rounding and from nothing else.  Not a product path and not
// fast (simple kernels, no tiling effort): small models / batches only.
//
// Same structure as model.cpp (reference src/model.rs:158-206,301-319,417-441,490-522); the decode loop, filters,
// beam search and ancestry table above it are the product code, unchanged.
#include "model.h"
#include <cmath>

namespace wrs {

typedef float floatx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum32(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max32(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// ------------------------------------------------------------------------------------------ layout helpers
// conv weight [N, C, 3] -> [N, 3*C] with k = tap*C + c (the implicit-GEMM order of the fp16 path), fp32
__global__ void conv_weight_f32_kernel(const float* __restrict__ w, int64_t n, int64_t c, float* __restrict__ out) {
    const int64_t total = n * 3 * c;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / (3 * c), kk = i % (3 * c);
        const int64_t tap = kk / c, cc = kk % c;
        out[i] = w[(row * c + cc) * 3 + tap];
    }
}

// mel [B, 80, F] f32 -> [B, 3002, 80] f32 time-major, rows 0 and 3001 zero, frames seek..seek+3000 (zero past the end)
__global__ void mel_prep_f32_kernel(const float* __restrict__ mel, int64_t n_frames, int64_t mel_bs, int64_t seek, float* __restrict__ out) {
    const int64_t b = blockIdx.y;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < 3002 * 80; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = i / 80 - 1;
        const int c = (int)(i % 80);
        float v = 0.f;
        if (t >= 0 && t < 3000 && seek + t < n_frames) v = mel[b * mel_bs + (int64_t)c * n_frames + seek + t];
        out[b * 3002 * 80 + i] = v;
    }
}

// x[r] = emb[token[r]] + pos[pos_idx]  (reference model.rs:503-506), fp32
__global__ void embed_f32_kernel(const float* __restrict__ emb, const float* __restrict__ pos, const StepParams* __restrict__ sp,
                                 const int* __restrict__ tokens, int D, float* __restrict__ x) {
    const int row = blockIdx.x;
    const int tok = tokens[row], pidx = sp->pos_idx;
    for (int c = threadIdx.x; c < D; c += blockDim.x) x[(int64_t)row * D + c] = emb[(int64_t)tok * D + c] + pos[(int64_t)pidx * D + c];
}

// append the new token's k / v (columns D..3D of qkv [R, 3D]) to the fp32 cache [phys_row][n_ctx][D] at sp->cache_pos
__global__ void append_kv_f32_kernel(const float* __restrict__ qkv, int D, int n_ctx, int phys_mul, const StepParams* __restrict__ sp,
                                     float* __restrict__ kc, float* __restrict__ vc) {
    const int row = blockIdx.x;
    const int64_t dst = ((int64_t)row * phys_mul * n_ctx + sp->cache_pos) * D;
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        kc[dst + c] = qkv[(int64_t)row * 3 * D + D + c];
        vc[dst + c] = qkv[(int64_t)row * 3 * D + 2 * D + c];
    }
}

// ------------------------------------------------------------------------------------------ attention, fp32
// One wave per (query row, head), head_dim 64.  Scores of all keys go to LDS (n_keys <= 1536), two-pass softmax in fp32
// (reference model.rs:191-201), then out[d] = sum_j p_j v[j][d] with lanes over d.
struct AttnF32Params {
    const float* q;      // [n_q, ldq], head h at column h*64
    int64_t ldq;
    const float* k;      // key j of query row r: k + kv_row(r, j) * kv_row_stride + j * ldk + h*64 (+ k_col_off)
    const float* v;
    int64_t ldk;         // floats between consecutive key positions
    int64_t kv_row_stride;
    int k_col_off, v_col_off;
    int n_q, H;
    int n_keys;          // fixed key count, or -1: sp->cache_pos + 1 (decode self-attention)
    int q_per_kv;        // kv row of query r = r / q_per_kv (cross-attention: beams of an audio; encoder: queries of a window)
    const int* anc_base; // decode self-attention: physical row of (row, position), or null
    int64_t anc_buf_stride;
    int n_ctx, phys_mul;
    const StepParams* sp;
    float* out;          // [n_q, ldo]
    int64_t ldo;
};

__global__ __launch_bounds__(256) void attn_f32_kernel(const AttnF32Params p) {
    __shared__ float sc[4][1536];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t unit = (int64_t)blockIdx.x * 4 + wave;
    if (unit >= (int64_t)p.n_q * p.H) return;  // no barriers below
    const int r = (int)(unit / p.H), h = (int)(unit % p.H);
    const int n_keys = p.n_keys >= 0 ? p.n_keys : p.sp->cache_pos + 1;
    const int* anc = p.anc_base ? p.anc_base + (int64_t)p.sp->parity * p.anc_buf_stride + (int64_t)r * p.phys_mul * p.n_ctx : nullptr;
    const int cpos = p.anc_base ? p.sp->cache_pos : 0;
    const float* qp = p.q + (int64_t)r * p.ldq + h * 64;
    float qv[64];
#pragma unroll
    for (int d = 0; d < 64; d += 4) {
        const floatx4 t = *reinterpret_cast<const floatx4*>(qp + d);
        qv[d] = t[0]; qv[d + 1] = t[1]; qv[d + 2] = t[2]; qv[d + 3] = t[3];
    }
    auto kv_row = [&](int j) -> int64_t {
        if (anc) return j == cpos ? (int64_t)r * p.phys_mul : (int64_t)anc[j];
        return (int64_t)(r / p.q_per_kv);
    };
    float mx = -INFINITY;
    for (int j = lane; j < n_keys; j += 64) {
        const float* kp = p.k + kv_row(j) * p.kv_row_stride + (int64_t)j * p.ldk + p.k_col_off + h * 64;
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < 64; d += 4) {
            const floatx4 t = *reinterpret_cast<const floatx4*>(kp + d);
            s += qv[d] * t[0] + qv[d + 1] * t[1] + qv[d + 2] * t[2] + qv[d + 3] * t[3];
        }
        sc[wave][j] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max32(mx);
    float su = 0.f;
    for (int j = lane; j < n_keys; j += 64) {
        const float e = expf(sc[wave][j] - mx);
        sc[wave][j] = e;
        su += e;
    }
    su = wave_sum32(su);
    __builtin_amdgcn_wave_barrier();
    float acc = 0.f;  // lane = output dim
    for (int j = 0; j < n_keys; ++j) {
        const float* vp = p.v + kv_row(j) * p.kv_row_stride + (int64_t)j * p.ldk + p.v_col_off + h * 64;
        acc += sc[wave][j] * vp[lane];
    }
    p.out[(int64_t)r * p.ldo + h * 64 + lane] = acc / su;
}

static void launch_attn_f32(const AttnF32Params& p, hipStream_t s) {
    WRS_REQUIRE(p.n_keys <= 1536 && p.n_ctx <= 1536, "attn_f32: at most 1536 keys");
    const int64_t units = (int64_t)p.n_q * p.H;
    attn_f32_kernel<<<(unsigned)cdiv(units, 4), 256, 0, s>>>(p);
}

// ------------------------------------------------------------------------------------------ weights
static float* dev_copy_f32(Model& m, const float* src, int64_t n, int64_t n_alloc = 0) {
    DevBuf b;
    b.alloc((size_t)std::max(n, n_alloc) * 4);  // zero-initialised: padded rows stay zero
    HIP_CHECK(hipMemcpyAsync(b.p, src, (size_t)n * 4, hipMemcpyDeviceToDevice, m.stream));
    float* p = b.as<float>();
    m.owned.push_back(std::move(b));
    return p;
}
static float* dev_concat_f32(Model& m, std::vector<const float*> parts, int64_t each) {
    DevBuf b;
    b.alloc(parts.size() * (size_t)each * 4);
    for (size_t i = 0; i < parts.size(); ++i)
        if (parts[i]) HIP_CHECK(hipMemcpyAsync(b.as<float>() + i * each, parts[i], (size_t)each * 4, hipMemcpyDeviceToDevice, m.stream));
    float* p = b.as<float>();
    m.owned.push_back(std::move(b));
    return p;
}

void build_block_f32(Model& m, BlockW& bw, const std::string& pre, int64_t d, bool cross, const std::map<std::string, DevBuf>& staging) {
    auto t = [&](const char* s) -> const float* {
        auto it = staging.find(pre + s);
        if (it == staging.end()) WRS_THROW(WRS_ESTATE, "finalize: tensor '%s%s' was never set", pre.c_str(), s);
        return it->second.as<float>();
    };
    bw.qkv_w32 = dev_concat_f32(m, {t(".attn.query.weight"), t(".attn.key.weight"), t(".attn.value.weight")}, d * d);
    bw.out_w32 = dev_copy_f32(m, t(".attn.out.weight"), d * d);
    if (cross) {
        bw.cq_w32 = dev_copy_f32(m, t(".cross_attn.query.weight"), d * d);
        bw.ckv_w32 = dev_concat_f32(m, {t(".cross_attn.key.weight"), t(".cross_attn.value.weight")}, d * d);
        bw.cout_w32 = dev_copy_f32(m, t(".cross_attn.out.weight"), d * d);
    }
    bw.mlp0_w32 = dev_copy_f32(m, t(".mlp.0.weight"), 4 * d * d);
    bw.mlp2_w32 = dev_copy_f32(m, t(".mlp.2.weight"), 4 * d * d);
    HIP_CHECK(hipStreamSynchronize(m.stream));
}

void Model::finalize_f32() {
    auto need = [&](const char* name) -> const float* {
        auto it = staging.find(name);
        if (it == staging.end()) WRS_THROW(WRS_ESTATE, "finalize: tensor '%s' was never set", name);
        return it->second.as<float>();
    };
    {
        DevBuf b1, b2;
        b1.alloc((size_t)D * 240 * 4);
        b2.alloc((size_t)D * 3 * D * 4);
        conv_weight_f32_kernel<<<256, 256, 0, stream>>>(need("encoder.conv1.weight"), D, WRS_N_MELS, b1.as<float>());
        conv_weight_f32_kernel<<<1024, 256, 0, stream>>>(need("encoder.conv2.weight"), D, D, b2.as<float>());
        conv1_w32 = b1.as<float>();
        conv2_w32 = b2.as<float>();
        owned.push_back(std::move(b1));
        owned.push_back(std::move(b2));
    }
    emb32 = dev_copy_f32(*this, need("decoder.token_embedding.weight"), (int64_t)V * D, round_up(V, 64) * D);
    HIP_CHECK(hipStreamSynchronize(stream));
}

// ------------------------------------------------------------------------------------------ GEMM wrapper
static void gemm32(const Model& m, const float* a, int64_t lda, const float* w, int64_t M, int64_t N, int64_t K, const GemmF32Epi& e,
                   int64_t rpb = 1ll << 40, int64_t bs = 0) {
    launch_gemm_f32_epi(a, lda, rpb, bs, w, K, M, N, K, e, m.stream);
}

// ------------------------------------------------------------------------------------------ encoder
void Model::ensure_encoder_ws_f32(int n) {
    EncoderWorkspaceF32& w = enc_ws32;
    if (n <= w.cap) return;
    const int64_t M = (int64_t)n * T;
    w.mel_t.alloc(((size_t)n * 3002 * 80 + 256) * 4);
    w.c1.alloc(((size_t)n * 3001 * D + 4 * D) * 4);
    w.xn.alloc((size_t)M * D * 4);
    w.qkv.alloc((size_t)M * 3 * D * 4);
    w.att.alloc((size_t)M * D * 4);
    w.h.alloc((size_t)M * 4 * D * 4);
    w.cap = n;
}

void Model::load_mel_f32(const float* mel_dev, int n, int64_t n_frames, int64_t mel_bs, int64_t seek, int slot) {
    ensure_encoder_ws(slot + n);
    ensure_encoder_ws_f32(enc_ws.cap);
    dim3 grid(64, (unsigned)n);
    mel_prep_f32_kernel<<<grid, 256, 0, stream>>>(mel_dev, n_frames, mel_bs, seek, enc_ws32.mel_t.as<float>() + (size_t)slot * 3002 * 80);
}

void Model::encoder_f32(int n) {
    WRS_REQUIRE(finalized && parity_f32, "encoder_f32: model is not in f32 parity mode");
    WRS_REQUIRE(n >= 1 && n <= enc_ws32.cap && n <= enc_ws.cap, "encoder: bad window count %d", n);
    EncoderWorkspaceF32& w = enc_ws32;
    enc_ws.n_encoded = n;
    ++enc_ws.epoch;
    const int64_t M = (int64_t)n * T;
    const float scale = std::pow((float)(D / H), -0.25f);
    float* x = enc_ws.x.as<float>();
    {   // conv1 + GELU (model.rs:421): rows of c1 are [n][3001][D], row 0 of a window stays zero (conv2's left padding)
        GemmF32Epi e;
        e.bias = conv1_b; e.act = 1; e.c = w.c1.as<float>(); e.ldc = D; e.c_rpb = 3000; e.c_brows = 3001; e.c_roff = 1;
        gemm32(*this, w.mel_t.as<float>(), 80, conv1_w32, (int64_t)n * 3000, D, 240, e, 3000, 3002 * 80);
    }
    {   // conv2 (stride 2) + GELU + positional embedding (model.rs:422-431)
        GemmF32Epi e;
        e.bias = conv2_b; e.act = 1; e.posemb = enc_pos; e.c = x; e.ldc = D; e.c_rpb = T; e.c_brows = T;
        gemm32(*this, w.c1.as<float>(), 2 * (int64_t)D, conv2_w32, M, D, 3 * (int64_t)D, e, T, 3001 * (int64_t)D);
    }
    for (int l = 0; l < L_enc; ++l) {
        const BlockW& b = enc[l];
        launch_layernorm(x, b.attn_ln_g, b.attn_ln_b, M, D, nullptr, w.xn.as<float>(), stream);
        {
            GemmF32Epi e;
            e.bias = b.qkv_b; e.scale = scale; e.scale_n_end = 2 * D; e.c = w.qkv.as<float>(); e.ldc = 3 * D;
            gemm32(*this, w.xn.as<float>(), D, b.qkv_w32, M, 3 * D, D, e);
        }
        {
            AttnF32Params p{};
            p.q = w.qkv.as<float>(); p.ldq = 3 * D; p.k = p.v = w.qkv.as<float>(); p.ldk = 3 * D; p.kv_row_stride = (int64_t)T * 3 * D;
            p.k_col_off = D; p.v_col_off = 2 * D; p.n_q = (int)M; p.H = H; p.n_keys = T; p.q_per_kv = T; p.n_ctx = T; p.phys_mul = 1;
            p.out = w.att.as<float>(); p.ldo = D;
            launch_attn_f32(p, stream);
        }
        {
            GemmF32Epi e;
            e.bias = b.out_b; e.residual = x; e.c = x; e.ldc = D;
            gemm32(*this, w.att.as<float>(), D, b.out_w32, M, D, D, e);
        }
        launch_layernorm(x, b.mlp_ln_g, b.mlp_ln_b, M, D, nullptr, w.xn.as<float>(), stream);
        {
            GemmF32Epi e;
            e.bias = b.mlp0_b; e.act = 1; e.c = w.h.as<float>(); e.ldc = 4 * D;
            gemm32(*this, w.xn.as<float>(), D, b.mlp0_w32, M, 4 * D, D, e);
        }
        {
            GemmF32Epi e;
            e.bias = b.mlp2_b; e.residual = x; e.c = x; e.ldc = D;
            gemm32(*this, w.h.as<float>(), 4 * (int64_t)D, b.mlp2_w32, M, D, 4 * (int64_t)D, e);
        }
    }
    launch_layernorm(x, ln_post_g, ln_post_b, M, D, enc_ws.xa16.as<half_t>(), enc_ws.xa32.as<float>(), stream);
}

// ------------------------------------------------------------------------------------------ decoder
void Model::init_state_f32(DecState& st) {
    st.f32.reset(new DecStateF32());
    DecStateF32& f = *st.f32;
    f.cross.alloc((size_t)L_dec * st.a_cap * T * 2 * D * 4);
    f.kc.alloc((size_t)L_dec * st.r_cap * st.ctx * D * 4);
    f.vc.alloc((size_t)L_dec * st.r_cap * st.ctx * D * 4);
    f.xn.alloc((size_t)st.r_cap * D * 4);
    f.qkv.alloc((size_t)st.r_cap * 3 * D * 4);
    f.att.alloc((size_t)st.r_cap * D * 4);
    f.h.alloc((size_t)st.r_cap * 4 * D * 4);
}

void Model::load_cross_kv_f32(DecState& st, int n_audio) {
    WRS_REQUIRE(st.f32 && n_audio >= 1 && n_audio <= st.a_cap, "cross K/V (f32): %d audios exceed capacity %d", n_audio, st.a_cap);
    const float scale = std::pow((float)(D / H), -0.25f);
    const size_t per_layer = (size_t)st.a_cap * T * 2 * D;
    for (int l = 0; l < L_dec; ++l) {
        GemmF32Epi e;
        e.bias = dec[l].ckv_b; e.scale = scale; e.scale_n_end = D; e.c = st.f32->cross.as<float>() + l * per_layer; e.ldc = 2 * D;
        gemm32(*this, enc_ws.xa32.as<float>(), D, dec[l].ckv_w32, (int64_t)n_audio * T, 2 * D, D, e);
    }
    st.n_audio = n_audio;
}

void Model::decoder_step_f32(DecState& st, int rows, int phys_mul, int q_per_audio, bool update_anc, bool want_logits) {
    WRS_REQUIRE(st.f32, "decoder_step_f32: state has no f32 buffers");
    DecStateF32& f = *st.f32;
    const float scale = std::pow((float)(D / H), -0.25f);
    const StepParams* sp = st.sp_d();
    const int64_t anc_stride = (int64_t)st.r_cap * st.ctx;
    const size_t kv_layer = (size_t)st.r_cap * st.ctx * D;
    const size_t cross_layer = (size_t)st.a_cap * T * 2 * D;
    float* x = st.x.as<float>();
    if (update_anc) launch_anc_update(st.anc.as<int>(), anc_stride, st.ctx, rows, sp, st.src_d(), stream);
    embed_f32_kernel<<<rows, 256, 0, stream>>>(emb32, dec_pos, sp, st.tokens_d(), D, x);
    for (int l = 0; l < L_dec; ++l) {
        const BlockW& b = dec[l];
        launch_layernorm(x, b.attn_ln_g, b.attn_ln_b, rows, D, nullptr, f.xn.as<float>(), stream);
        {
            GemmF32Epi e;
            e.bias = b.qkv_b; e.scale = scale; e.scale_n_end = 2 * D; e.c = f.qkv.as<float>(); e.ldc = 3 * D;
            gemm32(*this, f.xn.as<float>(), D, b.qkv_w32, rows, 3 * D, D, e);
        }
        float* kc = f.kc.as<float>() + l * kv_layer;
        float* vc = f.vc.as<float>() + l * kv_layer;
        append_kv_f32_kernel<<<rows, 256, 0, stream>>>(f.qkv.as<float>(), D, st.ctx, phys_mul, sp, kc, vc);
        {
            AttnF32Params p{};
            p.q = f.qkv.as<float>(); p.ldq = 3 * D; p.k = kc; p.v = vc; p.ldk = D; p.kv_row_stride = (int64_t)st.ctx * D;
            p.n_q = rows; p.H = H; p.n_keys = -1; p.q_per_kv = 1; p.anc_base = st.anc.as<int>(); p.anc_buf_stride = anc_stride;
            p.n_ctx = st.ctx; p.phys_mul = phys_mul; p.sp = sp; p.out = f.att.as<float>(); p.ldo = D;
            launch_attn_f32(p, stream);
        }
        {
            GemmF32Epi e;
            e.bias = b.out_b; e.residual = x; e.c = x; e.ldc = D;
            gemm32(*this, f.att.as<float>(), D, b.out_w32, rows, D, D, e);
        }
        launch_layernorm(x, b.cross_ln_g, b.cross_ln_b, rows, D, nullptr, f.xn.as<float>(), stream);
        {
            GemmF32Epi e;
            e.bias = b.cq_b; e.scale = scale; e.scale_n_end = D; e.c = f.qkv.as<float>(); e.ldc = D;
            gemm32(*this, f.xn.as<float>(), D, b.cq_w32, rows, D, D, e);
        }
        {
            AttnF32Params p{};
            const float* ckv = f.cross.as<float>() + l * cross_layer;
            p.q = f.qkv.as<float>(); p.ldq = D; p.k = p.v = ckv; p.ldk = 2 * D; p.kv_row_stride = (int64_t)T * 2 * D;
            p.k_col_off = 0; p.v_col_off = D; p.n_q = rows; p.H = H; p.n_keys = T; p.q_per_kv = q_per_audio; p.n_ctx = T; p.phys_mul = 1;
            p.out = f.att.as<float>(); p.ldo = D;
            launch_attn_f32(p, stream);
        }
        {
            GemmF32Epi e;
            e.bias = b.cout_b; e.residual = x; e.c = x; e.ldc = D;
            gemm32(*this, f.att.as<float>(), D, b.cout_w32, rows, D, D, e);
        }
        launch_layernorm(x, b.mlp_ln_g, b.mlp_ln_b, rows, D, nullptr, f.xn.as<float>(), stream);
        {
            GemmF32Epi e;
            e.bias = b.mlp0_b; e.act = 1; e.c = f.h.as<float>(); e.ldc = 4 * D;
            gemm32(*this, f.xn.as<float>(), D, b.mlp0_w32, rows, 4 * D, D, e);
        }
        {
            GemmF32Epi e;
            e.bias = b.mlp2_b; e.residual = x; e.c = x; e.ldc = D;
            gemm32(*this, f.h.as<float>(), 4 * (int64_t)D, b.mlp2_w32, rows, D, 4 * (int64_t)D, e);
        }
    }
    if (want_logits) {
        launch_layernorm(x, ln_g, ln_b, rows, D, nullptr, f.xn.as<float>(), stream);
        GemmF32Epi e;
        e.c = st.logits.as<float>(); e.ldc = st.logits_ld;
        gemm32(*this, f.xn.as<float>(), D, emb32, rows, V, D, e);
    }
}

}  // namespace wrs
