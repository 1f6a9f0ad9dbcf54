// Kernel launcher declarations (gfx950 HIP).  All launchers enqueue on the given stream and
// never synchronise.  Shapes are validated on the host before launch (a faulting kernel can
// reset the whole node).
#pragma once
#include "common.h"

namespace wrs {

#ifdef __HIPCC__
// erf-based GELU (reference model.rs:225-232, `gelu("none")`).  erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, i.e.
// fp32 round-off level; the result is stored as fp16): a third of the instructions of libm's erff, which cost the MLP-1
// epilogue more than its k-loop at decode-step sizes (64 values per lane).
// The reciprocal is v_rcp_f32 (1 ulp; the argument is in [1, inf)): `__frcp_rn` expands to the 11-instruction IEEE
// division sequence, a third of the whole function (ISA of the MLP-1 epilogue, round 3).
__device__ __forceinline__ float gelu_exact(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
#ifdef WRS_GELU_IEEE_RCP
    const float t = __frcp_rn(fmaf(0.3275911f, z, 1.f));
#else
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));
#endif
    float poly = fmaf(t, 1.061405429f, -1.453152027f);
    poly = fmaf(t, poly, 1.421413741f);
    poly = fmaf(t, poly, -0.284496736f);
    poly = fmaf(t, poly, 0.254829592f);
    const float erfz = 1.f - poly * t * __expf(-z * z);  // erf(|x| / sqrt 2)
    return 0.5f * x * (1.f + copysignf(erfz, x));
}
#endif

// Load of data the kernel reads ONCE (split-K slabs, the residual stream, logits, K/V rows of one (row, head)): the
// non-temporal hint makes the load bypass the CU's L1, whose miss handling — not the bytes — bounded the gathers of the
// decode self-attention (profiles/r03_self_attention_nontemporal.log).  Never for operands other workgroups of the CU or
// XCD re-read (weights, bias vectors, LayerNorm weights).  -DWRS_NT_STREAMS=0 builds the plain loads for comparison.
#ifndef WRS_NT_STREAMS
#define WRS_NT_STREAMS 1
#endif
#ifdef __HIPCC__
template <typename T>
__device__ __forceinline__ T ld_stream(const T* ptr) {
#if WRS_NT_STREAMS
    return __builtin_nontemporal_load(ptr);
#else
    return *ptr;
#endif
}
#endif

// per-step scalars that live in device memory so that one captured hipGraph serves every step
struct StepParams {
    int cache_pos;       // self-attention K/V slot written by this step (= tokens so far - 1)
    int pos_idx;         // positional-embedding row (reference quirk Q2: loop index)
    int parity;          // ancestry ping-pong: read anc[parity ^ 1], write anc[parity]
    int first_step;      // tokens.size(1) == sample_begin (filters)
    int want_no_speech;  // compute softmax(raw logits)[no_speech]
    int n_rows;          // live rows this step
    int pad0, pad1;
};
// followed in the same allocation by: int tokens[cap]; int src[cap]; int ts_mode[cap];

// ---------------------------------------------------------------- packed activations (decode step)
// Decode-step activations X [M, K] fp16 are kept in MFMA B-operand fragment order
//   [M_pad/16][K/32][64 lanes][8 halfs],  element (m, k) -> lane = ((k & 31) >> 3) * 16 + (m & 15), j = k & 7
// so that a wave fetches a 16x32 fragment as ONE contiguous 1 KiB read (lane-linear 16 B per lane).
static inline __host__ __device__ int64_t pa_off(int m, int k, int ksteps) {
    return ((int64_t)((m >> 4) * ksteps + (k >> 5)) * 64 + ((k & 31) >> 3) * 16 + (m & 15)) * 8 + (k & 7);
}

// ---------------------------------------------------------------- weights packing
// packed MFMA-fragment layout of a [N, K] row-major matrix: [Npad/16][K/32][64 lanes][8 halfs],
// element (nt, ks, l, j) = W[nt*16 + (l & 15)][ks*32 + 8*(l >> 4) + j]; rows >= N are zero.
size_t packed_weight_halfs(int64_t n_pad, int64_t k_pad);
void launch_pack_weight_f32(const float* w, int64_t n, int64_t k, int64_t ldw, half_t* out, int64_t n_pad,
                            int64_t k_pad, hipStream_t s);
// conv weight [N, C, 3] -> [N, 3*C] with k = tap*C + c  (implicit-GEMM order), then packed
void launch_pack_conv_weight_f32(const float* w, int64_t n, int64_t c, half_t* out, int64_t n_pad, int64_t k_pad,
                                 hipStream_t s);
void launch_f32_to_f16(const float* in, half_t* out, int64_t n, hipStream_t s);
void launch_f16_to_f32(const half_t* in, float* out, int64_t n, hipStream_t s);

// ---------------------------------------------------------------- big GEMM (encoder, cross-KV)
struct GemmEpi {
    const float* bias = nullptr;      // [N]
    float scale = 1.f;                // applied to columns n < scale_n_end after bias
    int scale_n_end = 0;
    int act = 0;                      // 1 = exact (erf) GELU
    const float* residual = nullptr;  // fp32, same addressing as C; may alias C
    const float* posemb = nullptr;    // fp32 [c_rows_per_batch, N] added after activation
    void* c = nullptr;
    int c_f32 = 0;
    int64_t ldc = 0;
    int64_t c_rows_per_batch = 1ll << 40;  // C row of m = (m / rpb) * c_batch_rows + m % rpb + c_row_off
    int64_t c_batch_rows = 0;
    int64_t c_row_off = 0;
    int c_packed = 0;                      // fp16 output in packed-activation order (pa_off with K' = N) instead of row-major
    int ksplit = 1;                        // > 1: split-K, raw fp32 partial sums to c + slice * slab_stride (no epilogue math)
    int64_t slab_stride = 0;
    // QKV projection of a decode step (N = 3 d_model): columns [0, D) -> c as fp16 [M, ldc] (the query, bias and scale
    // applied), [D, 2D) / [2D, 3D) -> the new token's key / value rows of the self-attention cache at position sp->cache_pos
    // (row m of the GEMM = physical cache row m); replaces the fp32 slab + the attention kernel's reduction of it
    int qkv_d_model = 0;                   // > 0 selects this epilogue
    half_t* kcache = nullptr;              // [rows][n_ctx][D]
    half_t* vcache = nullptr;
    int64_t cache_row_stride = 0;          // halfs between physical rows (n_ctx * D)
    const StepParams* sp = nullptr;
    // K / V projections of the encoder and of the decoder's cross-attention written STRAIGHT into the packed per-(audio, head)
    // formats of attention.hip (Kp / Vp; row m of the GEMM = key m % pack_T of audio m / pack_T): columns [pack_k_col0,
    // pack_v_col0) are K, [pack_v_col0, N) are V, columns below pack_k_col0 (the encoder's Q) go to c as usual.  Replaces the
    // fp16 [M, N] round trip through launch_pack_kv.  256x256 ping-pong kernel only (gemm_pack_kv_supported).
    half_t* pack_k = nullptr;
    half_t* pack_v = nullptr;
    int pack_T = 0, pack_H = 0;
    int pack_k_col0 = 0, pack_v_col0 = 0;
};
// true when launch_gemm_f16 would run (M, N) on the 256x256 ping-pong kernel, the one that has the pack_k / pack_v epilogue
bool gemm_pack_kv_supported(int64_t M, int64_t N);
struct GemmA {
    const half_t* a = nullptr;
    int64_t lda = 0;                       // halfs between consecutive rows (may be < K: overlapping rows)
    int64_t rows_per_batch = 1ll << 40;    // A row of m = a + (m / rpb) * batch_stride + (m % rpb) * lda
    int64_t batch_stride = 0;
    int packed = 0;                        // 1: A is in packed-activation order (pa_off), K/32 k-steps per row tile
    int64_t packed_rows = 0;               // rows the packed buffer holds (multiple of 16; must cover round_up(M, 128))
};
// C[M, N] = A[M, K] * W[N, K]^T ; K multiple of 64; W packed with n_pad multiple of 128
// kernel variants of launch_gemm_f16 (test hook: gemm_debug_force picks one, gemm_debug_last reports the one taken)
enum GemmKernel { GEMM_AUTO = 0, GEMM_PP2 = 1, GEMM_TILE128 = 2, GEMM_TILE128_PACKED = 3, GEMM_RING = 4 };
void gemm_debug_force(int kernel);  // GEMM_AUTO restores the launcher's own choice; GEMM_TILE128 also stands for the packed-A form
int gemm_debug_last();
void launch_gemm_f16(const GemmA& A, const half_t* w_packed, int64_t M, int64_t N, int64_t K, const GemmEpi& epi,
                     hipStream_t s);

// ---------------------------------------------------------------- persistent ring GEMM (decode step, >= 480 rows; gemm_pring.hip)
enum PringMode { PRING_RESID = 0, PRING_F16 = 1, PRING_GELU_PACKED = 2, PRING_QKV = 3 };
struct PringEpi {
    int mode = PRING_RESID;
    const float* bias = nullptr;   // [N] or null
    float scale = 1.f;             // applied after the bias to columns n < scale_n_end
    int scale_n_end = 0;
    float* x = nullptr;            // RESID: fp32 [M, ldc], x += acc + bias in place (the residual stream)
    half_t* c16 = nullptr;         // F16: fp16 [M, ldc]; GELU_PACKED: packed-activation order (K' = N); QKV: q fp16 [M, ldc]
    int64_t ldc = 0;
    // QKV: N = 3 * d_model; columns [D, 2D) -> kcache, [2D, 3D) -> vcache at [m * phys_mul][sp->cache_pos][.]
    half_t* kcache = nullptr;
    half_t* vcache = nullptr;
    int d_model = 0;
    int64_t cache_row_stride = 0;  // halfs per physical row (= n_ctx * D)
    int phys_mul = 1;
    const StepParams* sp = nullptr;
};
// tile rows bm = 64 or 128 (x 128 columns), no split-K; packed_rows >= round_up(M, bm); W packed with n_pad multiple of 128
void launch_gemm_pring(const half_t* a_packed, int64_t packed_rows, const half_t* w_packed, int64_t M, int64_t N, int64_t K, int bm,
                       const PringEpi& epi, hipStream_t s);
int pring_debug_last_wgs();

// ---------------------------------------------------------------- skinny GEMM (decode, M <= 320)
enum SkinnyKind { SK_F16 = 0, SK_F32 = 1, SK_SLAB = 2, SK_QKV_SELF = 3 };
struct SkinnyEpi {
    int kind = SK_F16;
    const float* bias = nullptr;
    float scale = 1.f;
    int scale_n_end = 0;
    int act = 0;
    void* c = nullptr;   // SK_F16: half [M, N]; SK_F32: float [M, N]; SK_SLAB: float [ksplit][m_pad][N]; SK_QKV_SELF: q half [M, D]
    int64_t ldc = 0;
    int c_packed = 0;    // SK_F16 only: write C in packed-activation order (K' = N) instead of row-major
    // SK_QKV_SELF: N = 3*D; n in [D,2D) -> kcache, [2D,3D) -> vcache at [phys_row][cache_pos][n]
    half_t* kcache = nullptr;
    half_t* vcache = nullptr;
    int64_t d_model = 0;
    int64_t cache_row_stride = 0;  // halfs per physical row (= n_ctx * D)
    int phys_mul = 1;
    const StepParams* sp = nullptr;
};
// C[M, N] = A[M, K] (PACKED activation order, see pa_off) * W^T ; W packed (n_pad multiple of 32)
void launch_gemm_skinny(const half_t* a, const half_t* w_packed, int M, int64_t N, int64_t n_pad, int64_t K,
                        int ksplit, const SkinnyEpi& epi, hipStream_t s);
int skinny_m_pad(int M);
// latency-optimised decode GEMM: slab[ks][m_pad][N] (fp32 partial sums) = A (packed activations) * W^T split over
// ksplit = gemm_dec_plan(...) k-slices; the consumer kernel reduces the slabs in order.
int gemm_dec_plan(int M, int64_t N, int64_t K, int* kps_out, int* mt_block_out);
void launch_gemm_dec(const half_t* a, const half_t* w_packed, int M, int64_t N, int64_t n_pad, int64_t K, float* slab, hipStream_t s);

// decode GEMM for <= 16 rows with the producer fused into its prologue (gemm.hip): mode 0 = split-K finish + residual +
// LayerNorm (x_out = x_in + bias_in + sum slab_in; A = LN(x_out)), mode 1 = split-K finish + bias + GELU.  Returns the slabs written.
int launch_gemm_dec_fused(int mode, const float* slab_in, int ksplit_in, int m_pad_in, const float* bias_in, const float* x_in, float* x_out,
                          const float* ln_g, const float* ln_b, const half_t* w_packed, int M, int64_t N, int64_t n_pad, int64_t K, float* slab,
                          hipStream_t s);

// ---------------------------------------------------------------- f32 GEMM (f32 parity mode)
// mode 2 (f32 parity mode, parity_f32.hip): C[crow(m)][n] = act((acc + bias[n]) * (n < scale_n_end ? scale : 1)) + posemb + residual
struct GemmF32Epi {
    const float* bias = nullptr;
    float scale = 1.f;
    int scale_n_end = 0;
    int act = 0;                       // 1 = exact GELU (erff)
    const float* residual = nullptr;   // addressed like C; may alias C
    const float* posemb = nullptr;     // [c_rpb, N]
    float* c = nullptr;
    int64_t ldc = 0;
    int64_t c_rpb = 1ll << 40, c_brows = 0, c_roff = 0;  // C row of m = (m / c_rpb) * c_brows + m % c_rpb + c_roff
};
// W row-major [round_up(N, 64), ldw] (rows >= N zero); K % 16 == 0
void launch_gemm_f32_epi(const float* a, int64_t lda, int64_t rows_per_batch, int64_t batch_stride, const float* w, int64_t ldw,
                         int64_t M, int64_t N, int64_t K, const GemmF32Epi& epi, hipStream_t s);

// ---------------------------------------------------------------- mel
// fused log-mel (mel_fused.hip): audio [n_batch, n] f32 -> out [n_batch, 80, n/160] f32, reference audio.rs:230-259.
// twp: DFT matrix packed by mel_pack_twiddle; filt [80][201]; krange [80][2] non-zero band per filter row; maxkey [n_batch]
// preset to the key of a very negative float (0x80808080).
size_t mel_twiddle_packed_floats();
void mel_pack_twiddle(float* out);
void launch_mel_fused(const float* audio, int64_t n_batch, int64_t n, const float* twp, const float* filt, const int* krange, float* out,
                      int* maxkey, hipStream_t s);
// mel [B, 80, F] f32 -> encoder input [B, 3002, 80] fp16 (time-major, one zero row each side), frames
// seek..seek+3000, zero (0.0 in log-mel space, reference audio.rs:226) past the end
void launch_mel_prep(const float* mel, int64_t n_batch, int64_t n_frames, int64_t mel_batch_stride, int64_t seek,
                     half_t* out, hipStream_t s);

// ---------------------------------------------------------------- norms / embedding
// y = LN(x) * g + b (eps 1e-5); x fp32 [M, D]; out16 and/or out32 may be null
void launch_layernorm(const float* x, const float* g, const float* b, int64_t M, int D, half_t* out16, float* out32,
                      hipStream_t s);
// xn = LN(x) in PACKED activation order (decode step: the GEMM epilogues update x in place, this only normalises)
void launch_layernorm_packed(const float* x, const float* g, const float* b, int M, int D, half_t* xn, hipStream_t s);
// x[m] += bias + sum_s slab[s][m]; xn = LN(x) in PACKED activation order.  slab [ksplit][m_pad][D]
void launch_finish_residual_ln(float* x, const float* slab, int ksplit, int m_pad, const float* bias, const float* g,
                               const float* b, int M, int D, half_t* xn, hipStream_t s);
// x[r] = emb[token[r]] + pos[pos_idx]; xn = LN(x) in PACKED activation order
// out (PACKED activation order, K' = N) = act(sum_s slab[s][m][:] + bias); slab [ksplit][m_pad][N]; act 1 = exact GELU
void launch_finish_act(const float* slab, int ksplit, int m_pad, const float* bias, int act, int M, int N, half_t* out,
                       hipStream_t s);
void launch_embed_ln(const half_t* emb, const float* pos, const StepParams* sp, const int* tokens, const float* g,
                     const float* b, int M, int D, float* x, half_t* xn, hipStream_t s);

// prompt prefill (all sample_begin tokens in one pass, decode/mod.rs:236-245): rows are (audio, position) = a * S + p
void launch_prefill_embed_ln(const half_t* emb, const float* pos, const int* tokens, int S, int pos0, const float* g, const float* b,
                             int M, int D, float* x, half_t* xn, hipStream_t s);
void launch_prefill_store_kv(const half_t* qkv, int n_audio, int S, int pos0, int D, int n_ctx, int phys_mul, half_t* kc, half_t* vc,
                             hipStream_t s);
void launch_gather_ln_packed(const float* x, int64_t row_stride, int64_t row_off, const float* g, const float* b, int n, int D, half_t* xn,
                             hipStream_t s);

// ---------------------------------------------------------------- attention
// packed K/V for one (batch, head): Kp [Tp/16][2][64][8], Vp [Tp/32][4][64][8] (Tp = round_up(T, 32))
size_t packed_kv_halfs_per_head(int T);
// src [n_batch*T, ld] fp16 row-major; K at column k_off + h*64, V at v_off + h*64
void launch_pack_kv(const half_t* src, int64_t ld, int64_t k_off, int64_t v_off, int n_batch, int T, int H,
                    half_t* kp, half_t* vp, hipStream_t s);
// encoder-style: every 16-query tile of [n_batch, Tq] attends to all T keys. q [n_batch*Tq, ldq] at col h*64
// causal: query i attends to keys 0..i (Tq == T): the decoder's prompt prefill
void launch_attn_full(const half_t* q, int64_t ldq, const half_t* kp, const half_t* vp, int n_batch, int Tq, int T,
                      int H, half_t* out, int64_t ldo, hipStream_t s, bool causal = false);
// decode cross-attention: the G (<=16) rows of one audio share that audio's packed K/V; out in PACKED activation
// order.  Q = (sum_s q_slab[s] + q_bias) * q_scale: the split-K slabs [ksplit][m_pad][H*64] of the query projection.
void launch_attn_cross(const float* q_slab, int q_ksplit, int q_m_pad, const float* q_bias, float q_scale, const half_t* kp,
                       const half_t* vp, int n_audio, int G, int T, int H, half_t* out, hipStream_t s);
// decode self-attention over the ancestry-indirected K/V cache; one wave per (row, head); out in PACKED activation
// order.  q/k/v of the new token come from the QKV projection's split-K slabs [ksplit][m_pad][3D]; k/v are appended
// to kc/vc [phys_row][n_ctx][D] at sp->cache_pos.
void launch_attn_self(const float* qkv_slab, int ksplit, int m_pad, const float* qkv_bias, float scale, half_t* kc, half_t* vc,
                      const int* anc_base, int64_t anc_buf_stride, int n_ctx, int n_rows, int rows_per_audio, int phys_mul, int H, int D,
                      const StepParams* sp, half_t* out, hipStream_t s);
// same with the new token's q given as fp16 [n_rows, D] (bias and scale applied) and its k / v already appended to the
// cache at sp->cache_pos by the QKV GEMM's epilogue (PRING_QKV)
void launch_attn_self_q16(const half_t* q16, half_t* kc, half_t* vc, const int* anc_base, int64_t anc_buf_stride, int n_ctx, int n_rows,
                          int rows_per_audio, int phys_mul, int H, int D, const StepParams* sp, half_t* out, hipStream_t s);
// cross-attention with the query projection given as fp16 [n_audio * G, D] (bias and scale applied)
void launch_attn_cross_q16(const half_t* q16, const half_t* kp, const half_t* vp, int n_audio, int G, int T, int H, half_t* out, hipStream_t s);
// beam-de-duplicating form on the matrix cores (attn_self_mfma.hip): one wave per (audio, head) over the distinct
// (position, physical row) pairs of the G beams' histories.  launch_self_keys builds that list once per step (it depends on
// the ancestry table only): klist [n_audio][klist_stride >= G * n_ctx], kcount [n_audio]; phys_mul must be 1.
// include_new: the list also holds the step's own position (one key per beam, owned by that beam): for the form of the
// attention kernel that finds q as fp16 and the new k / v already in the cache (launch_attn_self_mfma_q16)
void launch_self_keys(const int* anc_base, int64_t anc_buf_stride, int n_ctx, int n_audio, int G, const StepParams* sp, int* klist,
                      int klist_stride, int* kcount, bool include_new, hipStream_t s);
void launch_attn_self_mfma_q16(const half_t* q16, half_t* kc, half_t* vc, int n_ctx, int n_rows, int G, int H, int D, const StepParams* sp,
                               const int* klist, int klist_stride, const int* kcount, half_t* out, hipStream_t s);
void launch_attn_self_mfma(const float* qkv_slab, int ksplit, int m_pad, const float* qkv_bias, float scale, half_t* kc, half_t* vc, int n_ctx,
                           int n_rows, int G, int H, int D, const StepParams* sp, const int* klist, int klist_stride, const int* kcount,
                           half_t* out, hipStream_t s);
void launch_anc_init(int* anc_base, int64_t anc_buf_stride, int n_ctx, int n_phys_rows, int G, hipStream_t s);
void launch_anc_update(int* anc_base, int64_t anc_buf_stride, int n_ctx, int n_rows, const StepParams* sp,
                       const int* src, hipStream_t s);

// ---------------------------------------------------------------- logits: filters + log-softmax + top-k
struct FilterCfg {
    int n_vocab;
    int ld = 0;                 // floats between logits rows (0: n_vocab)
    int ts_begin, eot, no_timestamps, no_speech, space;
    int suppress_blank;         // SuppressBlank enabled
    int timestamps;             // TimestampTokens enabled
    int max_initial_ts_index;   // -1: None
    const int* suppress_ids;    // device
    int n_suppress;
    // optional workspace (device, filter_ws_floats(rows) floats): with it and few rows (<= FILTER_SPLIT_MAX_ROWS) the vocabulary
    // of every row is split over FILTER_SPLIT workgroups + one merge wave per row (one workgroup per row leaves a 5-row step on 5 CUs)
    float* ws = nullptr;
    int ws_rows = 0;
};
constexpr int FILTER_SPLIT = 16, FILTER_SPLIT_MAX_ROWS = 64, FILTER_WS_STRIDE = 40;
inline size_t filter_ws_floats(int rows) { return (size_t)rows * FILTER_SPLIT * FILTER_WS_STRIDE; }
// logits [n_rows, V] f32 device.  ts_mode[r]: 0 none, 1 mask [ts_begin, V), 2 mask [0, eot)   (timestamp_tokens.rs:47-66)
// out_lp/out_id [n_rows, k] (device-visible), no_speech [n_rows] (written when sp->want_no_speech)
void launch_filter_topk(const float* logits, int n_rows, const FilterCfg& cfg, const StepParams* sp, const int* ts_mode,
                        int k, float* out_lp, int* out_id, float* no_speech, float* out_filtered, hipStream_t s);

// softmax over logits[r, first_id : first_id + n_lang] -> out_probs [n_rows, n_lang] (optional), out_id[r] = argmax id
void launch_lang_detect(const float* logits, int n_rows, int ld, int first_id, int n_lang, int* out_id, float* out_probs, hipStream_t s);

}  // namespace wrs
