/*
 * wrs_debug.h — test hooks.  NOT part of the drop-in boundary (include/wrs.h) and NOT exported by the product library
 * libwrs_mi355x.so: they are linked into libwrs_mi355x_debug.so only (the product's objects + csrc/debug_api.cpp), the build
 * tests/ and __graft_entry__.smoke() load; that build also honours the WRS_* experiment switches (csrc/common.h: exp_env),
 * which the product library ignores.  Nothing here has a counterpart in laptou/whisper-rs.  They exist so that tests/ can (a) drive single kernels of the hot path on caller
 * data at the benchmark's shapes and compare them with an independent fp32 product, (b) make the device follow the
 * oracle's beam decisions (teacher forcing / re-synchronisation after a near-tie), and (c) exercise the host sequencer
 * and the long-form bookkeeping without a device.
 */
#ifndef WRS_DEBUG_H
#define WRS_DEBUG_H

#include "wrs.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- kernel-level hooks (device pointers) */

/* kernel variants of the tiled fp16 GEMM launcher (csrc/gemm.hip) */
#define WRS_GEMM_AUTO 0            /* the launcher's own choice (what the model / bench run) */
#define WRS_GEMM_PP2 1             /* gemm_f16_pp2_kernel, 256x256x32 ping-pong (row-major A, N % 256 == 0) */
#define WRS_GEMM_TILE128 2         /* gemm_f16_kernel<false>, 128x128x64 (row-major A) */
#define WRS_GEMM_TILE128_PACKED 3  /* gemm_f16_kernel<true> (packed A) */
#define WRS_GEMM_RING 4            /* gemm_f16_ring_kernel<4,8> (packed A) */
/* paths other than the tiled launcher */
#define WRS_GEMM_PATH_TILED 0      /* launch_gemm_f16 */
#define WRS_GEMM_PATH_DEC 1        /* launch_gemm_dec: decode GEMM for < 480 rows, fp32 split-K slabs */
#define WRS_GEMM_PATH_SKINNY_F32 2 /* launch_gemm_skinny, fp32 row-major output (logits at < 480 rows) */
#define WRS_GEMM_PATH_PRING 3      /* launch_gemm_pring: persistent ring GEMM of the decode step at >= 480 rows (csrc/gemm_pring.hip) */
/* epilogue modes of the persistent ring GEMM */
#define WRS_PRING_RESID 0          /* x[m][n] += acc + bias, fp32 in place */
#define WRS_PRING_F16 1            /* c = fp16((acc + bias) * scale) row-major */
#define WRS_PRING_GELU_PACKED 2    /* c = fp16(gelu(acc + bias)) in packed-activation order */
#define WRS_PRING_QKV 3            /* q -> c (fp16 [M, D]), k / v -> caches at cache_pos */

/* C[M, N] = epilogue(A[M, K] * W[N, K]^T) through the library's own weight packer and GEMM launchers.
 * All pointers are DEVICE pointers on device `device_id`; the call synchronises before returning. */
typedef struct wrs_debug_gemm {
    int32_t device_id;
    int32_t path;            /* WRS_GEMM_PATH_* */
    int32_t force_kernel;    /* WRS_GEMM_* (tiled path) */
    int32_t kernel_used;     /* out: WRS_GEMM_* the launcher took */
    int64_t M, N, K;
    /* A: fp16.  a_packed = 0: row-major, row m at a + (m / rows_per_batch) * batch_stride + (m % rows_per_batch) * lda
     * (lda < K gives overlapping rows: conv1d as implicit GEMM).  a_packed = 1: packed-activation order
     * [packed_rows/16][K/32][64][8] (csrc/kernels.h pa_off), packed_rows a multiple of 128 covering M. */
    const void* a;
    int32_t a_packed;
    int64_t lda, rows_per_batch, batch_stride, packed_rows;
    /* W: fp32 row-major [N_w, K_w]; conv_c > 0: a conv1d weight [N_w, conv_c, 3] (K_w = 3 * conv_c).  Packed by the
     * library exactly as Model::finalize packs it (rows padded to 128, K to 64; N may exceed N_w: zero rows). */
    const float* w;
    int64_t n_w, k_w;
    int32_t conv_c;
    /* epilogue */
    const float* bias;       /* [N] or NULL */
    float scale;
    int32_t scale_n_end;
    int32_t act;             /* 1 = exact GELU */
    const float* residual;   /* fp32, addressed like C, or NULL */
    const float* posemb;     /* fp32 [c_rows_per_batch, N] or NULL */
    void* c;                 /* fp32 (c_f32) or fp16; packed-activation order if c_packed */
    int32_t c_f32, c_packed;
    int64_t ldc, c_rows_per_batch, c_batch_rows, c_row_off;
    int32_t ksplit;          /* tiled path: > 1 = split-K, raw fp32 partial sums [ksplit][slab_stride]; path DEC: out = slabs used */
    int64_t slab_stride;     /* tiled path in; path DEC: out = m_pad * N */
    /* path PRING: epilogue mode, tile rows (64 / 128); RESID updates `c` (fp32 [M, ldc]) in place; QKV writes q to `c`
     * (fp16 [M, ldc], ldc >= N/3) and k / v to kcache / vcache (fp16 [M * phys_mul][n_ctx][N/3]) at position cache_pos.
     * Path TILED with kcache != NULL: the tiled kernel's QKV-to-cache epilogue (same outputs, phys_mul 1, c fp16 with ldc = N/3) */
    int32_t pring_mode, pring_bm;
    void* kcache;
    void* vcache;
    int32_t n_ctx, cache_pos, phys_mul;
    int32_t pring_workgroups; /* out */
    /* path TILED with pack_k != NULL: the K / V projection epilogue of the 256x256 kernel — columns [pack_k_col0, pack_v_col0) go to
     * pack_k and [pack_v_col0, N) to pack_v in the packed per-(audio, head) attention formats (csrc/attention.hip header; row m = key
     * m % pack_T of audio m / pack_T; buffers of (M / pack_T) * pack_H * round_up(pack_T, 32) * 64 halfs, zero-filled by the caller),
     * columns below pack_k_col0 to c (fp16 [M, ldc]) */
    void* pack_k;
    void* pack_v;
    int32_t pack_T, pack_H, pack_k_col0, pack_v_col0;
} wrs_debug_gemm;
int wrs_debug_gemm_f16(wrs_debug_gemm* g);

/* decode self-attention kernel on caller data (csrc/attention.hip attn_self): see tests/test_gpu_attention.py */
typedef struct wrs_debug_attn_self {
    int32_t device_id;
    int32_t n_rows, rows_per_audio, H, D, n_ctx, cache_pos;
    const float* qkv;        /* [n_rows, 3D] fp32: the un-biased QKV projection of the new token */
    const float* qkv_bias;   /* [3D] */
    float scale;
    void* kc;                /* fp16 [n_rows][n_ctx][D] in/out (position cache_pos is written) */
    void* vc;
    const int32_t* anc;      /* [n_rows][n_ctx]: physical row holding (row, position) */
    void* out_packed;        /* fp16 packed-activation order, round_up(n_rows,128) x D */
    int32_t q16_mode;        /* 1: the >= 480-row form of the kernel (q as fp16, new k / v already in the cache);
                              * 2: the beam-de-duplicating MFMA kernel (attn_self_mfma.hip; rows_per_audio <= 8);
                              * 3: the same kernel in its >= 480-row form (q as fp16, new k / v in the cache and in the key list) */
} wrs_debug_attn_self;
int wrs_debug_attn_self_run(wrs_debug_attn_self* a);

/* decode cross-attention kernel on caller data: q [n_audio*G, D] fp32 (un-biased projection), K/V row-major fp16
 * [n_audio, T, D] (already scaled / biased as load_cross_kv leaves them), packed by the library's pack_kv. */
typedef struct wrs_debug_attn_cross {
    int32_t device_id;
    int32_t n_audio, G, H, D, T;
    const float* q;
    const float* q_bias;
    float q_scale;
    const void* k;
    const void* v;
    void* out_packed;
    int32_t q16_mode;        /* 1: the >= 480-row form (query projection given as fp16, bias and scale applied) */
} wrs_debug_attn_cross;
int wrs_debug_attn_cross_run(wrs_debug_attn_cross* a);

/* ---------------------------------------------------------------- decode-level hooks */

/* teacher-forced decode of mel [n_audio,80,3000] (host): step s applies forced tokens / sources [s][rows]; the device's
 * candidates for every step are returned ([n_steps][rows][k]). */
int wrs_debug_decode_forced(wrs_decode_task* t, const float* mel, int n_audio, int n_steps, const int32_t* ftok, const int32_t* fsrc,
                            float* out_lp, int32_t* out_id, double* out_no_speech);
/* same, starting from encoder output [n_audio, n_audio_ctxs, n_state] f32 (host) */
int wrs_debug_decode_forced_features(wrs_decode_task* t, const float* feats, int n_audio, int n_steps, const int32_t* ftok,
                                     const int32_t* fsrc, float* out_lp, int32_t* out_id, double* out_no_speech);
/* re-synchronised run: the first n_forced steps take the given decisions (tokens, sources, f32 cumulative scores,
 * [n_forced][rows], e.g. the oracle's trace up to and including a near-tie), the remaining steps run free; the result
 * and trace are those of wrs_decode_task_run.  n_forced = 0 is wrs_decode_task_run. */
int wrs_debug_decode_resync(wrs_decode_task* t, const float* mel, int n_audio, int n_forced, const int32_t* ftok, const int32_t* fsrc,
                            const float* fslp, wrs_decode_result* result);
/* per-step candidates of the last run (free-running or resync): lp/id [n_steps][rows][k]; returns k through *k */
int wrs_debug_decode_candidates(wrs_decode_task* t, int32_t* n_steps, int32_t* n_rows, int32_t* k, const float** lp, const int32_t** id);

/* host sequencer alone (no device): candidate lists in, beam search / finalize / rank out */
int wrs_debug_sequencer_run(const wrs_token_ids* ids, const wrs_decode_options* opt, const int64_t* initial, int32_t n_initial, int32_t n_audio,
                            int32_t n_steps, const float* lp, const int32_t* id, wrs_decode_result* out, int32_t* out_tokens,
                            int32_t* out_sources, float* out_slp);
/* long-form bookkeeping alone (no device) */
int wrs_debug_transcribe_bookkeeping(int32_t ts_begin, int64_t n_audio_ctxs, int64_t n_frames, const int64_t* prompt, int64_t n_prompt,
                                     const int64_t* window_tokens, const int32_t* window_lens, int32_t n_windows, const wrs_tokenizer* tok,
                                     wrs_transcribe_output* out);
/* f32 PARITY MODE (csrc/parity_f32.hip): every weight, activation, K/V cache entry and product of the model in fp32
 * (v_mfma_f32_16x16x4_f32 GEMMs, fp32 attention).  Exists to demonstrate "token ids bit-exact": in this mode the
 * free-running decode must reproduce the fp32 oracle's ids and beam sources to the last step.  Slow, small models only;
 * choose before wrs_model_finalize. */
int wrs_debug_model_set_parity_f32(wrs_model* m, int enabled);
/* which forms of Model::decoder_step this model has taken since it was created (OR; steps replayed from a captured graph
 * count through their capture): bit 0 tiled GEMMs (>= dec_big_gemm_rows rows), 1 persistent ring GEMMs
 * (WRS_DEC_PRING), 2 fused prologue (WRS_DEC_FUSED, <= 16 rows), 3 MFMA self-attention, 4 QKV handed over without slabs.
 * The experiment switches are read from the environment when the MODEL is created. */
int wrs_debug_model_step_paths(const wrs_model* m, int32_t* path);
int wrs_debug_set_encoder_stop(wrs_model* m, int n_launches);
int wrs_debug_read_encoder(wrs_model* m, int which, float* out, int64_t n);

#ifdef __cplusplus
}
#endif
#endif /* WRS_DEBUG_H */
