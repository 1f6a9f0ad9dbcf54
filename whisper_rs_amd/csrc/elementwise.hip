// Memory-bound helper kernels (gfx950): mel pre/post-processing, LayerNorm (one 64-lane wave
// per row, fp32 statistics), split-K finish + residual + LayerNorm fusion, embedding lookup,
// beam-ancestry table maintenance.  All loads/stores are 16-byte vectors where the layout allows.
#include "kernels.h"

namespace wrs {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ------------------------------------------------------------------------------------------ mel
// reference transcribe.rs:152 + audio.rs:213-228 (pad_or_trim of mel[:, seek:]) fused with the layout change the
// conv1 implicit GEMM wants: [3002, 80] fp16 time-major, rows 0 and 3001 zero (conv padding=1).
__global__ void mel_prep_kernel(const float* __restrict__ mel, int64_t n_frames, int64_t mel_bs, int64_t seek,
                                half_t* __restrict__ out) {
    __shared__ float tile[32][33];
    const int64_t b = blockIdx.z;
    const int64_t t0 = (int64_t)blockIdx.x * 32;  // window-relative frame
    const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {  // r: mel bin within tile, tx: frame within tile
        const int c = c0 + r;
        const int64_t t = t0 + tx, f = seek + t;
        float v = 0.f;
        if (c < 80 && t < 3000 && f < n_frames) v = mel[b * mel_bs + (int64_t)c * n_frames + f];
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {  // r: frame within tile, tx: mel bin
        const int64_t t = t0 + r;
        const int c = c0 + tx;
        if (t < 3000 && c < 80) out[(b * 3002 + t + 1) * 80 + c] = (_Float16)tile[tx][r];
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 160) {
        const int which = threadIdx.x / 80, c = threadIdx.x % 80;
        out[(b * 3002 + (which ? 3001 : 0)) * 80 + c] = (_Float16)0.f;
    }
}
void launch_mel_prep(const float* mel, int64_t n_batch, int64_t n_frames, int64_t mel_batch_stride, int64_t seek,
                     half_t* out, hipStream_t s) {
    dim3 grid((unsigned)cdiv(3000, 32), 3, (unsigned)n_batch);
    mel_prep_kernel<<<grid, 256, 0, s>>>(mel, n_frames, mel_batch_stride, seek, out);
}

// ------------------------------------------------------------------------------------------ LayerNorm
// tch nn::LayerNorm defaults (eps 1e-5, affine) — reference model.rs:255-259,305,310,315,439,518.
// One wave per row; the row lives in registers (D <= 64*4*NV), two-pass mean / variance in fp32.
template <int NV>  // float4 vectors per lane
__device__ __forceinline__ void ln_row(floatx4 (&v)[NV], int D, int lane, const float* __restrict__ g,
                                       const float* __restrict__ b, half_t* o16, float* o32, int packed_row = -1) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < D) s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < D) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d = v[i][r] - mean;
                q += d * d;
            }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + 1e-5f);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < D) {
            const floatx4 gg = *reinterpret_cast<const floatx4*>(g + c);
            const floatx4 bb = *reinterpret_cast<const floatx4*>(b + c);
            floatx4 y;
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] = (v[i][r] - mean) * rstd * gg[r] + bb[r];
            if (o16) {
                half_t* dst = packed_row >= 0 ? o16 + pa_off(packed_row, c, D >> 5) : o16 + c;
                *reinterpret_cast<half4*>(dst) = half4{(_Float16)y[0], (_Float16)y[1], (_Float16)y[2], (_Float16)y[3]};
            }
            if (o32) *reinterpret_cast<floatx4*>(o32 + c) = y;
        }
    }
}

template <int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                        const float* __restrict__ b, int64_t M, int D,
                                                        half_t* __restrict__ out16, float* __restrict__ out32) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    floatx4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        v[i] = c < D ? ld_stream(reinterpret_cast<const floatx4*>(x + row * D + c)) : floatx4{0.f, 0.f, 0.f, 0.f};
    }
    ln_row<NV>(v, D, lane, g, b, out16 ? out16 + row * D : nullptr, out32 ? out32 + row * D : nullptr);
}

static int ln_nv(int D) {
    WRS_REQUIRE(D % 4 == 0 && D <= 2048, "layernorm: D=%d unsupported", D);
    return (int)cdiv(D, 256);
}

void launch_layernorm(const float* x, const float* g, const float* b, int64_t M, int D, half_t* out16, float* out32,
                      hipStream_t s) {
    const unsigned grid = (unsigned)cdiv(M, 4);
    switch (ln_nv(D)) {
        case 1: layernorm_kernel<1><<<grid, 256, 0, s>>>(x, g, b, M, D, out16, out32); break;
        case 2: layernorm_kernel<2><<<grid, 256, 0, s>>>(x, g, b, M, D, out16, out32); break;
        case 3: layernorm_kernel<3><<<grid, 256, 0, s>>>(x, g, b, M, D, out16, out32); break;
        case 4: layernorm_kernel<4><<<grid, 256, 0, s>>>(x, g, b, M, D, out16, out32); break;
        case 5: layernorm_kernel<5><<<grid, 256, 0, s>>>(x, g, b, M, D, out16, out32); break;
        default: layernorm_kernel<8><<<grid, 256, 0, s>>>(x, g, b, M, D, out16, out32); break;
    }
}

template <int NV>
__global__ __launch_bounds__(256) void layernorm_packed_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                               const float* __restrict__ b, int M, int D, half_t* __restrict__ xn) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    floatx4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        v[i] = c < D ? ld_stream(reinterpret_cast<const floatx4*>(x + (int64_t)row * D + c)) : floatx4{0.f, 0.f, 0.f, 0.f};
    }
    ln_row<NV>(v, D, lane, g, b, xn, nullptr, row);  // packed activation order
}
void launch_layernorm_packed(const float* x, const float* g, const float* b, int M, int D, half_t* xn, hipStream_t s) {
    const unsigned grid = (unsigned)cdiv(M, 4);
    switch (ln_nv(D)) {
#define WRS_LNP(NV) case NV: layernorm_packed_kernel<NV><<<grid, 256, 0, s>>>(x, g, b, M, D, xn); break;
        WRS_LNP(1) WRS_LNP(2) WRS_LNP(3) WRS_LNP(4) WRS_LNP(5)
#undef WRS_LNP
        default: layernorm_packed_kernel<8><<<grid, 256, 0, s>>>(x, g, b, M, D, xn); break;
    }
}

// decode step: x[m] += bias + sum_s slab[s][m] (fixed order => deterministic), then xn = LN(x).
// Fuses the split-K reduction of the out-projection / MLP-2 GEMM, the residual add
// (reference model.rs:302-318) and the next LayerNorm into one pass over [M, D].
// One 256-thread workgroup per row: parallelism comes from splitting the row, and all ksplit slab loads of a thread are
// independent and in flight together.  (A wave-per-row variant — no LDS, no barrier — was measured 2 % slower on the
// whole decode at 1280 rows.)
// NT threads per row: every thread owns float4 vectors c = (i * NT + tid) * 4 < D.  NT = D / 4 when that is a whole number of
// waves (D = 1280: 320 threads, one vector each, nobody idle; the 256-thread form left 3 of 4 waves with half the work).
template <int NV, int NT>
__global__ __launch_bounds__(NT) void finish_residual_ln_kernel(float* __restrict__ x, const float* __restrict__ slab,
                                                                int ksplit, int m_pad, const float* __restrict__ bias,
                                                                const float* __restrict__ g, const float* __restrict__ b,
                                                                int M, int D, half_t* __restrict__ xn, int xcd_rows) {
    constexpr int NW = NT / 64;
    __shared__ float red[2 * NW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware row order (workgroups are dealt round-robin over the 8 XCDs): XCD x takes the contiguous rows
    // [x * per, (x + 1) * per), per a multiple of 8.  A 128-byte line of the packed activation output holds the 16-byte
    // chunks of 8 consecutive rows: with row = blockIdx.x eight different L2s each wrote back an eighth of every line.
    // Speed only.  xcd_rows = 0: plain order.
    int row = blockIdx.x;
    if (xcd_rows > 0) {
        row = (int)(blockIdx.x & 7) * xcd_rows + (int)(blockIdx.x >> 3);
        if (row >= M) return;  // whole workgroup, before any barrier
    }
    floatx4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * NT + tid) * 4;
        v[i] = floatx4{0.f, 0.f, 0.f, 0.f};
        if (c < D) {
            // (plain loads: the slabs were written by the kernel before and x by the finish kernel before that; the
            // non-temporal form measured 9.8 us against 9.1 us per launch here, profiles/r03_streaming_loads_ab.log)
            floatx4 acc = *reinterpret_cast<const floatx4*>(slab + (int64_t)row * D + c);
            for (int sidx = 1; sidx < ksplit; ++sidx)
                acc += *reinterpret_cast<const floatx4*>(slab + ((int64_t)sidx * m_pad + row) * D + c);
            if (bias) acc += *reinterpret_cast<const floatx4*>(bias + c);
            const floatx4 a = *reinterpret_cast<const floatx4*>(x + (int64_t)row * D + c) + acc;
            *reinterpret_cast<floatx4*>(x + (int64_t)row * D + c) = a;
            v[i] = a;
            s += (a[0] + a[1]) + (a[2] + a[3]);
        }
    }
    if (!xn) return;
    // LayerNorm weights requested now: behind the two barriers below they would be one more dependent memory round trip
    floatx4 gg[NV], bb[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * NT + tid) * 4;
        gg[i] = c < D ? *reinterpret_cast<const floatx4*>(g + c) : floatx4{0.f, 0.f, 0.f, 0.f};
        bb[i] = c < D ? *reinterpret_cast<const floatx4*>(b + c) : floatx4{0.f, 0.f, 0.f, 0.f};
    }
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) tot += red[w];
    const float mean = tot / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * NT + tid) * 4;
        if (c < D) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d = v[i][r] - mean;
                q += d * d;
            }
        }
    }
    q = wave_sum(q);
    if (lane == 0) red[NW + wave] = q;
    __syncthreads();
    float qt = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) qt += red[NW + w];
    const float rstd = rsqrtf(qt / (float)D + 1e-5f);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * NT + tid) * 4;
        if (c < D) {
            half4 y;
#pragma unroll
            for (int r = 0; r < 4; ++r) y[r] = (_Float16)((v[i][r] - mean) * rstd * gg[i][r] + bb[i][r]);
            *reinterpret_cast<half4*>(xn + pa_off(row, c, D >> 5)) = y;  // packed activation order
        }
    }
}

void launch_finish_residual_ln(float* x, const float* slab, int ksplit, int m_pad, const float* bias, const float* g,
                               const float* b, int M, int D, half_t* xn, hipStream_t s) {
    WRS_REQUIRE(D % 32 == 0 && D <= 2048, "finish_residual_ln: D=%d unsupported", D);
    static const bool plain = exp_env("WRS_FINISH_PLAIN_ROWS") && exp_env("WRS_FINISH_PLAIN_ROWS")[0] == '1';  // experiment switch
    const int per = (M >= 64 && !plain) ? (int)round_up(cdiv(M, 8), 8) : 0;  // rows per XCD (few rows: plain order)
    const unsigned grid = per ? 8u * (unsigned)per : (unsigned)M;
    if (D == 1280) finish_residual_ln_kernel<1, 320><<<grid, 320, 0, s>>>(x, slab, ksplit, m_pad, bias, g, b, M, D, xn, per);
    else if (D == 768) finish_residual_ln_kernel<1, 192><<<grid, 192, 0, s>>>(x, slab, ksplit, m_pad, bias, g, b, M, D, xn, per);
    else if (D <= 1024) finish_residual_ln_kernel<1, 256><<<grid, 256, 0, s>>>(x, slab, ksplit, m_pad, bias, g, b, M, D, xn, per);
    else finish_residual_ln_kernel<2, 256><<<grid, 256, 0, s>>>(x, slab, ksplit, m_pad, bias, g, b, M, D, xn, per);
}

// MLP-1 finish: split-K reduction + bias + exact GELU (reference model.rs:279-286) -> packed fp16 activations
__global__ __launch_bounds__(256) void finish_act_kernel(const float* __restrict__ slab, int ksplit, int m_pad,
                                                         const float* __restrict__ bias, int act, int M, int N,
                                                         half_t* __restrict__ out) {
    const int nq = N >> 2;
    const int64_t total = (int64_t)M * nq;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(i / nq), n = (int)(i % nq) * 4;
        floatx4 a = *reinterpret_cast<const floatx4*>(slab + (int64_t)m * N + n);
        for (int sidx = 1; sidx < ksplit; ++sidx) a += *reinterpret_cast<const floatx4*>(slab + ((int64_t)sidx * m_pad + m) * N + n);
        a += *reinterpret_cast<const floatx4*>(bias + n);
        half4 y;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float x = a[r];
            if (act) x = gelu_exact(x);
            y[r] = (_Float16)x;
        }
        *reinterpret_cast<half4*>(out + pa_off(m, n, N >> 5)) = y;
    }
}
void launch_finish_act(const float* slab, int ksplit, int m_pad, const float* bias, int act, int M, int N, half_t* out,
                       hipStream_t s) {
    WRS_REQUIRE(N % 32 == 0 && M >= 1 && ksplit >= 1, "finish_act: bad shape");
    const int64_t total = (int64_t)M * (N / 4);
    finish_act_kernel<<<(unsigned)std::min<int64_t>(cdiv(total, 256), 4096), 256, 0, s>>>(slab, ksplit, m_pad, bias, act, M, N, out);
}

// reference model.rs:503-506: token_emb(tokens) + position_emb[offset..]; fused with the first block's attn_ln.
template <int NV>
__global__ __launch_bounds__(256) void embed_ln_kernel(const half_t* __restrict__ emb, const float* __restrict__ pos,
                                                       const StepParams* __restrict__ sp, const int* __restrict__ tokens,
                                                       const float* __restrict__ g, const float* __restrict__ b, int M,
                                                       int D, float* __restrict__ x, half_t* __restrict__ xn) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int tok = tokens[row];
    const int pidx = sp->pos_idx;
    floatx4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < D) {
            const half4 e = *reinterpret_cast<const half4*>(emb + (int64_t)tok * D + c);
            const floatx4 pp = *reinterpret_cast<const floatx4*>(pos + (int64_t)pidx * D + c);
            floatx4 a = floatx4{(float)e[0], (float)e[1], (float)e[2], (float)e[3]} + pp;
            *reinterpret_cast<floatx4*>(x + (int64_t)row * D + c) = a;
            v[i] = a;
        } else {
            v[i] = floatx4{0.f, 0.f, 0.f, 0.f};
        }
    }
    ln_row<NV>(v, D, lane, g, b, xn, nullptr, row);  // packed activation order
}

void launch_embed_ln(const half_t* emb, const float* pos, const StepParams* sp, const int* tokens, const float* g,
                     const float* b, int M, int D, float* x, half_t* xn, hipStream_t s) {
    const unsigned grid = (unsigned)cdiv(M, 4);
    switch (ln_nv(D)) {
#define WRS_EMB(NV) case NV: embed_ln_kernel<NV><<<grid, 256, 0, s>>>(emb, pos, sp, tokens, g, b, M, D, x, xn); break;
        WRS_EMB(1) WRS_EMB(2) WRS_EMB(3) WRS_EMB(4) WRS_EMB(5)
#undef WRS_EMB
        default: embed_ln_kernel<8><<<grid, 256, 0, s>>>(emb, pos, sp, tokens, g, b, M, D, x, xn); break;
    }
}

// ------------------------------------------------------------------------------------------ prompt prefill
// The decoder's first forward pass takes all sample_begin prompt tokens at once (reference decode/mod.rs:236-245,
// model.rs:503-506).  Rows are (audio, position): row = a * S + p.  x = emb[token] + pos[p] (fp32), xn = LN(x) fp16 row-major.
template <int NV>
__global__ __launch_bounds__(256) void prefill_embed_ln_kernel(const half_t* __restrict__ emb, const float* __restrict__ pos,
                                                               const int* __restrict__ tokens, int S, int pos0, const float* __restrict__ g,
                                                               const float* __restrict__ b, int M, int D, float* __restrict__ x,
                                                               half_t* __restrict__ xn) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int tok = tokens[row];
    const int pidx = pos0 + row % S;
    floatx4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < D) {
            const half4 e = *reinterpret_cast<const half4*>(emb + (int64_t)tok * D + c);
            const floatx4 pp = *reinterpret_cast<const floatx4*>(pos + (int64_t)pidx * D + c);
            floatx4 a = floatx4{(float)e[0], (float)e[1], (float)e[2], (float)e[3]} + pp;
            *reinterpret_cast<floatx4*>(x + (int64_t)row * D + c) = a;
            v[i] = a;
        } else {
            v[i] = floatx4{0.f, 0.f, 0.f, 0.f};
        }
    }
    ln_row<NV>(v, D, lane, g, b, xn + (int64_t)row * D, nullptr);
}
void launch_prefill_embed_ln(const half_t* emb, const float* pos, const int* tokens, int S, int pos0, const float* g, const float* b,
                             int M, int D, float* x, half_t* xn, hipStream_t s) {
    const unsigned grid = (unsigned)cdiv(M, 4);
    switch (ln_nv(D)) {
#define WRS_PEMB(NV) case NV: prefill_embed_ln_kernel<NV><<<grid, 256, 0, s>>>(emb, pos, tokens, S, pos0, g, b, M, D, x, xn); break;
        WRS_PEMB(1) WRS_PEMB(2) WRS_PEMB(3) WRS_PEMB(4) WRS_PEMB(5)
#undef WRS_PEMB
        default: prefill_embed_ln_kernel<8><<<grid, 256, 0, s>>>(emb, pos, tokens, S, pos0, g, b, M, D, x, xn); break;
    }
}

// k / v of the prompt positions (columns D..3D of qkv [n_audio*S, 3D] fp16) into the decode cache
// [phys_row = a * phys_mul][n_ctx][D] at positions pos0 .. pos0+S-1 (Cached::forward's concatenation, model.rs:83-87)
__global__ void prefill_store_kv_kernel(const half_t* __restrict__ qkv, int S, int pos0, int D, int n_ctx, int phys_mul, int64_t total8,
                                        half_t* __restrict__ kc, half_t* __restrict__ vc) {
    const int d8 = D >> 3;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total8; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / d8;
        const int c = (int)(i % d8) * 8;
        const int a = (int)(row / S), p = (int)(row % S);
        const int64_t dst = ((int64_t)a * phys_mul * n_ctx + pos0 + p) * D + c;
        *reinterpret_cast<half8*>(kc + dst) = *reinterpret_cast<const half8*>(qkv + row * 3 * D + D + c);
        *reinterpret_cast<half8*>(vc + dst) = *reinterpret_cast<const half8*>(qkv + row * 3 * D + 2 * D + c);
    }
}
void launch_prefill_store_kv(const half_t* qkv, int n_audio, int S, int pos0, int D, int n_ctx, int phys_mul, half_t* kc, half_t* vc,
                             hipStream_t s) {
    WRS_REQUIRE(D % 8 == 0 && pos0 >= 0 && pos0 + S <= n_ctx, "prefill_store_kv: bad shape");
    const int64_t total8 = (int64_t)n_audio * S * (D / 8);
    prefill_store_kv_kernel<<<(unsigned)std::min<int64_t>(cdiv(total8, 256), 4096), 256, 0, s>>>(qkv, S, pos0, D, n_ctx, phys_mul, total8, kc, vc);
}

// xn[r] (PACKED activation order, decode-step layout) = LN(x[r * row_stride + row_off]) for r < n: picks one prompt
// position per audio (sot_idx / the last one) for the logits (decode/mod.rs:250-258)
template <int NV>
__global__ __launch_bounds__(256) void gather_ln_packed_kernel(const float* __restrict__ x, int64_t row_stride, int64_t row_off,
                                                               const float* __restrict__ g, const float* __restrict__ b, int n, int D,
                                                               half_t* __restrict__ xn) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const float* src = x + ((int64_t)r * row_stride + row_off) * D;
    floatx4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        v[i] = c < D ? *reinterpret_cast<const floatx4*>(src + c) : floatx4{0.f, 0.f, 0.f, 0.f};
    }
    ln_row<NV>(v, D, lane, g, b, xn, nullptr, r);
}
void launch_gather_ln_packed(const float* x, int64_t row_stride, int64_t row_off, const float* g, const float* b, int n, int D, half_t* xn,
                             hipStream_t s) {
    const unsigned grid = (unsigned)cdiv(n, 4);
    switch (ln_nv(D)) {
#define WRS_GLN(NV) case NV: gather_ln_packed_kernel<NV><<<grid, 256, 0, s>>>(x, row_stride, row_off, g, b, n, D, xn); break;
        WRS_GLN(1) WRS_GLN(2) WRS_GLN(3) WRS_GLN(4) WRS_GLN(5)
#undef WRS_GLN
        default: gather_ln_packed_kernel<8><<<grid, 256, 0, s>>>(x, row_stride, row_off, g, b, n, D, xn); break;
    }
}

// ------------------------------------------------------------------------------------------ beam ancestry
// Replaces `Cached::update_cache` (reference model.rs:53-62: cache.index([src]) gathers every K/V
// tensor of every block each step).  Instead of moving K/V, each row keeps, per cached position, the
// physical row that holds it; a beam re-index only gathers this int table (448 ints per row).
__global__ void anc_init_kernel(int* anc_base, int64_t buf_stride, int n_ctx, int n_phys_rows, int G) {
    const int64_t total = (int64_t)n_phys_rows * n_ctx;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / n_ctx);
        const int v = (r / G) * G;
        anc_base[i] = v;
        anc_base[buf_stride + i] = v;
    }
}
void launch_anc_init(int* anc_base, int64_t anc_buf_stride, int n_ctx, int n_phys_rows, int G, hipStream_t s) {
    const int64_t total = (int64_t)n_phys_rows * n_ctx;
    anc_init_kernel<<<(int)std::min<int64_t>(cdiv(total, 256), 2048), 256, 0, s>>>(anc_base, anc_buf_stride, n_ctx, n_phys_rows, G);
}

__global__ void anc_update_kernel(int* anc_base, int64_t buf_stride, int n_ctx, int n_rows,
                                  const StepParams* __restrict__ sp, const int* __restrict__ src) {
    const int r = blockIdx.x;
    if (r >= n_rows) return;
    const int pos = sp->cache_pos;
    const int* rd = anc_base + (int64_t)(sp->parity ^ 1) * buf_stride + (int64_t)src[r] * n_ctx;
    int* wr = anc_base + (int64_t)sp->parity * buf_stride + (int64_t)r * n_ctx;
    for (int p = threadIdx.x; p <= pos && p < n_ctx; p += blockDim.x) wr[p] = (p == pos) ? r : rd[p];
}
void launch_anc_update(int* anc_base, int64_t anc_buf_stride, int n_ctx, int n_rows, const StepParams* sp,
                       const int* src, hipStream_t s) {
    anc_update_kernel<<<n_rows, 128, 0, s>>>(anc_base, anc_buf_stride, n_ctx, n_rows, sp, src);
}

}  // namespace wrs
