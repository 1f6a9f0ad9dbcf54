// GEMM kernels for gfx950 (CDNA4), 64-wide wavefronts, MFMA.
//
//  * gemm_f16_kernel     C[M,N] = A[M,K] W[N,K]^T, fp16 in / fp32 accumulate, 128x128x64 tiles,
//                        v_mfma_f32_16x16x32_f16.  A rows may overlap (lda < K): conv1d as implicit GEMM
//                        (reference model.rs:421-422 conv1/conv2).  Replaces the tch `linear`/`conv1d`
//                        call sites of the encoder and the cross-attention K/V projection
//                        (reference model.rs:159-161,205,279-292).
//  * gemm_skinny_kernel  decode-step GEMM, M <= 320 rows: weight-streaming (HBM-bound), every
//                        weight fragment is read once with lane-linear 16-byte loads.
//  * gemm_f32_kernel     exact-f32 MFMA (v_mfma_f32_16x16x4_f32) for the STFT-as-GEMM and the mel
//                        filterbank (reference audio.rs:236-252).
//
// Weights are pre-packed in MFMA fragment order (kernels.h) so a wave reads 1 KiB contiguous per
// fragment and LDS fragment reads are lane-linear ds_read_b128 (conflict-free, no swizzle needed).
// MFMA operand roles: A-operand = weight fragment (rows = n), B-operand = activation fragment
// (cols = m); D[n][m]: lane l holds m = l&15 and n = 4*(l>>4)+r, r=0..3 -> 4 consecutive output
// columns per lane = one vector store.
#include "kernels.h"
#include <atomic>

namespace wrs {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// hipFuncSetAttribute is per device: a process that drives several GPUs (window pool) must raise the dynamic-LDS limit of a
// kernel on each of them, once.  `done` is the kernel's own table of 64 flags
// (atomic: worker threads of a window pool reach here concurrently).
static void allow_dynamic_lds(const void* kernel, size_t bytes, std::atomic<bool> (&done)[64]) {
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || done[dev].load(std::memory_order_acquire)) return;
    // concurrent first calls from several worker threads (window pool) may both set the attribute: idempotent
    HIP_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    done[dev].store(true, std::memory_order_release);
}

// ------------------------------------------------------------------------------------------ packing
size_t packed_weight_halfs(int64_t n_pad, int64_t k_pad) { return (size_t)n_pad * (size_t)k_pad; }

template <bool CONV>
__global__ void pack_weight_kernel(const float* __restrict__ w, int64_t n, int64_t k, int64_t ldw, int64_t conv_c,
                                   half_t* __restrict__ out, int64_t n_pad, int64_t k_pad) {
    const int64_t ksteps = k_pad / 32;
    const int64_t total = (n_pad / 16) * ksteps * 64;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int l = (int)(i & 63);
        const int64_t f = i >> 6;
        const int64_t nt = f / ksteps, ks = f % ksteps;
        const int64_t row = nt * 16 + (l & 15);
        const int64_t k0 = ks * 32 + 8 * (l >> 4);
        half8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t kk = k0 + j;
            float x = 0.f;
            if (row < n && kk < k) {
                if (CONV) {
                    const int64_t tap = kk / conv_c, c = kk % conv_c;  // w[n][c][tap]
                    x = w[(row * conv_c + c) * 3 + tap];
                } else {
                    x = w[row * ldw + kk];
                }
            }
            v[j] = (_Float16)x;
        }
        *reinterpret_cast<half8*>(out + i * 8) = v;
    }
}

void launch_pack_weight_f32(const float* w, int64_t n, int64_t k, int64_t ldw, half_t* out, int64_t n_pad, int64_t k_pad,
                            hipStream_t s) {
    const int64_t total = (n_pad / 16) * (k_pad / 32) * 64;
    const int blocks = (int)std::min<int64_t>(cdiv(total, 256), 8192);
    pack_weight_kernel<false><<<blocks, 256, 0, s>>>(w, n, k, ldw, 0, out, n_pad, k_pad);
}
void launch_pack_conv_weight_f32(const float* w, int64_t n, int64_t c, half_t* out, int64_t n_pad, int64_t k_pad,
                                 hipStream_t s) {
    const int64_t total = (n_pad / 16) * (k_pad / 32) * 64;
    const int blocks = (int)std::min<int64_t>(cdiv(total, 256), 8192);
    pack_weight_kernel<true><<<blocks, 256, 0, s>>>(w, n, 3 * c, 0, c, out, n_pad, k_pad);
}

__global__ void f32_to_f16_kernel(const float* __restrict__ in, half_t* __restrict__ out, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (_Float16)in[i];
}
__global__ void f16_to_f32_kernel(const half_t* __restrict__ in, float* __restrict__ out, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (float)in[i];
}
void launch_f32_to_f16(const float* in, half_t* out, int64_t n, hipStream_t s) {
    f32_to_f16_kernel<<<(int)std::min<int64_t>(cdiv(n, 256), 8192), 256, 0, s>>>(in, out, n);
}
void launch_f16_to_f32(const half_t* in, float* out, int64_t n, hipStream_t s) {
    f16_to_f32_kernel<<<(int)std::min<int64_t>(cdiv(n, 256), 8192), 256, 0, s>>>(in, out, n);
}

// ------------------------------------------------------------------------------------------ big GEMM
// cache-policy bits of the tiled kernels' direct-to-LDS operand loads (0 default, 1 sc0, 2 nt, 16 sc1 = L1 bypass): experiment
// switches of scripts/micro/gemm_f16_bench (profiles/r03_glds_cache_policy.log)
#ifndef WRS_GLDS_AUX_A
#define WRS_GLDS_AUX_A 0
#endif
#ifndef WRS_GLDS_AUX_W
#define WRS_GLDS_AUX_W 0
#endif
struct GemmParams {
    const half_t* a;
    int64_t lda, a_rpb, a_bs;
    const half_t* wp;
    int M, N, K, ksteps;
    // epilogue
    const float* bias;
    float scale;
    int scale_n_end;
    int act;
    const float* residual;
    const float* posemb;
    void* c;
    int c_f32;
    int64_t ldc, c_rpb, c_brows, c_roff;
    int c_packed;
    int ksplit;              // > 1: blockIdx.y = k-slice, fp32 partial sums go to slab blockIdx.y (no epilogue math)
    int64_t slab_stride;     // floats between slabs
    int qkv_d_model;         // > 0: decode-step QKV epilogue (GemmEpi)
    half_t* kcache;
    half_t* vcache;
    int64_t cache_row_stride;
    const StepParams* sp;
    half_t* pack_k;          // != null: K / V columns go to the packed attention formats (GemmEpi)
    half_t* pack_v;
    int pack_T, pack_Tp, pack_H, pack_k_col0, pack_v_col0;
};

// Two 4-column groups of a lane's row (fp32, columns 16 i + 4 q .. and 16 (i + 1) + 4 q .. with q = lane >> 4) -> fp16, then
// v_permlane16_swap exchanges the odd lane rows of group i with the even lane rows of group i + 1: afterwards a lane holds 8
// CONSECUTIVE columns — of group i + (q & 1), starting at column 8 (q >> 1) of it — as one 16-byte vector (columns ascending).
// Every lane of the wave must call it (the exchange reads the other rows' registers whatever their EXEC state).
__device__ __forceinline__ uint4 epi_pair16(const floatx4 a, const floatx4 b) {
    typedef _Float16 half2v __attribute__((ext_vector_type(2)));
    unsigned ax = __builtin_bit_cast(unsigned, half2v{(_Float16)a[0], (_Float16)a[1]});
    unsigned ay = __builtin_bit_cast(unsigned, half2v{(_Float16)a[2], (_Float16)a[3]});
    unsigned bx = __builtin_bit_cast(unsigned, half2v{(_Float16)b[0], (_Float16)b[1]});
    unsigned by = __builtin_bit_cast(unsigned, half2v{(_Float16)b[2], (_Float16)b[3]});
    const auto x = __builtin_amdgcn_permlane16_swap(ax, bx, false, false);  // {rows a0 b0 a2 b2, rows a1 b1 a3 b3}
    const auto y = __builtin_amdgcn_permlane16_swap(ay, by, false, false);
    return uint4{x[0], y[0], x[1], y[1]};
}

// Epilogue of one wave's 64-row x (16 NI)-column block (acc[i][j]: n = ncolw + 16 i + 4 (lane >> 4) + r, m = mrow0 + 16 j + (lane & 15)).
// PF: request the residual of the next row group before the current one is stored (kernels with registers to spare: the
// row-major kernels of the encoder; the 12-wave ring kernel has 168 registers per lane and would spill).
template <int NI, bool PF = false>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, floatx4 (&acc)[NI][4], const int64_t mrow0, const int ncolw, const int lane) {
    // epilogue.  Vector path (N, ldc multiples of 4: every shape of the model except an odd vocabulary): all
    // bias / positional / residual accesses are 16-byte loads requested together, branches are wave-uniform.
    const bool vec_ok = ((p.ldc & 3) == 0) && ((p.N & 3) == 0);
    const int ncol0 = ncolw + 4 * (lane >> 4);
    if (p.ksplit > 1) {  // split-K: raw partial sums to this slice's slab; the consumer kernel reduces in fixed order
        float* slab = reinterpret_cast<float*>(p.c) + (int64_t)blockIdx.y * p.slab_stride;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t m = mrow0 + j * 16 + (lane & 15);
            if (m >= p.M) continue;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int n = ncol0 + i * 16;
                if (n < p.N) *reinterpret_cast<floatx4*>(slab + m * p.ldc + n) = acc[i][j];
            }
        }
        return;
    }
    const int q16 = lane >> 4;
    if (p.qkv_d_model > 0) {  // q -> fp16 [M, ldc]; k / v -> the cache row of m at the step's position (launcher checks the shape)
        const int Dm = p.qkv_d_model;
        const int cpos = p.sp->cache_pos;
        static_assert(NI % 2 == 0, "column groups are stored in pairs");
        // after the row exchange (epi_pair16) this lane owns 8 consecutive columns of group i + (q & 1): one 16-byte store
        // instead of two 8-byte ones; a 16-column group never straddles q / k / v (D % 16 == 0, checked by the launcher)
#pragma unroll
        for (int i = 0; i < NI; i += 2) {
            const int na = ncol0 + i * 16, nb = na + 16;
            const floatx4 ba = (p.bias && na < p.N) ? *reinterpret_cast<const floatx4*>(p.bias + na) : floatx4{0.f, 0.f, 0.f, 0.f};
            const floatx4 bb = (p.bias && nb < p.N) ? *reinterpret_cast<const floatx4*>(p.bias + nb) : floatx4{0.f, 0.f, 0.f, 0.f};
            const float sa = na < p.scale_n_end ? p.scale : 1.f, sb = nb < p.scale_n_end ? p.scale : 1.f;
            const int n = ncolw + 16 * (i + (q16 & 1)) + 8 * (q16 >> 1);  // first of this lane's 8 columns after the exchange
            const int region = n / Dm, nl = n - region * Dm;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t m = mrow0 + j * 16 + (lane & 15);
                const uint4 w = epi_pair16((acc[i][j] + ba) * sa, (acc[i + 1][j] + bb) * sb);
                if (m >= p.M || n >= p.N) continue;
                half_t* dst = region == 0 ? reinterpret_cast<half_t*>(p.c) + m * p.ldc + nl
                                          : (region == 1 ? p.kcache : p.vcache) + m * p.cache_row_stride + (int64_t)cpos * Dm + nl;
                *reinterpret_cast<uint4*>(dst) = w;
            }
        }
        return;
    }
    if (vec_ok) {
        floatx4 bias4[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int n = ncol0 + i * 16;
            bias4[i] = (p.bias && n < p.N) ? *reinterpret_cast<const floatx4*>(p.bias + n) : floatx4{0.f, 0.f, 0.f, 0.f};
        }
        // rows are batched only for the conv outputs; elsewhere c_rpb is "infinite" and the 64-bit division and modulo
        // (software routines of a few hundred cycles each, 8 per lane per tile) are skipped
        const bool batched = p.c_rpb <= (int64_t)p.M;
        // fp16 outputs: 16-byte stores (8 consecutive columns per lane after a row exchange) when the groups pair up
        const bool wide = !p.c_f32 && (NI % 2 == 0) && (p.N & 15) == 0 && (p.ldc & 7) == 0;
        // The residual (usually C itself: x += ...) of row group j + 1 is requested BEFORE row group j is stored: behind the
        // stores the compiler has to assume they alias, and every row group paid a full memory round trip with nothing else
        // resident on the CU to hide it (encoder out-projection / MLP-2: +17 us per 256x256 tile, round 3).
        bool mok_n;
        int64_t mr_n, crow_n;
        floatx4 res_n[NI];
        auto rows = [&](int j, bool& mok, int64_t& mr, int64_t& crow) {
            const int64_t m = mrow0 + j * 16 + (lane & 15);
            mok = m < p.M;
            const int64_t mc = mok ? m : p.M - 1;
            const int64_t mb = batched ? mc / p.c_rpb : 0;
            mr = batched ? mc % p.c_rpb : mc;
            crow = mb * p.c_brows + mr + p.c_roff;
        };
        auto fetch = [&](int j) {
            rows(j, mok_n, mr_n, crow_n);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int n = ncol0 + i * 16;
                res_n[i] = (p.residual && mok_n && n < p.N) ? ld_stream(reinterpret_cast<const floatx4*>(p.residual + crow_n * p.ldc + n)) : floatx4{0.f, 0.f, 0.f, 0.f};
            }
        };
        if (PF) fetch(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (!PF) fetch(j);
            const bool mok = mok_n;
            const int64_t mr = mr_n, crow = crow_n;
            floatx4 res4[NI], v4[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) res4[i] = res_n[i];
            if (PF && j + 1 < 4) fetch(j + 1);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int n = ncol0 + i * 16;
                floatx4 v = acc[i][j] + bias4[i];
                if (n < p.scale_n_end) v *= p.scale;  // scale_n_end is a multiple of 4
                if (p.act) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = gelu_exact(v[r]);
                }
                if (p.posemb && mok && n < p.N) v += *reinterpret_cast<const floatx4*>(p.posemb + mr * p.N + n);
                v4[i] = v + res4[i];
            }
            if (wide) {
#pragma unroll
                for (int i = 0; i + 1 < NI; i += 2) {
                    const uint4 w = epi_pair16(v4[i], v4[i + 1]);  // every lane takes part in the exchange
                    const int n = ncolw + 16 * (i + (q16 & 1)) + 8 * (q16 >> 1);
                    if (mok && n < p.N)
                        *reinterpret_cast<uint4*>(reinterpret_cast<half_t*>(p.c) + (p.c_packed ? pa_off((int)crow, n, p.N >> 5) : crow * p.ldc + n)) = w;
                }
                continue;
            }
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int n = ncol0 + i * 16;
                if (!(mok && n < p.N)) continue;
                const floatx4 v = v4[i];
                if (p.c_f32) *reinterpret_cast<floatx4*>(reinterpret_cast<float*>(p.c) + crow * p.ldc + n) = v;
                else *reinterpret_cast<half4*>(reinterpret_cast<half_t*>(p.c) + (p.c_packed ? pa_off((int)crow, n, p.N >> 5) : crow * p.ldc + n)) =
                        half4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t m = mrow0 + j * 16 + (lane & 15);
        if (m >= p.M) continue;
        const bool batched = p.c_rpb <= (int64_t)p.M;
        const int64_t mb = batched ? m / p.c_rpb : 0, mr = batched ? m % p.c_rpb : m;
        const int64_t crow = mb * p.c_brows + mr + p.c_roff;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int n = ncol0 + i * 16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int nn = n + r;
                if (nn < p.N) {
                    float x = acc[i][j][r];
                    if (p.bias) x += p.bias[nn];
                    if (nn < p.scale_n_end) x *= p.scale;
                    if (p.act) x = gelu_exact(x);
                    if (p.posemb) x += p.posemb[mr * p.N + nn];
                    if (p.residual) x += p.residual[crow * p.ldc + nn];
                    if (p.c_f32) reinterpret_cast<float*>(p.c)[crow * p.ldc + nn] = x;
                    else reinterpret_cast<half_t*>(p.c)[crow * p.ldc + nn] = (_Float16)x;
                }
            }
        }
    }
}

// K columns of one wave's 64-row x 64-column block -> packed K (attention.hip: Kp [Tp/16][2][64 lanes][8], element (kt, ks, l, j) =
// K[kt*16 + (l & 15)][ks*32 + 8*(l >> 4) + j] per (audio, head)).  After the row exchange a lane owns 8 consecutive dims of one key:
// exactly one 16-byte entry of the format.  ncolw is a multiple of 64 (one head per call).
__device__ __forceinline__ void gemm_epilogue_pack_k(const GemmParams& p, floatx4 (&acc)[4][4], const int64_t mrow0, const int ncolw, const int lane) {
    const int q16 = lane >> 4;
    const int ncol0 = ncolw + 4 * q16;
    const int64_t head_halfs = (int64_t)p.pack_Tp * 64;
#pragma unroll
    for (int i = 0; i < 4; i += 2) {
        const int na = ncol0 + i * 16, nb = na + 16;
        const floatx4 ba = p.bias ? *reinterpret_cast<const floatx4*>(p.bias + na) : floatx4{0.f, 0.f, 0.f, 0.f};
        const floatx4 bb = p.bias ? *reinterpret_cast<const floatx4*>(p.bias + nb) : floatx4{0.f, 0.f, 0.f, 0.f};
        const float sa = na < p.scale_n_end ? p.scale : 1.f, sb = nb < p.scale_n_end ? p.scale : 1.f;
        const int nk = ncolw + 16 * (i + (q16 & 1)) + 8 * (q16 >> 1) - p.pack_k_col0;  // this lane's first K column after the exchange
        const int h = nk >> 6, dd = nk & 63;
        const int frag = (dd >> 5) * 64 + ((dd & 31) >> 3) * 16;  // (ks, 16-byte chunk) -> lane slot of the fragment
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = (int)mrow0 + j * 16 + (lane & 15);
            const uint4 w = epi_pair16((acc[i][j] + ba) * sa, (acc[i + 1][j] + bb) * sb);
            if (m >= p.M) continue;
            const int b = m / p.pack_T, t = m - b * p.pack_T;
            *reinterpret_cast<uint4*>(p.pack_k + ((int64_t)b * p.pack_H + h) * head_halfs + ((int64_t)((t >> 4) * 128 + frag + (t & 15)) << 3)) = w;
        }
    }
}
// V columns, accumulated with the MFMA operands SWAPPED (activation fragment as the A operand): a lane then holds 4 consecutive
// keys (rows) of one dim (column) — the transposed order of packed V (Vp [Tp/32][4][64 lanes][8], element (kb, dt, l, j) =
// V[kb*32 + kperm(l >> 4, j)][dt*16 + (l & 15)], kperm(q, j) = j < 4 ? 4q + j : 16 + 4q + (j - 4)): one 8-byte store, no transpose.
// acc[i][j][r]: column ncolw + 16 i + (lane & 15), row mrow0 + 16 j + 4 (lane >> 4) + r.  pack_T % 4 == 0: a lane's 4 rows never straddle two audios.
__device__ __forceinline__ void gemm_epilogue_pack_v(const GemmParams& p, floatx4 (&acc)[4][4], const int64_t mrow0, const int ncolw, const int lane) {
    const int64_t head_halfs = (int64_t)p.pack_Tp * 64;
    const int d = lane & 15;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = ncolw + 16 * i + d, nv = n - p.pack_v_col0;
        const int h = nv >> 6, dt = (nv & 63) >> 4;
        const float bias = p.bias ? p.bias[n] : 0.f;
        const float sc = n < p.scale_n_end ? p.scale : 1.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m0 = (int)mrow0 + j * 16 + 4 * (lane >> 4);
            if (m0 >= p.M) continue;
            const int b = m0 / p.pack_T, t = m0 - b * p.pack_T;
            const int pp = t & 31;
            const floatx4 v = (acc[i][j] + bias) * sc;
            *reinterpret_cast<half4*>(p.pack_v + ((int64_t)b * p.pack_H + h) * head_halfs +
                                      ((int64_t)(((t >> 5) * 4 + dt) * 64 + ((pp & 15) >> 2) * 16 + d) << 3) + (pp >> 4) * 4) =
                half4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
        }
    }
}

template <bool APACKED>  // APACKED: A is in packed-activation (fragment) order, as produced by the decode-step kernels
__global__ __launch_bounds__(256) void gemm_f16_kernel(const GemmParams p) {
    // [buf][0 = activations, 1 = weights][fragment][lane] ; 2*2*16*64*16 B = 64 KiB
    __shared__ half8 lds[2][2][16][64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // XCD-aware grouped tile order.  Workgroups are dealt round-robin over the 8 XCDs (each with a private 4 MiB
    // L2), so workgroup id -> (xcd = id % 8, j = id / 8) and XCD x walks the contiguous chunk [x*S, (x+1)*S) of
    // a tile sequence in which 8 row tiles x all column tiles form a band walked column-major: the ~64 tiles in
    // flight on one XCD share 8 A row-tiles and 8 W column-tiles.  (Measured before: FETCH_SIZE 15-30x the operand
    // bytes, A re-streamed once per column tile.)  Speed only: any placement gives the same result.
    const int mt_n = (p.M + 127) >> 7, nt_n = (p.N + 127) >> 7;
    const int tiles = mt_n * nt_n, per_xcd = (tiles + 7) >> 3;
    const int seq = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (seq >= tiles) return;
    const int band = seq / (8 * nt_n);
    const int band_m0 = band * 8;
    const int band_rows = min(8, mt_n - band_m0);
    const int r = seq - band * 8 * nt_n;
    const int m0 = (band_m0 + r % band_rows) * 128, n0 = (r / band_rows) * 128;

    // Activation tile: 128 rows x 64 halfs (one 128-B line per row per k-tile).  Each wave instruction fetches 8 FULL
    // rows (lane -> row l/8, 16-B chunk l%8): whole cache lines instead of 16 half lines.  LDS image is row-major
    // [128][128 B]; to keep the fragment reads (16 rows x one chunk per ds_read_b128) conflict-free the chunk index is
    // XOR-swizzled with (row >> 1) & 7 — on the SOURCE address (the direct-to-LDS destination is always lane-linear)
    // and again on the read.  Weight tile: fragment-packed in HBM, lane-linear as before.
    const half_t* a_ptr[4];
    const half_t* w_ptr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int f = wave * 4 + j, t16 = f >> 1, ks = f & 1;
        if (APACKED) {  // fragment (row tile, k-step) is 1 KiB contiguous: lane-linear, LDS image fragment-ordered
            a_ptr[j] = p.a + ((int64_t)(m0 / 16 + t16) * p.ksteps + ks) * 512 + lane * 8;
        } else {
            const int rl = (wave * 4 + j) * 8 + (lane >> 3);       // row within the tile
            const int chunk = (lane & 7) ^ ((rl >> 1) & 7);         // global 16-B chunk stored at LDS chunk (lane & 7)
            int64_t m = m0 + rl;
            if (m > p.M - 1) m = p.M - 1;
            a_ptr[j] = p.a + (m / p.a_rpb) * p.a_bs + (m % p.a_rpb) * p.lda + chunk * 8;
        }
        w_ptr[j] = p.wp + ((int64_t)(n0 / 16 + t16) * p.ksteps + ks) * 512 + lane * 8;
    }

    floatx4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    const int wy = wave >> 1, wx = wave & 1;
    const int nk = p.K / 64 / p.ksplit;          // k-tiles of this workgroup's slice
    const int kt0 = blockIdx.y * nk;
    // Operand staging by direct-to-LDS loads (global_load_lds_dwordx4): the LDS image is fragment-ordered, so the
    // wave-uniform-base + lane*16 destination rule is exactly one fragment per instruction; the per-lane SOURCE
    // address does the gather for the row-major activation operand.  No staging VGPRs, no ds_write pass.
    // Tile kt+1 is requested before the MFMAs of tile kt and waited for (vmcnt(0)) right before the barrier.
#define WRS_STAGE(buf_, kt_)                                                                                                   \
    do {                                                                                                                       \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                        \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_ptr[j] + (int64_t)(kt_) * (APACKED ? 1024 : 64)),  \
                                             (__attribute__((address_space(3))) void*)&lds[buf_][0][wave * 4 + j][0], 16, 0, WRS_GLDS_AUX_A); \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_ptr[j] + (int64_t)(kt_) * 1024), \
                                             (__attribute__((address_space(3))) void*)&lds[buf_][1][wave * 4 + j][0], 16, 0, WRS_GLDS_AUX_W); \
        }                                                                                                                      \
    } while (0)
    WRS_STAGE(0, kt0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) WRS_STAGE(buf ^ 1, kt0 + kt + 1);
        // all 16 fragment reads of the k-tile are issued before its 32 MFMAs: LDS latency is exposed once per
        // k-tile (the compiler's own schedule drained lgkmcnt four times with 4 MFMAs in between)
        half8 wf[2][4], af[2][4];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < 4; ++i) wf[ks][i] = lds[buf][1][(wy * 4 + i) * 2 + ks][lane];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (APACKED) {
                    af[ks][j] = lds[buf][0][(wx * 4 + j) * 2 + ks][lane];
                } else {
                    const int r = wx * 64 + j * 16 + (lane & 15);
                    const int q = (ks * 4 + (lane >> 4)) ^ ((r >> 1) & 7);
                    af[ks][j] = (&lds[buf][0][0][0])[r * 8 + q];  // row r = 8 x 16-B chunks
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks][i], af[ks][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#undef WRS_STAGE

#ifdef WRS_EXP_NOSTORE  // micro-benchmark hook (scripts/micro): leave out the epilogue to price it
    {
        float sink = 0.f;  // keeps every accumulator (and so every MFMA) alive
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) sink += (acc[i][j][0] + acc[i][j][1]) + (acc[i][j][2] + acc[i][j][3]);
        if (sink == 12345.678f) reinterpret_cast<float*>(p.c)[0] = 1.f;
        return;
    }
#endif
    // Split-K slabs of full tiles: transpose the 128x128 fp32 tile through LDS (the operand ring is free now) so that every
    // store instruction writes two whole 512-byte rows instead of 16 separate 64-byte pieces.
    // (the last row tile may reach past M when the slab has the rows: slabs are padded to the consumer's row stride, and an
    // output row depends on its own activation row only)
    if (p.ksplit > 1 && (int64_t)(m0 + 128) * p.ldc <= p.slab_stride && n0 + 128 <= p.N && (p.ldc & 3) == 0) {
        float* tile = reinterpret_cast<float*>(&lds[0][0][0][0]);  // [128][128] fp32 = 64 KiB
        // chunk swizzle: 16-byte chunk c of row r is kept at chunk c ^ (r & 7) (writes and reads spread over the banks)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int r = wx * 64 + jj * 16 + (lane & 15);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = (wy * 64 + i * 16 + 4 * (lane >> 4)) >> 2;  // 16-byte chunk index within the row (0..31)
                *reinterpret_cast<floatx4*>(tile + r * 128 + ((c ^ (r & 7)) << 2)) = acc[i][jj];
            }
        }
        __syncthreads();
        float* slab = reinterpret_cast<float*>(p.c) + (int64_t)blockIdx.y * p.slab_stride;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int r = wave * 32 + t * 2 + (lane >> 5), c = lane & 31;
            const floatx4 v = *reinterpret_cast<const floatx4*>(tile + r * 128 + ((c ^ (r & 7)) << 2));
            *reinterpret_cast<floatx4*>(slab + (int64_t)(m0 + r) * p.ldc + n0 + (c << 2)) = v;
        }
        return;
    }
    gemm_epilogue<4, !APACKED>(p, acc, (int64_t)m0 + wx * 64, n0 + wy * 64, lane);
}

// ------------------------------------------------------------------------------------------ ring GEMM (decode-step shapes)
// 128x128x64 tiles with SEPARATE loader and MFMA waves: 4 consumer waves (64x64 each, as in gemm_f16_kernel) + 8 loader waves
// that do nothing but issue direct-to-LDS requests (4 of a k-tile's 32 pieces each).  The N = D GEMMs of a decode step
// (M = rows <= 1280, packed activations) have 100 tiles, 200 workgroups with split-K 2: a CU never holds two of them and
// the 4-wave kernel's in-order waves pay request issue (~95 cycles per piece) and MFMA time back to back, 0.55 us per
// k-tile against 0.23 us of MFMA.  Here the requests of tile t+2 are issued while tile t is multiplied (measured 0.35 us per
// k-tile, loader-bound); 3-slot LDS ring (96 KiB), one workgroup barrier per k-tile.  Used when the grid has at most one
// workgroup per CU (a 12-wave, 96-KiB workgroup fills a CU; with more tiles the 4-wave kernel's two workgroups per CU win;
// a 128x256 variant for the N = 3D / 4D shapes spilled and lost: profiles/r01_gemm_ring_decode.log).
// NC consumer waves (4: one 128x128 tile; 8: 128x256, two consumer waves per SIMD) and NL loader waves.
template <int NC, int NL>
__global__ __launch_bounds__(64 * (NC + NL)) void gemm_f16_ring_kernel(const GemmParams p) {
    constexpr int WFR = 4 * NC;                 // weight fragments per k-tile (16 per 128 columns)
    constexpr int BN = 32 * NC;                 // tile width
    constexpr int STAGE = (16 + WFR) * 64;      // half8 units per ring slot
    constexpr int APL = 16 / NL, WPL = WFR / NL;  // pieces per loader wave per k-tile
    static_assert(APL * NL == 16 && WPL * NL == WFR, "pieces must divide evenly over the loader waves");
    extern __shared__ __attribute__((aligned(16))) unsigned char gemm_smem[];
    half8* smem = reinterpret_cast<half8*>(gemm_smem);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int mt_n = (p.M + 127) >> 7, nt_n = (p.N + BN - 1) / BN;
    const int tiles = mt_n * nt_n, per_xcd = (tiles + 7) >> 3;
    const int seq = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (seq >= tiles) return;  // whole workgroup, before any barrier
    const int band = seq / (8 * nt_n);
    const int band_m0 = band * 8;
    const int band_rows = min(8, mt_n - band_m0);
    const int r = seq - band * 8 * nt_n;
    const int m0 = (band_m0 + r % band_rows) * 128, n0 = (r / band_rows) * BN;
    const int nk = p.K / 64 / p.ksplit;
    const int kt0 = blockIdx.y * nk;
#define WRS_RING_BARRIER() asm volatile("s_barrier" ::: "memory")
    floatx4 acc[4][4];
    const int wy = wave >> 1, wx = wave & 1;  // consumer wave: 64-column block, 64-row half
    if (wave >= NC) {
        // ---------------- loader waves
        const int lw = wave - NC;
        const half_t* a_ptr[APL];
        const half_t* w_ptr[WPL];
#pragma unroll
        for (int j = 0; j < APL; ++j) {
            const int f = lw * APL + j;  // fragment: (tile of 16 rows f>>1, k-step f&1)
            a_ptr[j] = p.a + ((int64_t)(m0 / 16 + (f >> 1)) * p.ksteps + (f & 1)) * 512 + lane * 8;
        }
#pragma unroll
        for (int j = 0; j < WPL; ++j) {
            const int f = lw * WPL + j;  // fragment: (tile of 16 columns f>>1, k-step f&1)
            w_ptr[j] = p.wp + ((int64_t)(n0 / 16 + (f >> 1)) * p.ksteps + (f & 1)) * 512 + lane * 8;
        }
        auto issue = [&](int u) {
            half8* st_ = smem + (u % 3) * STAGE;
#pragma unroll
            for (int j = 0; j < APL; ++j)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_ptr[j] + (int64_t)(kt0 + u) * 1024),
                                                 (__attribute__((address_space(3))) void*)(st_ + (lw * APL + j) * 64), 16, 0, WRS_GLDS_AUX_A);
#pragma unroll
            for (int j = 0; j < WPL; ++j)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_ptr[j] + (int64_t)(kt0 + u) * 1024),
                                                 (__attribute__((address_space(3))) void*)(st_ + 1024 + (lw * WPL + j) * 64), 16, 0, WRS_GLDS_AUX_W);
        };
        static_assert(APL + WPL == 4 || APL + WPL == 12, "vmcnt immediates below");
#define WRS_RING_KEEP_ONE() do { if (APL + WPL == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); } while (0)
        issue(0);
        if (nk > 1) { issue(1); WRS_RING_KEEP_ONE(); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        WRS_RING_BARRIER();  // tile 0 has landed
        for (int t = 0; t < nk; ++t) {
            // slot of tile t+2 = slot of tile t-1, which the consumers left at the previous barrier
            if (t + 2 < nk) {
                issue(t + 2);
                WRS_RING_KEEP_ONE();  // tile t+1 landed, tile t+2 in flight
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            WRS_RING_BARRIER();
        }
    } else {
        // ---------------- consumer waves
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
        WRS_RING_BARRIER();
        for (int t = 0; t < nk; ++t) {
            const half8* As = smem + (t % 3) * STAGE;
            const half8* Ws = As + 1024;
            half8 wf[2][4], af[2][4];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int i = 0; i < 4; ++i) wf[ks][i] = Ws[((wy * 4 + i) * 2 + ks) * 64 + lane];
#pragma unroll
                for (int j = 0; j < 4; ++j) af[ks][j] = As[((wx * 4 + j) * 2 + ks) * 64 + lane];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks][i], af[ks][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            WRS_RING_BARRIER();
        }
    }
    // every wave is past the last barrier of the loop: the ring is free
    if (NC == 4 && p.ksplit > 1 && (int64_t)(m0 + 128) * p.ldc <= p.slab_stride && n0 + 128 <= p.N && (p.ldc & 3) == 0) {
        // split-K slab of a full tile: transposed through LDS, stored as whole 512-byte rows by all waves
        float* tile = reinterpret_cast<float*>(smem);  // [128][128] fp32 = 64 KiB
        if (wave < NC) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int rr = wx * 64 + jj * 16 + (lane & 15);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = (wy * 64 + i * 16 + 4 * (lane >> 4)) >> 2;
                    *reinterpret_cast<floatx4*>(tile + rr * 128 + ((c ^ (rr & 7)) << 2)) = acc[i][jj];
                }
            }
        }
        __syncthreads();
        float* slab = reinterpret_cast<float*>(p.c) + (int64_t)blockIdx.y * p.slab_stride;
        for (int t = wave; t < 64; t += NC + NL) {
            const int rr = t * 2 + (lane >> 5), c = lane & 31;
            const floatx4 v = *reinterpret_cast<const floatx4*>(tile + rr * 128 + ((c ^ (rr & 7)) << 2));
            *reinterpret_cast<floatx4*>(slab + (int64_t)(m0 + rr) * p.ldc + n0 + (c << 2)) = v;
        }
        return;
    }
    if (wave < NC) gemm_epilogue(p, acc, (int64_t)m0 + wx * 64, n0 + wy * 64, lane);
#undef WRS_RING_BARRIER
#undef WRS_RING_KEEP_ONE
}

// ------------------------------------------------------------------------------------------ ping-pong GEMM, 256x256x32
// Measured on the 4-wave kernel above (profiles/r01_gemm_*): a wave spends ~95 cycles of ISSUE per 1-KiB direct-to-LDS
// piece — its 8 pieces per k-tile cost more than the tile's 32 MFMAs — and, being in order, cannot overlap the two; a CU
// pulls 90 GB/s from L2 with 4 waves issuing and 140 GB/s with 8 (scripts/micro/l2_to_lds_bw.cpp), while that kernel uses
// 45.  Here 8 waves form two groups of four (group g owns rows 128 g .. 128 g + 127 of a 256x256 tile, each wave a
// 64x128 block, acc 128 registers) that alternate roles every half k-tile ("slot"): while one group's 4 waves (one per
// SIMD) run their 32 MFMAs, the other group's read the fragments of their next k-tile from LDS and issue their share of a
// future tile's loads: 4 pieces + 12 fragment reads per slot against 32 MFMAs, half the L2->LDS bytes per flop of a
// 128x128 tile.  (A 256x128x64 version — 6 pieces + 16 reads per slot — only matched the 4-wave kernel:
// profiles/r01_gemm_pingpong_256x128.log; a 128x256x32 packed-activation version for the decode-step shapes, M = 1280,
// is 1.4x more efficient per CU but leaves 40 % of the CUs without a tile and loses 10-18 % to the 4-wave kernel:
// profiles/r01_gemm_pingpong_128x256_decode.log.)  Ring of 4 x 32 KiB; T(u) = loads of k-tile u, R = fragment reads, M = MFMAs:
//     slot 2t   : group 0  M_t                          group 1  R_t, issue T(t+3)
//     slot 2t+1 : group 0  R_(t+1), issue T(t+4)        group 1  M_t
// T(u) reuses the ring slot of T(u-4), which both groups have read by then, and is waited for (counted vmcnt: 4 pieces per
// wave per tile, the younger tiles stay in flight) at the end of slot 2u-2, one slot before its first reader.  Slot
// boundaries are bare `s_waitcnt lgkmcnt(0); s_barrier`: __syncthreads() carries a workgroup-scope fence that the compiler
// lowers to vmcnt(0), which would wait for the requests just issued and serialise the ring.
// Activation tile: 256 rows x 64 B; a piece is 16 rows x 64 B (lane -> row l/4, chunk l%4).  Fragment reads take chunk
// l>>4 of rows l&15: the chunk position is XOR-swizzled with (row >> 2) & 3 (on the source address of the request, the
// LDS destination of a direct-to-LDS load being lane-linear) so the 16 lanes of a read group hit 64 distinct banks.
// one 256x256 tile (number `seq` of the XCD-aware sequence, k-slice `kslice`) by the 8 waves of a workgroup
// VSWAP: the MFMA operands swapped (D^T per 16x16 block: a lane holds 4 consecutive rows of one column) and the packed-V epilogue
// (gemm_epilogue_pack_v): the V columns of a K/V projection run as a launch of their own (a runtime switch between the two operand
// orders in one kernel made the register allocator spill 417 registers).
template <bool VSWAP>
__device__ __forceinline__ void pp2_tile(const GemmParams& p, const int seq, const int kslice, half8* smem) {
    constexpr int STAGE = 2048;  // half8 units per ring slot: activations 256 x 64 B (1024) + weights 16 fragments (1024)
    int tid = threadIdx.x;
    // opaque to the optimiser: inside a tile LOOP (scripts/micro/cu_confine.h) it would otherwise hoist every lane-dependent
    // address term out of the loop and keep it in registers across the tile (213 -> 256 VGPRs + 124 B of scratch per lane)
    asm volatile("" : "+v"(tid));
    const int wave = tid >> 6, lane = tid & 63;
    const int grp = wave >> 2, w4 = wave & 3;
    const int mt_n = (p.M + 255) >> 8, nt_n = p.N >> 8;
    const int band = seq / (8 * nt_n);
    const int band_m0 = band * 8;
    const int band_rows = min(8, mt_n - band_m0);
    const int r = seq - band * 8 * nt_n;
    const int m0 = (band_m0 + r % band_rows) * 256, n0 = (r / band_rows) * 256;

    const half_t* a_ptr[2];
    const half_t* w_ptr[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int f = wave * 2 + j;                       // piece index 0..15
        const int rl = f * 16 + (lane >> 2);              // row within the tile
        const int chunk = (lane & 3) ^ ((rl >> 2) & 3);   // global 16-B chunk stored at LDS chunk (lane & 3)
        int64_t m = m0 + rl;
        if (m > p.M - 1) m = p.M - 1;
        a_ptr[j] = (p.a_rpb <= (int64_t)p.M) ? p.a + (m / p.a_rpb) * p.a_bs + (m % p.a_rpb) * p.lda + chunk * 8
                                             : p.a + m * p.lda + chunk * 8;
        w_ptr[j] = p.wp + ((int64_t)(n0 / 16 + f) * p.ksteps) * 512 + lane * 8;
    }
    const int nk = p.K / 32 / p.ksplit;
    const int kt0 = kslice * nk;
#define WRS_PP2_ISSUE(u_)                                                                                                      \
    do {                                                                                                                       \
        half8* st_ = smem + ((u_) & 3) * STAGE;                                                                                \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                          \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_ptr[j] + (int64_t)(kt0 + (u_)) * 32), \
                                             (__attribute__((address_space(3))) void*)(st_ + (wave * 2 + j) * 64), 16, 0, WRS_GLDS_AUX_A);  \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                                          \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_ptr[j] + (int64_t)(kt0 + (u_)) * 512), \
                                             (__attribute__((address_space(3))) void*)(st_ + 1024 + (wave * 2 + j) * 64), 16, 0, WRS_GLDS_AUX_W); \
    } while (0)
#define WRS_PP2_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

    floatx4 acc0[4][4], acc1[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc0[i][j] = floatx4{0.f, 0.f, 0.f, 0.f}; acc1[i][j] = floatx4{0.f, 0.f, 0.f, 0.f}; }
    const int wy = w4 >> 1, wx = w4 & 1;  // 128-column half, 64-row block inside the group's 128 rows
    half8 wf[8], af[4];
    auto read_frags = [&](int u) {
        const half8* As = smem + (u & 3) * STAGE;
        const half8* Ws = As + 1024;
#pragma unroll
        for (int i = 0; i < 8; ++i) wf[i] = Ws[(wy * 8 + i) * 64 + lane];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int rr = grp * 128 + wx * 64 + j * 16 + (lane & 15);
            const int q = (lane >> 4) ^ ((rr >> 2) & 3);
            af[j] = As[rr * 4 + q];
        }
    };
    auto mfmas = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (VSWAP) {
                    acc0[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[j], wf[i], acc0[i][j], 0, 0, 0);
                    acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[j], wf[4 + i], acc1[i][j], 0, 0, 0);
                } else {
                    acc0[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[i], af[j], acc0[i][j], 0, 0, 0);
                    acc1[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[4 + i], af[j], acc1[i][j], 0, 0, 0);
                }
            }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
    };
    // wait until every request of this wave older than its `younger` newest tiles (4 pieces each) has landed
    auto wait_keep = [&](int younger) {
        if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    for (int u = 0; u < 4 && u < nk; ++u) WRS_PP2_ISSUE(u);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WRS_PP2_BARRIER();
    if (grp == 0) read_frags(0);
    WRS_PP2_BARRIER();
    if (grp == 0) {
        for (int t = 0; t < nk; ++t) {
            mfmas();                                                  // slot 2t: M_t
            wait_keep(min(2, max(0, min(nk - 1, t + 3) - max(t + 1, 3))));  // T(t+1) landed; issued so far: T(4..t+3)
            WRS_PP2_BARRIER();
            if (t + 1 < nk) read_frags(t + 1);                        // slot 2t+1: R_(t+1), T(t+4) into the slot of T(t)
            if (t + 4 < nk) WRS_PP2_ISSUE(t + 4);
            WRS_PP2_BARRIER();
        }
    } else {
        for (int t = 0; t < nk; ++t) {
            read_frags(t);                                            // slot 2t: R_t, T(t+3) into the slot of T(t-1)
            if (t >= 1 && t + 3 < nk) WRS_PP2_ISSUE(t + 3);
            wait_keep(min(2, max(0, min(nk - 1, t + 3) - max(t + 1, 3))));
            WRS_PP2_BARRIER();
            mfmas();                                                  // slot 2t+1: M_t
            WRS_PP2_BARRIER();
        }
    }
#undef WRS_PP2_ISSUE
#undef WRS_PP2_BARRIER
    const int64_t mrow = (int64_t)m0 + grp * 128 + wx * 64;
    if (VSWAP) {
        gemm_epilogue_pack_v(p, acc0, mrow, n0 + wy * 128, lane);
        gemm_epilogue_pack_v(p, acc1, mrow, n0 + wy * 128 + 64, lane);
        return;
    }
    if (p.pack_k != nullptr && n0 >= p.pack_k_col0) {
        gemm_epilogue_pack_k(p, acc0, mrow, n0 + wy * 128, lane);
        gemm_epilogue_pack_k(p, acc1, mrow, n0 + wy * 128 + 64, lane);
        return;
    }
    gemm_epilogue<4, true>(p, acc0, mrow, n0 + wy * 128, lane);
    gemm_epilogue<4, true>(p, acc1, mrow, n0 + wy * 128 + 64, lane);
}

__global__ __launch_bounds__(512) void gemm_f16_pp2_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gemm_smem[];
    const int mt_n = (p.M + 255) >> 8, nt_n = p.N >> 8;
    const int tiles = mt_n * nt_n, per_xcd = (tiles + 7) >> 3;
    const int seq = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (seq >= tiles) return;  // whole workgroup, before any barrier
    pp2_tile<false>(p, seq, blockIdx.y, reinterpret_cast<half8*>(gemm_smem));
}
__global__ __launch_bounds__(512) void gemm_f16_pp2v_kernel(const GemmParams p) {  // V columns -> packed V (operands swapped)
    extern __shared__ __attribute__((aligned(16))) unsigned char gemm_smem[];
    const int mt_n = (p.M + 255) >> 8, nt_n = p.N >> 8;
    const int tiles = mt_n * nt_n, per_xcd = (tiles + 7) >> 3;
    const int seq = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (seq >= tiles) return;
    pp2_tile<true>(p, seq, blockIdx.y, reinterpret_cast<half8*>(gemm_smem));
}

static int gemm_pp2_min_wgs = 256;

// test hook (wrs_debug_gemm_f16): force one kernel variant / report the variant the last launch took
static int g_gemm_force = GEMM_AUTO;
static int g_gemm_last = GEMM_AUTO;
void gemm_debug_force(int kernel) { g_gemm_force = kernel; }
int gemm_debug_last() { return g_gemm_last; }

bool gemm_pack_kv_supported(int64_t M, int64_t N) {
    return N % 256 == 0 && cdiv(M, 256) * (N / 256) >= gemm_pp2_min_wgs;
}

void launch_gemm_f16(const GemmA& A, const half_t* w_packed, int64_t M, int64_t N, int64_t K, const GemmEpi& e,
                     hipStream_t s) {
    WRS_REQUIRE(M > 0 && N > 0 && K > 0 && K % 64 == 0, "gemm_f16: bad shape M=%ld N=%ld K=%ld", (long)M, (long)N, (long)K);
    WRS_REQUIRE(A.packed || (A.lda % 8 == 0 && A.batch_stride % 8 == 0), "gemm_f16: A strides must be multiples of 8 halfs");
    WRS_REQUIRE(M < (1ll << 31) && N < (1ll << 31), "gemm_f16: shape too large");
    WRS_REQUIRE(e.scale_n_end % 4 == 0, "gemm_f16: scale_n_end must be a multiple of 4");
    GemmParams p;
    p.a = A.a; p.lda = A.lda; p.a_rpb = A.rows_per_batch; p.a_bs = A.batch_stride;
    p.wp = w_packed; p.M = (int)M; p.N = (int)N; p.K = (int)K; p.ksteps = (int)(K / 32);
    p.bias = e.bias; p.scale = e.scale; p.scale_n_end = e.scale_n_end; p.act = e.act;
    p.residual = e.residual; p.posemb = e.posemb; p.c = e.c; p.c_f32 = e.c_f32; p.ldc = e.ldc;
    p.c_rpb = e.c_rows_per_batch; p.c_brows = e.c_batch_rows; p.c_roff = e.c_row_off; p.c_packed = e.c_packed;
    p.qkv_d_model = e.qkv_d_model; p.kcache = e.kcache; p.vcache = e.vcache; p.cache_row_stride = e.cache_row_stride; p.sp = e.sp;
    p.pack_k = e.pack_k; p.pack_v = e.pack_v; p.pack_T = e.pack_T; p.pack_Tp = (int)round_up(std::max(e.pack_T, 1), 32); p.pack_H = e.pack_H;
    p.pack_k_col0 = e.pack_k_col0; p.pack_v_col0 = e.pack_v_col0;
    if (e.pack_k)
        WRS_REQUIRE(e.pack_v && !A.packed && gemm_pack_kv_supported(M, N) && e.ksplit <= 1 && e.pack_T >= 4 && e.pack_T % 4 == 0 && M % e.pack_T == 0 && e.pack_H >= 1 &&
                        e.pack_k_col0 % 256 == 0 && e.pack_v_col0 % 256 == 0 && e.pack_k_col0 < e.pack_v_col0 && e.pack_v_col0 < N &&
                        e.pack_v_col0 - e.pack_k_col0 == (int64_t)e.pack_H * 64 && N - e.pack_v_col0 == (int64_t)e.pack_H * 64 && !e.residual && !e.posemb && !e.act &&
                        !e.c_f32 && !e.c_packed && e.qkv_d_model == 0 && (e.pack_k_col0 == 0 || e.c), "gemm_f16: bad packed-K/V epilogue");
    if (e.qkv_d_model > 0)
        WRS_REQUIRE(N == 3 * (int64_t)e.qkv_d_model && e.qkv_d_model % 64 == 0 && e.c && !e.c_f32 && !e.c_packed && e.ldc >= e.qkv_d_model && e.ldc % 8 == 0 &&
                        e.kcache && e.vcache && e.sp && e.cache_row_stride >= e.qkv_d_model && e.cache_row_stride % 8 == 0 && e.ksplit <= 1 && !e.residual && !e.posemb && !e.act &&
                        e.scale_n_end % 4 == 0, "gemm_f16: bad QKV epilogue");
    if (e.c_packed) WRS_REQUIRE(!e.c_f32 && N % 32 == 0 && e.ldc % 4 == 0, "gemm_f16: packed output needs fp16 and N %% 32 == 0");
    const int64_t tiles = cdiv(M, 128) * cdiv(N, 128);
    p.ksplit = e.ksplit > 1 ? e.ksplit : 1;
    p.slab_stride = e.slab_stride;
    if (p.ksplit > 1)
        WRS_REQUIRE((K / 64) % p.ksplit == 0 && e.c_f32 && !e.bias && !e.residual && !e.posemb && !e.act && e.ldc % 4 == 0 && N % 4 == 0 &&
                        e.slab_stride >= M * e.ldc, "gemm_f16: bad split-K configuration");
    if (A.packed) WRS_REQUIRE(A.packed_rows >= round_up(M, 128), "gemm_f16: packed A must hold a multiple of 128 rows (%ld < %ld)", (long)A.packed_rows, (long)round_up(M, 128));
    const int force = g_gemm_force;
    // 256x256 ping-pong kernel: row-major activations, N a multiple of 256, grid fills the chip
    static const int pp2_mode = exp_env("WRS_GEMM_PP2") ? atoi(exp_env("WRS_GEMM_PP2")) : 1;  // 0 off, 1 auto, 2 always
    const int64_t pp2_tiles = cdiv(M, 256) * (N / 256);
    const bool pp2_ok = !A.packed && N % 256 == 0 && (K / 32) % p.ksplit == 0;
    if (force == GEMM_PP2) WRS_REQUIRE(pp2_ok, "gemm_f16: the 256x256 kernel needs row-major A and N %% 256 == 0");
    if (pp2_ok && (force == GEMM_PP2 || e.pack_k || (force == GEMM_AUTO && (pp2_mode == 2 || (pp2_mode == 1 && pp2_tiles * p.ksplit >= gemm_pp2_min_wgs))))) {
        const size_t shmem = 4 * 2048 * 16;
        static std::atomic<bool> attr[64];
        allow_dynamic_lds(reinterpret_cast<const void*>(gemm_f16_pp2_kernel), shmem, attr);
        if (e.pack_k) {
            // two launches: columns [0, pack_v_col0) on the kernel above (Q, if any, to c; K to packed K), the V columns on the
            // operand-swapped instantiation (a sub-GEMM of its own: weight rows, bias and scale range shifted by pack_v_col0)
            static std::atomic<bool> attr_v[64];
            allow_dynamic_lds(reinterpret_cast<const void*>(gemm_f16_pp2v_kernel), shmem, attr_v);
            GemmParams pk = p;
            pk.N = e.pack_v_col0;
            gemm_f16_pp2_kernel<<<dim3((unsigned)(8 * cdiv(cdiv(M, 256) * (pk.N / 256), 8)), 1), 512, shmem, s>>>(pk);
            GemmParams pv = p;
            pv.N = (int)(N - e.pack_v_col0);
            pv.wp = w_packed + (int64_t)e.pack_v_col0 * K;
            pv.bias = e.bias ? e.bias + e.pack_v_col0 : nullptr;
            pv.scale_n_end = std::max(0, e.scale_n_end - e.pack_v_col0);
            pv.pack_v_col0 = 0;
            gemm_f16_pp2v_kernel<<<dim3((unsigned)(8 * cdiv(cdiv(M, 256) * (pv.N / 256), 8)), 1), 512, shmem, s>>>(pv);
            g_gemm_last = GEMM_PP2;
            return;
        }
        const dim3 g((unsigned)(8 * cdiv(pp2_tiles, 8)), (unsigned)p.ksplit);
        gemm_f16_pp2_kernel<<<g, 512, shmem, s>>>(p);
        g_gemm_last = GEMM_PP2;
        return;
    }
    const dim3 grid((unsigned)(8 * cdiv(tiles, 8)), (unsigned)p.ksplit);
    // decode-step shapes (packed activations) whose grid gives a CU at most one workgroup: loader/consumer ring kernels
    static const int ring_mode = exp_env("WRS_GEMM_RING") ? atoi(exp_env("WRS_GEMM_RING")) : 1;  // 0 off, 1 auto, 2 always
    if (force == GEMM_RING) WRS_REQUIRE(A.packed, "gemm_f16: the ring kernel needs packed activations");
    if (A.packed && (force == GEMM_RING || (force == GEMM_AUTO && ring_mode != 0 && (tiles * p.ksplit <= 256 || (ring_mode == 2 && p.ksplit > 1))))) {
        const size_t shmem = 3 * 2048 * 16;
        static std::atomic<bool> attr[64];
        allow_dynamic_lds(reinterpret_cast<const void*>(gemm_f16_ring_kernel<4, 8>), shmem, attr);
        gemm_f16_ring_kernel<4, 8><<<grid, 768, shmem, s>>>(p);
        g_gemm_last = GEMM_RING;
        return;
        // (a 128x256 instantiation, 8 consumer + 4 loader waves, only equals the 4-wave kernel on the N = 3D / 4D shapes —
        // 24 vs 24 us, 29 vs 30 us: both are dominated by the fp32 / GELU epilogue — and is not used:
        // profiles/r01_gemm_ring_decode.log)
    }
    if (A.packed) {
        gemm_f16_kernel<true><<<grid, 256, 0, s>>>(p);
        g_gemm_last = GEMM_TILE128_PACKED;
    } else {
        gemm_f16_kernel<false><<<grid, 256, 0, s>>>(p);
        g_gemm_last = GEMM_TILE128;
    }
}

// ------------------------------------------------------------------------------------------ skinny GEMM
struct SkinnyParams {
    const half_t* a;
    const half_t* wp;
    int M, N, K, ksteps, ksplit, m_pad;
    int kind;
    const float* bias;
    float scale;
    int scale_n_end;
    int act;
    void* c;
    int64_t ldc;
    int c_packed;
    half_t* kcache;
    half_t* vcache;
    int d_model;
    int64_t cache_row_stride;
    int phys_mul;
    const StepParams* sp;
};

int skinny_m_pad(int M) {
    static const int opts[] = {1, 2, 4, 6, 8, 10, 12, 16, 20};
    for (int o : opts)
        if (o * 16 >= M) return o * 16;
    if (M <= 2560) return (M + 159) / 160 * 160;  // beyond 320 rows: multiples of 10 row tiles (row blocks)
    return -1;
}

// Per wave: up to SK_NKS k-steps (the launcher picks ksplit accordingly).  All weight fragments of the
// wave (the HBM stream, ~1-2 us latency) are requested up front; activation fragments (L2-resident,
// shared by every workgroup) are double-buffered one k-step ahead.  The k-loop is fully unrolled so the
// compiler can keep every load in flight behind the MFMAs.
#define SK_NKS 10
template <int MT>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(const SkinnyParams p) {
    __shared__ float red[4][MT][4][64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int kps = p.ksteps / p.ksplit;
    const int ks_begin = blockIdx.y * kps, ks_end = ks_begin + kps;

    floatx4 acc[2][MT];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[t][mt] = floatx4{0.f, 0.f, 0.f, 0.f};

    const half_t* w0 = p.wp + ((int64_t)(blockIdx.x * 2 + 0) * p.ksteps) * 512 + lane * 8;
    const half_t* w1 = p.wp + ((int64_t)(blockIdx.x * 2 + 1) * p.ksteps) * 512 + lane * 8;
    const half_t* ab = p.a + lane * 8;  // packed activations: fragment (mt, ks) at ((mt * ksteps + ks) * 512) halfs

    half8 wf0[SK_NKS], wf1[SK_NKS];
#pragma unroll
    for (int i = 0; i < SK_NKS; ++i) {
        const int ks = ks_begin + wave + 4 * i;
        if (ks < ks_end) {
            wf0[i] = *reinterpret_cast<const half8*>(w0 + (int64_t)ks * 512);
            wf1[i] = *reinterpret_cast<const half8*>(w1 + (int64_t)ks * 512);
        }
    }
    half8 af[2][MT];
    if (ks_begin + wave < ks_end) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            af[0][mt] = *reinterpret_cast<const half8*>(ab + ((int64_t)mt * p.ksteps + (ks_begin + wave)) * 512);
    }
#pragma unroll
    for (int i = 0; i < SK_NKS; ++i) {
        const int ks = ks_begin + wave + 4 * i;
        if (ks < ks_end) {
            if (i + 1 < SK_NKS && ks + 4 < ks_end) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    af[(i + 1) & 1][mt] = *reinterpret_cast<const half8*>(ab + ((int64_t)mt * p.ksteps + (ks + 4)) * 512);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                acc[0][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf0[i], af[i & 1][mt], acc[0][mt], 0, 0, 0);
                acc[1][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf1[i], af[i & 1][mt], acc[1][mt], 0, 0, 0);
            }
        }
    }

    const int cache_pos = (p.kind == SK_QKV_SELF) ? p.sp->cache_pos : 0;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][mt][r][lane] = acc[t][mt][r];
        __syncthreads();
        for (int vl = tid; vl < MT * 64; vl += 256) {
            const int mt = vl >> 6, ll = vl & 63;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (red[0][mt][r][ll] + red[1][mt][r][ll]) + (red[2][mt][r][ll] + red[3][mt][r][ll]);
            const int m = mt * 16 + (ll & 15);
            const int n = (blockIdx.x * 2 + t) * 16 + 4 * (ll >> 4);
            if (m >= p.M || n >= p.N) continue;
            if (p.kind == SK_SLAB) {
                float* dst = reinterpret_cast<float*>(p.c) + ((int64_t)blockIdx.y * p.m_pad + m) * p.N + n;
                *reinterpret_cast<floatx4*>(dst) = floatx4{v[0], v[1], v[2], v[3]};
                continue;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int nn = n + r;
                if (nn < p.N) {
                    float x = v[r];
                    if (p.bias) x += p.bias[nn];
                    if (nn < p.scale_n_end) x *= p.scale;
                    if (p.act) x = gelu_exact(x);
                    v[r] = x;
                }
            }
            if (p.kind == SK_F32) {
                float* dst = reinterpret_cast<float*>(p.c) + (int64_t)m * p.ldc + n;
                if (((p.ldc & 3) == 0) && n + 4 <= p.N) {
                    *reinterpret_cast<floatx4*>(dst) = floatx4{v[0], v[1], v[2], v[3]};
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (n + r < p.N) dst[r] = v[r];
                }
            } else if (p.kind == SK_F16) {
                half_t* dst = reinterpret_cast<half_t*>(p.c) + (p.c_packed ? pa_off(m, n, p.N >> 5) : (int64_t)m * p.ldc + n);
                *reinterpret_cast<half4*>(dst) = half4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
            } else {  // SK_QKV_SELF
                const int region = n / p.d_model, nl = n - region * p.d_model;
                half_t* dst;
                if (region == 0) dst = reinterpret_cast<half_t*>(p.c) + (int64_t)m * p.ldc + nl;
                else {
                    half_t* base = region == 1 ? p.kcache : p.vcache;
                    dst = base + (int64_t)m * p.phys_mul * p.cache_row_stride + (int64_t)cache_pos * p.d_model + nl;
                }
                *reinterpret_cast<half4*>(dst) = half4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
            }
        }
        __syncthreads();
    }
}

static void launch_gemm_skinny_block(const half_t* a, const half_t* w_packed, int M, int64_t N, int64_t n_pad, int64_t K, int ksplit,
                                     const SkinnyEpi& e, hipStream_t s) {
    const int m_pad = skinny_m_pad(M);
    WRS_REQUIRE(m_pad > 0 && m_pad <= 320, "gemm_skinny: M=%d exceeds 320 rows per block", M);
    WRS_REQUIRE(K % 32 == 0 && n_pad % 32 == 0 && N <= n_pad, "gemm_skinny: bad shape N=%ld n_pad=%ld K=%ld", (long)N,
                (long)n_pad, (long)K);
    WRS_REQUIRE(ksplit >= 1 && (K / 32) % ksplit == 0, "gemm_skinny: ksplit %d does not divide K/32=%ld", ksplit, (long)(K / 32));
    WRS_REQUIRE(ksplit == 1 || e.kind == SK_SLAB, "gemm_skinny: split-K needs the slab epilogue");
    WRS_REQUIRE(cdiv(K / 32 / ksplit, 4) <= SK_NKS, "gemm_skinny: K=%ld with ksplit %d exceeds %d k-steps per wave", (long)K, ksplit, SK_NKS);
    if (e.kind == SK_F16 || e.kind == SK_QKV_SELF || e.kind == SK_SLAB)
        WRS_REQUIRE(N % 4 == 0 && (e.kind == SK_SLAB || e.ldc % 4 == 0), "gemm_skinny: N and ldc must be multiples of 4");
    if (e.c_packed) WRS_REQUIRE(e.kind == SK_F16 && N % 32 == 0, "gemm_skinny: packed output needs SK_F16 and N %% 32 == 0");
    if (e.kind == SK_QKV_SELF)
        WRS_REQUIRE(e.sp && e.kcache && e.vcache && e.d_model > 0 && N == 3 * e.d_model && e.d_model % 4 == 0, "gemm_skinny: bad QKV epilogue");
    SkinnyParams p;
    p.a = a; p.wp = w_packed; p.M = M; p.N = (int)N; p.K = (int)K; p.ksteps = (int)(K / 32); p.ksplit = ksplit;
    p.m_pad = m_pad; p.kind = e.kind; p.bias = e.bias; p.scale = e.scale; p.scale_n_end = e.scale_n_end;
    p.act = e.act; p.c = e.c; p.ldc = e.ldc; p.c_packed = e.c_packed; p.kcache = e.kcache; p.vcache = e.vcache; p.d_model = (int)e.d_model;
    p.cache_row_stride = e.cache_row_stride; p.phys_mul = e.phys_mul; p.sp = e.sp;
    dim3 grid((unsigned)(n_pad / 32), (unsigned)ksplit);
    switch (m_pad / 16) {
#define WRS_SK(MT) case MT: gemm_skinny_kernel<MT><<<grid, 256, 0, s>>>(p); break;
        WRS_SK(1) WRS_SK(2) WRS_SK(4) WRS_SK(6) WRS_SK(8) WRS_SK(10) WRS_SK(12) WRS_SK(16) WRS_SK(20)
#undef WRS_SK
        default: WRS_THROW(WRS_EINVAL, "gemm_skinny: unsupported m_pad %d", m_pad);
    }
}

// More than 320 rows: row blocks of 320 (only the row-major fp32 epilogue used by the logits GEMM supports this)
void launch_gemm_skinny(const half_t* a, const half_t* w_packed, int M, int64_t N, int64_t n_pad, int64_t K, int ksplit,
                        const SkinnyEpi& e, hipStream_t s) {
    if (M <= 320) { launch_gemm_skinny_block(a, w_packed, M, N, n_pad, K, ksplit, e, s); return; }
    WRS_REQUIRE(e.kind == SK_F32 && ksplit == 1, "gemm_skinny: > 320 rows only with the fp32 row-major epilogue");
    for (int m0 = 0; m0 < M; m0 += 320) {
        SkinnyEpi eb = e;
        eb.c = reinterpret_cast<float*>(e.c) + (int64_t)m0 * e.ldc;
        launch_gemm_skinny_block(a + (int64_t)(m0 / 16) * (K / 32) * 512, w_packed, std::min(320, M - m0), N, n_pad, K, 1, eb, s);
    }
}

// ------------------------------------------------------------------------------------------ decode GEMM
// Latency-optimised split-K GEMM for the decode step (M <= 320 rows, weights streamed once from HBM).
// The step's GEMMs are tiny (3-13 MB of weights, ~0.5 GFLOP): a kernel costs launch + memory round trips, so
// the structure is "request everything, wait once, compute":
//   workgroup = 64 output columns (one 16-column tile per wave) x KSW k-steps x MT row tiles
//   * activation panel [MT x kps] fragments (<= 100 KiB) goes global -> LDS with direct-to-LDS loads
//     (global_load_lds_dwordx4: no VGPRs, 1 KiB contiguous per wave instruction thanks to the packed
//     activation layout), shared by the 4 waves;
//   * each wave requests its kps weight fragments straight into registers (lane-linear 1 KiB reads);
//   * one barrier, then kps x MT MFMAs per wave with activation fragments read lane-linearly from LDS
//     (conflict-free ds_read_b128);
//   * partial sums go to slab[blockIdx.y] with one 16-byte store per lane per row tile; the consumer kernel
//     reduces the slabs in fixed order (deterministic, unlike float atomics).
#ifndef DEC_W_AUX
#define DEC_W_AUX 0  // cache policy of the weight stream's direct-to-LDS loads (2 = non-temporal)
#endif
template <int MT, int KPS>
__global__ __launch_bounds__(256) void gemm_dec_kernel(const half_t* __restrict__ a, const half_t* __restrict__ wp,
                                                       float* __restrict__ slab, int M, int N, int ksteps, int m_pad) {
    extern __shared__ __attribute__((aligned(16))) unsigned char dec_smem[];
    // [4 waves * KPS weight fragments][MT * KPS activation fragments], 64 lanes x 16 B each
    half8* wlds = reinterpret_cast<half8*>(dec_smem);
    half8* alds = wlds + (size_t)4 * KPS * 64;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int ntile = blockIdx.x * 4 + wave;
    const int ks0 = blockIdx.y * KPS;
    const int mt0 = blockIdx.z * MT;  // activation buffers always hold a multiple of MT row tiles (skinny_m_pad)

#if !(defined(DEC_ABLATE) && DEC_ABLATE == 2)
    // request everything: this wave's weight fragments, then its share of the activation panel.
    // Direct-to-LDS loads are asynchronous, use no VGPRs and cannot be sunk below the barrier by the compiler.
    const half_t* wbase = wp + ((int64_t)ntile * ksteps + ks0) * 512 + lane * 8;
#pragma unroll
    for (int i = 0; i < KPS; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wbase + (int64_t)i * 512),
                                         (__attribute__((address_space(3))) void*)(wlds + (size_t)(wave * KPS + i) * 64), 16, 0, DEC_W_AUX);
    // row tile mt is fetched by wave (mt & 3): KPS contiguous KiB per row tile in the packed activation layout
    for (int mt = wave; mt < MT; mt += 4) {
        const half_t* src = a + ((int64_t)(mt0 + mt) * ksteps + ks0) * 512 + lane * 8;
#pragma unroll
        for (int kk = 0; kk < KPS; ++kk)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (int64_t)kk * 512),
                                             (__attribute__((address_space(3))) void*)(alds + (size_t)(mt * KPS + kk) * 64), 16, 0, 0);
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // every wave's direct-to-LDS loads have landed and are visible
#if defined(DEC_ABLATE) && DEC_ABLATE == 1
    return;
#endif

    floatx4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = floatx4{0.f, 0.f, 0.f, 0.f};
    // no per-tile guards: every LDS read of a k-step is issued before its MFMAs (rows >= M are only masked at the store)
#pragma unroll
    for (int i = 0; i < KPS; ++i) {
        const half8 wf = wlds[(size_t)(wave * KPS + i) * 64 + lane];
        half8 af[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) af[mt] = alds[(size_t)(mt * KPS + i) * 64 + lane];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, af[mt], acc[mt], 0, 0, 0);
    }
    const int n = ntile * 16 + 4 * (lane >> 4);
    if (n < N) {
        float* dst = slab + ((int64_t)blockIdx.y * m_pad + mt0 * 16 + (lane & 15)) * N + n;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            if ((mt0 + mt) * 16 + (lane & 15) < M) *reinterpret_cast<floatx4*>(dst + (int64_t)mt * 16 * N) = acc[mt];
    }
}

// k-steps per workgroup: the largest supported divisor of `ksteps` (the kernel is instantiated for these)
static int pick_kps(int ksteps, int cap) {
    static const int opts[] = {16, 10, 8, 6, 5, 4, 3, 2, 1};
    for (int d : opts)
        if (d <= cap && ksteps % d == 0) return d;
    return 1;
}

// row tiles per workgroup: m_pad/16 is 1, 2, 4, 6, 8, 10, 12, 16 or 20 (skinny_m_pad)
static int pick_mt(int mtiles, int cap) {
    static const int opts[] = {10, 8, 6, 5, 4, 3, 2, 1};
    for (int d : opts)
        if (d <= cap && mtiles % d == 0) return d;
    return 1;
}

int gemm_dec_plan(int M, int64_t N, int64_t K, int* kps_out, int* mt_block_out) {
    const int ksteps = (int)(K / 32);
    const int mtiles = skinny_m_pad(M) / 16;
    const bool deep = ksteps >= 80;  // K >= 2560 (MLP-2): 16 k-steps x <= 5 row tiles per workgroup, rows split over grid.z
    const int kps = pick_kps(ksteps, deep ? 16 : 10);
    const int mt = pick_mt(mtiles, kps > 10 ? 5 : 10);  // LDS: (mt + 4) * kps KiB <= 160 KiB
    if (kps_out) *kps_out = kps;
    if (mt_block_out) *mt_block_out = mt;
    return ksteps / kps;  // ksplit = number of slabs
}

template <int MT, int KPS>
static void launch_dec(const half_t* a, const half_t* wp, float* slab, int M, int N, int n_pad, int ksteps, int mtiles, int m_pad,
                       hipStream_t s) {
    const size_t shmem = (size_t)(MT + 4) * KPS * 1024;  // weight fragments of 4 waves + activation panel
    static std::atomic<bool> attr_set[64];
    allow_dynamic_lds(reinterpret_cast<const void*>(&gemm_dec_kernel<MT, KPS>), shmem, attr_set);
    dim3 grid((unsigned)(n_pad / 64), (unsigned)(ksteps / KPS), (unsigned)(mtiles / MT));
    gemm_dec_kernel<MT, KPS><<<grid, 256, shmem, s>>>(a, wp, slab, M, N, ksteps, m_pad);
}

template <int MT>
static void launch_dec_k(int kps, const half_t* a, const half_t* wp, float* slab, int M, int N, int n_pad, int ksteps, int mtiles,
                         int m_pad, hipStream_t s) {
    switch (kps) {
#define WRS_DK(K_) case K_: launch_dec<MT, K_>(a, wp, slab, M, N, n_pad, ksteps, mtiles, m_pad, s); break;
        WRS_DK(1) WRS_DK(2) WRS_DK(3) WRS_DK(4) WRS_DK(5) WRS_DK(6) WRS_DK(8) WRS_DK(10)
#undef WRS_DK
        case 16:
            if constexpr (MT <= 5) launch_dec<MT, 16>(a, wp, slab, M, N, n_pad, ksteps, mtiles, m_pad, s);
            else WRS_THROW(WRS_EINVAL, "gemm_dec: kps 16 needs <= 5 row tiles per workgroup");
            break;
        default: WRS_THROW(WRS_EINVAL, "gemm_dec: unsupported k-steps per workgroup %d", kps);
    }
}

void launch_gemm_dec(const half_t* a, const half_t* w_packed, int M, int64_t N, int64_t n_pad, int64_t K, float* slab, hipStream_t s) {
    const int m_pad = skinny_m_pad(M);
    WRS_REQUIRE(m_pad > 0, "gemm_dec: M=%d exceeds 1280 rows", M);
    WRS_REQUIRE(K % 32 == 0 && n_pad % 64 == 0 && N <= n_pad && N % 4 == 0, "gemm_dec: bad shape N=%ld n_pad=%ld K=%ld", (long)N, (long)n_pad, (long)K);
    int kps = 0, mt = 0;
    gemm_dec_plan(M, N, K, &kps, &mt);
    const int ksteps = (int)(K / 32), mtiles = m_pad / 16;
    switch (mt) {
#define WRS_DM(M_) case M_: launch_dec_k<M_>(kps, a, w_packed, slab, M, (int)N, (int)n_pad, ksteps, mtiles, m_pad, s); break;
        WRS_DM(1) WRS_DM(2) WRS_DM(3) WRS_DM(4) WRS_DM(5) WRS_DM(6) WRS_DM(8) WRS_DM(10)
#undef WRS_DM
        default: WRS_THROW(WRS_EINVAL, "gemm_dec: unsupported row tiles per workgroup %d", mt);
    }
}

// ------------------------------------------------------------------------------------------ decode GEMM with a fused producer (<= 16 rows)
// One window (beam 5 -> 5 rows) is the reference's own usage mode (TranscribeTask::run decodes window after window), and
// there a decode step is launch-bound: 12 dependent launches per layer at ~6.5 us each against 0.25 ms of weight stream per
// step.  For up to 16 rows (one MFMA row tile) the "finish" kernels disappear into the NEXT GEMM's prologue: every
// workgroup rebuilds the activation fragments of its k-slice itself —
//   DECF_LN    x_new = x_in + bias_in + sum_s slab_in[s]   (split-K reduction of the previous GEMM + residual, model.rs:302-318),
//              a = LayerNorm(x_new) * g + b (full-row statistics, two passes, fp32); the workgroups of n-tile group 0 write
//              x_new to x_out (ping-pong with x_in: other workgroups are still reading x_in)
//   DECF_GELU  a = gelu(sum_s slab_in[s] + bias_in)          (MLP-1 finish, model.rs:279-286)
// — from a few KB of L2-resident data (5 rows), so the redundancy costs nothing; 12 launches per layer become 8.
enum { DECF_LN = 0, DECF_GELU = 1 };
struct DecFusedParams {
    const half_t* wp;
    float* slab;              // out: [ksteps / KPS][16][N]
    int M, N, ksteps;
    // producer
    const float* slab_in;     // [ksplit_in][m_pad_in][K]
    int ksplit_in, m_pad_in;
    const float* bias_in;     // [K]
    const float* x_in;        // DECF_LN: [M, K] fp32
    float* x_out;             // DECF_LN: [M, K] fp32 (!= x_in)
    const float* ln_g;
    const float* ln_b;
};

static constexpr int DECF_SLAB_CHUNK = 4;  // producer slabs read per round trip

template <int KPS, int MODE>
__global__ __launch_bounds__(256) void gemm_dec_fused_kernel(const DecFusedParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char decf_smem[];
    half8* wlds = reinterpret_cast<half8*>(decf_smem);            // [4 waves * KPS][64]
    half8* alds = wlds + (size_t)4 * KPS * 64;                     // [KPS][64]
    float* rowbuf = reinterpret_cast<float*>(alds + (size_t)KPS * 64);  // DECF_LN: the producer's rows [M][K] fp32
    __shared__ float stat_mean[16], stat_rstd[16];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int ntile = blockIdx.x * 4 + wave;
    const int ks0 = blockIdx.y * KPS;
    const int K = p.ksteps * 32;

    // this wave's weight fragments: requested first, they land while the activations are rebuilt
    const half_t* wbase = p.wp + ((int64_t)ntile * p.ksteps + ks0) * 512 + lane * 8;
#pragma unroll
    for (int i = 0; i < KPS; ++i)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wbase + (int64_t)i * 512),
                                         (__attribute__((address_space(3))) void*)(wlds + (size_t)(wave * KPS + i) * 64), 16, 0, DEC_W_AUX);

    // value of the producer's output at (row m, columns k .. k+3)
    // (slabs are read four at a time with independent loads: one dependent L2-miss round trip per slab made this prologue 14 us)
    auto produced4 = [&](int m, int k) -> floatx4 {
        const floatx4 bi = *reinterpret_cast<const floatx4*>(p.bias_in + k);
        floatx4 xr = floatx4{0.f, 0.f, 0.f, 0.f};
        if (MODE == DECF_LN) xr = *reinterpret_cast<const floatx4*>(p.x_in + (int64_t)m * K + k);
        floatx4 a = floatx4{0.f, 0.f, 0.f, 0.f};
        for (int s0 = 0; s0 < p.ksplit_in; s0 += DECF_SLAB_CHUNK) {
            floatx4 v[DECF_SLAB_CHUNK];
#pragma unroll
            for (int j = 0; j < DECF_SLAB_CHUNK; ++j)
                v[j] = s0 + j < p.ksplit_in ? *reinterpret_cast<const floatx4*>(p.slab_in + ((int64_t)(s0 + j) * p.m_pad_in + m) * K + k) : floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < DECF_SLAB_CHUNK; ++j) a += v[j];
        }
        a += bi;
        if (MODE == DECF_LN) a += xr;
        return a;
    };
    if (MODE == DECF_LN) {
        // the producer's M full rows -> LDS, all 256 threads, every load independent (one memory round trip), then the
        // LayerNorm statistics (eps 1e-5, two passes like layernorm_kernel) from LDS: 16 threads per row
        const int kq = K >> 2;
#pragma unroll 2
        for (int i = tid; i < p.M * kq; i += 256) {
            const int m = i / kq, k = (i - m * kq) * 4;
            *reinterpret_cast<floatx4*>(rowbuf + (size_t)m * K + k) = produced4(m, k);
        }
        __syncthreads();
        const int r = tid >> 4, sub = tid & 15;
        float mean = 0.f, rstd = 0.f;
        if (r < p.M) {
            const float* row = rowbuf + (size_t)r * K;
            float sacc = 0.f;
            for (int k = sub * 4; k < K; k += 64) {
                const floatx4 a = *reinterpret_cast<const floatx4*>(row + k);
                sacc += (a[0] + a[1]) + (a[2] + a[3]);
            }
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) sacc += __shfl_xor(sacc, o);
            mean = sacc / (float)K;
            float q = 0.f;
            for (int k = sub * 4; k < K; k += 64) {
                const floatx4 a = *reinterpret_cast<const floatx4*>(row + k);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = a[e] - mean;
                    q += d * d;
                }
            }
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) q += __shfl_xor(q, o);
            rstd = rsqrtf(q / (float)K + 1e-5f);
        }
        if (sub == 0) { stat_mean[r] = mean; stat_rstd[r] = rstd; }
        __syncthreads();
    }
    // activation fragments of this k-slice in MFMA B-operand order: fragment kk, lane l holds A[l & 15][(ks0+kk)*32 + 8*(l>>4) .. +7]
    for (int kk = wave; kk < KPS; kk += 4) {
        const int m = lane & 15, k0 = (ks0 + kk) * 32 + 8 * (lane >> 4);
        half8 h = {0, 0, 0, 0, 0, 0, 0, 0};
        if (m < p.M) {
            floatx4 a0, a1;
            if (MODE == DECF_LN) {
                a0 = *reinterpret_cast<const floatx4*>(rowbuf + (size_t)m * K + k0);
                a1 = *reinterpret_cast<const floatx4*>(rowbuf + (size_t)m * K + k0 + 4);
            } else {
                a0 = produced4(m, k0);
                a1 = produced4(m, k0 + 4);
            }
            float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            if (MODE == DECF_LN) {
                if (blockIdx.x == 0) {  // one workgroup per k-slice publishes the new residual stream
                    *reinterpret_cast<floatx4*>(p.x_out + (int64_t)m * K + k0) = a0;
                    *reinterpret_cast<floatx4*>(p.x_out + (int64_t)m * K + k0 + 4) = a1;
                }
                const float mean = stat_mean[m], rstd = stat_rstd[m];
                const floatx4 g0 = *reinterpret_cast<const floatx4*>(p.ln_g + k0), g1 = *reinterpret_cast<const floatx4*>(p.ln_g + k0 + 4);
                const floatx4 b0 = *reinterpret_cast<const floatx4*>(p.ln_b + k0), b1 = *reinterpret_cast<const floatx4*>(p.ln_b + k0 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = (v[e] - mean) * rstd * g0[e] + b0[e];
                    v[4 + e] = (v[4 + e] - mean) * rstd * g1[e] + b1[e];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = gelu_exact(v[e]);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) h[e] = (_Float16)v[e];
        }
        alds[(size_t)kk * 64 + lane] = h;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();  // weight fragments landed, activation fragments written

    floatx4 acc = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < KPS; ++i) {
        const half8 wf = wlds[(size_t)(wave * KPS + i) * 64 + lane];
        const half8 af = alds[(size_t)i * 64 + lane];
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, af, acc, 0, 0, 0);
    }
    const int n = ntile * 16 + 4 * (lane >> 4);
    if (n < p.N && (lane & 15) < p.M)
        *reinterpret_cast<floatx4*>(p.slab + ((int64_t)blockIdx.y * 16 + (lane & 15)) * p.N + n) = acc;
}

template <int KPS, int MODE>
static void launch_decf(const DecFusedParams& p, int n_pad, hipStream_t s) {
    const size_t shmem = (size_t)(1 + 4) * KPS * 1024 + (MODE == DECF_LN ? (size_t)p.M * p.ksteps * 32 * 4 : 0);
    WRS_REQUIRE(shmem <= 159 * 1024, "gemm_dec_fused: %zu bytes of LDS (rows %d, K %d)", shmem, p.M, p.ksteps * 32);
    static std::atomic<bool> attr_set[64];
    static std::atomic<size_t> attr_max[64];
    {   // the limit must cover the largest request seen on this device (rows vary between calls)
        int dev = 0;
        HIP_CHECK(hipGetDevice(&dev));
        if (dev >= 0 && dev < 64 && attr_max[dev].load() < shmem) {
            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_dec_fused_kernel<KPS, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
            attr_max[dev].store(shmem);
            attr_set[dev].store(true);
        }
    }
    dim3 grid((unsigned)(n_pad / 64), (unsigned)(p.ksteps / KPS));
    gemm_dec_fused_kernel<KPS, MODE><<<grid, 256, shmem, s>>>(p);
}

// slab[ks][16][N] = A W^T with A rebuilt in the prologue (mode 0: finish + LayerNorm, mode 1: finish + GELU); M <= 16 rows.
// Returns the number of slabs written (= gemm_dec_plan of the shape).
int launch_gemm_dec_fused(int mode, const float* slab_in, int ksplit_in, int m_pad_in, const float* bias_in, const float* x_in, float* x_out,
                          const float* ln_g, const float* ln_b, const half_t* w_packed, int M, int64_t N, int64_t n_pad, int64_t K, float* slab,
                          hipStream_t s) {
    WRS_REQUIRE(M >= 1 && M <= 16, "gemm_dec_fused: %d rows (at most 16)", M);
    WRS_REQUIRE(K % 64 == 0 && n_pad % 64 == 0 && N <= n_pad && N % 4 == 0, "gemm_dec_fused: bad shape N=%ld n_pad=%ld K=%ld", (long)N, (long)n_pad, (long)K);
    WRS_REQUIRE(slab_in && bias_in && ksplit_in >= 1 && m_pad_in >= M && slab, "gemm_dec_fused: bad producer arguments");
    if (mode == DECF_LN) WRS_REQUIRE(x_in && x_out && x_in != x_out && ln_g && ln_b, "gemm_dec_fused: LN mode needs x_in != x_out and the LayerNorm parameters");
    int kps = 0, mt = 0;
    const int ksplit = gemm_dec_plan(M, N, K, &kps, &mt);
    WRS_REQUIRE(mt == 1, "gemm_dec_fused: unexpected row tiling");
    DecFusedParams p{w_packed, slab, M, (int)N, (int)(K / 32), slab_in, ksplit_in, m_pad_in, bias_in, x_in, x_out, ln_g, ln_b};
    switch (kps) {
#define WRS_DF(K_) case K_: if (mode == DECF_LN) launch_decf<K_, DECF_LN>(p, (int)n_pad, s); else launch_decf<K_, DECF_GELU>(p, (int)n_pad, s); break;
        WRS_DF(1) WRS_DF(2) WRS_DF(3) WRS_DF(4) WRS_DF(5) WRS_DF(6) WRS_DF(8) WRS_DF(10) WRS_DF(16)
#undef WRS_DF
        default: WRS_THROW(WRS_EINVAL, "gemm_dec_fused: unsupported k-steps per workgroup %d", kps);
    }
    return ksplit;
}

// ------------------------------------------------------------------------------------------ f32 GEMM
__device__ __forceinline__ int float_order_key(float f) {
    const int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7fffffff;
}

template <int MODE>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ a, int64_t lda, int64_t rpb, int64_t bs,
                                                       const float* __restrict__ w, int64_t ldw, int M, int N, int K,
                                                       float* __restrict__ c, int64_t ldc, int* __restrict__ maxbuf,
                                                       const GemmF32Epi e = GemmF32Epi()) {
    __shared__ float As[64][17];
    __shared__ float Ws[64][17];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
    const int wy = wave >> 1, wx = wave & 1;
    const int lrow = tid >> 2, kq = (tid & 3) * 4;
    int64_t m = m0 + lrow;
    if (m > M - 1) m = M - 1;
    const float* ap = a + (m / rpb) * bs + (m % rpb) * lda + kq;
    const float* wp = w + (int64_t)(n0 + lrow) * ldw + kq;  // W allocated with round_up(N, 64) rows

    floatx4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

    for (int k0 = 0; k0 < K; k0 += 16) {
        const floatx4 av = *reinterpret_cast<const floatx4*>(ap + k0);
        const floatx4 wv = *reinterpret_cast<const floatx4*>(wp + k0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            As[lrow][kq + r] = av[r];
            Ws[lrow][kq + r] = wv[r];
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            float wf[2], af[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) wf[i] = Ws[wy * 32 + i * 16 + (lane & 15)][kk * 4 + (lane >> 4)];
#pragma unroll
            for (int j = 0; j < 2; ++j) af[j] = As[wx * 32 + j * 16 + (lane & 15)][kk * 4 + (lane >> 4)];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i], af[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int64_t mm_raw = m0 + wx * 32 + j * 16 + (lane & 15);
        const bool valid = mm_raw < M;            // no early exit: the wave-level reduction below needs every lane
        const int64_t mm = valid ? mm_raw : M - 1;
        float row_max = -3.0e38f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int n = n0 + wy * 32 + i * 16 + 4 * (lane >> 4);
            if (MODE == 2) {  // f32 parity mode: bias, scale, exact GELU, positional embedding, residual; remapped C rows
                if (valid) {
                    const int64_t mb = mm / e.c_rpb, mr = mm % e.c_rpb;
                    const int64_t crow = mb * e.c_brows + mr + e.c_roff;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int nn = n + r;
                        if (nn < N) {
                            float x = acc[i][j][r];
                            if (e.bias) x += e.bias[nn];
                            if (nn < e.scale_n_end) x *= e.scale;
                            if (e.act) x = 0.5f * x * (1.f + erff(x * 0.70710678118654752f));
                            if (e.posemb) x += e.posemb[mr * N + nn];
                            if (e.residual) x += e.residual[crow * e.ldc + nn];
                            e.c[crow * e.ldc + nn] = x;
                        }
                    }
                }
            } else if (MODE == 0) {
                const int b0 = n >> 1;
                if (valid && b0 < ldc) c[mm * ldc + b0] = acc[i][j][0] * acc[i][j][0] + acc[i][j][1] * acc[i][j][1];
                if (valid && b0 + 1 < ldc) c[mm * ldc + b0 + 1] = acc[i][j][2] * acc[i][j][2] + acc[i][j][3] * acc[i][j][3];
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (valid && n + r < N) {
                        const float x = log10f(fmaxf(acc[i][j][r], 1e-10f));
                        c[mm * ldc + n + r] = x;
                        row_max = fmaxf(row_max, x);
                    }
                }
            }
        }
        if (MODE == 1) {
            // one atomic per wave when the whole wave belongs to one recording (the common case); a
            // per-lane atomicMax on one address serialises the chip (measured 9.5 ms for 32 windows)
            const int64_t bidx = mm / rpb;
            const int64_t b0 = __shfl(bidx, 0);
            if (__all(bidx == b0)) {
                float wm = row_max;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) wm = fmaxf(wm, __shfl_xor(wm, o));
                if (lane == 0 && wm > -3.0e38f) atomicMax(maxbuf + b0, float_order_key(wm));
            } else if (row_max > -3.0e38f) {
                atomicMax(maxbuf + bidx, float_order_key(row_max));
            }
        }
    }
}

void launch_gemm_f32_epi(const float* a, int64_t lda, int64_t rows_per_batch, int64_t batch_stride, const float* w, int64_t ldw,
                         int64_t M, int64_t N, int64_t K, const GemmF32Epi& e, hipStream_t s) {
    WRS_REQUIRE(K % 16 == 0 && lda % 4 == 0 && ldw % 4 == 0 && batch_stride % 4 == 0, "gemm_f32: alignment");
    WRS_REQUIRE(M > 0 && M < (1ll << 31) && N > 0 && e.c && e.ldc >= N, "gemm_f32: bad shape");
    dim3 grid((unsigned)cdiv(M, 64), (unsigned)cdiv(N, 64));
    gemm_f32_kernel<2><<<grid, 256, 0, s>>>(a, lda, rows_per_batch, batch_stride, w, ldw, (int)M, (int)N, (int)K, e.c, e.ldc, nullptr, e);
}

}  // namespace wrs
