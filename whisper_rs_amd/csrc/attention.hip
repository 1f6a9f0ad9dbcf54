// Attention kernels for gfx950 (CDNA4).  head_dim = 64 for every Whisper size.
// Replaces reference model.rs:169-201 (view/permute, q.matmul(k), softmax, matmul(v)).
//
// Layout idea (MI355X-first, not a port of a warp-32 tiling): K and V of one (batch, head) are
// stored pre-packed in MFMA operand order, so that
//   * every wave-level load is 1 KiB contiguous (lane-linear 16 B) straight into VGPRs — no LDS,
//     no swizzle, full 128-B lines from HBM (decode cross-attention streams 246 MB/audio/step at
//     large-v2, the dominant HBM stream of the whole path);
//   * S^T = K Q^T is computed with keys as MFMA rows, so the fp32 score registers of two 16-key
//     tiles ARE the B-operand fragment of the P·V product after an in-register fp16 convert: no
//     cross-lane movement, no LDS round trip for P.
// Packed formats for Tp = round_up(T, 32) keys:
//   Kp [Tp/16][2][64 lanes][8]: (kt, ks, l, j) = K[kt*16 + (l&15)][ks*32 + 8*(l>>4) + j]
//   Vp [Tp/32][4][64 lanes][8]: (kb, dt, l, j) = V[kb*32 + kperm(l>>4, j)][dt*16 + (l&15)]
//        kperm(q, j) = j < 4 ? 4q + j : 16 + 4q + (j - 4)
// (the k-order inside the P·V MFMA is permuted to match the score-register order).
#include "kernels.h"

namespace wrs {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// cross-lane adds / maxima inside an 8-lane group and between the two groups of a 16-lane row as DPP modifiers of the
// VALU instruction itself (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_ror:8) instead of ds_bpermute round trips
// through the LDS pipeline (what __shfl_xor compiles to): the decode self-attention kernel issues 15 of them per 32 positions
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// max over the lanes l, l^16, l^32, l^48 with v_permlane16_swap / v_permlane32_swap (gfx950 VALU row swaps) instead of two
// ds_bpermute round trips through the LDS pipeline
__device__ __forceinline__ float max_rows(float x) {
    const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float sum8(float x) {  // every lane gets the sum over its 8-lane group
    x += dpp_f<0xB1>(x);   // lane ^ 1
    x += dpp_f<0x4E>(x);   // lane ^ 2
    x += dpp_f<0x141>(x);  // mirror inside 8 lanes: the other quad's total
    return x;
}
size_t packed_kv_halfs_per_head(int T) { return (size_t)round_up(T, 32) * 64; }

// ------------------------------------------------------------------------------------------ pack
// grid (Tp/32, H, n_batch), 256 threads.  One block packs 32 keys of one head: K is a permuted
// 16-byte-chunk copy, V needs a transpose through LDS.
__global__ __launch_bounds__(256) void pack_kv_kernel(const half_t* __restrict__ src, int64_t ld, int64_t k_off,
                                                      int64_t v_off, int T, int H, half_t* __restrict__ kp,
                                                      half_t* __restrict__ vp) {
    __shared__ _Float16 vt[32][64 + 8];
    const int kb = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int tid = threadIdx.x;
    const int Tp = (T + 31) / 32 * 32;
    const int64_t head_base = ((int64_t)b * H + h) * (int64_t)Tp * 64;
    // K: 2 key tiles x 2 ksteps x 64 lanes = 256 chunks of 16 B
    {
        const int l = tid & 63, f = tid >> 6;  // f = tile_in_block*2 + ks
        const int kt = kb * 2 + (f >> 1), ks = f & 1;
        const int key = kt * 16 + (l & 15);
        half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (key < T) v = ld_stream(reinterpret_cast<const half8*>(src + ((int64_t)b * T + key) * ld + k_off + h * 64 + ks * 32 + 8 * (l >> 4)));
        *reinterpret_cast<half8*>(kp + head_base + ((int64_t)(kt * 2 + ks) * 64 + l) * 8) = v;
    }
    // V: stage [32 keys][64 dims] in LDS
    {
        const int key_l = tid >> 3, c = tid & 7;
        const int key = kb * 32 + key_l;
        half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (key < T) v = ld_stream(reinterpret_cast<const half8*>(src + ((int64_t)b * T + key) * ld + v_off + h * 64 + c * 8));
#pragma unroll
        for (int j = 0; j < 8; ++j) vt[key_l][c * 8 + j] = v[j];
    }
    __syncthreads();
    {
        const int l = tid & 63, dt = tid >> 6;
        const int q = l >> 4, d = dt * 16 + (l & 15);
        half8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int key_l = j < 4 ? 4 * q + j : 16 + 4 * q + (j - 4);
            v[j] = vt[key_l][d];
        }
        *reinterpret_cast<half8*>(vp + head_base + ((int64_t)(kb * 4 + dt) * 64 + l) * 8) = v;
    }
}

void launch_pack_kv(const half_t* src, int64_t ld, int64_t k_off, int64_t v_off, int n_batch, int T, int H, half_t* kp,
                    half_t* vp, hipStream_t s) {
    WRS_REQUIRE(ld % 8 == 0 && k_off % 8 == 0 && v_off % 8 == 0, "pack_kv: alignment");
    dim3 grid((unsigned)cdiv(T, 32), (unsigned)H, (unsigned)n_batch);
    pack_kv_kernel<<<grid, 256, 0, s>>>(src, ld, k_off, v_off, T, H, kp, vp);
}

// ------------------------------------------------------------------------------------------ MFMA attention
// One wave = 16 query columns.  SPLIT = false: the wave walks all key blocks (encoder; 4 waves of a
// block take 4 consecutive query tiles).  SPLIT = true: the 4 waves of a block share ONE query
// tile (the G beams of one audio) and each takes every 4th key block; partial (m, l, O) are merged
// through LDS (flash-decoding inside the workgroup).
struct AttnParams {
    const half_t* q;
    int64_t ldq;
    const half_t* kp;
    const half_t* vp;
    int T, Tp, H;
    int q_rows_per_batch;  // rows per batch item in q / out
    int q_valid;           // SPLIT: number of valid query rows (G); else unused
    half_t* out;
    int64_t ldo;
    // SPLIT only: Q = (sum_s q_slab[s][row][:] + q_bias) * q_scale, the un-reduced split-K output of the
    // cross-attention query projection (fuses the slab reduction + bias + scale into this kernel)
    const float* q_slab;
    int q_ksplit, q_m_pad;
    const float* q_bias;
    float q_scale;
    int n_units;  // !SPLIT: n_batch * H
    int causal;   // !SPLIT: query i of a batch item attends to keys 0..i only (decoder prompt prefill, reference model.rs:191-201 mask)
};

#ifndef XA_WAVES
#define XA_WAVES 8   // waves per workgroup in the SPLIT (decode cross-attention) kernel (measured: 2/4/8/12/16 -> 90/94/77-82/90/95 us)
#endif
#ifndef ENC_ATTN_NQ
#define ENC_ATTN_NQ 2  // query tiles per wave in the encoder attention (measured 1 / 2 / 4: 392 / 408 / 300 TFLOP/s; 4 drops to one wave per SIMD)
#endif
#ifndef XA_NT
#define XA_NT 1      // non-temporal K/V loads in the SPLIT kernel: every byte is read once per launch (measured -6 us)
#endif
template <bool SPLIT>
__device__ __forceinline__ half8 kv_load(const half_t* ptr) {
#if XA_NT
    if (SPLIT) return __builtin_nontemporal_load(reinterpret_cast<const half8*>(ptr));
#endif
    return *reinterpret_cast<const half8*>(ptr);
}

// NQ (encoder only): query tiles of 16 per wave; every K/V fragment fetched is used NQ times from registers.  The kernel
// turned out to be bound by the softmax VALU work per (query tile, key block), not by the loads: NQ = 2 gains 4 %.
// XW (SPLIT only): waves per workgroup
template <bool SPLIT, int NQ, int XW = XA_WAVES>
__global__ __launch_bounds__(SPLIT ? XW * 64 : 256) void attn_mfma_kernel(const AttnParams p) {
    static_assert(!SPLIT || NQ == 1, "the decode kernel has one query tile (the beams of an audio)");
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    int h, b, qblk = 0;
    if (SPLIT) {
        h = blockIdx.y;
        b = blockIdx.z;
    } else {
        // 1-D grid, XCD-aware: all query blocks of one (batch, head) run on ONE XCD so its packed K/V (384 KiB) is
        // fetched into one L2 once instead of by all 8 (measured FETCH_SIZE 8x the K/V bytes before).  Speed only.
        const int wq = (p.q_rows_per_batch + 64 * NQ - 1) / (64 * NQ);  // query blocks per (batch, head)
        const int j = (int)(blockIdx.x >> 3);
        const int unit = (j / wq) * 8 + (int)(blockIdx.x & 7);
        if (unit >= p.n_units) return;
        qblk = j % wq;
        h = unit % p.H;
        b = unit / p.H;
    }
    const int qcol = lane & 15, qg = lane >> 4;

    int q0, n_valid;  // n_valid: valid query rows of this wave counted from q0 (over its NQ tiles)
    if (SPLIT) {
        q0 = 0;
        n_valid = p.q_valid;
    } else {
        q0 = (qblk * 4 + wave) * 16 * NQ;
        n_valid = p.q_rows_per_batch - q0;
        if (n_valid <= 0) return;  // whole wave out of range (no barriers in the non-split kernel)
        if (n_valid > 16 * NQ) n_valid = 16 * NQ;
    }
    const int64_t qrow0 = (int64_t)b * p.q_rows_per_batch + q0;

    // Q^T as B operand: lane holds Q[q0 + 16 t + qcol][ks*32 + 8*qg .. +7]
    half8 qf[NQ][2];
#pragma unroll
    for (int t = 0; t < NQ; ++t)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (t * 16 + qcol < n_valid) {
                if (SPLIT && p.q_slab) {
                    const int N = p.H * 64;
                    const int col = h * 64 + ks * 32 + 8 * qg;
                    const float* src = p.q_slab + (qrow0 + qcol) * N + col;
                    floatx4 a0 = *reinterpret_cast<const floatx4*>(src), a1 = *reinterpret_cast<const floatx4*>(src + 4);
                    for (int sidx = 1; sidx < p.q_ksplit; ++sidx) {
                        a0 += *reinterpret_cast<const floatx4*>(src + (int64_t)sidx * p.q_m_pad * N);
                        a1 += *reinterpret_cast<const floatx4*>(src + (int64_t)sidx * p.q_m_pad * N + 4);
                    }
                    a0 += *reinterpret_cast<const floatx4*>(p.q_bias + col);
                    a1 += *reinterpret_cast<const floatx4*>(p.q_bias + col + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        v[j] = (_Float16)(a0[j] * p.q_scale);
                        v[4 + j] = (_Float16)(a1[j] * p.q_scale);
                    }
                } else {
                    v = *reinterpret_cast<const half8*>(p.q + (qrow0 + t * 16 + qcol) * p.ldq + h * 64 + ks * 32 + 8 * qg);
                }
            }
            qf[t][ks] = v;
        }

    const int64_t head_base = ((int64_t)b * p.H + h) * (int64_t)p.Tp * 64;
    const half_t* kbase = p.kp + head_base + lane * 8;
    const half_t* vbase = p.vp + head_base + lane * 8;
    int n_blocks = p.Tp / 32;
    if (!SPLIT && p.causal) n_blocks = min(n_blocks, (q0 + 16 * NQ - 1) / 32 + 1);  // key blocks past the wave's last query are fully masked
    const int kb_begin = SPLIT ? wave : 0, kb_step = SPLIT ? XW : 1;

    floatx4 o[NQ][4];
    float m_run[NQ], l_run[NQ];
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[t][dt] = floatx4{0.f, 0.f, 0.f, 0.f};
        m_run[t] = -INFINITY;
        l_run[t] = 0.f;
    }

    half8 kf[4], vf[4];
    if (kb_begin < n_blocks) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            kf[i] = kv_load<SPLIT>(kbase + ((int64_t)kb_begin * 4 + i) * 512);
            vf[i] = kv_load<SPLIT>(vbase + ((int64_t)kb_begin * 4 + i) * 512);
        }
    }
    for (int kb = kb_begin; kb < n_blocks; kb += kb_step) {
        half8 kc[4], vc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            kc[i] = kf[i];
            vc[i] = vf[i];
        }
        const int nxt = kb + kb_step;
        if (nxt < n_blocks) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                kf[i] = kv_load<SPLIT>(kbase + ((int64_t)nxt * 4 + i) * 512);
                vf[i] = kv_load<SPLIT>(vbase + ((int64_t)nxt * 4 + i) * 512);
            }
        }
        const bool tail = kb * 32 + 32 > p.T;
#pragma unroll
        for (int t = 0; t < NQ; ++t) {
            // S^T tiles: rows = keys, cols = queries.  kc[tile*2 + ks]
            floatx4 s0 = floatx4{0.f, 0.f, 0.f, 0.f}, s1 = floatx4{0.f, 0.f, 0.f, 0.f};
            s0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(kc[0], qf[t][0], s0, 0, 0, 0);
            s0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(kc[1], qf[t][1], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(kc[2], qf[t][0], s1, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(kc[3], qf[t][1], s1, 0, 0, 0);
            float sv[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
            if (tail) {  // tail block: mask padded keys
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int key = kb * 32 + (j < 4 ? 4 * qg + j : 16 + 4 * qg + (j - 4));
                    if (key >= p.T) sv[j] = -INFINITY;
                }
            }
            if (!SPLIT && p.causal && kb * 32 + 31 > q0 + t * 16) {  // block reaches past the tile's first query: mask key > query
                const int qi = q0 + t * 16 + qcol;                    // (block 0 always holds key 0 <= query: the running max stays finite)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int key = kb * 32 + (j < 4 ? 4 * qg + j : 16 + 4 * qg + (j - 4));
                    if (key > qi) sv[j] = -INFINITY;
                }
            }
            float bm = fmaxf(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])), fmaxf(fmaxf(sv[4], sv[5]), fmaxf(sv[6], sv[7])));
            bm = fmaxf(bm, __shfl_xor(bm, 16));
            bm = fmaxf(bm, __shfl_xor(bm, 32));
            const float m_new = fmaxf(m_run[t], bm);  // every block holds >= 1 valid key, so m_new is finite
            const float alpha = __expf(m_run[t] - m_new);
            float ps = 0.f;
            half8 pf;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float e = __expf(sv[j] - m_new);
                ps += e;
                pf[j] = (_Float16)e;
            }
            ps += __shfl_xor(ps, 16);
            ps += __shfl_xor(ps, 32);
            l_run[t] = l_run[t] * alpha + ps;
            m_run[t] = m_new;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                o[t][dt] *= alpha;
                o[t][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vc[dt], pf, o[t][dt], 0, 0, 0);
            }
        }
    }

    if (!SPLIT) {
#pragma unroll
        for (int t = 0; t < NQ; ++t) {
            if (t * 16 + qcol < n_valid) {
                const float inv = 1.f / l_run[t];
                half_t* dst = p.out + (qrow0 + t * 16 + qcol) * p.ldo + h * 64 + 4 * qg;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
                    *reinterpret_cast<half4*>(dst + dt * 16) = half4{(_Float16)(o[t][dt][0] * inv), (_Float16)(o[t][dt][1] * inv),
                                                                     (_Float16)(o[t][dt][2] * inv), (_Float16)(o[t][dt][3] * inv)};
            }
        }
        return;
    } else {
        __shared__ float sm[XW][16];
        __shared__ float sl[XW][16];
        __shared__ float so[XW][64][17];
        if (qg == 0) {
            sm[wave][qcol] = m_run[0];
            sl[wave][qcol] = l_run[0];
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) so[wave][dt * 16 + 4 * qg + r][qcol] = o[0][dt][r];
        __syncthreads();
        // 16 queries x 64 dims = 1024 outputs, 4 consecutive dims per thread (threads 0..255)
        const int qq = tid >> 4, d0 = (tid & 15) * 4;
        if (tid < 256 && qq < n_valid) {
            float mm = sm[0][qq];
#pragma unroll
            for (int w = 1; w < XW; ++w) mm = fmaxf(mm, sm[w][qq]);
            float wgt[XW], lsum = 0.f;
#pragma unroll
            for (int w = 0; w < XW; ++w) {
                wgt[w] = __expf(sm[w][qq] - mm);  // waves that saw no key block hold m = -inf, l = 0 -> weight 0
                lsum += wgt[w] * sl[w][qq];
            }
            const float inv = 1.f / lsum;
            float r4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float a = 0.f;
#pragma unroll
                for (int w = 0; w < XW; ++w) a += wgt[w] * so[w][d0 + r][qq];
                r4[r] = a * inv;
            }
            // decode output feeds the out-projection GEMM: packed activation order (K = H*64)
            *reinterpret_cast<half4*>(p.out + pa_off((int)(qrow0 + qq), h * 64 + d0, p.H * 2)) =
                half4{(_Float16)r4[0], (_Float16)r4[1], (_Float16)r4[2], (_Float16)r4[3]};
        }
    }
}

// ------------------------------------------------------------------------------------------ encoder attention, K/V shared through LDS
// attn_mfma_kernel<false> above lets every wave stream the whole packed K/V of its (batch, head) from L2 by itself: with
// T = 1500 keys that is 385 KB per wave, 47 waves per (batch, head), 92 GB of L2->CU traffic per encoder layer at 256
// windows — the kernel ran at 420 TFLOP/s (17 % of the MFMA peak) bound by that traffic, not by its arithmetic.  Here the 8
// waves of a workgroup (256 queries of one (batch, head)) share every 32-key block: each wave fetches ONE of the block's 8
// fragments (4 K + 4 V, 1 KiB each, already in MFMA operand order) with a direct-to-LDS load into a 3-slot ring, one
// barrier per key block, and all 8 waves read the fragments back with lane-linear ds_read_b128.  L2->CU traffic drops 8x;
// the running-max rescale of the output accumulators is skipped while the maximum does not move (wave-uniform test).
// Same arithmetic as above (S^T = K Q^T on MFMA, score registers reused as the P.V operand, online softmax in fp32).
// Also serves the decoder's prompt prefill: causal self-attention (Tq == T) and cross-attention (Tq = prompt length).
template <int NQ>
__global__ __launch_bounds__(512) void attn_enc_kernel(const AttnParams p) {
    __shared__ half8 ring[3][8][64];  // [slot][K fragments 0..3 | V fragments 4..7][lane]: 24 KiB
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // 1-D grid, XCD-aware: all query blocks of one (batch, head) run on ONE XCD so its packed K/V (384 KiB) is fetched into one L2 once
    const int wq = (p.q_rows_per_batch + 128 * NQ - 1) / (128 * NQ);  // query blocks (8 waves x 16 NQ queries) per (batch, head)
    const int j = (int)(blockIdx.x >> 3);
    const int unit = (j / wq) * 8 + (int)(blockIdx.x & 7);
    if (unit >= p.n_units) return;  // whole workgroup, before any barrier
    const int qblk = j % wq;
    const int h = unit % p.H, b = unit / p.H;
    const int qcol = lane & 15, qg = lane >> 4;
    const int q0 = (qblk * 8 + wave) * 16 * NQ;
    int n_valid = p.q_rows_per_batch - q0;  // valid query rows of this wave (<= 0: the wave only helps staging)
    if (n_valid > 16 * NQ) n_valid = 16 * NQ;
    const int64_t qrow0 = (int64_t)b * p.q_rows_per_batch + q0;

    half8 qf[NQ][2];
#pragma unroll
    for (int t = 0; t < NQ; ++t)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (t * 16 + qcol < n_valid) v = *reinterpret_cast<const half8*>(p.q + (qrow0 + t * 16 + qcol) * p.ldq + h * 64 + ks * 32 + 8 * qg);
            qf[t][ks] = v;
        }
    const int64_t head_base = ((int64_t)b * p.H + h) * (int64_t)p.Tp * 64;
    // fragment this wave stages for every key block: waves 0..3 -> K fragment `wave`, waves 4..7 -> V fragment `wave - 4`
    const half_t* my_frag = (wave < 4 ? p.kp : p.vp) + head_base + (int64_t)(wave & 3) * 512 + lane * 8;
    int n_blocks = p.Tp / 32;
    if (p.causal) n_blocks = min(n_blocks, (min(p.q_rows_per_batch, (qblk + 1) * 128 * NQ) - 1) / 32 + 1);  // past the workgroup's last query: masked
    const int my_blocks = (p.causal && n_valid > 0) ? min(n_blocks, (q0 + n_valid - 1) / 32 + 1) : n_blocks;

    floatx4 o[NQ][4], lsum[NQ];  // lsum: the softmax denominator, accumulated by a fifth MFMA against a fragment of ones
    float m_run[NQ];
    const half8 ones = {1, 1, 1, 1, 1, 1, 1, 1};
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[t][dt] = floatx4{0.f, 0.f, 0.f, 0.f};
        lsum[t] = floatx4{0.f, 0.f, 0.f, 0.f};
        m_run[t] = -INFINITY;
    }
#define WRS_ENC_ISSUE(kb_)                                                                                                     \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(my_frag + (int64_t)(kb_) * 2048),           \
                                     (__attribute__((address_space(3))) void*)&ring[(kb_) % 3][wave][0], 16, 0, 0)
    WRS_ENC_ISSUE(0);
    if (n_blocks > 1) WRS_ENC_ISSUE(1);
    for (int kb = 0; kb < n_blocks; ++kb) {
        // block kb has landed (the younger request of block kb+1 stays in flight); every wave is past compute(kb-1)
        if (kb + 1 < n_blocks) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (kb + 2 < n_blocks) WRS_ENC_ISSUE(kb + 2);  // into the slot of block kb-1, which nobody reads any more
        if (n_valid <= 0 || kb >= my_blocks) continue;
        half8 kc[4], vc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            kc[i] = ring[kb % 3][i][lane];
            vc[i] = ring[kb % 3][4 + i][lane];
        }
        const bool tail = kb * 32 + 32 > p.T;
#pragma unroll
        for (int t = 0; t < NQ; ++t) {
            floatx4 s0 = floatx4{0.f, 0.f, 0.f, 0.f}, s1 = floatx4{0.f, 0.f, 0.f, 0.f};
            s0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(kc[0], qf[t][0], s0, 0, 0, 0);
            s0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(kc[1], qf[t][1], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(kc[2], qf[t][0], s1, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(kc[3], qf[t][1], s1, 0, 0, 0);
            float sv[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
            if (tail) {  // tail block: mask padded keys
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const int key = kb * 32 + (jj < 4 ? 4 * qg + jj : 16 + 4 * qg + (jj - 4));
                    if (key >= p.T) sv[jj] = -INFINITY;
                }
            }
            if (p.causal && kb * 32 + 31 > q0 + t * 16) {  // block reaches past the tile's first query: mask key > query
                const int qi = q0 + t * 16 + qcol;           // (block 0 always holds key 0 <= query: the running max stays finite)
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const int key = kb * 32 + (jj < 4 ? 4 * qg + jj : 16 + 4 * qg + (jj - 4));
                    if (key > qi) sv[jj] = -INFINITY;
                }
            }
            float bm = fmaxf(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])), fmaxf(fmaxf(sv[4], sv[5]), fmaxf(sv[6], sv[7])));
            bm = max_rows(bm);
            const float m_new = fmaxf(m_run[t], bm);
            if (__any(m_new != m_run[t])) {  // the running maximum moved for some query of the tile: rescale (rare after the first blocks)
                const float alpha = __expf(m_run[t] - m_new);
                lsum[t] *= alpha;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) o[t][dt] *= alpha;
                m_run[t] = m_new;
            }
            half8 pf;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) pf[jj] = (_Float16)__expf(sv[jj] - m_new);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[t][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vc[dt], pf, o[t][dt], 0, 0, 0);
            // sum_k P[k][q] for every output row: each lane ends up with its query column's denominator (of the fp16-rounded
            // weights, i.e. exactly the weights the P.V product used)
            lsum[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, pf, lsum[t], 0, 0, 0);
        }
    }
#undef WRS_ENC_ISSUE
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
        if (t * 16 + qcol < n_valid) {
            const float inv = 1.f / lsum[t][0];
            half_t* dst = p.out + (qrow0 + t * 16 + qcol) * p.ldo + h * 64 + 4 * qg;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                *reinterpret_cast<half4*>(dst + dt * 16) = half4{(_Float16)(o[t][dt][0] * inv), (_Float16)(o[t][dt][1] * inv),
                                                                 (_Float16)(o[t][dt][2] * inv), (_Float16)(o[t][dt][3] * inv)};
        }
    }
}

void launch_attn_full(const half_t* q, int64_t ldq, const half_t* kp, const half_t* vp, int n_batch, int Tq, int T, int H,
                      half_t* out, int64_t ldo, hipStream_t s, bool causal) {
    WRS_REQUIRE(ldq % 8 == 0 && ldo % 4 == 0 && T >= 1 && Tq >= 1, "attn_full: bad shape");
    WRS_REQUIRE(!causal || Tq == T, "attn_full: causal attention needs as many keys as queries");
    AttnParams p{q, ldq, kp, vp, T, (int)round_up(T, 32), H, Tq, 0, out, ldo, nullptr, 0, 0, nullptr, 1.f, n_batch * H, causal ? 1 : 0};
    const int64_t units8 = cdiv((int64_t)n_batch * H, 8);
    static const int nq = exp_env("WRS_ATTN_NQ") ? atoi(exp_env("WRS_ATTN_NQ")) : ENC_ATTN_NQ;
    static const int lds_mode = exp_env("WRS_ATTN_LDS") ? atoi(exp_env("WRS_ATTN_LDS")) : 2;  // 0: per-wave K/V streams (round 1); 1/2/4: LDS-shared, NQ tiles per wave
    if (lds_mode == 4) { attn_enc_kernel<4><<<(unsigned)(8 * units8 * cdiv(Tq, 512)), 512, 0, s>>>(p); return; }
    if (lds_mode == 2) { attn_enc_kernel<2><<<(unsigned)(8 * units8 * cdiv(Tq, 256)), 512, 0, s>>>(p); return; }
    if (lds_mode == 1) { attn_enc_kernel<1><<<(unsigned)(8 * units8 * cdiv(Tq, 128)), 512, 0, s>>>(p); return; }
    if (nq == 4) attn_mfma_kernel<false, 4><<<(unsigned)(8 * units8 * cdiv(Tq, 256)), 256, 0, s>>>(p);
    else if (nq == 2) attn_mfma_kernel<false, 2><<<(unsigned)(8 * units8 * cdiv(Tq, 128)), 256, 0, s>>>(p);
    else attn_mfma_kernel<false, 1><<<(unsigned)(8 * units8 * cdiv(Tq, 64)), 256, 0, s>>>(p);
}

void launch_attn_cross(const float* q_slab, int q_ksplit, int q_m_pad, const float* q_bias, float q_scale, const half_t* kp,
                       const half_t* vp, int n_audio, int G, int T, int H, half_t* out, hipStream_t s) {
    WRS_REQUIRE(G >= 1 && G <= 16, "attn_cross: group size %d not in 1..16", G);
    WRS_REQUIRE(T >= 1 && q_slab && q_bias && q_ksplit >= 1, "attn_cross: bad argument");
    AttnParams p{nullptr, 0, kp, vp, T, (int)round_up(T, 32), H, G, G, out, 0, q_slab, q_ksplit, q_m_pad, q_bias, q_scale, 0, 0};
    dim3 grid(1, (unsigned)H, (unsigned)n_audio);
    // (16 waves per workgroup when the launch has only a window's 20 workgroups: 14.4 us against 12.9 us, round 3: not used)
    attn_mfma_kernel<true, 1><<<grid, XA_WAVES * 64, 0, s>>>(p);
}

// ------------------------------------------------------------------------------------------ decode self-attention
// One wave per (row, head).  Fuses, for the new token: the split-K reduction of the QKV projection
// (slabs [ksplit][m_pad][3D]) + bias + the (D/H)^-0.25 scale on q and k (reference model.rs:159-182), the
// append of k/v to the cache (Cached::forward, model.rs:83-87) and the attention over the
// ancestry-indirected cache kc/vc [phys_row][n_ctx][D] (Cached::update_cache, model.rs:53-62).
// Lanes = 8 position groups x 8 dim chunks (16 B each): every wave load instruction fetches 8 full 128-B
// lines; 4 such loads are kept in flight.  Length <= 448, so scores live in LDS.
struct SelfAttnParams {
    const float* slab;
    int ksplit, m_pad;
    const float* bias;  // [3D]
    float scale;
    half_t* kc;
    half_t* vc;
    const int* anc_base;
    int64_t anc_buf_stride;
    int n_ctx, n_rows, phys_mul, H, D;
    const StepParams* sp;
    half_t* out;  // packed activation order
    int rows_per_audio;  // beams of one audio are consecutive rows and share most of their cached history
    // q16 != null: the new token's q is fp16 [n_rows, D] with bias and scale applied, and its k / v are ALREADY in the cache
    // at sp->cache_pos (written by the QKV GEMM's epilogue); slab / bias are unused
    const half_t* q16;
};

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef int intx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void slab_chunk8(const float* __restrict__ slab, int ksplit, int64_t slab_stride, int64_t off,
                                            const float* __restrict__ bias, int col, float (&o)[8]) {
    floatx4 a0 = *reinterpret_cast<const floatx4*>(slab + off), a1 = *reinterpret_cast<const floatx4*>(slab + off + 4);
    for (int sidx = 1; sidx < ksplit; ++sidx) {
        a0 += *reinterpret_cast<const floatx4*>(slab + sidx * slab_stride + off);
        a1 += *reinterpret_cast<const floatx4*>(slab + sidx * slab_stride + off + 4);
    }
    a0 += *reinterpret_cast<const floatx4*>(bias + col);
    a1 += *reinterpret_cast<const floatx4*>(bias + col + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        o[j] = a0[j];
        o[4 + j] = a1[j];
    }
}

// PSPLIT = false: one workgroup = (up to) 4 heads of ONE row, a wave per head.  PSPLIT = true (a window or two: a few dozen rows):
// one workgroup = ONE (row, head), its 4 waves take every 4th block of 32 cached positions and are merged through LDS — a wave's
// walk is a chain of dependent memory round trips (one per 32 positions), 4 waves cut it to a quarter.
template <bool PSPLIT>
__global__ __launch_bounds__(256) void attn_self_kernel(const SelfAttnParams p) {
    __shared__ int anc_s[512];
    __shared__ float ps_m[4], ps_l[4], ps_o[4][64];  // PSPLIT: the waves' partial (max, denominator, output)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int hpb = (p.H + 3) >> 2;                 // head groups per row
    int row, h;
    if (PSPLIT) {
        row = (int)blockIdx.x / p.H;
        h = (int)blockIdx.x % p.H;
    } else {
        // XCD-aware order: the G beams of one (audio, head group) run back-to-back on ONE XCD, so the K/V rows they
        // share through the ancestry table (common beam prefixes) are fetched into that L2 once.  Speed only.
        const int G = p.rows_per_audio;
        const int j = (int)(blockIdx.x >> 3);
        const int unit = (j / G) * 8 + (int)(blockIdx.x & 7);  // (audio, head group)
        if (unit >= (p.n_rows / G) * hpb) return;               // whole workgroup, before any barrier
        row = (unit / hpb) * G + j % G;
        h = (unit % hpb) * 4 + wave;
    }
    const int g = lane >> 3, c = lane & 7;
    const int D = p.D;
    const int cpos = p.sp->cache_pos;  // the new token's slot; cached positions are 0..cpos-1
    const int phys = row * p.phys_mul;
    const int* anc = p.anc_base + (int64_t)p.sp->parity * p.anc_buf_stride + (int64_t)phys * p.n_ctx;
    for (int i = threadIdx.x; i < cpos; i += 256) anc_s[i] = anc[i];
    const int64_t row_stride = (int64_t)p.n_ctx * D;
    const int col = h * 64 + c * 8;
    if (h >= p.H) {  // surplus wave of the last head group: only takes part in the barrier
        __syncthreads();
        return;
    }

    // new token: q, k, v chunk c — from the cache / the fp16 query buffer, or from the QKV slabs
    float qv[8], kn[8], vn[8];
    if (p.q16) {
        const half8 q8 = *reinterpret_cast<const half8*>(p.q16 + (int64_t)row * D + col);
        const half8 k8 = *reinterpret_cast<const half8*>(p.kc + (int64_t)phys * row_stride + (int64_t)cpos * D + col);
        const half8 v8 = *reinterpret_cast<const half8*>(p.vc + (int64_t)phys * row_stride + (int64_t)cpos * D + col);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            qv[j] = (float)q8[j];
            kn[j] = (float)k8[j];
            vn[j] = (float)v8[j];
        }
    } else {
        const int64_t ss = (int64_t)p.m_pad * 3 * D, off = (int64_t)row * 3 * D + col;
        slab_chunk8(p.slab, p.ksplit, ss, off, p.bias, col, qv);
        slab_chunk8(p.slab, p.ksplit, ss, off + D, p.bias, D + col, kn);
        slab_chunk8(p.slab, p.ksplit, ss, off + 2 * D, p.bias, 2 * D + col, vn);
        half8 k16, v16;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            qv[j] *= p.scale;
            k16[j] = (_Float16)(kn[j] * p.scale);
            v16[j] = (_Float16)vn[j];
            kn[j] = (float)k16[j];  // use exactly what later steps will read back from the fp16 cache
            vn[j] = (float)v16[j];
        }
        if (g == 0 && (!PSPLIT || wave == 0)) {
            *reinterpret_cast<half8*>(p.kc + (int64_t)phys * row_stride + (int64_t)cpos * D + col) = k16;
            *reinterpret_cast<half8*>(p.vc + (int64_t)phys * row_stride + (int64_t)cpos * D + col) = v16;
        }
    }
    __syncthreads();  // anc_s ready

    // single pass, online softmax.  State per lane: running max m (wave-uniform), partial denominator l and
    // partial output acc[8] over the positions this lane's group g has seen; groups are merged at the end.
    // q is rounded to fp16 (the cross-attention kernel does the same): scores are v_dot2_f32_f16 products, fp32 accumulate
    half2_t qh[4];
    float s_new = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        qh[j] = half2_t{(_Float16)qv[2 * j], (_Float16)qv[2 * j + 1]};
        s_new += (float)qh[j][0] * kn[2 * j] + (float)qh[j][1] * kn[2 * j + 1];
    }
    s_new = sum8(s_new);
    const bool seed = !PSPLIT || wave == 0;  // the new position is counted once: by group 0 (of wave 0)
    // the new position seeds the running max (weight exp(0) = 1); a wave without it starts from a finite floor so that
    // exp(m_run - m_new) is exp(-huge) = 0, never exp(-inf + inf)
    float m_run = seed ? s_new : -1.0e30f;
    float l_run = (seed && g == 0) ? 1.f : 0.f;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = (seed && g == 0) ? vn[j] : 0.f;

    const unsigned row_stride_b = (unsigned)(p.n_ctx * D * 2);
    for (int p0 = PSPLIT ? wave * 32 : 0; p0 < cpos; p0 += PSPLIT ? 128 : 32) {
        half8 kv[4], vv[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int pos = p0 + u * 8 + g;
            ok[u] = pos < cpos;
            kv[u] = half8{0, 0, 0, 0, 0, 0, 0, 0};
            vv[u] = half8{0, 0, 0, 0, 0, 0, 0, 0};
            if (ok[u]) {
                // 32-bit byte offset from the (wave-uniform) cache base: one v_mad_u32_u24 + add per load instead of 64-bit
                // pointer arithmetic (a layer's cache is < 4 GiB: checked by the launcher)
                const unsigned boff = (unsigned)anc_s[pos] * row_stride_b + (unsigned)pos * (unsigned)(D * 2) + (unsigned)(col * 2);
                kv[u] = ld_stream(reinterpret_cast<const half8*>(reinterpret_cast<const char*>(p.kc) + boff));
                vv[u] = ld_stream(reinterpret_cast<const half8*>(reinterpret_cast<const char*>(p.vc) + boff));
            }
        }
        float t[4];
        float cm = -INFINITY;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float x = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) x = __builtin_amdgcn_fdot2(qh[j], half2_t{kv[u][2 * j], kv[u][2 * j + 1]}, x, false);
            x = sum8(x);
            t[u] = ok[u] ? x : -INFINITY;
            cm = fmaxf(cm, t[u]);
        }
        cm = fmaxf(cm, dpp_f<0x128>(cm));  // row_ror:8: the other 8-lane group of the 16-lane row
        cm = max_rows(cm);
        const float m_new = fmaxf(m_run, cm);
        const float alpha = __expf(m_run - m_new);
        l_run *= alpha;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] *= alpha;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float w = __expf(t[u] - m_new);  // exp(-inf) = 0 for padded slots
            l_run += w;
            // acc += w * v with the fp16 -> fp32 conversion inside the FMA (v_fma_mix_f32: 8 instructions instead of 8 cvt + 8 fma)
            // (the s_nop covers the transcendental-result hazard: w comes straight from v_exp_f32 and the compiler does not
            // see VALU consumers inside inline asm)
            const intx4 vi = __builtin_bit_cast(intx4, vv[u]);
            asm("s_nop 1\n\t"
                "v_fma_mix_f32 %0, %8, %12, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
                "v_fma_mix_f32 %1, %8, %12, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                "v_fma_mix_f32 %2, %9, %12, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
                "v_fma_mix_f32 %3, %9, %12, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                "v_fma_mix_f32 %4, %10, %12, %4 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
                "v_fma_mix_f32 %5, %10, %12, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                "v_fma_mix_f32 %6, %11, %12, %6 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
                "v_fma_mix_f32 %7, %11, %12, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7])
                : "v"(vi[0]), "v"(vi[1]), "v"(vi[2]), "v"(vi[3]), "v"(w));
        }
        m_run = m_new;
    }
    // merge the 8 position groups (lanes differing in bits 3..5); l_run is replicated over c, so reduce over g only
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {
        l_run += __shfl_xor(l_run, o);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += __shfl_xor(acc[j], o);
    }
    if (PSPLIT) {  // the 4 waves' partial results -> wave 0
        if (g == 0) {
            if (c == 0) { ps_m[wave] = m_run; ps_l[wave] = l_run; }
#pragma unroll
            for (int j = 0; j < 8; ++j) ps_o[wave][c * 8 + j] = acc[j];
        }
        __syncthreads();
        if (wave != 0) return;
        const float mm = fmaxf(fmaxf(ps_m[0], ps_m[1]), fmaxf(ps_m[2], ps_m[3]));  // wave 0 holds the new position: finite
        l_run = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float wgt = __expf(ps_m[w] - mm);
            l_run += wgt * ps_l[w];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += wgt * ps_o[w][c * 8 + j];
        }
    }
    if (g == 0) {
        const float inv = 1.f / l_run;
        half8 r;
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = (_Float16)(acc[j] * inv);
        *reinterpret_cast<half8*>(p.out + pa_off(row, col, D >> 5)) = r;  // packed activation order
    }
}

// rows at or below this count take the position-split form (one workgroup per (row, head)); WRS_SELF_PSPLIT_ROWS (test build) overrides
static constexpr int SELF_PSPLIT_MAX_ROWS = 40;
static int self_psplit_max_rows() {
    static const int v = exp_env("WRS_SELF_PSPLIT_ROWS") ? atoi(exp_env("WRS_SELF_PSPLIT_ROWS")) : SELF_PSPLIT_MAX_ROWS;
    return v;
}

void launch_attn_self(const float* qkv_slab, int ksplit, int m_pad, const float* qkv_bias, float scale, half_t* kc, half_t* vc,
                      const int* anc_base, int64_t anc_buf_stride, int n_ctx, int n_rows, int rows_per_audio, int phys_mul, int H, int D,
                      const StepParams* sp, half_t* out, hipStream_t s) {
    WRS_REQUIRE(n_ctx <= 512 && D == H * 64 && qkv_slab && qkv_bias && ksplit >= 1, "attn_self: bad shape n_ctx=%d D=%d H=%d", n_ctx, D, H);
    WRS_REQUIRE((int64_t)n_rows * phys_mul * n_ctx * D * 2 < (1ll << 32), "attn_self: a layer's K/V cache must stay below 4 GiB (32-bit offsets)");
    WRS_REQUIRE(rows_per_audio >= 1 && n_rows % rows_per_audio == 0, "attn_self: rows %d not a multiple of %d", n_rows, rows_per_audio);
    SelfAttnParams p{qkv_slab, ksplit, m_pad, qkv_bias, scale, kc, vc, anc_base, anc_buf_stride, n_ctx, n_rows, phys_mul, H, D, sp, out, rows_per_audio, nullptr};
    if (n_rows <= self_psplit_max_rows()) {
        attn_self_kernel<true><<<(unsigned)(n_rows * H), 256, 0, s>>>(p);
        return;
    }
    const int64_t units = (int64_t)(n_rows / rows_per_audio) * ((H + 3) / 4);
    const unsigned grid = (unsigned)(8 * cdiv(units, 8) * rows_per_audio);
    attn_self_kernel<false><<<grid, 256, 0, s>>>(p);
}

void launch_attn_self_q16(const half_t* q16, half_t* kc, half_t* vc, const int* anc_base, int64_t anc_buf_stride, int n_ctx, int n_rows,
                          int rows_per_audio, int phys_mul, int H, int D, const StepParams* sp, half_t* out, hipStream_t s) {
    WRS_REQUIRE(n_ctx <= 512 && D == H * 64 && q16, "attn_self: bad shape n_ctx=%d D=%d H=%d", n_ctx, D, H);
    WRS_REQUIRE((int64_t)n_rows * phys_mul * n_ctx * D * 2 < (1ll << 32), "attn_self: a layer's K/V cache must stay below 4 GiB (32-bit offsets)");
    WRS_REQUIRE(rows_per_audio >= 1 && n_rows % rows_per_audio == 0, "attn_self: rows %d not a multiple of %d", n_rows, rows_per_audio);
    SelfAttnParams p{nullptr, 1, 0, nullptr, 1.f, kc, vc, anc_base, anc_buf_stride, n_ctx, n_rows, phys_mul, H, D, sp, out, rows_per_audio, q16};
    if (n_rows <= self_psplit_max_rows()) {
        attn_self_kernel<true><<<(unsigned)(n_rows * H), 256, 0, s>>>(p);
        return;
    }
    const int64_t units = (int64_t)(n_rows / rows_per_audio) * ((H + 3) / 4);
    const unsigned grid = (unsigned)(8 * cdiv(units, 8) * rows_per_audio);
    attn_self_kernel<false><<<grid, 256, 0, s>>>(p);
}

void launch_attn_cross_q16(const half_t* q16, const half_t* kp, const half_t* vp, int n_audio, int G, int T, int H, half_t* out, hipStream_t s) {
    WRS_REQUIRE(G >= 1 && G <= 16, "attn_cross: group size %d not in 1..16", G);
    WRS_REQUIRE(T >= 1 && q16, "attn_cross: bad argument");
    AttnParams p{q16, (int64_t)H * 64, kp, vp, T, (int)round_up(T, 32), H, G, G, out, 0, nullptr, 0, 0, nullptr, 1.f, 0, 0};
    dim3 grid(1, (unsigned)H, (unsigned)n_audio);
    // (16 waves per workgroup when the launch has only a window's 20 workgroups: 14.4 us against 12.9 us, round 3: not used)
    attn_mfma_kernel<true, 1><<<grid, XA_WAVES * 64, 0, s>>>(p);
}

}  // namespace wrs
