// Decode self-attention on the matrix cores, one wave per (audio, head), beams de-duplicated (gfx950).
//
// Reference: MultiHeadAttention::forward_ext on the self-attention K/V cache (model.rs:158-206), with the cache re-indexed
// by beam search every step (Cached::update_cache, model.rs:53-62; beam_search.rs:111-117).  The cache is never moved
// here: the ancestry table anc[row][position] names the physical row that holds (row, position) (elementwise.hip).
//
// attn_self_kernel (attention.hip) gives every (row, head) its own wave, so the G beams of an audio each pull the history
// they SHARE (common beam prefixes: 2.4x duplication at beam 5, profiles/r01_pmc_fetch_write_b256.json) from L2 again and do
// their dot products on the vector ALU (63 % VALU-busy, 79 us per layer at 1280 rows).  Here:
//   * once per STEP (the table is the same for all layers) self_keys_kernel lists, per audio, every DISTINCT
//     (position, physical row) pair of the G beams' histories with the set of beams that own it: entry = t | row << 9 | beams << 12;
//   * one wave per (audio, head) walks that list in blocks of 32 keys: K rows gathered straight into MFMA A-operand
//     fragments (a key's 64 dims are contiguous), S^T = K Q^T for all beams at once (the beams are the 16-wide query tile,
//     as in the cross-attention kernel), scores of pairs a beam does not own masked out, online softmax in fp32, and
//     O^T += V^T P^T with V^T read through LDS by ds_read_b64_tr_b16 (the PV product sums over KEYS, which are rows in memory);
//   * the new token's q / k / v come from the QKV GEMM's split-K slabs (bias, scale, cache append fused as before) and
//     form the first key block (every beam owns exactly its own new key, so the running max is finite from then on).
// Every distinct K/V row is fetched once per (audio, head): L2->CU traffic equals the HBM traffic.
#include "kernels.h"

namespace wrs {

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef short short4v __attribute__((__vector_size__(4 * sizeof(short))));

// reductions over the lanes l, l^16, l^32, l^48 (the four key groups of a query column) with v_permlane32_swap /
// v_permlane16_swap instead of two ds_bpermute round trips each
__device__ __forceinline__ float rows_max(float x) {
    const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float rows_sum(float x) {
    const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

static constexpr float SK_NEG = -1.0e30f;  // "masked": finite, so a query tile column with no owned key never produces inf - inf

// ------------------------------------------------------------------------------------------ per-step key list
// klist[audio][0] = number of entries, [1 ..]: distinct (t, row) pairs of positions 0 .. cache_pos - 1 in position order
// (kcount[audio]: the same count, for the host / tests).
__global__ __launch_bounds__(256) void self_keys_kernel(const int* __restrict__ anc_base, int64_t anc_buf_stride, int n_ctx, int G,
                                                        const StepParams* __restrict__ sp, int* __restrict__ klist, int klist_stride,
                                                        int* __restrict__ kcount, int include_new) {
    __shared__ int wave_tot[4];
    const int a = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int cpos = sp->cache_pos;
    const int* anc = anc_base + (int64_t)sp->parity * anc_buf_stride + (int64_t)a * G * n_ctx;
    int* out = klist + (int64_t)a * klist_stride;
    int base = 0;
    for (int t0 = 0; t0 < cpos; t0 += 256) {  // block-uniform trip count
        const int t = t0 + tid;
        int ent[8], n = 0;
        if (t < cpos) {
            int r[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = j < G ? anc[j * n_ctx + t] - a * G : -1 - j;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (j >= G) continue;
                bool first = true;
                int mask = 0;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (r[i] == r[j]) {
                        if (i < j) first = false;
                        mask |= 1 << i;
                    }
                }
                if (first) ent[n++] = t | ((r[j] & 7) << 9) | (mask << 12);
            }
        }
        int incl = n;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int woff = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w < wave) woff += wave_tot[w];
            tot += wave_tot[w];
        }
        const int off = base + woff + incl - n;
        for (int i = 0; i < n; ++i) out[1 + off + i] = ent[i];
        base += tot;
        __syncthreads();
    }
    // include_new: the step's own position (every beam owns exactly its own new key, already appended to the cache by the
    // QKV GEMM's epilogue) closes the list; otherwise the attention kernel builds the new keys from the QKV slabs itself
    if (include_new && tid < G) out[1 + base + tid] = cpos | (tid << 9) | ((1 << tid) << 12);
    if (tid == 0) {
        kcount[a] = base + (include_new ? G : 0);
        out[0] = base + (include_new ? G : 0);  // the attention kernel reads count and first entries with one round trip
    }
}

void launch_self_keys(const int* anc_base, int64_t anc_buf_stride, int n_ctx, int n_audio, int G, const StepParams* sp, int* klist,
                      int klist_stride, int* kcount, bool include_new, hipStream_t s) {
    WRS_REQUIRE(G >= 1 && G <= 8 && n_ctx <= 512 && n_audio >= 1 && klist_stride >= G * n_ctx + 1, "self_keys: bad shape (G %d, n_ctx %d)", G, n_ctx);
    self_keys_kernel<<<n_audio, 256, 0, s>>>(anc_base, anc_buf_stride, n_ctx, G, sp, klist, klist_stride, kcount, include_new ? 1 : 0);
}

// ------------------------------------------------------------------------------------------ attention
struct SelfMfmaParams {
    const float* slab;   // [ksplit][m_pad][3D] fp32 split-K partial sums of the QKV projection
    int ksplit, m_pad;
    const float* bias;   // [3D]
    float scale;
    half_t* kc;          // [rows][n_ctx][D]
    half_t* vc;
    int n_ctx, G, H, D;
    const StepParams* sp;
    const int* klist;
    int klist_stride;
    const int* kcount;
    half_t* out;         // packed activation order
    const half_t* q16;   // != null: the query as fp16 [rows, D] (bias and scale applied) and the new keys / values already in the
                         // cache AND in the key list (QKV GEMM epilogue + self_keys include_new); slab / bias / scale unused
};

static constexpr int SM_VIMG = 4096;              // V image: 32 keys x 128 B, 16-byte chunks XOR-swizzled with (key & 7)
static constexpr int SM_STAGE = 4096;             // K image of the same form; before the loop: 16 rows x 128 B of staging for q, then the new keys
static constexpr int SM_WAVE = SM_VIMG + SM_STAGE + 64;  // + 32 beam masks

__device__ __forceinline__ int vimg_off(int key, int chunk) { return key * 128 + ((chunk ^ (key & 7)) << 4); }

__device__ __forceinline__ void slab_sum8(const float* __restrict__ slab, int ksplit, int64_t slab_stride, int64_t off,
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

// NT: the cached K / V rows are requested with the non-temporal hint (every row is used once per launch by one wave)
template <bool NT>
__global__ __launch_bounds__(256) void attn_self_mfma_kernel(const SelfMfmaParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[4][SM_WAVE];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int hgroups = (p.H + 3) >> 2;
    const int a = (int)blockIdx.x / hgroups, h = ((int)blockIdx.x % hgroups) * 4 + wave;
    if (h >= p.H) return;  // the kernel has no workgroup barrier: every wave works on its own LDS slice
    unsigned char* my = lds[wave];
    unsigned char* vimg = my;
    unsigned char* stage = my + SM_VIMG;
    unsigned char* mbuf = my + SM_VIMG + SM_STAGE;
    const int qcol = lane & 15, qg = lane >> 4;
    const int G = p.G, D = p.D;
    const int cpos = p.sp->cache_pos;
    const int* list = p.klist + (int64_t)a * p.klist_stride;
    const int count = list[0];
    const int nb = (count + 31) >> 5;
    const unsigned row_stride_b = (unsigned)(p.n_ctx * D * 2);
    const unsigned audio_b = (unsigned)(a * G) * row_stride_b + (unsigned)(h * 128);
    const char* kcb = reinterpret_cast<const char*>(p.kc);
    const char* vcb = reinterpret_cast<const char*>(p.vc);

    // requests of key block 0 first: they fly while the new token is prepared
    half8 kf[4], vf[4];
    int e_cur = 0, e_nxt = 0;
    // list entries are requested without waiting for `count` (the buffer covers every index asked for) and validated only
    // when used: a select on the freshly loaded value would make the compiler wait for it — and, loads returning in order,
    // for every K / V request issued before it
    auto entry_raw = [&](int kb) -> int { return list[1 + min(kb * 32 + (lane & 31), p.klist_stride - 2)]; };
    auto entry_ok = [&](int kb, int e) -> int {
        return kb * 32 + (lane & 31) < count ? e : 0;  // 0 = (t 0, first row of the audio, no owner): a valid address, masked out
    };
    // A load instruction covers 8 keys x 128 contiguous bytes (lane = key l >> 3, chunk l & 7): whole cache lines.  (In
    // MFMA-fragment order — key l & 15, chunk l >> 4 — an instruction touches 16 rows for 64 bytes each and the second half
    // of every row arrives as a separate request that hits a line with a miss pending: PMC showed the address unit stalled
    // by the L1 70x more often than in a loads-only kernel.)  Both K and V therefore go to LDS row-major and are read back
    // as fragments (K: ds_read_b128, V^T: ds_read_b64_tr_b16).
    const int lkey = lane >> 3, lchunk = lane & 7;
    auto issue = [&](int e) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ek = __shfl(e, i * 8 + lkey);
            const unsigned boff = audio_b + (unsigned)((ek >> 9) & 7) * row_stride_b + (unsigned)(ek & 511) * (unsigned)(D * 2) + (unsigned)(lchunk * 16);
            if (NT) {
                kf[i] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(kcb + boff));
                vf[i] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(vcb + boff));
            } else {
                kf[i] = *reinterpret_cast<const half8*>(kcb + boff);
                vf[i] = *reinterpret_cast<const half8*>(vcb + boff);
            }
        }
    };
    e_nxt = entry_raw(1);
    e_cur = entry_ok(0, entry_raw(0));
    if (nb > 0) issue(e_cur);

    // ---- the new token of every beam: q, k, v from the slabs (+ bias, scale), k / v appended to the cache
    half8 qf[2], knf[2];
    const bool from_slab = p.q16 == nullptr;
    if (!from_slab) {
        // B operand Q[query = beam qcol][dims]: straight from the fp16 query buffer; log2(e) folded in for the exp2 softmax
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (qcol < G) v = *reinterpret_cast<const half8*>(p.q16 + (int64_t)(a * G + qcol) * D + h * 64 + ks * 32 + 8 * qg);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (_Float16)((float)v[j] * 1.44269504088896341f);
            qf[ks] = v;
            knf[ks] = half8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    } else {
        const int beam = lane >> 3, c = lane & 7;
        half8 q16 = {0, 0, 0, 0, 0, 0, 0, 0}, k16 = q16, v16 = q16;
        const float qscale = p.scale * 1.44269504088896341f;  // scores in the log2 domain: the softmax below is exp2 (one v_exp_f32 each)
        if (beam < G) {
            const int row = a * G + beam, col = h * 64 + c * 8;
            const int64_t ss = (int64_t)p.m_pad * 3 * D, off = (int64_t)row * 3 * D + col;
            float qv[8], kn[8], vn[8];
            slab_sum8(p.slab, p.ksplit, ss, off, p.bias, col, qv);
            slab_sum8(p.slab, p.ksplit, ss, off + D, p.bias, D + col, kn);
            slab_sum8(p.slab, p.ksplit, ss, off + 2 * D, p.bias, 2 * D + col, vn);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                q16[j] = (_Float16)(qv[j] * qscale);
                k16[j] = (_Float16)(kn[j] * p.scale);
                v16[j] = (_Float16)vn[j];
            }
            const int64_t coff = ((int64_t)row * p.n_ctx + cpos) * D + col;
            *reinterpret_cast<half8*>(p.kc + coff) = k16;
            *reinterpret_cast<half8*>(p.vc + coff) = v16;
        }
        const half8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
        // q -> stage rows 0..7 (zero for beam >= G), rows 8..15 zero; read back as the B operand Q[query][dims]
        *reinterpret_cast<half8*>(stage + beam * 128 + c * 16) = q16;
        *reinterpret_cast<half8*>(stage + (8 + beam) * 128 + c * 16) = zero;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf[ks] = *reinterpret_cast<const half8*>(stage + qcol * 128 + (ks * 4 + qg) * 16);
        // new keys -> the same rows (LDS operations of a wave execute in order), read back as the A operand K[key][dims]
        *reinterpret_cast<half8*>(stage + beam * 128 + c * 16) = k16;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) knf[ks] = *reinterpret_cast<const half8*>(stage + qcol * 128 + (ks * 4 + qg) * 16);
        // new values -> V image rows 0..7, rows 8..31 zero (0 x garbage could be NaN), masks: key b is owned by beam b
        *reinterpret_cast<half8*>(vimg + vimg_off(beam, c)) = v16;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int idx = i * 64 + lane;  // 24 rows x 8 chunks
            *reinterpret_cast<half8*>(vimg + vimg_off(8 + (idx >> 3), idx & 7)) = zero;
        }
        if (lane < 32) mbuf[lane] = (unsigned char)(lane < G ? (1 << lane) : 0);
    }

    floatx4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = floatx4{0.f, 0.f, 0.f, 0.f};
    float m_run = SK_NEG, l_run = 0.f;

    // one 32-key block, second half: masks, online softmax, P.V (V image and masks are in LDS)
    auto block = [&](const floatx4 s0, const floatx4 s1) {
        const unsigned m0 = *reinterpret_cast<const unsigned*>(mbuf + 4 * qg);
        const unsigned m1 = *reinterpret_cast<const unsigned*>(mbuf + 16 + 4 * qg);
        float sv[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sv[j] = (((m0 >> (8 * j)) & 0xffu) >> qcol) & 1u ? s0[j] : SK_NEG;
            sv[4 + j] = (((m1 >> (8 * j)) & 0xffu) >> qcol) & 1u ? s1[j] : SK_NEG;
        }
        float bm = fmaxf(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])), fmaxf(fmaxf(sv[4], sv[5]), fmaxf(sv[6], sv[7])));
        bm = rows_max(bm);
        const float m_new = fmaxf(m_run, bm);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        float ps = 0.f;
        half8 pf, pl;  // probabilities as fp16 hi + lo parts: the P.V product keeps ~22 bits of P, like the fp32 x fp16 FMAs of attn_self_kernel
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float e = __builtin_amdgcn_exp2f(sv[j] - m_new);  // masked: exp2(-1e30 - m) = 0 once the column has seen an owned key
            ps += e;
            pf[j] = (_Float16)e;
            pl[j] = (_Float16)(e - (float)pf[j]);
        }
        ps = rows_sum(ps);
        l_run = l_run * alpha + ps;
        const bool moved = __builtin_amdgcn_ballot_w64(m_new > m_run) != 0;  // wave-uniform: rescale only when some column's max moved
        m_run = m_new;
        // V^T fragments: lane (dim d = qcol of tile dt, key group qg) needs V[keys 4qg..4qg+3 | 16+4qg..][dt*16 + d]: two
        // transposed 4x16 block reads; lane 4q+p of a 16-lane group supplies the address of row q, columns 4p..4p+3
        const int tq = qcol >> 2, tp = qcol & 3;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const int chunk = 2 * dt + (tp >> 1);
            const int key0 = 4 * qg + tq, key1 = 16 + 4 * qg + tq;
            const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) short4v*)(vimg + vimg_off(key0, chunk) + 8 * (tp & 1)));
            const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) short4v*)(vimg + vimg_off(key1, chunk) + 8 * (tp & 1)));
            const half4 l4 = __builtin_bit_cast(half4, lo), h4 = __builtin_bit_cast(half4, hi);
            const half8 vfrag = {l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]};
            if (moved) o[dt] *= alpha;
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vfrag, pf, o[dt], 0, 0, 0);
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vfrag, pl, o[dt], 0, 0, 0);
        }
    };

    if (from_slab) {  // block of the new keys (tile 1 is empty: its masks are 0)
        floatx4 s0 = floatx4{0.f, 0.f, 0.f, 0.f};
        s0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(knf[0], qf[0], s0, 0, 0, 0);
        s0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(knf[1], qf[1], s0, 0, 0, 0);
        block(s0, floatx4{0.f, 0.f, 0.f, 0.f});
    }
    unsigned char* kimg = stage;  // the staging rows are dead once the loop starts
    for (int kb = 0; kb < nb; ++kb) {
        // this block's K and V rows -> LDS images (the previous block's reads were consumed by its MFMAs), masks
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<half8*>(kimg + vimg_off(i * 8 + lkey, lchunk)) = kf[i];
            *reinterpret_cast<half8*>(vimg + vimg_off(i * 8 + lkey, lchunk)) = vf[i];
        }
        if (lane < 32) mbuf[lane] = (unsigned char)((e_cur >> 12) & 0xff);
        // the row registers are free: the next block's requests go out before this block's arithmetic
        e_cur = entry_ok(kb + 1, e_nxt);
        if (kb + 1 < nb) {
            e_nxt = entry_raw(kb + 2);  // requested BEFORE the block's rows: waiting for it later does not wait for them
            issue(e_cur);
        }
        // S^T = K Q^T (two 16-key tiles)
        half8 kfr[4];
#pragma unroll
        for (int tile = 0; tile < 2; ++tile)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) kfr[tile * 2 + ks] = *reinterpret_cast<const half8*>(kimg + vimg_off(tile * 16 + qcol, ks * 4 + qg));
        floatx4 s0 = floatx4{0.f, 0.f, 0.f, 0.f}, s1 = floatx4{0.f, 0.f, 0.f, 0.f};
        s0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(kfr[0], qf[0], s0, 0, 0, 0);
        s0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(kfr[1], qf[1], s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(kfr[2], qf[0], s1, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(kfr[3], qf[1], s1, 0, 0, 0);
        block(s0, s1);
    }

    // O^T: lane holds dims dt*16 + 4 qg + r of query (beam) qcol
    if (qcol < G) {
        const float inv = 1.f / l_run;
        const int row = a * G + qcol;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
            *reinterpret_cast<half4*>(p.out + pa_off(row, h * 64 + dt * 16 + 4 * qg, D >> 5)) =
                half4{(_Float16)(o[dt][0] * inv), (_Float16)(o[dt][1] * inv), (_Float16)(o[dt][2] * inv), (_Float16)(o[dt][3] * inv)};
    }
}

// Non-temporal K / V loads (they bypass the CU's L1: every row is used once, by one wave): 65.2 -> 50.6 us on the kernel
// test's shapes (profiles/r03_self_attention_nontemporal.log).  The PMC comparison of round 2 had shown the texture-address
// unit stalled by the L1 70x more often than in a loads-only kernel of the same address pattern; this is that stall.
// In the decode step (colder caches than the kernel test) the gain is 74.9 -> 71.9 us (profiles/r03_insitu_ab_start_vs_epilogues.log).
// A fragment-order gather with these loads (K rows straight into MFMA operands, no K image in LDS) was 101 us against 71.8 us
// in situ (profiles/r03_self_attention_fragment_order.log): not kept.
// WRS_SELF_NT=0 (test build, read once per process) selects the plain loads for comparison.
static bool self_mfma_nt() {
    static const bool nt = !(exp_env("WRS_SELF_NT") && exp_env("WRS_SELF_NT")[0] == '0');
    return nt;
}

void launch_attn_self_mfma_q16(const half_t* q16, half_t* kc, half_t* vc, int n_ctx, int n_rows, int G, int H, int D, const StepParams* sp,
                               const int* klist, int klist_stride, const int* kcount, half_t* out, hipStream_t s) {
    WRS_REQUIRE(n_ctx <= 512 && D == H * 64 && q16, "attn_self_mfma: bad shape n_ctx=%d D=%d H=%d", n_ctx, D, H);
    WRS_REQUIRE(G >= 1 && G <= 8 && n_rows % G == 0 && klist && kcount, "attn_self_mfma: bad group size %d for %d rows", G, n_rows);
    WRS_REQUIRE((int64_t)n_rows * n_ctx * D * 2 < (1ll << 32), "attn_self_mfma: a layer's K/V cache must stay below 4 GiB (32-bit offsets)");
    SelfMfmaParams p{nullptr, 1, 0, nullptr, 1.f, kc, vc, n_ctx, G, H, D, sp, klist, klist_stride, kcount, out, q16};
    if (self_mfma_nt()) attn_self_mfma_kernel<true><<<(n_rows / G) * ((H + 3) / 4), 256, 0, s>>>(p);
    else attn_self_mfma_kernel<false><<<(n_rows / G) * ((H + 3) / 4), 256, 0, s>>>(p);
}

void launch_attn_self_mfma(const float* qkv_slab, int ksplit, int m_pad, const float* qkv_bias, float scale, half_t* kc, half_t* vc, int n_ctx,
                           int n_rows, int G, int H, int D, const StepParams* sp, const int* klist, int klist_stride, const int* kcount,
                           half_t* out, hipStream_t s) {
    WRS_REQUIRE(n_ctx <= 512 && D == H * 64 && qkv_slab && qkv_bias && ksplit >= 1, "attn_self_mfma: bad shape n_ctx=%d D=%d H=%d", n_ctx, D, H);
    WRS_REQUIRE(G >= 1 && G <= 8 && n_rows % G == 0 && klist && kcount, "attn_self_mfma: bad group size %d for %d rows", G, n_rows);
    WRS_REQUIRE((int64_t)n_rows * n_ctx * D * 2 < (1ll << 32), "attn_self_mfma: a layer's K/V cache must stay below 4 GiB (32-bit offsets)");
    SelfMfmaParams p{qkv_slab, ksplit, m_pad, qkv_bias, scale, kc, vc, n_ctx, G, H, D, sp, klist, klist_stride, kcount, out, nullptr};
    const int n_audio = n_rows / G;
    if (self_mfma_nt()) attn_self_mfma_kernel<true><<<n_audio * ((H + 3) / 4), 256, 0, s>>>(p);
    else attn_self_mfma_kernel<false><<<n_audio * ((H + 3) / 4), 256, 0, s>>>(p);
}

}  // namespace wrs
