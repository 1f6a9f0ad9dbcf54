// Fused logit filters + log-softmax + top-k over the 51 864/51 865-entry vocabulary (gfx950).
//
// Replaces, per decode step and per row, the reference's chain of separate ATen calls and host
// round-trips:
//   SuppressBlank      (filter/suppress_blanks.rs:38-43)   index_fill_(-inf) at first step
//   SuppressTokens     (filter/suppress_tokens.rs:21-23)   index_fill_(-inf)
//   TimestampTokens    (filter/timestamp_tokens.rs:33-97)  pairing masks, first-step masks, and the
//                      "timestamp mass > best text token" rule (log_softmax + logsumexp + max,
//                      4 scalar device reads per row in the reference)
//   log_softmax        (extractor/beam_search.rs:37, greedy.rs:31)
//   topk(beam+1)       (extractor/beam_search.rs:62-65), argmax (greedy.rs:25)
//   no-speech prob     (decode/mod.rs:250-255) softmax(raw logits)[no_speech] at step 0
// One 256-thread workgroup per row; every reduction is a 64-lane wave reduction followed by a
// 4-wave LDS combine.  The row (207 KB f32) is read ONCE with 16-byte loads (online max / sum-exp and per-thread top-8
// lists for the text and the timestamp range); nothing is written back except k (logprob, id) pairs.
#include "kernels.h"

namespace wrs {

#define TOPK_MAX 8

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_add(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// 64-lane maximum on the VALU (DPP inside the 16-lane rows, v_permlane swaps between them): the K-th-largest bound below is
// refreshed inside the pass, where six ds_bpermute round trips per maximum would cost what it saves
template <int CTRL>
__device__ __forceinline__ float dpp_max(float v) {
    return fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xF, 0xF, false)));
}
__device__ __forceinline__ float wave_max_valu(float v) {
    v = dpp_max<0xB1>(v);   // quad_perm [1,0,3,2]
    v = dpp_max<0x4E>(v);   // quad_perm [2,3,0,1]
    v = dpp_max<0x141>(v);  // row_half_mirror
    v = dpp_max<0x140>(v);  // row_mirror: every lane of a 16-lane row holds the row's maximum
    const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    const auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
// K-th largest of the 64 lanes' values (wave-uniform; -inf when fewer than K lanes hold a finite value).  With x = every lane's
// best candidate so far this is a LOWER bound of the wave's K-th best candidate: a value below it cannot be among the K best of
// the row, whatever comes later.
__device__ __forceinline__ float wave_kth_largest(float x, int K) {
    int need = K;
#pragma unroll 1
    for (int r = 0; r < TOPK_MAX; ++r) {
        const float m = wave_max_valu(x);
        if (m == -INFINITY) break;
        need -= __popcll(__ballot(x == m));
        if (need <= 0) return m;
        if (x == m) x = -INFINITY;
    }
    return -INFINITY;
}
__device__ __forceinline__ float block_max(float v, float* sh) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}
__device__ __forceinline__ float block_add(float v, float* sh) {
    v = wave_add(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

struct FilterParams {
    const float* logits;
    int ld;
    int n_vocab, ts_begin, eot, no_timestamps, no_speech, space;
    int suppress_blank, timestamps, mit;
    const int* suppress_ids;
    int n_suppress;
    const StepParams* sp;
    const int* ts_mode;
    int k;
    float* out_lp;
    int* out_id;
    float* no_speech_out;
    float* out_filtered;
    float* ws;   // PARTIAL form: per (row, chunk) FILTER_WS_STRIDE floats: Mt St Ms Ss Mr Sr, then k (value, id) pairs text, k pairs timestamps
    int chunk;   // PARTIAL form: columns per workgroup (multiple of 4)
};

// insertion of (cv, ci) into a list sorted by value descending, index ascending on ties (static indexing: registers)
__device__ __forceinline__ void topk_insert(float (&tv)[TOPK_MAX], int (&ti)[TOPK_MAX], float cv, int ci) {
#pragma unroll
    for (int j = 0; j < TOPK_MAX; ++j) {
        const bool better = cv > tv[j] || (cv == tv[j] && ci < ti[j]);
        if (better) {
            const float ov = tv[j];
            const int oi = ti[j];
            tv[j] = cv;
            ti[j] = ci;
            cv = ov;
            ci = oi;
        }
    }
}

// ONE pass over the row (round 1 streamed it three times with 4-byte loads): 16-byte loads, and for each of the two index
// ranges the timestamp rule distinguishes — text [0, ts_begin) and timestamps [ts_begin, V) — an online (max, sum-exp)
// pair and a per-thread top-8 list.  The rule's decision (timestamp_tokens.rs:83-96) then only picks which lists and
// which normaliser apply; nothing has to be re-read.
// k rounds of block arg-max over the threads' sorted lists (value descending, index ascending on ties): thread 0 emits the
// winners through `emit(round, value, index)`
template <typename Emit>
__device__ __forceinline__ void block_topk(float (&tv)[TOPK_MAX], int (&ti)[TOPK_MAX], int K, float* cand_v, int* cand_i, Emit emit) {
    const int tid = threadIdx.x;
    int head = 0;
    for (int round = 0; round < K; ++round) {
        float hv = -INFINITY;
        int hi_ = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < TOPK_MAX; ++j)
            if (j == head) {
                hv = tv[j];
                hi_ = ti[j];
            }
        cand_v[tid] = hv;
        cand_i[tid] = hi_;
        __syncthreads();
        for (int stride = 128; stride > 0; stride >>= 1) {
            if (tid < stride) {
                const float ov = cand_v[tid + stride];
                const int oi = cand_i[tid + stride];
                if (ov > cand_v[tid] || (ov == cand_v[tid] && oi < cand_i[tid])) {
                    cand_v[tid] = ov;
                    cand_i[tid] = oi;
                }
            }
            __syncthreads();
        }
        const float bv = cand_v[0];
        const int bi = cand_i[0];
        __syncthreads();
        if (bi == hi_ && hi_ != 0x7fffffff) head++;
        if (tid == 0) emit(round, bv, bi);
    }
}

// PARTIAL: blockIdx.y = chunk of the vocabulary; the workgroup leaves its (max, sum-exp) pairs and its k best text and k best
// timestamp candidates in the workspace, filter_merge_kernel finishes the row (few rows: one workgroup per row would use
// rows CUs of 256).  The candidates of a row are the same set either way, so ids and their order are identical.
template <bool PARTIAL>
__global__ __launch_bounds__(256) void filter_topk_kernel(const FilterParams p) {
    extern __shared__ unsigned int smem[];  // bitmask of statically suppressed ids: ceil(V/32) words
    __shared__ float red[4];
    __shared__ float cand_v[256];
    __shared__ int cand_i[256];
    const int row = blockIdx.x, tid = threadIdx.x;
    const int V = p.n_vocab;
    const int v_begin = PARTIAL ? (int)blockIdx.y * p.chunk : 0;
    const int v_end = PARTIAL ? min(V, v_begin + p.chunk) : V;
    const float* x = p.logits + (int64_t)row * p.ld;
    const int first = p.sp->first_step;
    const int nwords = (V + 31) / 32;
    const bool want_ns = p.sp->want_no_speech && p.no_speech_out;

    // ---- static suppression mask
    for (int w = tid; w < nwords; w += 256) smem[w] = 0u;
    __syncthreads();
    for (int i = tid; i < p.n_suppress; i += 256) {
        const int id = p.suppress_ids[i];
        if (id >= 0 && id < V) atomicOr(&smem[id >> 5], 1u << (id & 31));
    }
    if (tid == 0) {
        if (p.suppress_blank && first) {  // suppress_blanks.rs:38-43
            atomicOr(&smem[p.space >> 5], 1u << (p.space & 31));
            atomicOr(&smem[p.eot >> 5], 1u << (p.eot & 31));
        }
        if (p.timestamps) atomicOr(&smem[p.no_timestamps >> 5], 1u << (p.no_timestamps & 31));  // timestamp_tokens.rs:37-41
    }
    __syncthreads();

    // ---- range masks of TimestampTokens (timestamp_tokens.rs:47-78)
    int lo = 0, hi = V;         // allowed index range [lo, hi) after the pairing / first-step rules
    if (p.timestamps) {
        const int mode = p.ts_mode[row];
        if (mode == 1) hi = min(hi, p.ts_begin);   // last two sampled were timestamps: must be text
        else if (mode == 2) lo = max(lo, p.eot);   // last was a lone timestamp: no text below eot
        if (first) {
            lo = max(lo, p.ts_begin);
            if (p.mit >= 0) hi = min(hi, p.ts_begin + p.mit + 1);
        }
    }
    const int tsb = p.timestamps ? p.ts_begin : V;

    // ---- the pass
    float mt = -INFINITY, st = 0.f, ms = -INFINITY, ss = 0.f;  // text / timestamp ranges: running max, sum of exp(val - max)
    float mr = -INFINITY, sr = 0.f;                            // raw logits (no-speech probability, step 0 only)
    float tvt[TOPK_MAX], tvs[TOPK_MAX];
    int tit[TOPK_MAX], tis[TOPK_MAX];
#pragma unroll
    for (int j = 0; j < TOPK_MAX; ++j) {
        tvt[j] = tvs[j] = -INFINITY;
        tit[j] = tis[j] = 0x7fffffff;
    }
    typedef float floatx4 __attribute__((ext_vector_type(4)));
    const bool vec_ok = (p.ld & 3) == 0 && ((reinterpret_cast<uintptr_t>(p.logits) & 15) == 0);
    // tau_t: wave-uniform lower bound of the K-th best allowed text value (refreshed after iterations 1, 3, 7, 15, 31 from the
    // lanes' best candidates): text values below it skip the top-8 insertion.  Without it the insertion — an 8-step shift that
    // the whole wave executes whenever ANY lane's value beats that lane's own 8th best — ran ~200 times per wave and made a
    // full row's pass 4x its memory time (round 3: 77 us for rows that allow nearly nothing, 305 us for full rows).
    float tau_t = -INFINITY;
    int it = 0;
    for (int v0 = v_begin + tid * 4; v0 < v_end; v0 += 1024, ++it) {
        float val4[4];
        if (vec_ok && v0 + 4 <= p.ld) {
            const floatx4 t = ld_stream(reinterpret_cast<const floatx4*>(x + v0));  // a row's logits are read once
            val4[0] = t[0]; val4[1] = t[1]; val4[2] = t[2]; val4[3] = t[3];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) val4[e] = v0 + e < v_end ? x[v0 + e] : -INFINITY;
        }
        const unsigned bits = smem[v0 >> 5] >> (v0 & 31);  // 4 consecutive ids never straddle a 32-bit word (v0 % 4 == 0)
        // The 4 values as one group per range: the running maximum is raised once per group (rarely, after the first few), ONE
        // exponential per value (per value, both sides of `if (val > m) rescale else add` ran under divergence: two), the top-8
        // list touched only when the group's maximum reaches the wave's bound and the lane's last entry.  Masked values enter as
        // -inf: exp(-inf - m) = 0.
        float raw[4], tx[4], ts[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int v = v0 + e;
            const bool in = v < v_end;
            const bool ok = in && v >= lo && v < hi && !((bits >> e) & 1u);
            raw[e] = in ? val4[e] : -INFINITY;
            tx[e] = (ok && v < tsb) ? val4[e] : -INFINITY;
            ts[e] = (ok && v >= tsb) ? val4[e] : -INFINITY;
        }
        if (want_ns) {  // softmax of the RAW logits (decode/mod.rs:250-255)
            const float g = fmaxf(fmaxf(raw[0], raw[1]), fmaxf(raw[2], raw[3]));
            if (g > mr) { sr *= __expf(mr - g); mr = g; }  // mr = -inf: sr = 0 * exp(-inf) = 0
            if (mr > -INFINITY) sr += (__expf(raw[0] - mr) + __expf(raw[1] - mr)) + (__expf(raw[2] - mr) + __expf(raw[3] - mr));
        }
        if (v0 < tsb) {  // the group holds text ids (all of it, except the one group ts_begin may fall into)
            const float g = fmaxf(fmaxf(tx[0], tx[1]), fmaxf(tx[2], tx[3]));
            if (g > -INFINITY) {
                if (g > mt) { st *= __expf(mt - g); mt = g; }
                st += (__expf(tx[0] - mt) + __expf(tx[1] - mt)) + (__expf(tx[2] - mt) + __expf(tx[3] - mt));
                if (g >= tau_t && g >= tvt[TOPK_MAX - 1]) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (tx[e] >= tau_t && (tx[e] > tvt[TOPK_MAX - 1] || (tx[e] == tvt[TOPK_MAX - 1] && tx[e] > -INFINITY && v0 + e < tit[TOPK_MAX - 1])))
                            topk_insert(tvt, tit, tx[e], v0 + e);
                }
            }
        }
        if (v0 + 3 >= tsb) {  // ... timestamp ids
            const float g = fmaxf(fmaxf(ts[0], ts[1]), fmaxf(ts[2], ts[3]));
            if (g > -INFINITY) {
                if (g > ms) { ss *= __expf(ms - g); ms = g; }
                ss += (__expf(ts[0] - ms) + __expf(ts[1] - ms)) + (__expf(ts[2] - ms) + __expf(ts[3] - ms));
                if (g >= tvs[TOPK_MAX - 1]) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (ts[e] > tvs[TOPK_MAX - 1] || (ts[e] == tvs[TOPK_MAX - 1] && ts[e] > -INFINITY && v0 + e < tis[TOPK_MAX - 1])) topk_insert(tvs, tis, ts[e], v0 + e);
                }
            }
        }
        // `it` is the same in every lane.  A lane's trip count is the wave's or one less, so the only refresh that can run with lanes
        // missing is the one at the end of the wave's LAST iteration — whose result nobody uses: every bound that is used was
        // computed by 64 active lanes (the cross-lane exchanges in wave_kth_largest need that)
        if (it >= 1 && it <= 31 && (it & (it + 1)) == 0) tau_t = wave_kth_largest(tvt[0], p.k);
    }
    // ---- combine the threads' (max, sum) pairs
    if (PARTIAL) {
        float* w = p.ws + ((int64_t)row * gridDim.y + blockIdx.y) * FILTER_WS_STRIDE;
        const float Mr = block_max(mr, red);
        const float Sr = block_add(mr == -INFINITY ? 0.f : sr * __expf(mr - Mr), red);
        const float Mt = block_max(mt, red);
        const float St = block_add(mt == -INFINITY ? 0.f : st * __expf(mt - Mt), red);
        const float Ms = block_max(ms, red);
        const float Ss = block_add(ms == -INFINITY ? 0.f : ss * __expf(ms - Ms), red);
        if (tid == 0) { w[0] = Mt; w[1] = St; w[2] = Ms; w[3] = Ss; w[4] = Mr; w[5] = Sr; }
        const int K = p.k;
        block_topk(tvt, tit, K, cand_v, cand_i, [&](int r, float v, int i) { w[6 + 2 * r] = v; w[7 + 2 * r] = __int_as_float(i); });
        block_topk(tvs, tis, K, cand_v, cand_i, [&](int r, float v, int i) { w[6 + 2 * K + 2 * r] = v; w[7 + 2 * K + 2 * r] = __int_as_float(i); });
        return;
    }
    if (want_ns) {
        const float Mr = block_max(mr, red);
        const float Sr = block_add(mr == -INFINITY ? 0.f : sr * __expf(mr - Mr), red);
        if (tid == 0) p.no_speech_out[row] = __expf(x[p.no_speech] - Mr) / Sr;
    }
    const float Mt = block_max(mt, red);
    const float St = block_add(mt == -INFINITY ? 0.f : st * __expf(mt - Mt), red);  // sum over text of exp(val - Mt)
    const float Ms = block_max(ms, red);
    const float Ss = block_add(ms == -INFINITY ? 0.f : ss * __expf(ms - Ms), red);  // sum over timestamps of exp(val - Ms)
    const float mall = fmaxf(Mt, Ms);
    const float tot = (Mt == -INFINITY ? 0.f : St * __expf(Mt - mall)) + (Ms == -INFINITY ? 0.f : Ss * __expf(Ms - mall));
    bool mask_text = false;
    float lse = mall + __logf(tot);  // log_softmax normaliser (timestamp_tokens.rs:82)
    if (p.timestamps) {
        // timestamp_tokens.rs:83-96: logsumexp(logprobs[ts_begin:]) > max(logprobs[:ts_begin]) ?
        const float ts_lp = (Ms == -INFINITY) ? -INFINITY : (Ms - lse) + __logf(Ss);
        const float text_lp = Mt - lse;
        if (ts_lp > text_lp) {
            mask_text = true;
            lse = Ms + __logf(Ss);  // log_softmax after masking the text range (beam_search.rs:37)
        }
    }
    if (mask_text) lo = max(lo, p.ts_begin);

    if (p.out_filtered) {  // test hook only
        float* o = p.out_filtered + (int64_t)row * V;
        for (int v = tid; v < V; v += 256) {
            float val = x[v];
            if (v < lo || v >= hi || ((smem[v >> 5] >> (v & 31)) & 1u)) val = -INFINITY;
            o[v] = val;
        }
    }

    // ---- top-k.  The thread's candidates: its timestamp list, plus its text list unless the rule masked the text range
    // (text ids are all below timestamp ids, so the merged order keeps "index ascending on ties").  Then k rounds of block
    // arg-max over the threads' current heads.
    float tv[TOPK_MAX];
    int ti[TOPK_MAX];
#pragma unroll
    for (int j = 0; j < TOPK_MAX; ++j) {
        tv[j] = tvs[j];
        ti[j] = tis[j];
    }
    if (!mask_text) {
#pragma unroll
        for (int j = 0; j < TOPK_MAX; ++j)
            if (tit[j] != 0x7fffffff) topk_insert(tv, ti, tvt[j], tit[j]);
    }
    block_topk(tv, ti, p.k, cand_v, cand_i, [&](int round, float bv, int bi) {
        p.out_lp[row * p.k + round] = (bi == 0x7fffffff) ? -INFINITY : bv - lse;
        p.out_id[row * p.k + round] = (bi == 0x7fffffff) ? -1 : bi;
    });
}

// second half of the split form: one wave per row merges the FILTER_SPLIT partial results exactly as the one-workgroup
// kernel decides (timestamp_tokens.rs:83-96) and selects the k best of the chunks' candidates
__global__ __launch_bounds__(64) void filter_merge_kernel(const FilterParams p, const int n_chunks) {
    const int row = blockIdx.x, lane = threadIdx.x;
    const int K = p.k;
    const float* w = p.ws + (int64_t)row * n_chunks * FILTER_WS_STRIDE;
    const bool has = lane < n_chunks;
    const float mt = has ? w[lane * FILTER_WS_STRIDE + 0] : -INFINITY, st = has ? w[lane * FILTER_WS_STRIDE + 1] : 0.f;
    const float ms = has ? w[lane * FILTER_WS_STRIDE + 2] : -INFINITY, ss = has ? w[lane * FILTER_WS_STRIDE + 3] : 0.f;
    const float mr = has ? w[lane * FILTER_WS_STRIDE + 4] : -INFINITY, sr = has ? w[lane * FILTER_WS_STRIDE + 5] : 0.f;
    if (p.sp->want_no_speech && p.no_speech_out) {
        const float Mr = wave_max(mr);
        const float Sr = wave_add(mr == -INFINITY ? 0.f : sr * __expf(mr - Mr));
        if (lane == 0) p.no_speech_out[row] = __expf(p.logits[(int64_t)row * p.ld + p.no_speech] - Mr) / Sr;
    }
    const float Mt = wave_max(mt);
    const float St = wave_add(mt == -INFINITY ? 0.f : st * __expf(mt - Mt));
    const float Ms = wave_max(ms);
    const float Ss = wave_add(ms == -INFINITY ? 0.f : ss * __expf(ms - Ms));
    const float mall = fmaxf(Mt, Ms);
    const float tot = (Mt == -INFINITY ? 0.f : St * __expf(Mt - mall)) + (Ms == -INFINITY ? 0.f : Ss * __expf(Ms - mall));
    bool mask_text = false;
    float lse = mall + __logf(tot);
    if (p.timestamps) {
        const float ts_lp = (Ms == -INFINITY) ? -INFINITY : (Ms - lse) + __logf(Ss);
        const float text_lp = Mt - lse;
        if (ts_lp > text_lp) {
            mask_text = true;
            lse = Ms + __logf(Ss);
        }
    }
    // candidates: n_chunks * K text + n_chunks * K timestamp entries, 4 per lane (n_chunks * K * 2 <= 256)
    float cv[4];
    int ci[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int e = lane + 64 * j;
        cv[j] = -INFINITY;
        ci[j] = 0x7fffffff;
        if (e < 2 * n_chunks * K) {
            const int list = e / (n_chunks * K), rem = e % (n_chunks * K);  // 0: text, 1: timestamps
            const int c = rem / K, r = rem % K;
            const float* q = w + c * FILTER_WS_STRIDE + 6 + list * 2 * K + 2 * r;
            const int idx = __float_as_int(q[1]);
            if (idx != 0x7fffffff && !(list == 0 && mask_text)) { cv[j] = q[0]; ci[j] = idx; }
        }
    }
    for (int round = 0; round < K; ++round) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (cv[j] > bv || (cv[j] == bv && ci[j] < bi)) { bv = cv[j]; bi = ci[j]; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o);
            const int oi = __shfl_xor(bi, o);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (ci[j] == bi && bi != 0x7fffffff) { cv[j] = -INFINITY; ci[j] = 0x7fffffff; }
        if (lane == 0) {
            p.out_lp[row * K + round] = (bi == 0x7fffffff) ? -INFINITY : bv - lse;
            p.out_id[row * K + round] = (bi == 0x7fffffff) ? -1 : bi;
        }
    }
}

// Language detection (OpenAI whisper `detect_language`; the reference declares Task::LanguageId, tokenize.rs:135-139,
// but does not implement it — BASELINE config 4): softmax restricted to the language tokens of the logits at the
// <|startoftranscript|> position.  One wave per audio; the <= 128 language logits sit two per lane.
__global__ __launch_bounds__(64) void lang_detect_kernel(const float* __restrict__ logits, int ld, int first, int n,
                                                          int* __restrict__ out_id, float* __restrict__ out_probs) {
    const int row = blockIdx.x, lane = threadIdx.x;
    const float* x = logits + (int64_t)row * ld + first;
    const float a = lane < n ? x[lane] : -INFINITY, b = lane + 64 < n ? x[lane + 64] : -INFINITY;
    const float mx = wave_max(fmaxf(a, b));
    const float ea = lane < n ? __expf(a - mx) : 0.f, eb = lane + 64 < n ? __expf(b - mx) : 0.f;
    const float sum = wave_add(ea + eb);
    if (out_probs) {
        if (lane < n) out_probs[(int64_t)row * n + lane] = ea / sum;
        if (lane + 64 < n) out_probs[(int64_t)row * n + lane + 64] = eb / sum;
    }
    // argmax, lowest index on ties (torch.argmax)
    int idx = (a == mx && lane < n) ? lane : ((b == mx && lane + 64 < n) ? lane + 64 : 0x7fffffff);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) idx = min(idx, __shfl_xor(idx, o));
    if (lane == 0) out_id[row] = first + idx;
}

void launch_lang_detect(const float* logits, int n_rows, int ld, int first_id, int n_lang, int* out_id, float* out_probs, hipStream_t s) {
    WRS_REQUIRE(n_lang >= 1 && n_lang <= 128 && first_id >= 0, "lang_detect: %d languages not in 1..128", n_lang);
    lang_detect_kernel<<<n_rows, 64, 0, s>>>(logits, ld, first_id, n_lang, out_id, out_probs);
}

void launch_filter_topk(const float* logits, int n_rows, const FilterCfg& cfg, const StepParams* sp, const int* ts_mode,
                        int k, float* out_lp, int* out_id, float* no_speech, float* out_filtered, hipStream_t s) {
    WRS_REQUIRE(k >= 1 && k <= TOPK_MAX, "filter_topk: k=%d not in 1..%d", k, TOPK_MAX);
    WRS_REQUIRE(cfg.n_vocab > 0 && cfg.ts_begin >= 0 && cfg.ts_begin <= cfg.n_vocab && cfg.eot < cfg.n_vocab &&
                    cfg.no_timestamps < cfg.n_vocab && cfg.no_speech < cfg.n_vocab && cfg.space < cfg.n_vocab,
                "filter_topk: token ids out of range for vocab %d", cfg.n_vocab);
    FilterParams p;
    p.logits = logits; p.ld = cfg.ld > 0 ? cfg.ld : cfg.n_vocab; p.n_vocab = cfg.n_vocab; p.ts_begin = cfg.ts_begin; p.eot = cfg.eot;
    p.no_timestamps = cfg.no_timestamps; p.no_speech = cfg.no_speech; p.space = cfg.space;
    p.suppress_blank = cfg.suppress_blank; p.timestamps = cfg.timestamps; p.mit = cfg.max_initial_ts_index;
    p.suppress_ids = cfg.suppress_ids; p.n_suppress = cfg.n_suppress; p.sp = sp; p.ts_mode = ts_mode; p.k = k;
    p.out_lp = out_lp; p.out_id = out_id; p.no_speech_out = no_speech; p.out_filtered = out_filtered;
    const size_t shmem = (size_t)((cfg.n_vocab + 31) / 32) * 4;
    p.ws = nullptr; p.chunk = 0;
    static const bool split_off = exp_env("WRS_FILTER_SPLIT") && exp_env("WRS_FILTER_SPLIT")[0] == '0';
    if (cfg.ws && !out_filtered && !split_off && n_rows <= FILTER_SPLIT_MAX_ROWS && n_rows <= cfg.ws_rows && 2 * FILTER_SPLIT * k <= 256 &&
        6 + 4 * k <= FILTER_WS_STRIDE) {
        p.ws = cfg.ws;
        p.chunk = (int)round_up(cdiv(cfg.n_vocab, FILTER_SPLIT), 4);
        filter_topk_kernel<true><<<dim3((unsigned)n_rows, FILTER_SPLIT), 256, shmem, s>>>(p);
        filter_merge_kernel<<<n_rows, 64, 0, s>>>(p, FILTER_SPLIT);
        return;
    }
    filter_topk_kernel<false><<<n_rows, 256, shmem, s>>>(p);
}

}  // namespace wrs
