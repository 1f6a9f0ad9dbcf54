// Persistent loader/consumer ring GEMM for the decode step at >= 480 rows (gfx950).
//
//   C[M, N] = epilogue(A[M, K] W[N, K]^T), A in packed-activation order (kernels.h pa_off), W fragment-packed, fp32 accumulate.
//
// Round 1 ran the six GEMMs of a decoder layer as one tile per workgroup with split-K: every launch paid the pipeline fill,
// and every N = D GEMM wrote 2 fp32 partial-sum slabs (13 MB) that a "finish" kernel re-read, reduced, added to the
// residual stream and normalised (28 us per layer in finish kernels, ~8 us of every ~12 us N = D launch in the slab store:
// DESIGN.md section 3).  Here:
//   * NO split-K and NO slabs: the epilogue produces the consumer's operand directly —
//       RESID   x[m][n] += acc + bias                       (out / cross-out / MLP-2 projections: fp32 residual stream in place)
//       F16     c[m][n]  = fp16((acc + bias) * scale)        (cross-attention query projection)
//       QKV     q -> fp16 [M, D]; k, v -> the self-attention K/V cache at the step's position (Cached::forward, model.rs:83-87)
//       GELU    packed fp16 activations for MLP-2             (model.rs:279-286)
//   * tiles of BM x 128 with BM = 64 for the N = D shapes (200 tiles at 1280 rows: one per CU without splitting K),
//     BM = 128 for N = 3D / 4D;
//   * PERSISTENT workgroups: a workgroup walks several tiles; its 8 loader waves stream k-tiles across tile boundaries
//     into a 3-slot LDS ring with direct-to-LDS loads while its 4 MFMA waves multiply, so the pipeline fill is paid once per
//     workgroup instead of once per tile, and the loads of the next tile's first k-tiles are in flight during an epilogue.
// One barrier per k-tile for all 12 waves (bare s_barrier: __syncthreads() would add a vmcnt(0) that serialises the ring).
#include "kernels.h"
#include <atomic>

namespace wrs {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

struct PringParams {
    const half_t* a;      // packed activations
    const half_t* wp;     // packed weights
    int M, N, K, ksteps;
    int n_wg;             // workgroups launched (multiple of 8)
    PringEpi e;
};

// epilogue of one consumer wave: acc[i][j] holds n = ncol0 + 16 i + 4 (lane >> 4) + r, m = mrow0 + 16 j + (lane & 15)
template <int NJ>
__device__ __forceinline__ void pring_epilogue(const PringParams& p, floatx4 (&acc)[4][NJ], const int mrow0, const int ncolw, const int lane) {
    const PringEpi& e = p.e;
    const int ncol0 = ncolw + 4 * (lane >> 4);
    floatx4 bias4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = ncol0 + i * 16;
        bias4[i] = (e.bias && n < p.N) ? *reinterpret_cast<const floatx4*>(e.bias + n) : floatx4{0.f, 0.f, 0.f, 0.f};
    }
    const int cache_pos = (e.mode == PRING_QKV) ? e.sp->cache_pos : 0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int m = mrow0 + j * 16 + (lane & 15);
        if (m >= p.M) continue;
        if (e.mode == PRING_RESID) {
            float* xrow = e.x + (int64_t)m * e.ldc;
            floatx4 res4[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = ncol0 + i * 16;
                res4[i] = n < p.N ? *reinterpret_cast<const floatx4*>(xrow + n) : floatx4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = ncol0 + i * 16;
                if (n < p.N) *reinterpret_cast<floatx4*>(xrow + n) = res4[i] + (acc[i][j] + bias4[i]);
            }
            continue;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = ncol0 + i * 16;
            if (n >= p.N) continue;
            floatx4 v = acc[i][j] + bias4[i];
            if (n < e.scale_n_end) v *= e.scale;  // scale_n_end is a multiple of 4
            if (e.mode == PRING_GELU_PACKED) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = gelu_exact(v[r]);
            }
            const half4 h = half4{(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
            half_t* dst;
            if (e.mode == PRING_GELU_PACKED) {
                dst = e.c16 + pa_off(m, n, p.N >> 5);
            } else if (e.mode == PRING_QKV) {
                const int region = n / e.d_model, nl = n - region * e.d_model;  // a 4-column group never straddles a region (D % 4 == 0)
                if (region == 0) dst = e.c16 + (int64_t)m * e.ldc + nl;
                else dst = (region == 1 ? e.kcache : e.vcache) + (int64_t)m * e.phys_mul * e.cache_row_stride + (int64_t)cache_pos * e.d_model + nl;
            } else {  // PRING_F16
                dst = e.c16 + (int64_t)m * e.ldc + n;
            }
            *reinterpret_cast<half4*>(dst) = h;
        }
    }
}

template <int BM>
__global__ __launch_bounds__(768) void gemm_f16_pring_kernel(const PringParams p) {
    constexpr int NC = 4, NL = 8;
    constexpr int NJ = BM / 32;                  // 16-row fragments per consumer wave (wave tile = BM/2 rows x 64 columns)
    constexpr int AFR = BM / 8;                  // activation fragments per k-tile (BM/16 row tiles x 2 k-steps)
    constexpr int WFR = 16;                      // weight fragments per k-tile (8 column tiles x 2 k-steps)
    constexpr int STAGE = (AFR + WFR) * 64;      // half8 units per ring slot
    constexpr int PPW = (AFR + WFR) / NL;        // pieces per loader wave per k-tile (4 or 3)
    static_assert((AFR + WFR) % NL == 0, "pieces must divide evenly over the loader waves");
    extern __shared__ __attribute__((aligned(16))) unsigned char pring_smem[];
    half8* smem = reinterpret_cast<half8*>(pring_smem);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;

    // tile schedule.  Tiles are ordered in bands of 8 row tiles x all column tiles walked column-major (neighbours share
    // weight columns and activation rows); XCD x = blockIdx & 7 owns the contiguous chunk [x S, (x+1) S) of that order (one L2
    // per XCD) and its workgroups take the chunk's tiles round-robin.  Speed only: any assignment gives the same result.
    const int mt_n = (p.M + BM - 1) / BM, nt_n = (p.N + 127) >> 7;
    const int tiles = mt_n * nt_n, per_xcd = (tiles + 7) >> 3;
    const int xcd = (int)(blockIdx.x & 7), wg_in_xcd = (int)(blockIdx.x >> 3), wgs_per_xcd = p.n_wg >> 3;
    const int chunk_begin = xcd * per_xcd, chunk_end = min(tiles, chunk_begin + per_xcd);
    const int first = chunk_begin + wg_in_xcd;
    const int n_my = first < chunk_end ? (chunk_end - first + wgs_per_xcd - 1) / wgs_per_xcd : 0;
    if (n_my == 0) return;  // whole workgroup, before any barrier
    auto tile_origin = [&](int ti, int& m0, int& n0) {
        const int seq = first + ti * wgs_per_xcd;
        const int band = seq / (8 * nt_n);
        const int band_m0 = band * 8;
        const int band_rows = min(8, mt_n - band_m0);
        const int r = seq - band * 8 * nt_n;
        m0 = (band_m0 + r % band_rows) * BM;
        n0 = (r / band_rows) * 128;
    };
    const int nk = p.K / 64;
    const int total = n_my * nk;  // k-tiles this workgroup streams
#define WRS_PRING_BARRIER() asm volatile("s_barrier" ::: "memory")

    if (wave >= NC) {
        // ---------------- loader waves: pieces q = lw + NL i of every k-tile (q < AFR: activation fragment q, else weight fragment q - AFR)
        const int lw = wave - NC;
        int iss_tile = 0, iss_kt = 0;  // cursor of the next k-tile to request
        const half_t* src[PPW];
        auto set_tile = [&](int ti) {
            int m0, n0;
            tile_origin(ti, m0, n0);
#pragma unroll
            for (int i = 0; i < PPW; ++i) {
                const int q = lw + NL * i;
                if (q < AFR) src[i] = p.a + ((int64_t)(m0 / 16 + (q >> 1)) * p.ksteps + (q & 1)) * 512 + lane * 8;
                else src[i] = p.wp + ((int64_t)(n0 / 16 + ((q - AFR) >> 1)) * p.ksteps + ((q - AFR) & 1)) * 512 + lane * 8;
            }
        };
        set_tile(0);
        auto issue = [&](int u) {
            half8* st_ = smem + (u % 3) * STAGE;
#pragma unroll
            for (int i = 0; i < PPW; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (int64_t)iss_kt * 1024),
                                                 (__attribute__((address_space(3))) void*)(st_ + (lw + NL * i) * 64), 16, 0, 0);
            if (++iss_kt == nk) {
                iss_kt = 0;
                if (++iss_tile < n_my) set_tile(iss_tile);
            }
        };
#define WRS_PRING_KEEP_ONE() do { if (PPW == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); } while (0)
        issue(0);
        if (total > 1) { issue(1); WRS_PRING_KEEP_ONE(); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        WRS_PRING_BARRIER();  // k-tile 0 has landed
        for (int u = 0; u < total; ++u) {
            // slot of k-tile u+2 = slot of k-tile u-1, which the consumers left at the previous barrier
            if (u + 2 < total) {
                issue(u + 2);
                WRS_PRING_KEEP_ONE();  // k-tile u+1 landed, u+2 in flight
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            WRS_PRING_BARRIER();
        }
#undef WRS_PRING_KEEP_ONE
        return;
    }
    // ---------------- consumer waves
    const int wy = wave >> 1, wx = wave & 1;  // 64-column block, row half
    floatx4 acc[4][NJ];
    int u = 0;
    WRS_PRING_BARRIER();
    for (int ti = 0; ti < n_my; ++ti) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
        for (int kt = 0; kt < nk; ++kt, ++u) {
            const half8* As = smem + (u % 3) * STAGE;
            const half8* Ws = As + AFR * 64;
            half8 wf[2][4], af[2][NJ];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int i = 0; i < 4; ++i) wf[ks][i] = Ws[((wy * 4 + i) * 2 + ks) * 64 + lane];
#pragma unroll
                for (int j = 0; j < NJ; ++j) af[ks][j] = As[((wx * NJ + j) * 2 + ks) * 64 + lane];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks][i], af[ks][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            WRS_PRING_BARRIER();
        }
        int m0, n0;
        tile_origin(ti, m0, n0);
        pring_epilogue<NJ>(p, acc, m0 + wx * (BM / 2), n0 + wy * 64, lane);
    }
#undef WRS_PRING_BARRIER
}

static int g_pring_last_wgs = 0;
int pring_debug_last_wgs() { return g_pring_last_wgs; }

void launch_gemm_pring(const half_t* a_packed, int64_t packed_rows, const half_t* w_packed, int64_t M, int64_t N, int64_t K, int bm,
                       const PringEpi& e, hipStream_t s) {
    WRS_REQUIRE(M > 0 && N > 0 && K > 0 && K % 64 == 0 && N % 4 == 0 && M < (1ll << 24) && N < (1ll << 24), "gemm_pring: bad shape M=%ld N=%ld K=%ld", (long)M, (long)N, (long)K);
    WRS_REQUIRE(bm == 64 || bm == 128, "gemm_pring: tile rows must be 64 or 128");
    WRS_REQUIRE(packed_rows >= round_up(M, bm), "gemm_pring: packed A holds %ld rows, needs %ld", (long)packed_rows, (long)round_up(M, bm));
    WRS_REQUIRE(e.scale_n_end % 4 == 0, "gemm_pring: scale_n_end must be a multiple of 4");
    switch (e.mode) {
        case PRING_RESID: WRS_REQUIRE(e.x && e.ldc >= N && e.ldc % 4 == 0, "gemm_pring: RESID needs x and ldc"); break;
        case PRING_F16: WRS_REQUIRE(e.c16 && e.ldc >= N && e.ldc % 4 == 0, "gemm_pring: F16 needs c16 and ldc"); break;
        case PRING_GELU_PACKED: WRS_REQUIRE(e.c16 && N % 32 == 0, "gemm_pring: packed output needs N %% 32 == 0"); break;
        case PRING_QKV:
            WRS_REQUIRE(e.c16 && e.kcache && e.vcache && e.sp && e.d_model > 0 && N == 3 * (int64_t)e.d_model && e.d_model % 4 == 0 && e.ldc >= e.d_model &&
                            e.ldc % 4 == 0 && e.phys_mul >= 1 && e.cache_row_stride > 0, "gemm_pring: bad QKV epilogue");
            break;
        default: WRS_THROW(WRS_EINVAL, "gemm_pring: unknown epilogue mode %d", e.mode);
    }
    PringParams p;
    p.a = a_packed; p.wp = w_packed; p.M = (int)M; p.N = (int)N; p.K = (int)K; p.ksteps = (int)(K / 32); p.e = e;
    const int64_t tiles = cdiv(M, bm) * cdiv(N, 128);
    // workgroups: one per CU at most (a 12-wave workgroup with a 72-96 KiB ring fills a CU); with more tiles than CUs every
    // workgroup takes the same number of tiles (400 tiles -> 200 workgroups x 2, 300 -> 150 x 2) instead of 256 + a ragged tail
    const int64_t rounds = cdiv(tiles, 256);
    int64_t wgs = cdiv(tiles, rounds);
    wgs = round_up(wgs, 8);
    p.n_wg = (int)wgs;
    g_pring_last_wgs = p.n_wg;
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    if (bm == 64) {
        const size_t shmem = 3 * (8 + 16) * 1024;
        static std::atomic<bool> attr[64];
        if (dev >= 0 && dev < 64 && !attr[dev].load(std::memory_order_acquire)) {
            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_pring_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
            attr[dev].store(true, std::memory_order_release);
        }
        gemm_f16_pring_kernel<64><<<(unsigned)wgs, 768, shmem, s>>>(p);
    } else {
        const size_t shmem = 3 * (16 + 16) * 1024;
        static std::atomic<bool> attr[64];
        if (dev >= 0 && dev < 64 && !attr[dev].load(std::memory_order_acquire)) {
            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_pring_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
            attr[dev].store(true, std::memory_order_release);
        }
        gemm_f16_pring_kernel<128><<<(unsigned)wgs, 768, shmem, s>>>(p);
    }
}

}  // namespace wrs
