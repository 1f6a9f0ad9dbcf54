// CU confinement experiment (round 3; NOT part of the library — see profiles/r03_cu_sharing_experiments.log for what it showed).
// Kernels of a "confined" launch work only on a fixed set of physical CUs: the launch is persistent (one workgroup per CU), a
// workgroup looks its (XCC, SE, SH, CU) id up in a role table and, off the set, leaves at once; on it, it pulls tiles from a queue.
// Placement is used for speed only: whoever lands wherever, every tile is computed.  Include AFTER csrc/gemm.hip.
#pragma once
#include <algorithm>
namespace wrs {
// In overlapped mode the next window batch's encoder runs on a side stream while this batch decodes.  Left to the hardware
// dispatcher the two streams time-slice the CUs at workgroup granularity and the encoder's long-lived 256x256 GEMM workgroups
// (30-130 us each, a whole CU's LDS) hold up the decode chain's short dependent kernels (a 15-us GEMM takes 100 us,
// profiles/r03_overlap_trace_*.txt).  Confined kernels instead work only on a fixed set of physical CUs: every launch of them
// is persistent (one workgroup per CU), a workgroup looks its (XCC, SE, SH, CU) id up in a role table and, off the set, leaves at
// once; on it, it pulls tiles from a queue.  Placement is used for speed only: whoever lands wherever, every tile is computed.
constexpr int CU_ROLE_KEYS = 2048;   // (xcc << 8) | (se << 5) | (sh << 4) | cu
constexpr int CU_CTR_INTS = 16;      // counters of one confined launch: [0] workers started, [1..8] queue heads of the 8 XCD chunks
struct CuConfine {
    const unsigned char* roles = nullptr;  // device [CU_ROLE_KEYS]: 1 = a CU confined kernels work on
    int* ctr = nullptr;                    // device ring of n_slots x CU_CTR_INTS ints; a slot is zero when a launch takes it
    int n_slots = 0, next = 0;             // host: slot of the next launch (reset() zeroes the ring and rewinds)
    int n_cus = 256;                       // CUs of the device = workgroups of a confined launch
    int* take();
    void reset(hipStream_t s);
};
// roles_dev [CU_ROLE_KEYS] bytes: marks `per_xcd` CUs of every XCD (spread over its shader engines) with 1.  Returns the CUs seen.
int build_cu_roles(int per_xcd, unsigned char* roles_dev, int* n_marked, hipStream_t s);

struct CuWork {                  // kernel argument of a confined launch
    const unsigned char* roles;  // [CU_ROLE_KEYS], 1 = a CU this launch works on
    int* ctr;                    // CU_CTR_INTS ints, zero before the launch
};

__device__ __forceinline__ int hw_cu_key() {
    // HW_REG_HW_ID (id 4): CU_ID [11:8], SH_ID [12], SE_ID [15:13]; HW_REG_XCC_ID (id 20): XCC_ID [3:0]
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xF;
    return (int)(((xcc & 7) << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF));
}

// Does this workgroup work (true) or leave (false)?  On a CU of the set: work.  Elsewhere: leave as soon as a worker exists —
// workers stay until every queue is empty, so whatever leaves has seen somebody who finishes the job.  If none shows up
// within 20 us (bounded wait: nothing here can hang) the workgroup works where it is: slower, never wrong.
// All threads of the workgroup call this; s_tmp is one int of shared memory.
__device__ __forceinline__ bool cu_confine_enter(const CuWork& cw, const int key, int* s_tmp) {
    const int tid = threadIdx.x;
    if (!cw.roles[key]) {
        if (tid == 0) {
            int st = __hip_atomic_load(cw.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long t0 = wall_clock64();  // 100 MHz
            while (st == 0 && wall_clock64() - t0 < 2000) {
                __builtin_amdgcn_s_sleep(16);
                st = __hip_atomic_load(cw.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            *s_tmp = st;
        }
        __syncthreads();
        const int st = *s_tmp;
        __syncthreads();
        if (st > 0) return false;
    }
    if (tid == 0) atomicAdd(cw.ctr, 1);
    return true;
}

// Next item of a sequence of `total` items cut into 8 chunks of `per_xcd` (chunk x = the items an XCD-aware order gives XCD x):
// the own XCD's chunk first, then the other XCDs' (what is left when they are slower).  `v` = chunks this workgroup has given
// up on (start at 0).  Returns the item or -1.  All threads call it; s_tmp is one int of shared memory; it contains the
// barrier that makes s_tmp reusable, and one more before the caller may overwrite other shared memory is NOT implied.
__device__ __forceinline__ int cu_queue_next(const CuWork& cw, const int xcd, int& v, const int per_xcd, const int total, int* s_tmp) {
    for (;;) {
        if (v >= 8) return -1;
        const int x = (xcd + v) & 7;
        if (threadIdx.x == 0) *s_tmp = atomicAdd(&cw.ctr[1 + x], 1);
        __syncthreads();
        const int j = *s_tmp;
        __syncthreads();
        const int item = x * per_xcd + j;
        if (j < per_xcd && item < total) return item;
        ++v;
    }
}

__global__ void cu_census_kernel(int* seen) {
    if (threadIdx.x == 0) atomicAdd(&seen[hw_cu_key()], 1);
    const unsigned long long t0 = wall_clock64();  // 100 MHz: stay 20 us so that the grid spreads over every CU
    while (wall_clock64() - t0 < 2000) {}
}

int build_cu_roles(int per_xcd, unsigned char* roles_dev, int* n_marked, hipStream_t s) {
    DevBuf seen;
    seen.alloc(CU_ROLE_KEYS * sizeof(int));
    cu_census_kernel<<<16384, 64, 0, s>>>(seen.as<int>());
    std::vector<int> h(CU_ROLE_KEYS);
    HIP_CHECK(hipMemcpyAsync(h.data(), seen.p, CU_ROLE_KEYS * sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    std::vector<unsigned char> role(CU_ROLE_KEYS, 0);
    int n_cu = 0, n_res = 0;
    for (int xcc = 0; xcc < 8; ++xcc) {
        int cus = 0;
        for (int k = xcc << 8; k < (xcc + 1) << 8; ++k) cus += h[k] > 0;
        const int want = std::min(per_xcd, std::max(cus - 1, 0));  // never every CU of an XCD
        int i = 0;
        for (int k = xcc << 8; k < (xcc + 1) << 8; ++k)
            if (h[k] > 0) {
                // CU i of the XCD (in id order: shader engine major) is marked iff floor((i+1) * want / cus) > floor(i * want / cus)
                if (((i + 1) * want) / cus > (i * want) / cus) { role[k] = 1; ++n_res; }
                ++i;
            }
        n_cu += cus;
    }
    HIP_CHECK(hipMemcpyAsync(roles_dev, role.data(), CU_ROLE_KEYS, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipStreamSynchronize(s));
    if (n_marked) *n_marked = n_res;
    return n_cu;
}

int* CuConfine::take() {
    WRS_REQUIRE(ctr && next < n_slots, "CU confinement: counter ring exhausted (%d launches since the last reset)", next);
    return ctr + (size_t)(next++) * CU_CTR_INTS;
}
void CuConfine::reset(hipStream_t s) {
    static const bool stats = getenv("WRS_CONFINE_STATS") != nullptr;
    if (stats && next > 0) {  // diagnosis: workers per launch of the pass that just ended (waits for it)
        std::vector<int> h((size_t)next * CU_CTR_INTS);
        HIP_CHECK(hipStreamSynchronize(s));
        HIP_CHECK(hipMemcpy(h.data(), ctr, h.size() * sizeof(int), hipMemcpyDeviceToHost));
        int lo = 1 << 30, hi = 0;
        long sum = 0;
        for (int i = 0; i < next; ++i) { const int w = h[(size_t)i * CU_CTR_INTS]; lo = std::min(lo, w); hi = std::max(hi, w); sum += w; }
        fprintf(stderr, "[confine] %d launches: workers per launch min %d avg %.1f max %d\n", next, lo, (double)sum / next, hi);
    }
    HIP_CHECK(hipMemsetAsync(ctr, 0, (size_t)n_slots * CU_CTR_INTS * sizeof(int), s));
    next = 0;
}


// The same tiles by a CONFINED launch (kernels.h "CU confinement of the side stream"): one persistent workgroup per CU; the
// workgroups on the side stream's CUs pull tiles — their own XCD's chunk of the XCD-aware sequence first, so a chunk's tiles
// still share their operand panels through one L2 — and the others leave.  No split-K (the launcher checks).
__global__ __launch_bounds__(512) void gemm_f16_pp2_cu_kernel(const GemmParams p, const CuWork cw) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gemm_smem[];
    __shared__ int s_tmp;
    const int key = hw_cu_key();
    if (!cu_confine_enter(cw, key, &s_tmp)) return;
    const int mt_n = (p.M + 255) >> 8, nt_n = p.N >> 8;
    const int tiles = mt_n * nt_n, per_xcd = (tiles + 7) >> 3;
    int v = 0;
    for (;;) {
        const int seq = cu_queue_next(cw, key >> 8, v, per_xcd, tiles, &s_tmp);  // ends with a barrier: every wave has left the previous tile's ring
        if (seq < 0) break;
        pp2_tile<false>(p, seq, 0, reinterpret_cast<half8*>(gemm_smem));
    }
}


// C = A W^T by the confined 256x256 kernel (row-major A, N a multiple of 256, no split-K)
static void launch_gemm_f16_confined(const GemmA& A, const half_t* w_packed, int64_t M, int64_t N, int64_t K, const GemmEpi& e, CuConfine& cc, hipStream_t s) {
    WRS_REQUIRE(!A.packed && N % 256 == 0 && K % 64 == 0 && e.ksplit <= 1, "confined GEMM: unsupported shape");
    GemmParams p{};
    p.a = A.a; p.lda = A.lda; p.a_rpb = A.rows_per_batch; p.a_bs = A.batch_stride;
    p.wp = w_packed; p.M = (int)M; p.N = (int)N; p.K = (int)K; p.ksteps = (int)(K / 32);
    p.bias = e.bias; p.scale = e.scale; p.scale_n_end = e.scale_n_end; p.act = e.act;
    p.residual = e.residual; p.posemb = e.posemb; p.c = e.c; p.c_f32 = e.c_f32; p.ldc = e.ldc;
    p.c_rpb = e.c_rows_per_batch; p.c_brows = e.c_batch_rows; p.c_roff = e.c_row_off; p.c_packed = e.c_packed;
    p.ksplit = 1;
    const size_t shmem = 4 * 2048 * 16;
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_pp2_cu_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    const CuWork cw{cc.roles, cc.take()};
    gemm_f16_pp2_cu_kernel<<<(unsigned)cc.n_cus, 512, shmem, s>>>(p, cw);
}
}  // namespace wrs
