// Second form of scripts/micro/fused_xa_gemm.cpp.  There a GEMM workgroup and cross-attention workgroups shared a CU and the
// GEMM's L2 requests queued behind the attention's HBM requests in the CU's memory pipeline (GEMM 2-2.5x slower, no net gain).
// Here the roles are split by PHYSICAL CU inside one launch: 512 persistent workgroups (2 per CU); a workgroup reads its
// (XCC, SE, CU) id, looks its CU's role up in a table and pulls either GEMM tiles or attention units (audio, head, key part)
// from a queue.  Nobody waits for anybody; placement is used for speed only.
//   hipcc -O3 --offload-arch=gfx950 -o scripts/micro/fused_cu_roles scripts/micro/fused_cu_roles.cpp
//   scripts/micro/fused_cu_roles [audios 128] [layers 8] [gemm CUs per XCD 8] [key parts per unit 1]
#include "../../whisper_rs_amd/csrc/common.cpp"
#include "../../whisper_rs_amd/csrc/kernels.h"
#include <chrono>
#include <algorithm>
using namespace wrs;

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

struct XaP {
    const half_t* q16;  // [R, D]
    const half_t* kp;
    const half_t* vp;
    int T, Tp, H, G;
    half_t* out;        // packed
    int n_units;        // n_audio * H
    int* queue;         // next unit
    int parts;          // key parts per (audio, head): a unit is (audio, head, part); parts > 1 writes partial (m, l, o) to `partial`
    float* partial;     // [n_units][8][66]
};
struct GmP {
    const half_t* a;    // packed activations
    const half_t* wp;
    int M, N, K, ksteps, ksplit;
    float* c;           // [ksplit][M][N] fp32
    int64_t slab_stride;
    int n_blocks;       // tiles * ksplit
    int* done;          // GEMM tiles finished
    int* tile_queue;    // next tile
};

template <int PF>
__device__ __forceinline__ void xa_unit(const XaP& p, const int h, const int b, const int part, const int unit, float* smem) {
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int qcol = lane & 15, qg = lane >> 4;
    const int64_t qrow0 = (int64_t)b * p.G;
    half8 qf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        half8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (qcol < p.G) v = *reinterpret_cast<const half8*>(p.q16 + (qrow0 + qcol) * (p.H * 64) + h * 64 + ks * 32 + 8 * qg);
        qf[ks] = v;
    }
    const int64_t head_base = ((int64_t)b * p.H + h) * (int64_t)p.Tp * 64;
    const half_t* kbase = p.kp + head_base + lane * 8;
    const half_t* vbase = p.vp + head_base + lane * 8;
    const int nb_all = p.Tp / 32, nbp = (nb_all + p.parts - 1) / p.parts;
    const int kb_first = part * nbp, n_blocks = min(nb_all, kb_first + nbp);
    floatx4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = floatx4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    half8 kf[PF][4], vf[PF][4];
#pragma unroll
    for (int d = 0; d < PF; ++d) {
        const int kb = kb_first + wave + 8 * d;
        if (kb < n_blocks) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                kf[d][i] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(kbase + ((int64_t)kb * 4 + i) * 512));
                vf[d][i] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(vbase + ((int64_t)kb * 4 + i) * 512));
            }
        }
    }
    auto compute = [&](const half8 (&kc)[4], const half8 (&vc)[4], const int kb) {
        floatx4 s0 = floatx4{0.f, 0.f, 0.f, 0.f}, s1 = floatx4{0.f, 0.f, 0.f, 0.f};
        s0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(kc[0], qf[0], s0, 0, 0, 0);
        s0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(kc[1], qf[1], s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(kc[2], qf[0], s1, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(kc[3], qf[1], s1, 0, 0, 0);
        float sv[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
        if (kb * 32 + 32 > p.T) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int key = kb * 32 + (j < 4 ? 4 * qg + j : 16 + 4 * qg + (j - 4));
                if (key >= p.T) sv[j] = -INFINITY;
            }
        }
        float bm = fmaxf(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])), fmaxf(fmaxf(sv[4], sv[5]), fmaxf(sv[6], sv[7])));
        bm = fmaxf(bm, __shfl_xor(bm, 16));
        bm = fmaxf(bm, __shfl_xor(bm, 32));
        const float m_new = fmaxf(m_run, bm);
        const float alpha = __expf(m_run - m_new);
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
        l_run = l_run * alpha + ps;
        m_run = m_new;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            o[dt] *= alpha;
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vc[dt], pf, o[dt], 0, 0, 0);
        }
    };
    if (PF == 1) {
        for (int kb = kb_first + wave; kb < n_blocks; kb += 8) {
            half8 kc[4], vc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                kc[i] = kf[0][i];
                vc[i] = vf[0][i];
            }
            const int nxt = kb + 8;
            if (nxt < n_blocks) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    kf[0][i] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(kbase + ((int64_t)nxt * 4 + i) * 512));
                    vf[0][i] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(vbase + ((int64_t)nxt * 4 + i) * 512));
                }
            }
            compute(kc, vc, kb);
        }
    } else {
        for (int kb0 = kb_first + wave; kb0 < n_blocks; kb0 += 8 * PF) {
#pragma unroll
            for (int d = 0; d < PF; ++d) {
                const int kb = kb0 + 8 * d;
                if (kb < n_blocks) {
                    compute(kf[d], vf[d], kb);
                    const int nxt = kb + 8 * PF;
                    if (nxt < n_blocks) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            kf[d][i] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(kbase + ((int64_t)nxt * 4 + i) * 512));
                            vf[d][i] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(vbase + ((int64_t)nxt * 4 + i) * 512));
                        }
                    }
                }
            }
        }
    }
    // merge of the 8 waves through LDS: sm [8][16], sl [8][16], so [8][64][9] (G <= 8 query columns kept)
    float* sm = smem;
    float* sl = smem + 128;
    float* so = smem + 256;
    if (qg == 0) {
        sm[wave * 16 + qcol] = m_run;
        sl[wave * 16 + qcol] = l_run;
    }
    if (qcol < 8) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) so[(wave * 64 + dt * 16 + 4 * qg + r) * 9 + qcol] = o[dt][r];
    }
    __syncthreads();
    const int qq = tid >> 4, d0 = (tid & 15) * 4;
    if (tid < 256 && qq < p.G) {
        float mm = sm[qq];
#pragma unroll
        for (int w = 1; w < 8; ++w) mm = fmaxf(mm, sm[w * 16 + qq]);
        float wgt[8], lsum = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            wgt[w] = __expf(sm[w * 16 + qq] - mm);
            lsum += wgt[w] * sl[w * 16 + qq];
        }
        const float inv = p.parts > 1 ? 1.f : 1.f / lsum;
        float r4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) a += wgt[w] * so[(w * 64 + d0 + r) * 9 + qq];
            r4[r] = a * inv;
        }
        if (p.parts > 1) {
            float* dst = p.partial + ((int64_t)unit * 8 + qq) * 66;
            *reinterpret_cast<float2*>(dst + d0) = float2{r4[0], r4[1]};
            *reinterpret_cast<float2*>(dst + d0 + 2) = float2{r4[2], r4[3]};
            if (d0 == 0) { dst[64] = mm; dst[65] = lsum; }
        } else {
            *reinterpret_cast<half4*>(p.out + pa_off((int)(qrow0 + qq), h * 64 + d0, p.H * 2)) =
                half4{(_Float16)r4[0], (_Float16)r4[1], (_Float16)r4[2], (_Float16)r4[3]};
        }
    }
    __syncthreads();  // smem is reused by the next unit
}

// 8 waves: one 128x128 tile, k-tiles of 64, 2 stages of 32 KiB; wave = 64 rows x 32 columns (acc 32 registers), fp32 output.
// (the 4-wave body of gemm_f16_kernel<true> needs 152 registers: two 8-wave workgroups per CU leave 128)
__device__ __forceinline__ void gemm_tile(const GmP& p, const int blk, half8* lds) {
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int mt_n = (p.M + 127) >> 7, nt_n = (p.N + 127) >> 7;
    const int tiles = mt_n * nt_n;
    const int seq = blk % tiles, ksl = blk / tiles;
    const int m0 = (seq % mt_n) * 128, n0 = (seq / mt_n) * 128;
    const half_t* a_ptr[2];
    const half_t* w_ptr[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int f = wave * 2 + j, t16 = f >> 1, ks = f & 1;
        a_ptr[j] = p.a + ((int64_t)(m0 / 16 + t16) * p.ksteps + ks) * 512 + lane * 8;
        w_ptr[j] = p.wp + ((int64_t)(n0 / 16 + t16) * p.ksteps + ks) * 512 + lane * 8;
    }
    floatx4 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
    const int wy = wave >> 1, wx = wave & 1;
    const int nk = p.K / 64 / p.ksplit;
    const int kt0 = ksl * nk;
    // lds [buf][0 = activations, 1 = weights][16 fragments][64 lanes]
#define STAGE(buf_, kt_)                                                                                                       \
    do {                                                                                                                       \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                                        \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_ptr[j] + (int64_t)(kt_) * 1024), \
                                             (__attribute__((address_space(3))) void*)(lds + (((buf_) * 2 + 0) * 16 + wave * 2 + j) * 64), 16, 0, 0); \
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w_ptr[j] + (int64_t)(kt_) * 1024), \
                                             (__attribute__((address_space(3))) void*)(lds + (((buf_) * 2 + 1) * 16 + wave * 2 + j) * 64), 16, 0, 0); \
        }                                                                                                                      \
    } while (0)
    STAGE(0, kt0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) STAGE(buf ^ 1, kt0 + kt + 1);
        half8 wf[2][2], af[2][4];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < 2; ++i) wf[ks][i] = lds[((buf * 2 + 1) * 16 + (wy * 2 + i) * 2 + ks) * 64 + lane];
#pragma unroll
            for (int j = 0; j < 4; ++j) af[ks][j] = lds[((buf * 2 + 0) * 16 + (wx * 4 + j) * 2 + ks) * 64 + lane];
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks][i], af[ks][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#undef STAGE
    float* slab = p.c + (int64_t)ksl * p.slab_stride;
    const int ncol0 = n0 + wy * 32 + 4 * (lane >> 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t m = (int64_t)m0 + wx * 64 + j * 16 + (lane & 15);
        if (m >= p.M) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int n = ncol0 + i * 16;
            if (n < p.N) *reinterpret_cast<floatx4*>(slab + m * p.N + n) = acc[i][j];
        }
    }
}

struct Stamps {  // 100 MHz wall clock: [0] GEMM first start, [1] GEMM last end, [2] attention first start, [3] attention last end
    unsigned long long t[4];
};

__device__ __forceinline__ int hw_cu_key() {
    // HW_REG_HW_ID (id 4): CU_ID [11:8], SH_ID [12], SE_ID [15:13]; HW_REG_XCC_ID (id 20): XCC_ID [3:0]
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xF;
    return (int)(((xcc & 7) << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF));
}
__global__ void census_kernel(int* seen) {
    if (threadIdx.x == 0) atomicAdd(&seen[hw_cu_key()], 1);
    // stay a little so the launch spreads over every CU
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 2000) {}
}

// 512 persistent workgroups.  role[key] = 1: this CU multiplies (pulls GEMM tiles); 0: it streams (pulls attention units, and
// with stop_on_done stops pulling once every GEMM tile of the launch has finished).
template <int PF>
__global__ __launch_bounds__(512, 4) void fused_kernel(const XaP x, const GmP g, const unsigned char* __restrict__ role, const int stop_on_done, Stamps* st) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ int s_unit;
    const int tid = threadIdx.x;
    const int my_role = role[hw_cu_key()];
    if (my_role == 1) {
        if (g.n_blocks <= 0) return;
        if (st && tid == 0) atomicMin(&st->t[0], wall_clock64());
        int mine = 0;
        for (;;) {
            if (tid == 0) s_unit = atomicAdd(g.tile_queue, 1);
            __syncthreads();
            const int t = s_unit;
            if (t >= g.n_blocks) break;
            gemm_tile(g, t, reinterpret_cast<half8*>(smem_raw));
            ++mine;
            __syncthreads();
        }
        if (tid == 0) {
            if (mine) atomicAdd(g.done, mine);
            if (st) atomicMax(&st->t[1], wall_clock64());
        }
        return;
    }
    if (x.n_units <= 0) return;
    if (st && tid == 0) atomicMin(&st->t[2], wall_clock64());
    for (;;) {
        if (tid == 0) {
            int u = -1;
            const bool stop = stop_on_done && g.n_blocks > 0 && __hip_atomic_load(g.done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= g.n_blocks;
            if (!stop) {
                u = atomicAdd(x.queue, 1);
                if (u >= x.n_units) u = -1;
            }
            s_unit = u;
        }
        __syncthreads();
        const int u = s_unit;
        if (u < 0) break;
        const int uh = u / x.parts;
        xa_unit<PF>(x, uh % x.H, uh / x.H, u % x.parts, u, reinterpret_cast<float*>(smem_raw));  // ends with a barrier: s_unit may be rewritten
    }
    if (st && tid == 0) atomicMax(&st->t[3], wall_clock64());
}

static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Shape { const char* name; int N, K, ksplit; };

int main(int argc, char** argv) {
    const int A = argc > 1 ? atoi(argv[1]) : 128, G = 5, D = 1280, H = 20, T = 1500, Tp = 1504;
    const int LAYERS = argc > 2 ? atoi(argv[2]) : 8;        // distinct K/V buffers walked per timed graph (HBM-cold)
    const int GPX = argc > 3 ? atoi(argv[3]) : 8;           // CUs per XCD that multiply
    const int PARTS = argc > 4 ? atoi(argv[4]) : 1;         // key parts per (audio, head)
    const int LDS = 65536, WGS = 512;
    const int R = A * G, Rp = (int)round_up(R, 128);
    hipStream_t s;
    HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const size_t per_layer = (size_t)A * H * Tp * 64;
    DevBuf kp, vp, q16, xo, act, act4, w[4], cbuf, ctr, stamps, seen, role_split, role_all0, role_all1, partial;
    kp.alloc(per_layer * LAYERS * 2); vp.alloc(per_layer * LAYERS * 2);
    HIP_CHECK(hipMemset(kp.p, 0x11, kp.bytes)); HIP_CHECK(hipMemset(vp.p, 0x22, vp.bytes));
    q16.alloc((size_t)Rp * D * 2); xo.alloc((size_t)Rp * D * 2);
    act.alloc((size_t)Rp * D * 2); act4.alloc((size_t)Rp * 4 * D * 2);
    HIP_CHECK(hipMemset(q16.p, 0x2c, q16.bytes)); HIP_CHECK(hipMemset(act.p, 0x2c, act.bytes)); HIP_CHECK(hipMemset(act4.p, 0x2c, act4.bytes));
    const Shape shapes[6] = {{"out  N=D  K=D ", D, D, 2}, {"mlp1 N=4D K=D ", 4 * D, D, 1}, {"mlp2 N=D  K=4D", D, 4 * D, 4},
                             {"qkv  N=3D K=D ", 3 * D, D, 1}, {"out  N=D  K=D ", D, D, 2}, {"q    N=D  K=D ", D, D, 2}};
    w[0].alloc((size_t)D * D * 2); w[1].alloc((size_t)4 * D * D * 2); w[2].alloc((size_t)4 * D * D * 2); w[3].alloc((size_t)3 * D * D * 2);
    for (auto& b : w) HIP_CHECK(hipMemset(b.p, 0x1c, b.bytes));
    cbuf.alloc((size_t)4 * Rp * 4 * D * 4);
    ctr.alloc(4096); stamps.alloc(sizeof(Stamps) * 64);
    partial.alloc((size_t)A * H * PARTS * 8 * 66 * 4);
    seen.alloc(2048 * 4); role_split.alloc(2048); role_all0.alloc(2048); role_all1.alloc(2048);
    HIP_CHECK(hipMemset(role_all1.p, 1, 2048));
    HIP_CHECK(hipDeviceSynchronize());
    int* ctr_i = ctr.as<int>();

    // census of the physical CUs
    census_kernel<<<16384, 64, 0, s>>>(seen.as<int>());
    HIP_CHECK(hipStreamSynchronize(s));
    std::vector<int> hseen(2048);
    HIP_CHECK(hipMemcpy(hseen.data(), seen.p, 2048 * 4, hipMemcpyDeviceToHost));
    std::vector<unsigned char> hrole(2048, 0);
    int n_cu = 0, n_gemm_cu = 0, per_xcc[8] = {0};
    for (int xcc = 0; xcc < 8; ++xcc) {
        int taken = 0;
        for (int k = xcc << 8; k < (xcc + 1) << 8; ++k)
            if (hseen[k] > 0) {
                ++n_cu; ++per_xcc[xcc];
                // spread the multiplying CUs over the shader engines: take every (cus/GPX)-th CU of the XCD
                if (taken < GPX && (per_xcc[xcc] - 1) % std::max(1, 32 / std::max(1, GPX)) == 0) { hrole[k] = 1; ++taken; ++n_gemm_cu; }
            }
    }
    HIP_CHECK(hipMemcpy(role_split.p, hrole.data(), 2048, hipMemcpyHostToDevice));
    printf("census: %d CUs seen (per XCC:", n_cu);
    for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]);
    printf("); %d multiply, %d stream\n", n_gemm_cu, n_cu - n_gemm_cu);

    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fused_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fused_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS));

    auto gm = [&](int si) {
        const Shape& sh = shapes[si];
        GmP g{};
        g.a = (sh.K == D ? act : act4).as<half_t>();
        g.wp = w[si == 0 || si >= 4 ? 0 : si].as<half_t>();
        g.M = R; g.N = sh.N; g.K = sh.K; g.ksteps = sh.K / 32; g.ksplit = sh.ksplit;
        g.c = cbuf.as<float>(); g.slab_stride = (int64_t)Rp * sh.N;
        g.n_blocks = (int)(cdiv(R, 128) * cdiv(sh.N, 128)) * sh.ksplit;
        g.done = ctr_i + 16 + si;
        g.tile_queue = ctr_i + 32 + si;
        return g;
    };
    auto xa = [&](int layer) {
        XaP x{};
        x.q16 = q16.as<half_t>(); x.kp = kp.as<half_t>() + layer * per_layer; x.vp = vp.as<half_t>() + layer * per_layer;
        x.T = T; x.Tp = Tp; x.H = H; x.G = G; x.out = xo.as<half_t>(); x.n_units = A * H * PARTS; x.queue = ctr_i + 64 + layer;
        x.parts = PARTS; x.partial = partial.as<float>();
        return x;
    };
    GmP g_none{}; g_none.done = ctr_i; g_none.tile_queue = ctr_i + 2;
    XaP x_none{}; x_none.queue = ctr_i + 1; x_none.parts = 1;
    const unsigned char* RS = role_split.as<unsigned char>();
    const unsigned char* R0 = role_all0.as<unsigned char>();
    const unsigned char* R1 = role_all1.as<unsigned char>();

    auto time_graph = [&](auto&& body, int reps = 5) {
        hipGraph_t gr; hipGraphExec_t ge;
        HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        body();
        HIP_CHECK(hipStreamEndCapture(s, &gr));
        HIP_CHECK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
        double best = 1e30;
        for (int rep = 0; rep < reps; ++rep) {
            HIP_CHECK(hipMemsetAsync(ctr.p, 0, 4096, s));
            HIP_CHECK(hipStreamSynchronize(s));
            const double t0 = now();
            HIP_CHECK(hipGraphLaunch(ge, s));
            HIP_CHECK(hipStreamSynchronize(s));
            best = std::min(best, now() - t0);
        }
        HIP_CHECK(hipGraphExecDestroy(ge)); HIP_CHECK(hipGraphDestroy(gr));
        return best / LAYERS;
    };
    auto reset = [&]() { HIP_CHECK(hipMemsetAsync(ctr.p, 0, 1024, s)); };

    if (argc > 5 && atoi(argv[5]) == 1) {  // sweep: the cross-attention alone against the number of streaming CUs
        DevBuf role_n; role_n.alloc(2048);
        for (int PF = 1; PF <= 2; ++PF)
            for (int stream_per_xcc = 32; stream_per_xcc >= 8; stream_per_xcc -= 4) {
                std::vector<unsigned char> hr(2048, 1);
                for (int xcc = 0; xcc < 8; ++xcc) {
                    // streaming CUs spread over the XCD's CU list: CU i streams iff floor((i+1)*n/32) > floor(i*n/32)
                    int i = 0;
                    for (int k = xcc << 8; k < (xcc + 1) << 8; ++k)
                        if (hseen[k] > 0) { if (((i + 1) * stream_per_xcc) / 32 > (i * stream_per_xcc) / 32) hr[k] = 0; ++i; }
                }
                HIP_CHECK(hipMemcpy(role_n.p, hr.data(), 2048, hipMemcpyHostToDevice));
                const double t = time_graph([&]() {
                    for (int l = 0; l < LAYERS; ++l) {
                        if (PF == 1) fused_kernel<1><<<WGS, 512, LDS, s>>>(xa(l), g_none, role_n.as<unsigned char>(), 0, nullptr);
                        else fused_kernel<2><<<WGS, 512, LDS, s>>>(xa(l), g_none, role_n.as<unsigned char>(), 0, nullptr);
                    }
                });
                printf("prefetch %d  streaming CUs %3d: %7.1f us  %.2f TB/s\n", PF, stream_per_xcc * 8, t, (double)A * 2 * T * D * 2 / 1e6 / t);
            }
        return 0;
    }
    printf("half batch: %d audios x %d beams = %d rows; %d persistent workgroups, %d key part(s) per (audio, head), %d layers per graph\n", A, G, R, WGS, PARTS, LAYERS);
    for (int PF = 1; PF <= 2; ++PF) {
        auto launch = [&](const XaP& x, const GmP& g, const unsigned char* role, int stop, Stamps* st) {
            if (PF == 1) fused_kernel<1><<<WGS, 512, LDS, s>>>(x, g, role, stop, st);
            else fused_kernel<2><<<WGS, 512, LDS, s>>>(x, g, role, stop, st);
        };
        const double t_xa_all = time_graph([&]() { for (int l = 0; l < LAYERS; ++l) launch(xa(l), g_none, R0, 0, nullptr); });
        const double t_xa_sub = time_graph([&]() { for (int l = 0; l < LAYERS; ++l) launch(xa(l), g_none, RS, 0, nullptr); });
        const double t_gemm_all = time_graph([&]() { for (int l = 0; l < LAYERS; ++l) { reset(); for (int si = 0; si < 6; ++si) launch(x_none, gm(si), R1, 0, nullptr); } });
        const double t_gemm_sub = time_graph([&]() { for (int l = 0; l < LAYERS; ++l) { reset(); for (int si = 0; si < 6; ++si) launch(x_none, gm(si), RS, 0, nullptr); } });
        double t_each_all[6], t_each_sub[6];
        for (int si = 0; si < 6; ++si) {
            t_each_all[si] = time_graph([&]() { for (int l = 0; l < LAYERS; ++l) { reset(); launch(x_none, gm(si), R1, 0, nullptr); } });
            t_each_sub[si] = time_graph([&]() { for (int l = 0; l < LAYERS; ++l) { reset(); launch(x_none, gm(si), RS, 0, nullptr); } });
        }
        const double t_reset = time_graph([&]() { for (int l = 0; l < LAYERS; ++l) reset(); });
        const double t_seq = time_graph([&]() {
            for (int l = 0; l < LAYERS; ++l) { reset(); for (int si = 0; si < 6; ++si) launch(x_none, gm(si), R1, 0, nullptr); launch(xa(l), g_none, R0, 0, nullptr); }
        });
        const double t_fused = time_graph([&]() {
            for (int l = 0; l < LAYERS; ++l) {
                reset();
                for (int si = 0; si < 6; ++si) launch(xa(l), gm(si), RS, 1, nullptr);
                launch(xa(l), g_none, R0, 0, nullptr);  // drain on every CU: the units the fused launches left
            }
        });
        HIP_CHECK(hipMemsetAsync(ctr.p, 0, 4096, s));
        HIP_CHECK(hipStreamSynchronize(s));
        {
            std::vector<Stamps> init(8);
            for (auto& e : init) { e.t[0] = e.t[2] = ~0ull; e.t[1] = e.t[3] = 0; }
            HIP_CHECK(hipMemcpy(stamps.p, init.data(), sizeof(Stamps) * 8, hipMemcpyHostToDevice));
        }
        int taken[7];
        for (int si = 0; si < 6; ++si) {
            launch(xa(0), gm(si), RS, 1, stamps.as<Stamps>() + si);
            HIP_CHECK(hipStreamSynchronize(s));
            HIP_CHECK(hipMemcpy(&taken[si], ctr_i + 64, 4, hipMemcpyDeviceToHost));
        }
        std::vector<Stamps> hs(8);
        HIP_CHECK(hipMemcpy(hs.data(), stamps.p, sizeof(Stamps) * 8, hipMemcpyDeviceToHost));
        const double gb = (double)A * 2 * T * D * 2 / 1e6;
        printf("\nprefetch depth %d (blocks of 32 keys per wave in flight)   [a counter-reset node alone: %.1f us]\n", PF, t_reset);
        printf("  cross-attention alone, every CU          %7.1f us  (%.2f TB/s)\n", t_xa_all, gb / t_xa_all);
        printf("  cross-attention alone, streaming CUs     %7.1f us  (%.2f TB/s on %d CUs)\n", t_xa_sub, gb / t_xa_sub, n_cu - n_gemm_cu);
        printf("  GEMM chain alone, every CU               %7.1f us  [", t_gemm_all);
        for (int si = 0; si < 6; ++si) printf("%s%.1f", si ? " " : "", t_each_all[si]);
        printf("]\n  GEMM chain alone, multiplying CUs        %7.1f us  [", t_gemm_sub);
        for (int si = 0; si < 6; ++si) printf("%s%.1f", si ? " " : "", t_each_sub[si]);
        printf("]\n  one after the other (every CU each)      %7.1f us\n", t_seq);
        printf("  fused launches + drain                   %7.1f us  (%.0f %% of sequential)\n", t_fused, 100.0 * t_fused / t_seq);
        printf("  per fused launch (eager): units taken so far / GEMM span / attention span (us)\n");
        for (int si = 0; si < 6; ++si)
            printf("    %s  tiles %4d  units %5d   gemm %6.1f   attention %6.1f\n", shapes[si].name, gm(si).n_blocks, taken[si],
                   (double)(hs[si].t[1] - hs[si].t[0]) / 100.0, (double)(hs[si].t[3] - hs[si].t[2]) / 100.0);
    }
    return 0;
}
