// Why does the side stream's encoder GEMM run 2.7x slower next to the decode step's cross-attention even on CUs of its own
// (profiles/r03_overlap_trace_confined7.txt)?  Hypothesis: the 6.4 TB/s K/V stream turns the L2s (4 MiB per XCD: every 5 us)
// and the Infinity Cache (256 MiB: every 40 us) over faster than the GEMM re-reads its operand panels, so the panels come
// from HBM, which the stream has saturated.  Test: the real kernels side by side —
//   stream A: the decode cross-attention (256 audios, 8 layers' K/V in rotation), back to back;
//   stream B: the CU-confined 256x256 GEMM (encoder shapes, M = 384 000) on `side` CUs per XCD;
// with the K/V buffers in ordinary device memory and in UNCACHED device memory (hipDeviceMallocUncached: MTYPE_UC, no L2
// allocation), and with / without the non-temporal hint on the K/V loads.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DXA_NT=0] -o scripts/micro/share_l2 scripts/micro/share_l2.cpp && scripts/micro/share_l2 [side CUs per XCD 7]
#include "../../whisper_rs_amd/csrc/common.cpp"
#include "../../whisper_rs_amd/csrc/gemm.hip"
#include "cu_confine.h"
#include "../../whisper_rs_amd/csrc/attention.hip"
#include <chrono>
using namespace wrs;
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const int side = argc > 1 ? atoi(argv[1]) : 7;
    const int A = 256, G = 5, L = 8, D = 1280, H = 20, T = 1500;
    const int64_t M = 384000;
    hipStream_t sa, sb;
    int lo = 0, hi = 0;
    HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    HIP_CHECK(hipStreamCreateWithPriority(&sa, hipStreamNonBlocking, hi));
    HIP_CHECK(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, lo));
    const size_t per_layer = (size_t)A * H * packed_kv_halfs_per_head(T);
    const size_t kv_bytes = per_layer * L * 2;
    void* kv[2][2];  // [0 = cached, 1 = uncached][k, v]
    for (int i = 0; i < 2; ++i) HIP_CHECK(hipMalloc(&kv[0][i], kv_bytes));
    for (int i = 0; i < 2; ++i) HIP_CHECK(hipExtMallocWithFlags(&kv[1][i], kv_bytes, hipDeviceMallocUncached));
    for (int u = 0; u < 2; ++u) { HIP_CHECK(hipMemset(kv[u][0], 0x11, kv_bytes)); HIP_CHECK(hipMemset(kv[u][1], 0x22, kv_bytes)); }
    const int R = A * G, m_pad = (R + 127) / 128 * 128;
    DevBuf q16, xo, act, w, cbuf, roles, ring;
    q16.alloc((size_t)m_pad * D * 2); xo.alloc((size_t)m_pad * D * 2);
    HIP_CHECK(hipMemset(q16.p, 0x2c, q16.bytes));
    act.alloc((size_t)M * 4 * D * 2); w.alloc((size_t)4 * D * 4 * D * 2); cbuf.alloc((size_t)M * 4 * D * 2);
    HIP_CHECK(hipMemset(act.p, 0x2c, act.bytes)); HIP_CHECK(hipMemset(w.p, 0x1c, w.bytes));
    roles.alloc(CU_ROLE_KEYS); ring.alloc((size_t)64 * CU_CTR_INTS * 4);
    int n_marked = 0;
    const int n_cus = build_cu_roles(side, roles.as<unsigned char>(), &n_marked, sb);
    CuConfine cc; cc.roles = roles.as<unsigned char>(); cc.ctr = ring.as<int>(); cc.n_slots = 64; cc.n_cus = n_cus;
    printf("%d CUs, %d for the GEMM stream\n", n_cus, n_marked);
    HIP_CHECK(hipDeviceSynchronize());

    struct Shape { const char* name; int64_t N, K; };
    const Shape shapes[3] = {{"qkv  N=3D K=D ", 3 * D, D}, {"mlp1 N=4D K=D ", 4 * D, D}, {"mlp2 N=D  K=4D", D, 4 * D}};
    auto gemm = [&](const Shape& sh, bool confined) {
        GemmA Aop{act.as<half_t>(), sh.K};
        GemmEpi e; e.c = cbuf.p; e.ldc = sh.N;
        if (confined) { cc.reset(sb); launch_gemm_f16_confined(Aop, w.as<half_t>(), M, sh.N, sh.K, e, cc, sb); }
        else launch_gemm_f16(Aop, w.as<half_t>(), M, sh.N, sh.K, e, sb);
    };
    auto attn = [&](int u, int l) {
        launch_attn_cross_q16(q16.as<half_t>(), reinterpret_cast<half_t*>(kv[u][0]) + l * per_layer, reinterpret_cast<half_t*>(kv[u][1]) + l * per_layer,
                              A, G, T, H, xo.as<half_t>(), sa);
    };
    hipEvent_t e0, e1, a0, a1;
    HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1)); HIP_CHECK(hipEventCreate(&a0)); HIP_CHECK(hipEventCreate(&a1));
    // warm up
    for (int u = 0; u < 2; ++u) for (int l = 0; l < L; ++l) attn(u, l);
    for (const Shape& sh : shapes) { gemm(sh, false); gemm(sh, true); }
    HIP_CHECK(hipDeviceSynchronize());
    for (int u = 0; u < 2; ++u) {
        HIP_CHECK(hipEventRecord(a0, sa));
        for (int rep = 0; rep < 4; ++rep) for (int l = 0; l < L; ++l) attn(u, l);
        HIP_CHECK(hipEventRecord(a1, sa)); HIP_CHECK(hipEventSynchronize(a1));
        float ms; HIP_CHECK(hipEventElapsedTime(&ms, a0, a1));
        printf("cross-attention alone, K/V %s: %.1f us per launch (%.2f TB/s)\n", u ? "UNCACHED" : "cached  ", ms * 1e3 / (4 * L), (double)A * 2 * T * D * 2 / (ms * 1e3 / (4 * L)) / 1e6);
    }
    for (const Shape& sh : shapes) {
        for (int confined = 0; confined < 2; ++confined) {
            HIP_CHECK(hipEventRecord(e0, sb)); gemm(sh, confined); HIP_CHECK(hipEventRecord(e1, sb)); HIP_CHECK(hipEventSynchronize(e1));
            float alone; HIP_CHECK(hipEventElapsedTime(&alone, e0, e1));
            printf("%s %s: alone %7.2f ms (%4.0f TFLOP/s)", sh.name, confined ? "confined" : "classic ", alone, 2.0 * M * sh.N * sh.K / alone / 1e9);
            for (int u = 0; u < 2; ++u) {
                // the attention stream runs for the whole GEMM and a little longer; its launches are timed as a block
                const int n_attn = (int)(alone * 1e3 * 4.0 / 300.0) + 16;
                HIP_CHECK(hipEventRecord(a0, sa));
                for (int i = 0; i < n_attn; ++i) attn(u, i % L);
                HIP_CHECK(hipEventRecord(a1, sa));
                HIP_CHECK(hipEventRecord(e0, sb)); gemm(sh, confined); HIP_CHECK(hipEventRecord(e1, sb));
                HIP_CHECK(hipEventSynchronize(e1));
                float both; HIP_CHECK(hipEventElapsedTime(&both, e0, e1));
                HIP_CHECK(hipEventSynchronize(a1));
                float at; HIP_CHECK(hipEventElapsedTime(&at, a0, a1));
                printf("   | beside the attention (K/V %s): GEMM %7.2f ms, %d attention launches in %7.2f ms", u ? "UNCACHED" : "cached", both, n_attn, at);
            }
            printf("\n");
        }
    }
    return 0;
}
