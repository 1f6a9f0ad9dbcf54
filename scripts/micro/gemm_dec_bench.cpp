// Standalone timing of the decode GEMM kernels on large-v2 shapes: a captured graph of 32 dependent launches
// (one per "layer", each with its own cold weight matrix), replayed; reports us per launch.
#include "../../whisper_rs_amd/csrc/common.cpp"
#include "../../whisper_rs_amd/csrc/gemm.hip"
#include <chrono>
using namespace wrs;
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 160, L = 32, D = 1280;
    hipStream_t s; HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    struct Shape { const char* name; int N, K; } shapes[] = {{"qkv", 3 * D, D}, {"out/cq", D, D}, {"mlp1", 4 * D, D}, {"mlp2", D, 4 * D}};
    const int m_pad = skinny_m_pad(M);
    DevBuf a, slab, flush;
    a.alloc((size_t)m_pad * 4 * D * 2);
    slab.alloc((size_t)40 * m_pad * 4 * D * 4);
    flush.alloc((size_t)512 << 20);
    for (auto& sh : shapes) {
        const int64_t n_pad = round_up(sh.N, 128);
        std::vector<DevBuf> w(L);
        for (auto& b : w) b.alloc((size_t)n_pad * sh.K * 2);
        int kps, mtb;
        const int ks = gemm_dec_plan(M, sh.N, sh.K, &kps, &mtb);
        hipGraph_t g; hipGraphExec_t ge;
        HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int l = 0; l < L; ++l) launch_gemm_dec(a.as<half_t>(), w[l].as<half_t>(), M, sh.N, n_pad, sh.K, slab.as<float>(), s);
        HIP_CHECK(hipStreamEndCapture(s, &g));
        HIP_CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        double best = 1e30;
        for (int rep = 0; rep < 5; ++rep) {
            HIP_CHECK(hipMemsetAsync(flush.p, rep, flush.bytes, s));  // evict weights from L2 / Infinity Cache
            HIP_CHECK(hipStreamSynchronize(s));
            const double t0 = now();
            HIP_CHECK(hipGraphLaunch(ge, s));
            HIP_CHECK(hipStreamSynchronize(s));
            best = std::min(best, now() - t0);
        }
        const double us = best / L;
        printf("%-7s M=%d N=%d K=%d ksplit=%d kps=%d mt_block=%d grid=%ldx%dx%d : %.2f us/launch, weights %.0f GB/s\n", sh.name, M, sh.N, sh.K, ks, kps, mtb,
               (long)(n_pad / 64), ks, (m_pad / 16) / mtb, us, (double)sh.N * sh.K * 2 / us / 1e3);
        HIP_CHECK(hipGraphExecDestroy(ge)); HIP_CHECK(hipGraphDestroy(g));
    }
    return 0;
}
