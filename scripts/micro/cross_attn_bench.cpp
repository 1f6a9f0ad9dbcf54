// Standalone timing of the decode cross-attention kernel: graph of 32 launches (one per layer, own K/V), large-v2.
#include "../../whisper_rs_amd/csrc/common.cpp"
#include "../../whisper_rs_amd/csrc/attention.hip"
#include <chrono>
using namespace wrs;
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const int A = argc > 1 ? atoi(argv[1]) : 64, G = 5, L = 32, D = 1280, H = 20, T = 1500;
    hipStream_t s; HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const size_t per_layer = (size_t)A * H * packed_kv_halfs_per_head(T);
    DevBuf kp, vp, slab, bias, out;
    kp.alloc(per_layer * L * 2); vp.alloc(per_layer * L * 2);
    const int R = A * G, m_pad = (R + 15) / 16 * 16;
    slab.alloc((size_t)4 * m_pad * D * 4); bias.alloc(D * 4); out.alloc((size_t)m_pad * D * 2);
    HIP_CHECK(hipMemset(kp.p, 0x11, kp.bytes)); HIP_CHECK(hipMemset(vp.p, 0x22, vp.bytes)); HIP_CHECK(hipDeviceSynchronize());
    hipGraph_t g; hipGraphExec_t ge;
    HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int l = 0; l < L; ++l)
        launch_attn_cross(slab.as<float>(), 4, m_pad, bias.as<float>(), 0.35f, kp.as<half_t>() + l * per_layer, vp.as<half_t>() + l * per_layer,
                          A, G, T, H, out.as<half_t>(), s);
    HIP_CHECK(hipStreamEndCapture(s, &g));
    HIP_CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    double best = 1e30;
    for (int rep = 0; rep < 5; ++rep) {
        const double t0 = now();
        HIP_CHECK(hipGraphLaunch(ge, s));
        HIP_CHECK(hipStreamSynchronize(s));
        best = std::min(best, now() - t0);
    }
    const double us = best / L, bytes = (double)A * 2 * T * D * 2;
    printf("cross-attn A=%d: %.2f us/launch, %.0f GB/s algorithmic (%.1f%% of 8 TB/s)\n", A, us, bytes / us / 1e3, bytes / us / 1e3 / 80.0);
    return 0;
}
