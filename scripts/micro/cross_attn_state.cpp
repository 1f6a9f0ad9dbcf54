// Does the decode cross-attention's rate depend on WHERE its K/V landed (per allocation) or on the state the device is in (per process /
// per moment)?  In round 3 one box gave 349 us per launch in one process and 374 us in the next (307 windows).  Each trial allocates fresh
// K/V buffers for 8 layers (with a differently sized dummy allocation in front, so the physical placement changes), replays a graph of the
// 8 launches 6 times and prints every replay's time per launch.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o scripts/micro/cross_attn_state scripts/micro/cross_attn_state.cpp && scripts/micro/cross_attn_state [audios 307] [trials 5]
#include "../../whisper_rs_amd/csrc/common.cpp"
#include "../../whisper_rs_amd/csrc/attention.hip"
#include <chrono>
using namespace wrs;
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const int A = argc > 1 ? atoi(argv[1]) : 307, trials = argc > 2 ? atoi(argv[2]) : 5, G = 5, L = 8, D = 1280, H = 20, T = 1500;
    hipStream_t s; HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const size_t per_layer = (size_t)A * H * packed_kv_halfs_per_head(T);
    const int R = A * G, m_pad = (R + 15) / 16 * 16;
    DevBuf slab, bias, out;
    slab.alloc((size_t)4 * m_pad * D * 4); bias.alloc(D * 4); out.alloc((size_t)m_pad * D * 2);
    HIP_CHECK(hipMemset(slab.p, 0, slab.bytes)); HIP_CHECK(hipMemset(bias.p, 0, bias.bytes));
    for (int trial = 0; trial < trials; ++trial) {
        DevBuf dummy, kp, vp;
        dummy.alloc(((size_t)1 + 3 * trial) * (257ull << 20) + 4096 * trial);  // shifts where the next allocations land
        kp.alloc(per_layer * L * 2); vp.alloc(per_layer * L * 2);
        HIP_CHECK(hipMemset(kp.p, 0x11, kp.bytes)); HIP_CHECK(hipMemset(vp.p, 0x22, vp.bytes)); HIP_CHECK(hipDeviceSynchronize());
        printf("trial %d (K at %p, V at %p):", trial, kp.p, vp.p);
        hipGraph_t g; hipGraphExec_t ge;
        HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int l = 0; l < L; ++l)
            launch_attn_cross(slab.as<float>(), 4, m_pad, bias.as<float>(), 0.35f, kp.as<half_t>() + l * per_layer, vp.as<half_t>() + l * per_layer,
                              A, G, T, H, out.as<half_t>(), s);
        HIP_CHECK(hipStreamEndCapture(s, &g));
        HIP_CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 6; ++rep) {
            const double t0 = now();
            for (int i = 0; i < 4; ++i) HIP_CHECK(hipGraphLaunch(ge, s));
            HIP_CHECK(hipStreamSynchronize(s));
            printf(" %.1f", (now() - t0) / (4 * L));
        }
        printf(" us per launch (%.0f MB each)\n", (double)A * 2 * T * D * 2 / 1e6);
        HIP_CHECK(hipGraphExecDestroy(ge)); HIP_CHECK(hipGraphDestroy(g));
    }
    return 0;
}
