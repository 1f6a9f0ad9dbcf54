// Is the decode cross-attention's rate a property of the whole allocation or of where inside it a layer's K/V sits?
// One K and one V buffer of 32 layers each (2 x 37.7 GB at 307 audios = one of the step's two cross-K/V buffers); every layer's launch
// is timed on its own (8 back-to-back launches, HIP events).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o scripts/micro/cross_attn_slices scripts/micro/cross_attn_slices.cpp && scripts/micro/cross_attn_slices [audios 307] [trials 2]
#include "../../whisper_rs_amd/csrc/common.cpp"
#include "../../whisper_rs_amd/csrc/attention.hip"
using namespace wrs;
int main(int argc, char** argv) {
    const int A = argc > 1 ? atoi(argv[1]) : 307, trials = argc > 2 ? atoi(argv[2]) : 2, G = 5, L = 32, D = 1280, H = 20, T = 1500;
    const int order = argc > 3 ? atoi(argv[3]) : 0;  // 0: layers 0..31, 1: 31..0, 2: a fixed shuffle (is the fast part a place or a moment?)
    hipStream_t s; HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const size_t per_layer = (size_t)A * H * packed_kv_halfs_per_head(T);
    const int R = A * G, m_pad = (R + 15) / 16 * 16;
    DevBuf slab, bias, out;
    slab.alloc((size_t)4 * m_pad * D * 4); bias.alloc(D * 4); out.alloc((size_t)m_pad * D * 2);
    hipEvent_t e0, e1; HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
    for (int trial = 0; trial < trials; ++trial) {
        DevBuf dummy, kp, vp;
        dummy.alloc(((size_t)1 + 5 * trial) * (513ull << 20));
        kp.alloc(per_layer * L * 2); vp.alloc(per_layer * L * 2);
        HIP_CHECK(hipMemset(kp.p, 0x11, kp.bytes)); HIP_CHECK(hipMemset(vp.p, 0x22, vp.bytes)); HIP_CHECK(hipDeviceSynchronize());
        printf("trial %d (K at %p, V at %p), us per launch by layer:\n ", trial, kp.p, vp.p);
        double sum = 0;
        std::vector<double> us(L);
        for (int i_ = 0; i_ < L; ++i_) {
            const int l = order == 0 ? i_ : order == 1 ? L - 1 - i_ : (i_ * 13 + 5) % L;
            for (int rep = 0; rep < 2; ++rep) {  // first round warms up
                HIP_CHECK(hipEventRecord(e0, s));
                for (int i = 0; i < 8; ++i)
                    launch_attn_cross(slab.as<float>(), 4, m_pad, bias.as<float>(), 0.35f, kp.as<half_t>() + l * per_layer, vp.as<half_t>() + l * per_layer,
                                      A, G, T, H, out.as<half_t>(), s);
                HIP_CHECK(hipEventRecord(e1, s));
                HIP_CHECK(hipEventSynchronize(e1));
            }
            float ms; HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
            us[l] = ms * 1e3 / 8;
            sum += ms * 1e3 / 8;
        }
        for (int l = 0; l < L; ++l) printf(" %.0f", us[l]);
        printf("\n  mean %.1f us (timed in order %d; listed by layer)\n", sum / L, order);
    }
    return 0;
}
