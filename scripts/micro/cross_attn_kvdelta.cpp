// The decode cross-attention streams a K block and the V block of the same keys at the same time; both buffers come 2-MiB aligned from
// hipMalloc, so the two streams of every wave share their low 21 address bits.  Does the rate depend on that?  The V buffer is shifted by
// `pad` bytes (same allocation), every layer timed on its own.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o scripts/micro/cross_attn_kvdelta scripts/micro/cross_attn_kvdelta.cpp && scripts/micro/cross_attn_kvdelta [audios 307] [layers 12]
#include "../../whisper_rs_amd/csrc/common.cpp"
#include "../../whisper_rs_amd/csrc/attention.hip"
using namespace wrs;
int main(int argc, char** argv) {
    const int A = argc > 1 ? atoi(argv[1]) : 307, L = argc > 2 ? atoi(argv[2]) : 12, G = 5, D = 1280, H = 20, T = 1500;
    hipStream_t s; HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const size_t per_layer = (size_t)A * H * packed_kv_halfs_per_head(T);
    const int R = A * G, m_pad = (R + 15) / 16 * 16;
    DevBuf slab, bias, out, kp, vp;
    slab.alloc((size_t)4 * m_pad * D * 4); bias.alloc(D * 4); out.alloc((size_t)m_pad * D * 2);
    kp.alloc(per_layer * L * 2); vp.alloc(per_layer * L * 2 + (8u << 20));
    HIP_CHECK(hipMemset(kp.p, 0x11, kp.bytes)); HIP_CHECK(hipMemset(vp.p, 0x22, vp.bytes)); HIP_CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1; HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
    printf("K at %p, V at %p (+pad); us per launch by layer\n", kp.p, vp.p);
    const size_t pads[] = {0, 256, 1024, 4096, 4096 + 256, 16384, 65536, 65536 + 4096, 262144, 1 << 20, (1 << 20) + 4096 + 256, 2 << 20, 0};
    for (size_t pad : pads) {
        const half_t* vbase = reinterpret_cast<const half_t*>(reinterpret_cast<const char*>(vp.p) + pad);
        printf("pad %8zu:", pad);
        double sum = 0;
        for (int l = 0; l < L; ++l) {
            for (int rep = 0; rep < 2; ++rep) {
                HIP_CHECK(hipEventRecord(e0, s));
                for (int i = 0; i < 6; ++i)
                    launch_attn_cross(slab.as<float>(), 4, m_pad, bias.as<float>(), 0.35f, kp.as<half_t>() + l * per_layer, vbase + l * per_layer, A, G, T, H, out.as<half_t>(), s);
                HIP_CHECK(hipEventRecord(e1, s));
                HIP_CHECK(hipEventSynchronize(e1));
            }
            float ms; HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf(" %.0f", ms * 1e3 / 6);
            sum += ms * 1e3 / 6;
        }
        printf("   mean %.1f\n", sum / L);
    }
    return 0;
}
