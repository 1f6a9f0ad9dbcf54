// Physical placement and the decode cross-attention's rate: per-layer K / V buffers from hipMalloc against the same buffers from
// hipExtMallocWithFlags(hipDeviceMallocContiguous) (physically contiguous), every layer timed on its own.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o scripts/micro/cross_attn_contig scripts/micro/cross_attn_contig.cpp && scripts/micro/cross_attn_contig [audios 307] [layers 12]
#include "../../whisper_rs_amd/csrc/common.cpp"
#include "../../whisper_rs_amd/csrc/attention.hip"
using namespace wrs;
int main(int argc, char** argv) {
    const int A = argc > 1 ? atoi(argv[1]) : 307, L = argc > 2 ? atoi(argv[2]) : 12, G = 5, D = 1280, H = 20, T = 1500;
    hipStream_t s; HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const size_t per_layer = (size_t)A * H * packed_kv_halfs_per_head(T);
    const int R = A * G, m_pad = (R + 15) / 16 * 16;
    DevBuf slab, bias, out;
    slab.alloc((size_t)4 * m_pad * D * 4); bias.alloc(D * 4); out.alloc((size_t)m_pad * D * 2);
    hipEvent_t e0, e1; HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
    size_t gran_min = 0, gran_rec = 0;
    hipMemAllocationProp prop{}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    (void)hipMemGetAllocationGranularity(&gran_min, &prop, hipMemAllocationGranularityMinimum);
    (void)hipMemGetAllocationGranularity(&gran_rec, &prop, hipMemAllocationGranularityRecommended);
    printf("allocation granularity: minimum %zu, recommended %zu\n", gran_min, gran_rec);
    for (int mode = 0; mode < 4; ++mode) {  // 0, 2: hipMalloc per layer; 1, 3: contiguous per layer
        std::vector<void*> kb(L), vb(L);
        bool ok = true;
        for (int l = 0; l < L && ok; ++l) {
            hipError_t e1_, e2_;
            if (mode & 1) {
                e1_ = hipExtMallocWithFlags(&kb[l], per_layer * 2, hipDeviceMallocContiguous);
                e2_ = hipExtMallocWithFlags(&vb[l], per_layer * 2, hipDeviceMallocContiguous);
            } else {
                e1_ = hipMalloc(&kb[l], per_layer * 2);
                e2_ = hipMalloc(&vb[l], per_layer * 2);
            }
            if (e1_ != hipSuccess || e2_ != hipSuccess) { printf("mode %d: allocation of layer %d failed (%s)\n", mode, l, hipGetErrorString(e1_ != hipSuccess ? e1_ : e2_)); ok = false; (void)hipGetLastError(); break; }
            HIP_CHECK(hipMemset(kb[l], 0x11, per_layer * 2)); HIP_CHECK(hipMemset(vb[l], 0x22, per_layer * 2));
        }
        HIP_CHECK(hipDeviceSynchronize());
        if (ok) {
            printf("%s:", (mode & 1) ? "contiguous" : "hipMalloc ");
            double sum = 0;
            for (int l = 0; l < L; ++l) {
                for (int rep = 0; rep < 2; ++rep) {
                    HIP_CHECK(hipEventRecord(e0, s));
                    for (int i = 0; i < 6; ++i)
                        launch_attn_cross(slab.as<float>(), 4, m_pad, bias.as<float>(), 0.35f, (const half_t*)kb[l], (const half_t*)vb[l], A, G, T, H, out.as<half_t>(), s);
                    HIP_CHECK(hipEventRecord(e1, s));
                    HIP_CHECK(hipEventSynchronize(e1));
                }
                float ms; HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
                printf(" %.0f", ms * 1e3 / 6);
                sum += ms * 1e3 / 6;
            }
            printf("   mean %.1f\n", sum / L);
        }
        for (int l = 0; l < L; ++l) { if (kb[l]) (void)hipFree(kb[l]); if (vb[l]) (void)hipFree(vb[l]); }
    }
    return 0;
}
