// Decode-step GEMMs with COLD weights: in the decode step every GEMM reads weights that were last touched 17 ms (one step)
// and ~90 GB of other traffic ago; the shape micro-benchmark (gemm_f16_bench dec) re-uses one weight buffer (hot in L2 / MALL).
//   mode hot   : one weight buffer
//   mode cold  : 40 weight buffers in rotation, 2 GB of unrelated reads between launches
//   mode touch : as cold, but a small kernel reads the NEXT launch's weights first (what a prefetch from the preceding
//                finish kernel would do), then the unrelated reads are skipped for that buffer
#include "../../whisper_rs_amd/csrc/common.cpp"
#include "../../whisper_rs_amd/csrc/gemm.hip"
#include <vector>
using namespace wrs;
__global__ void touch(const float4* __restrict__ p, size_t n, float* out) {
    float acc = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i].x;
    if (acc == 12345.678f) out[0] = acc;
}
int main() {
    const int D = 1280, M = 1280;
    hipStream_t s; HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    struct Shape { const char* name; int N, K; bool f32out, gelu; int ksplit; };
    const Shape shapes[] = {{"d_qkv", 3 * D, D, true, false, 1}, {"d_mlp1", 4 * D, D, false, true, 1}, {"d_DxD", D, D, true, false, 2}, {"d_mlp2", D, 4 * D, true, false, 2}};
    DevBuf junk, sink; junk.alloc((size_t)2 << 30); sink.alloc(64);
    HIP_CHECK(hipMemset(junk.p, 0, junk.bytes));
    hipEvent_t e0, e1; HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
    const int NB = 40;
    for (auto& sh : shapes) {
        DevBuf a, c, bias;
        std::vector<DevBuf> w(NB);
        const int64_t mp = round_up(M, 128);
        a.alloc((size_t)mp * sh.K * 2 + 4096); HIP_CHECK(hipMemset(a.p, 0, a.bytes));
        const size_t wbytes = (size_t)round_up(sh.N, 128) * sh.K * 2;
        for (auto& b : w) { b.alloc(wbytes); HIP_CHECK(hipMemset(b.p, 0, wbytes)); }
        c.alloc((size_t)mp * sh.N * 4 * sh.ksplit); bias.alloc(sh.N * 4); HIP_CHECK(hipMemset(bias.p, 0, bias.bytes));
        GemmA A{a.as<half_t>(), sh.K}; A.packed = 1; A.packed_rows = mp;
        GemmEpi e; e.c = c.p; e.c_f32 = sh.f32out; e.ldc = sh.N;
        if (sh.ksplit > 1) { e.ksplit = sh.ksplit; e.slab_stride = mp * sh.N; }
        else { e.bias = bias.as<float>(); e.act = sh.gelu; if (!sh.f32out) e.c_packed = 1; }
        for (int mode = 0; mode < 3; ++mode) {
            double tot = 0; int cnt = 0;
            for (int it = 0; it < 24; ++it) {
                const int wi = mode == 0 ? 0 : it % NB;
                if (mode >= 1) touch<<<2048, 256, 0, s>>>(junk.as<float4>(), junk.bytes / 16, sink.as<float>());  // flush caches
                if (mode == 2) touch<<<256, 256, 0, s>>>(w[wi].as<float4>(), wbytes / 16, sink.as<float>());     // prefetch
                HIP_CHECK(hipEventRecord(e0, s));
                launch_gemm_f16(A, w[wi].as<half_t>(), M, sh.N, sh.K, e, s);
                HIP_CHECK(hipEventRecord(e1, s));
                HIP_CHECK(hipEventSynchronize(e1));
                float ms; HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (it >= 4) { tot += ms; ++cnt; }
            }
            printf("%-7s %s: %.1f us\n", sh.name, mode == 0 ? "hot  " : mode == 1 ? "cold " : "touch", tot / cnt * 1e3);
        }
    }
    return 0;
}
