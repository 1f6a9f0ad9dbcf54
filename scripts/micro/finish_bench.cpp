// finish_residual_ln (split-K slabs + bias + residual + LayerNorm) against LayerNorm alone on [rows, 1280] — what a decode-step
// N = D GEMM that adds into the residual stream itself (no slab) would leave for the kernel after it.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o scripts/micro/finish_bench scripts/micro/finish_bench.cpp && scripts/micro/finish_bench [rows 1535]
#include "../../whisper_rs_amd/csrc/common.cpp"
#include "../../whisper_rs_amd/csrc/elementwise.hip"
#include <chrono>
using namespace wrs;
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 1535, D = 1280, mp = (M + 159) / 160 * 160, rows_alloc = (mp + 127) / 128 * 128;
    hipStream_t s; HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    DevBuf x, slab, bias, g, b, xn;
    x.alloc((size_t)rows_alloc * D * 4); slab.alloc((size_t)2 * mp * D * 4); bias.alloc(D * 4); g.alloc(D * 4); b.alloc(D * 4); xn.alloc((size_t)rows_alloc * D * 2);
    HIP_CHECK(hipMemset(x.p, 0, x.bytes)); HIP_CHECK(hipMemset(slab.p, 0, slab.bytes)); HIP_CHECK(hipMemset(bias.p, 0, bias.bytes));
    HIP_CHECK(hipMemset(g.p, 0, g.bytes)); HIP_CHECK(hipMemset(b.p, 0, b.bytes));
    // a 64 MB write between launches evicts nothing useful but keeps the two kernels from running on an all-L2-resident working set
    auto time = [&](const char* name, auto fn) {
        double best = 1e30;
        for (int rep = 0; rep < 5; ++rep) {
            HIP_CHECK(hipStreamSynchronize(s));
            const double t0 = now();
            for (int i = 0; i < 100; ++i) fn();
            HIP_CHECK(hipStreamSynchronize(s));
            best = std::min(best, (now() - t0) / 100);
        }
        printf("%-40s rows %d: %.1f us\n", name, M, best);
    };
    time("finish_residual_ln (2 slabs)", [&] { launch_finish_residual_ln(x.as<float>(), slab.as<float>(), 2, mp, bias.as<float>(), g.as<float>(), b.as<float>(), M, D, xn.as<half_t>(), s); });
    time("finish_residual_ln (1 slab)", [&] { launch_finish_residual_ln(x.as<float>(), slab.as<float>(), 1, mp, bias.as<float>(), g.as<float>(), b.as<float>(), M, D, xn.as<half_t>(), s); });
    time("layernorm_packed", [&] { launch_layernorm_packed(x.as<float>(), g.as<float>(), b.as<float>(), M, D, xn.as<half_t>(), s); });
    return 0;
}
