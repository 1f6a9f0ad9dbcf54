// Standalone timing of the persistent ring GEMM (gemm_pring.hip) against the round-1 slab path (gemm.hip) on the
// decode-step shapes of large-v2 at `rows` rows (default 1280 = 256 windows x beam 5).
//   hipcc -O3 --offload-arch=gfx950 -o scripts/micro/gemm_pring_bench scripts/micro/gemm_pring_bench.cpp && scripts/micro/gemm_pring_bench [rows]
#include "../../whisper_rs_amd/csrc/common.cpp"
#include "../../whisper_rs_amd/csrc/gemm.hip"
#include "../../whisper_rs_amd/csrc/gemm_pring.hip"
#include <chrono>
#include <random>
using namespace wrs;
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const int D = 1280;
    const int64_t M = argc > 1 ? atoi(argv[1]) : 1280;
    hipStream_t s; HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    std::mt19937 rng(1);
    std::vector<_Float16> h((size_t)4 << 20);
    for (auto& x : h) x = (_Float16)(((int)(rng() % 2001) - 1000) / 1000.0f);
    auto fill = [&](DevBuf& b) { for (size_t off = 0; off < b.bytes; off += h.size() * 2) HIP_CHECK(hipMemcpy((char*)b.p + off, h.data(), std::min(h.size() * 2, b.bytes - off), hipMemcpyHostToDevice)); };
    const int64_t mp = round_up(M, 128);
    DevBuf a, a4, w_dd, w_qkv, w_m1, w_m2, x, q16, hbuf, kc, vc, bias, sp, slab;
    a.alloc((size_t)mp * D * 2); fill(a);
    a4.alloc((size_t)mp * 4 * D * 2); fill(a4);
    w_dd.alloc((size_t)D * D * 2); fill(w_dd);
    w_qkv.alloc((size_t)3 * D * D * 2); fill(w_qkv);
    w_m1.alloc((size_t)4 * D * D * 2); fill(w_m1);
    w_m2.alloc((size_t)4 * D * D * 2); fill(w_m2);
    x.alloc((size_t)mp * D * 4); q16.alloc((size_t)mp * D * 2); hbuf.alloc((size_t)mp * 4 * D * 2);
    const int n_ctx = 448;
    kc.alloc((size_t)M * n_ctx * D * 2); vc.alloc((size_t)M * n_ctx * D * 2);
    bias.alloc(4 * D * 4); sp.alloc(sizeof(StepParams)); slab.alloc((size_t)4 * mp * 4 * D * 4);
    auto time = [&](const char* name, double flops, auto&& fn) {
        double best = 1e30;
        for (int rep = 0; rep < 4; ++rep) {
            HIP_CHECK(hipStreamSynchronize(s));
            const double t0 = now();
            for (int i = 0; i < 50; ++i) fn();
            HIP_CHECK(hipStreamSynchronize(s));
            best = std::min(best, (now() - t0) / 50);
        }
        printf("%-34s M=%ld: %7.1f us  %6.0f TFLOP/s\n", name, (long)M, best, flops / best / 1e6);
    };
    const double fdd = 2.0 * M * D * D;
    for (int bm : {64, 128}) {
        PringEpi e; e.mode = PRING_RESID; e.bias = bias.as<float>(); e.x = x.as<float>(); e.ldc = D;
        char nm[64]; snprintf(nm, sizeof nm, "pring RESID N=D K=D bm=%d", bm);
        time(nm, fdd, [&] { launch_gemm_pring(a.as<half_t>(), mp, w_dd.as<half_t>(), M, D, D, bm, e, s); });
        snprintf(nm, sizeof nm, "pring RESID N=D K=4D bm=%d", bm);
        time(nm, 4 * fdd, [&] { launch_gemm_pring(a4.as<half_t>(), mp, w_m2.as<half_t>(), M, D, 4 * D, bm, e, s); });
        PringEpi f; f.mode = PRING_F16; f.bias = bias.as<float>(); f.scale = 0.35f; f.scale_n_end = D; f.c16 = q16.as<half_t>(); f.ldc = D;
        snprintf(nm, sizeof nm, "pring F16 N=D K=D bm=%d", bm);
        time(nm, fdd, [&] { launch_gemm_pring(a.as<half_t>(), mp, w_dd.as<half_t>(), M, D, D, bm, f, s); });
        PringEpi q; q.mode = PRING_QKV; q.bias = bias.as<float>(); q.scale = 0.35f; q.scale_n_end = 2 * D; q.c16 = q16.as<half_t>(); q.ldc = D;
        q.kcache = kc.as<half_t>(); q.vcache = vc.as<half_t>(); q.d_model = D; q.cache_row_stride = (int64_t)n_ctx * D; q.phys_mul = 1; q.sp = sp.as<StepParams>();
        snprintf(nm, sizeof nm, "pring QKV N=3D bm=%d", bm);
        time(nm, 3 * fdd, [&] { launch_gemm_pring(a.as<half_t>(), mp, w_qkv.as<half_t>(), M, 3 * D, D, bm, q, s); });
        PringEpi g; g.mode = PRING_GELU_PACKED; g.bias = bias.as<float>(); g.c16 = hbuf.as<half_t>(); g.ldc = 4 * D;
        snprintf(nm, sizeof nm, "pring GELU N=4D bm=%d", bm);
        time(nm, 4 * fdd, [&] { launch_gemm_pring(a.as<half_t>(), mp, w_m1.as<half_t>(), M, 4 * D, D, bm, g, s); });
    }
    // round-1 path for comparison: slab GEMMs as model.cpp picked them
    GemmA A{a.as<half_t>(), D}; A.packed = 1; A.packed_rows = mp;
    GemmA A4{a4.as<half_t>(), 4 * D}; A4.packed = 1; A4.packed_rows = mp;
    {
        GemmEpi e; e.c = slab.p; e.c_f32 = 1; e.ldc = D; e.ksplit = 2; e.slab_stride = mp * D;
        time("slab ring N=D K=D ksplit=2", fdd, [&] { launch_gemm_f16(A, w_dd.as<half_t>(), M, D, D, e, s); });
        time("slab ring N=D K=4D ksplit=2", 4 * fdd, [&] { launch_gemm_f16(A4, w_m2.as<half_t>(), M, D, 4 * D, e, s); });
        GemmEpi q; q.c = slab.p; q.c_f32 = 1; q.ldc = 3 * D;
        time("slab tile128 N=3D", 3 * fdd, [&] { launch_gemm_f16(A, w_qkv.as<half_t>(), M, 3 * D, D, q, s); });
        GemmEpi g; g.bias = bias.as<float>(); g.act = 1; g.c = hbuf.p; g.c_f32 = 0; g.ldc = 4 * D; g.c_packed = 1;
        time("tile128 GELU packed N=4D", 4 * fdd, [&] { launch_gemm_f16(A, w_m1.as<half_t>(), M, 4 * D, D, g, s); });
    }
    return 0;
}
