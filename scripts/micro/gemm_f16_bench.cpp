// Standalone timing of gemm_f16_kernel on large-v2 shapes (random fp16 data).
//   gemm_f16_bench enc  [windows]   encoder shapes, row-major A, M = windows*1500
//   gemm_f16_bench dec  [rows]      decode-step shapes, packed A, fp32 slab output (split-K as model.cpp picks it)
// The ring depth is forced with WRS_GEMM_STAGES=2|3|4 (default: the library's own choice).
#include "../../whisper_rs_amd/csrc/common.cpp"
#include "../../whisper_rs_amd/csrc/gemm.hip"
#include <chrono>
#include <random>
#include <cstring>
using namespace wrs;
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const bool dec = argc > 1 && !strcmp(argv[1], "dec");
    const int D = 1280;
    const int64_t M = dec ? (argc > 2 ? atoi(argv[2]) : 1280) : (int64_t)(argc > 2 ? atoi(argv[2]) : 32) * 1500;
    hipStream_t s; HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    struct Shape { const char* name; int N, K; bool f32out, res, gelu; int ksplit; };
    std::vector<Shape> shapes;
    if (dec) {
        if (getenv("FIXED_COST")) {  // fixed cost of a launch: K = 64 .. 1280 at N = D (100 workgroups) and N = 4D (400)
            for (int k : {64, 128, 320, 640, 1280}) shapes.push_back({"d_k", D, k, true, false, false, 1});
            for (int k : {64, 128, 320, 640, 1280}) shapes.push_back({"d_k4", 4 * D, k, true, false, false, 1});
        }
        for (int ks : {1, 2, 4}) shapes.push_back({"d_DxD", D, D, true, false, false, ks});
        shapes.push_back({"d_DxDr", D, D, true, true, false, 1});       // bias + in-place residual (no slab): launcher's choice
        shapes.push_back({"d_qkv", 3 * D, D, true, false, false, 1});
        shapes.push_back({"d_qkvd", 3 * D, D, false, false, false, -1});  // the step's own form: fp16 query + key / value rows into the cache
        shapes.push_back({"d_mlp1", 4 * D, D, false, false, true, 1});
        for (int ks : {1, 2, 4, 8}) shapes.push_back({"d_mlp2", D, 4 * D, true, false, false, ks});
        shapes.push_back({"d_logit", 51968, D, true, false, false, 1});
    } else {
        shapes = {{"qkv", 3 * D, D, false, false, false, 1}, {"out", D, D, true, true, false, 1}, {"mlp1", 4 * D, D, false, false, true, 1},
                  {"mlp2", D, 4 * D, true, true, false, 1}, {"longK", D, 16 * D, false, false, false, 1}};
    }
    std::mt19937 rng(1);
    std::vector<_Float16> h((size_t)4 << 20);
    for (auto& x : h) x = (_Float16)(((int)(rng() % 2001) - 1000) / 1000.0f);
    auto fill = [&](DevBuf& b) { for (size_t off = 0; off < b.bytes; off += h.size() * 2) HIP_CHECK(hipMemcpy((char*)b.p + off, h.data(), std::min(h.size() * 2, b.bytes - off), hipMemcpyHostToDevice)); };
    const int reps = dec ? 50 : 4;
    for (auto& sh : shapes) {
        DevBuf a, w, c, bias;
        const int64_t mp = round_up(M, 128);
        a.alloc((size_t)mp * sh.K * 2 + 4096); fill(a);
        w.alloc((size_t)round_up(sh.N, 128) * sh.K * 2); fill(w);
        c.alloc((size_t)mp * sh.N * 4 * std::max(sh.ksplit, 1)); bias.alloc(sh.N * 4);
        GemmA A{a.as<half_t>(), sh.K};
        if (dec) { A.packed = 1; A.packed_rows = mp; }
        GemmEpi e; e.c = c.p; e.c_f32 = sh.f32out; e.ldc = sh.N;
        DevBuf kc, vc, spb;
        if (sh.ksplit == -1) {  // QKV-to-cache epilogue: 32 cached positions per row, this step writes position 7
            const int ctx = 32;
            kc.alloc((size_t)mp * ctx * D * 2); vc.alloc((size_t)mp * ctx * D * 2); spb.alloc(sizeof(StepParams));
            StepParams sp{}; sp.cache_pos = 7; sp.n_rows = (int)M;
            HIP_CHECK(hipMemcpy(spb.p, &sp, sizeof(sp), hipMemcpyHostToDevice));
            e.bias = bias.as<float>(); e.scale = 0.35f; e.scale_n_end = 2 * D; e.ldc = D; e.qkv_d_model = D;
            e.kcache = kc.as<half_t>(); e.vcache = vc.as<half_t>(); e.cache_row_stride = (int64_t)ctx * D; e.sp = spb.as<StepParams>();
        } else
        if (sh.ksplit > 1) { e.ksplit = sh.ksplit; e.slab_stride = mp * sh.N; }
        else { e.bias = bias.as<float>(); e.act = sh.gelu; if (dec && !sh.f32out) e.c_packed = 1; }
        if (sh.res) e.residual = c.as<float>();
        double best = 1e30;
        for (int rep = 0; rep < 4; ++rep) {
            HIP_CHECK(hipStreamSynchronize(s));
            const double t0 = now();
            for (int i = 0; i < reps; ++i) launch_gemm_f16(A, w.as<half_t>(), M, sh.N, sh.K, e, s);
            HIP_CHECK(hipStreamSynchronize(s));
            best = std::min(best, (now() - t0) / reps);
        }
        printf("%-7s M=%ld N=%d K=%d ksplit=%d: %.1f us, %.0f TFLOP/s\n", sh.name, (long)M, sh.N, sh.K, sh.ksplit, best, 2.0 * M * sh.N * sh.K / best / 1e6);
    }
    return 0;
}
