// The decode self-attention's REAL address pattern, loads only: wave (audio a, head h) reads, for positions t = 0 .. len-1,
// 128 bytes of K and of V at (row * row_stride + t * 2560 + h * 128), row = one of the audio's 5 physical rows; 5 120 waves
// walk t in step.  Question: does the cache's row stride (448 * 2560 = 35 * 32 KiB) put all rows on the same HBM channels?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float floatx4 __attribute__((ext_vector_type(4)));
// swz 0: position-major rows, a load instruction = 8 keys x 128 B; 1: head-major rows; 2: position-major, MFMA-fragment lane
// order (a load instruction = 16 keys x 64 B: lane = key l & 15, 16-byte chunk (l >> 4) + 4 (i & 1)); 3: as 2, with the
// (position, row) pairs read from a list in memory (one entry per lane, __shfl to the loading lanes), as the kernel does
__global__ __launch_bounds__(256) void walk(const char* __restrict__ kc, const char* __restrict__ vc, size_t row_stride, int len, int swz, float* out, const int* __restrict__ list) {
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int a = w / 20, h = w % 20;
    floatx4 acc = {0, 0, 0, 0};
    floatx4 k[4], v[4];
    int e_nxt = 0;
    auto issue = [&](int b) {
        if (swz >= 2) {
            int e = 0;
            if (swz == 3) { e = e_nxt; e_nxt = list[(size_t)a * 512 + min((b + 2) * 32 + (lane & 31), 447)]; }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int key = (i >> 1) * 16 + (lane & 15);
                int t = b * 32 + key;
                int row = a * 5 + ((t / 7 + a) % 5) * (t > len - 56 ? 1 : 0);
                if (swz == 3) { const int ek = __shfl(e, key); t = ek & 511; row = a * 5 + ((ek >> 9) & 7); }
                const size_t off = (size_t)row * row_stride + (size_t)t * 2560 + h * 128 + (((lane >> 4) + 4 * (i & 1)) * 16);
                k[i] = *reinterpret_cast<const floatx4*>(kc + off);
                v[i] = *reinterpret_cast<const floatx4*>(vc + off);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int t = b * 32 + i * 8 + (lane >> 3);
            const int row = a * 5 + ((t / 7 + a) % 5) * (t > len - 56 ? 1 : 0);  // recent positions: beams differ; old ones: shared row
            size_t off = (size_t)row * row_stride + (size_t)t * 2560 + h * 128 + (lane & 7) * 16;
            if (swz) off = (size_t)row * row_stride + (size_t)h * (448 * 128) + (size_t)t * 128 + (lane & 7) * 16;  // head-major rows
            k[i] = *reinterpret_cast<const floatx4*>(kc + off);
            v[i] = *reinterpret_cast<const floatx4*>(vc + off);
        }
    };
    const int nb = len / 32;
    if (swz == 3) { e_nxt = list[(size_t)a * 512 + (lane & 31)]; issue(-1 + 1); }
    else issue(0);
    for (int b = 0; b < nb; ++b) {
        floatx4 s = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 4; ++i) s += k[i] + v[i];
        if (b + 1 < nb) issue(b + 1);
        acc += s;
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}
int main() {
    const int rows = 1280, len = 448;
    const size_t max_stride = (size_t)448 * 2560 + 65536;
    char *kc, *vc; float* out; int* list;
    {
        std::vector<int> hl(256 * 512);
        for (int a = 0; a < 256; ++a)
            for (int t = 0; t < 448; ++t) hl[a * 512 + t] = t | ((((t / 7 + a) % 5) * (t > 448 - 56 ? 1 : 0)) << 9);
        CK(hipMalloc(&list, hl.size() * 4));
        CK(hipMemcpy(list, hl.data(), hl.size() * 4, hipMemcpyHostToDevice));
    }
    CK(hipMalloc(&kc, rows * max_stride)); CK(hipMalloc(&vc, rows * max_stride)); CK(hipMalloc(&out, 4));
    CK(hipMemset(kc, 0, rows * max_stride)); CK(hipMemset(vc, 0, rows * max_stride));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t base = (size_t)448 * 2560;
    const size_t pads[] = {0, 128, 256, 512, 1024, 2560, 4096, 8192, 16384, 2560 + 256, 32768 + 256};
    for (int swz = 0; swz < 4; ++swz)
        for (size_t pad : pads) {
            float best = 1e9;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipEventRecord(e0));
                walk<<<5120 / 4, 256>>>(kc, vc, base + pad, len, swz, out, list);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best) best = ms;
            }
            const double gb = 5120.0 * len * 256 / 1e9;
            printf("mode %d %s row stride 448*2560 + %6zu: %7.1f us, %.2f TB/s (requested; distinct rows are fewer)\n", swz, swz == 1 ? "head-major" : "pos-major ", pad, best * 1e3, gb / best);
            if (pad > 0) break;
        }
    // occupancy: the attention kernel runs at 4 waves per SIMD (124 VGPRs); limit this kernel the same way through LDS
    {
        extern __shared__ char dummy[];
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&walk), hipFuncAttributeMaxDynamicSharedMemorySize, 81408));
        for (int wps : {8, 4, 3, 2})
            for (int swz : {0, 2, 3}) {
                const size_t lds = wps >= 8 ? 0 : 160 * 1024 / wps - 512;
                float best = 1e9;
                for (int rep = 0; rep < 4; ++rep) {
                    CK(hipEventRecord(e0));
                    walk<<<5120 / 4, 256, lds>>>(kc, vc, base, len, swz, out, list);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep && ms < best) best = ms;
                }
                printf("mode %d, %d waves per SIMD: %7.1f us, %.2f TB/s\n", swz, wps, best * 1e3, 5120.0 * len * 256 / 1e9 / best);
            }
    }
    // cold address translation: the decode step touches a different layer's cache (2 x 1.47 GB) at every launch, 32 layers
    // = 94 GB in rotation.  Same kernel, L regions walked round-robin; per-launch average.
    for (int L : {1, 4, 32}) {
        std::vector<char*> ks(L), vs(L);
        const size_t bytes = (size_t)rows * base;
        for (int l = 0; l < L; ++l) { CK(hipMalloc(&ks[l], bytes)); CK(hipMalloc(&vs[l], bytes)); CK(hipMemsetAsync(ks[l], 0, bytes)); CK(hipMemsetAsync(vs[l], 0, bytes)); }
        CK(hipDeviceSynchronize());
        for (int len2 : {448, 224, 112}) {
            float best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0));
                for (int l = 0; l < L; ++l) walk<<<5120 / 4, 256>>>(ks[l], vs[l], base, len2, 0, out, list);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms / L < best) best = ms / L;
            }
            printf("%2d regions in rotation, len %3d: %7.1f us per launch, %.2f TB/s\n", L, len2, best * 1e3, 5120.0 * len2 * 256 / 1e9 / best);
        }
        for (int l = 0; l < L; ++l) { CK(hipFree(ks[l])); CK(hipFree(vs[l])); }
    }
    return 0;
}
