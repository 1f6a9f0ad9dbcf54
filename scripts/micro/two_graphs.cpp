// Do two hipGraphExec launched on two streams overlap?  Each graph = chain of N dependent kernels that each
// take ~5 us on a fraction of the chip (latency-bound: one dependent global round trip chain inside).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void chase(const int* __restrict__ next, int* out, int hops) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int p = i;
    for (int h = 0; h < hops; ++h) p = next[p];  // dependent loads: latency-bound
    out[i] = p;
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const int N = 200, n = 1 << 22;
    std::vector<int> h(n);
    for (int i = 0; i < n; ++i) h[i] = (int)(((long long)i * 7919 + 104729) % n);
    int *next, *out[4];
    CK(hipMalloc(&next, n * 4)); CK(hipMemcpy(next, h.data(), n * 4, hipMemcpyHostToDevice));
    hipStream_t s[4]; hipGraph_t g[4]; hipGraphExec_t ge[4];
    for (int c = 0; c < 4; ++c) {
        CK(hipMalloc(&out[c], n * 4));
        CK(hipStreamCreateWithFlags(&s[c], hipStreamNonBlocking));
        CK(hipStreamBeginCapture(s[c], hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < N; ++i) chase<<<80, 256, 0, s[c]>>>(next, out[c], 24);
        CK(hipStreamEndCapture(s[c], &g[c]));
        CK(hipGraphInstantiate(&ge[c], g[c], nullptr, nullptr, 0));
    }
    for (int chains = 1; chains <= 4; ++chains) {
        double best = 1e30;
        for (int rep = 0; rep < 4; ++rep) {
            double t0 = now();
            for (int c = 0; c < chains; ++c) CK(hipGraphLaunch(ge[c], s[c]));
            for (int c = 0; c < chains; ++c) CK(hipStreamSynchronize(s[c]));
            best = std::min(best, now() - t0);
        }
        printf("%d graph(s) on %d stream(s), %d kernels each: %.1f us total, %.2f us per kernel-slot\n", chains, chains, N, best, best / N);
    }
    return 0;
}
