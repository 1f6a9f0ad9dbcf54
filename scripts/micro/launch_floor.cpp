// Measures the per-kernel cost of a chain of dependent small kernels: eager launches vs hipGraph replay,
// single chain vs several independent chains captured as parallel graph branches.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void tiny(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.f; }
__global__ void stream_k(const float4* __restrict__ in, float4* __restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const int N = 400, blocks = 160;
    float* buf; CK(hipMalloc(&buf, 4 << 20));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        for (int i = 0; i < N; ++i) tiny<<<blocks, 256, 0, s>>>(buf, blocks * 256);
        CK(hipStreamSynchronize(s));
        if (rep) printf("eager chain: %.2f us per kernel\n", (now() - t0) / N);
    }
    for (int chains = 1; chains <= 4; chains *= 2) {
        std::vector<hipStream_t> ss(chains);
        std::vector<hipEvent_t> ev(chains);
        for (auto& x : ss) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
        for (auto& evt : ev) CK(hipEventCreateWithFlags(&evt, hipEventDisableTiming));
        hipEvent_t fork; CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        CK(hipEventRecord(fork, s));
        for (int c = 0; c < chains; ++c) {
            CK(hipStreamWaitEvent(ss[c], fork, 0));
            for (int i = 0; i < N / chains; ++i) tiny<<<blocks / chains, 256, 0, ss[c]>>>(buf + c * 65536, blocks / chains * 256);
            CK(hipEventRecord(ev[c], ss[c]));
            CK(hipStreamWaitEvent(s, ev[c], 0));
        }
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 3; ++rep) {
            double t0 = now();
            CK(hipGraphLaunch(ge, s));
            CK(hipStreamSynchronize(s));
            if (rep == 2) printf("graph, %d parallel chain(s) of %d kernels: total %.1f us = %.2f us per kernel (of %d)\n", chains, N / chains, now() - t0, (now() - t0) / N, N);
        }
    }
    // HBM copy bandwidth sanity
    size_t bytes = (size_t)1 << 30;
    float4 *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now();
        stream_k<<<2048, 256, 0, s>>>(a, b, bytes / 16);
        CK(hipStreamSynchronize(s));
        if (rep == 2) printf("copy 1 GiB: %.1f us -> %.2f TB/s (read+write)\n", now() - t0, 2.0 * bytes / (now() - t0) / 1e6);
    }
    return 0;
}
