// How fast can one CU pull L2-resident operand tiles into LDS?  (The 128x128x64 GEMM needs 32 KiB per k-tile per
// workgroup; its k-tile takes ~0.7 us per CU whatever the staging form.)  Every workgroup streams `tiles` tiles of 32 KiB
// (8 x 1 KiB direct-to-LDS pieces per wave, fragment-packed like the GEMM's weight operand) out of a 4 MiB buffer that
// stays in L2, with `inflight` tiles outstanding and a barrier per tile as the GEMM has.
//   l2_to_lds_bw <wgs_per_cu> <inflight 1|2|3> <mode 0=dma 1=regs>
#include <hip/hip_runtime.h>
#ifndef GLDS_AUX
#define GLDS_AUX 0  // cache-policy bits of the direct-to-LDS load: 0 default, 1 sc0, 2 nt, 16 sc1 (L1 bypass)
#endif
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

template <int INFLIGHT, bool DMA>
__global__ __launch_bounds__(256) void stream_kernel(const _Float16* __restrict__ src, int tiles, int src_tiles, float* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    half8* lds = reinterpret_cast<half8*>(smem);  // [4 stages][32 pieces][64 lanes]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int first = (blockIdx.x * 7) % src_tiles;
    auto issue = [&](int t, int stage) {
        const _Float16* base = src + ((size_t)((first + t) % src_tiles) * 32 + wave * 8) * 512 + lane * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (DMA) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + j * 512),
                                                      (__attribute__((address_space(3))) void*)(lds + (stage * 32 + wave * 8 + j) * 64), 16, 0, GLDS_AUX);
            else lds[(stage * 32 + wave * 8 + j) * 64 + lane] = *reinterpret_cast<const half8*>(base + j * 512);
        }
    };
    for (int s = 0; s < INFLIGHT; ++s) issue(s, s);
    half8 acc = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int t = 0; t < tiles; ++t) {
        if (DMA) {
            if (INFLIGHT == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (INFLIGHT == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            if (INFLIGHT == 3) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        }
        __syncthreads();
        acc += lds[((t & 3) * 32 + wave * 8) * 64 + lane];  // touch the tile
        __syncthreads();
        if (t + INFLIGHT < tiles + INFLIGHT) issue(t + INFLIGHT, (t + INFLIGHT) & 3);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((float)acc[0] == 12345.f) sink[0] = 1.f;
}

int main(int argc, char** argv) {
    const int per_cu = argc > 1 ? atoi(argv[1]) : 1, inflight = argc > 2 ? atoi(argv[2]) : 1, mode = argc > 3 ? atoi(argv[3]) : 0;
    const int src_tiles = 96;  // 3 MiB: L2-resident on every XCD
    _Float16* src; float* sink;
    CK(hipMalloc(&src, (size_t)src_tiles * 32768)); CK(hipMemset(src, 0, (size_t)src_tiles * 32768)); CK(hipMalloc(&sink, 4));
    const int tiles = 400, grid = 256 * per_cu;
    const size_t shmem = per_cu == 1 ? 131072 : 65536;  // pins the residency: 1 or 2 workgroups per CU
    auto run = [&](auto k) {
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        k<<<grid, 256, shmem, 0>>>(src, tiles, src_tiles, sink);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        k<<<grid, 256, shmem, 0>>>(src, tiles, src_tiles, sink);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        const double bytes = (double)grid * tiles * 32768;
        printf("wgs/cu %d inflight %d %s: %.1f us, %.1f us per 32-KiB tile per WG, %.1f GB/s per CU, %.2f TB/s chip\n", per_cu, inflight, mode ? "regs" : "dma",
               ms * 1e3, ms * 1e3 / tiles, bytes / 256 / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / 1e12);
    };
    if (mode == 0) { if (inflight == 1) run(stream_kernel<1, true>); else if (inflight == 2) run(stream_kernel<2, true>); else run(stream_kernel<3, true>); }
    else run(stream_kernel<1, false>);
    return 0;
}
