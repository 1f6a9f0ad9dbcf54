// HBM gather bandwidth against the size of the contiguous piece: every wave reads `iters` blocks of 8 KB, a block being
// 8192 / piece pieces of `piece` contiguous bytes at pseudo-random places of a 4 GiB buffer (16 B per lane per load, 8 lanes
// per 128 B).  Motivation: the decode self-attention gathers 128-byte K / V rows (one head of one cached position) and runs
// at 4.2 TB/s, the cross-attention streams contiguous tiles at 6.3 TB/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float floatx4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int PIECE>
__global__ __launch_bounds__(256) void gather(const char* __restrict__ buf, uint64_t n_pieces, int iters, float* out) {
    const int lane = threadIdx.x & 63;
    const uint32_t wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    floatx4 acc = {0, 0, 0, 0};
    constexpr int LANES_PER_PIECE = PIECE >= 1024 ? 64 : PIECE / 16;  // lanes covering one piece in one load
    constexpr int LOADS = 8;                                            // 8 loads x 64 lanes x 16 B = 8 KB per block
    for (int it = 0; it < iters; ++it) {
        floatx4 v[LOADS];
#pragma unroll
        for (int i = 0; i < LOADS; ++i) {
            // piece index of this lane's 16 bytes within the block
            const int byte_in_block = (i * 64 + lane) * 16;
            const int piece_in_block = byte_in_block / PIECE;
            const uint64_t piece = mix(wid * 7919u + it * 104729u + piece_in_block * 31u) % n_pieces;
            v[i] = __builtin_nontemporal_load(reinterpret_cast<const floatx4*>(buf + piece * PIECE + byte_in_block % PIECE));
        }
#pragma unroll
        for (int i = 0; i < LOADS; ++i) acc += v[i];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = acc[0];
}
template <int PIECE>
int run(const char* buf, uint64_t bytes, float* out) {
    const int waves = 5120 * 2, iters = 40;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        gather<PIECE><<<waves / 4, 256>>>(buf, bytes / PIECE, iters, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    const double gb = (double)waves * iters * 8192 / 1e9;
    printf("piece %5d B: %.3f ms, %.2f TB/s\n", PIECE, best, gb / best);
    return 0;
}
int main() {
    const uint64_t bytes = 4ull << 30;
    char* buf; float* out;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 4));
    CK(hipMemset(buf, 0, bytes));
    if (run<128>(buf, bytes, out) || run<256>(buf, bytes, out) || run<512>(buf, bytes, out) || run<1024>(buf, bytes, out) || run<4096>(buf, bytes, out) || run<8192>(buf, bytes, out)) return 1;
    return 0;
}
