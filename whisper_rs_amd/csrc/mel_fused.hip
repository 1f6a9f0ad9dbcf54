// Fused log-mel front-end (gfx950): reference src/audio.rs:230-259 `log_mel_spectrogram`
//   stft(n_fft 400, hop 160, periodic Hann, center = true / reflect padding, onesided) -> drop the last frame -> |.|^2 ->
//   mel filterbank (80 x 201, 391 non-zeros) -> log10(max(x, 1e-10)) -> floor at (max - 8) -> (x + 4) / 4.
//
// ONE kernel does frames -> DFT -> power -> mel -> log10 -> per-recording max with the intermediate tiles in LDS: HBM sees
// the audio once (coalesced frame loads: a workgroup reads the contiguous 10 480-sample span of its 64 frames) and the
// [80, F] log-mel once; a second, elementwise kernel applies the floor / scale once the recording's max is known (the
// reference's `log_spec.max()` is global over the input, audio.rs:256).  Round 1 wrote the reflect-padded signal, a
// 208-wide power spectrum and a time-major log-mel to HBM between three kernels (~6x the algorithmic bytes).
//
// The DFT is a GEMM on exact-fp32 matrix cores (v_mfma_f32_16x16x4_f32): frames are OVERLAPPING rows of the signal
// (row f = samples f*160 .. f*160+399, read straight from the LDS copy of the span), the [402 x 400] Hann-windowed
// cos/-sin matrix streams from L2 in MFMA-fragment order.  It is MFMA-bound by construction (2 * 400 * 402 flops per
// frame in fp32), not HBM-bound: see DESIGN.md section 3 for the roofline of this stage.
#include "kernels.h"
#include <atomic>
#include <cmath>

namespace wrs {

typedef float floatx4 __attribute__((ext_vector_type(4)));

static constexpr int MEL_FT = 64;                                // frames per workgroup
static constexpr int MEL_SPAN = (MEL_FT - 1) * 160 + 400;        // 10 480 samples
static constexpr int MEL_SIG_LDS = MEL_SPAN + MEL_SPAN / 160 + 1;  // skewed: one pad float per 160 samples (bank spread)
static constexpr int MEL_P_LD = 209;                             // power tile row pitch (201 bins, odd pitch: conflict-free column reads)
static constexpr int MEL_NT = 28;                                // 448 / 16 column tiles of the DFT matrix (402 used)
static constexpr int MEL_K4 = 100;                               // 400 / 4 k-steps

__device__ __forceinline__ int mel_order_key(float f) {
    const int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7fffffff;
}
__device__ __forceinline__ float mel_key_to_float(int k) { return __int_as_float(k >= 0 ? k : k ^ 0x7fffffff); }

struct MelFusedParams {
    const float* audio;    // [n_batch, n]
    int64_t n;             // samples per recording
    int64_t n_frames;      // F = n / 160
    const float* twp;      // DFT matrix in fragment order [28][100][64]
    const float* filt;     // [80][201] dense mel filterbank
    const int* krange;     // [80][2]: first and one-past-last non-zero bin of every mel row
    float* out;            // [n_batch, 80, F] log10 domain (floor / scale applied by mel_scale_kernel)
    int* maxkey;           // [n_batch] order-preserving int key of the running max
};

__global__ __launch_bounds__(256) void mel_fused_kernel(const MelFusedParams p) {
    extern __shared__ __attribute__((aligned(16))) float mel_smem[];  // signal span (skewed), later the power tile [64][209]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t b = blockIdx.y;
    const int64_t f0 = (int64_t)blockIdx.x * MEL_FT;
    const float* src = p.audio + b * p.n;

    // ---- span of the 64 frames -> LDS, reflect padding at both ends (torch.stft center = true, pad_mode = "reflect")
    const int64_t s0 = f0 * 160 - 200;  // sample index of the span's first element
    for (int i = tid; i < MEL_SPAN; i += 256) {
        int64_t j = s0 + i;
        if (j < 0) j = -j;
        if (j >= p.n) j = 2 * (p.n - 1) - j;
        j = j < 0 ? 0 : (j >= p.n ? p.n - 1 : j);  // frames past the end of the recording (never stored) stay in range
        mel_smem[i + i / 160] = src[j];
    }
    __syncthreads();

    // ---- DFT: acc[mt][nt] (16 frames x 16 columns each) over k = 400; wave w owns column tiles 7w .. 7w+6
    floatx4 acc[4][7];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 7; ++nt) acc[mt][nt] = floatx4{0.f, 0.f, 0.f, 0.f};
    const float* wbase = p.twp + ((int64_t)(wave * 7) * MEL_K4) * 64 + lane;
    const int fl = lane & 15, kq = lane >> 4;
    float bcur[7], bnext[7];
#pragma unroll
    for (int nt = 0; nt < 7; ++nt) bcur[nt] = wbase[((int64_t)nt * MEL_K4) * 64];
    for (int k4 = 0; k4 < MEL_K4; ++k4) {
        if (k4 + 1 < MEL_K4) {
#pragma unroll
            for (int nt = 0; nt < 7; ++nt) bnext[nt] = wbase[((int64_t)nt * MEL_K4 + k4 + 1) * 64];
        }
        const int k = k4 * 4 + kq;
        float a[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int i = (mt * 16 + fl) * 160 + k;  // sample offset inside the span
            a[mt] = mel_smem[i + i / 160];
        }
#pragma unroll
        for (int nt = 0; nt < 7; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bcur[nt], a[mt], acc[mt][nt], 0, 0, 0);
#pragma unroll
        for (int nt = 0; nt < 7; ++nt) bcur[nt] = bnext[nt];
    }
    __syncthreads();  // every wave is done with the signal: the LDS becomes the power tile

    // ---- power: lane holds columns n = 16 nt' + 4 kq + r of frame 16 mt + fl; (r = 0, 1) and (2, 3) are (re, im) of two bins
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int f = mt * 16 + fl;
#pragma unroll
        for (int nt = 0; nt < 7; ++nt) {
            const int bin = ((wave * 7 + nt) * 16 + 4 * kq) >> 1;
            if (bin < 201) mel_smem[f * MEL_P_LD + bin] = acc[mt][nt][0] * acc[mt][nt][0] + acc[mt][nt][1] * acc[mt][nt][1];
            if (bin + 1 < 201) mel_smem[f * MEL_P_LD + bin + 1] = acc[mt][nt][2] * acc[mt][nt][2] + acc[mt][nt][3] * acc[mt][nt][3];
        }
    }
    __syncthreads();

    // ---- mel projection (only the non-zero band of each filter row), log10, running max, [80, F] store (64 consecutive frames per wave)
    float vmax = -3.0e38f;
    for (int idx = tid; idx < MEL_FT * 80; idx += 256) {
        const int f = idx & (MEL_FT - 1), m = idx >> 6;
        const int k0 = p.krange[2 * m], k1 = p.krange[2 * m + 1];
        const float* fr = p.filt + m * 201;
        float s = 0.f;
        for (int k = k0; k < k1; ++k) s += fr[k] * mel_smem[f * MEL_P_LD + k];
        if (f0 + f < p.n_frames) {
            const float v = log10f(fmaxf(s, 1e-10f));
            p.out[(b * 80 + m) * p.n_frames + f0 + f] = v;
            vmax = fmaxf(vmax, v);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
    if (lane == 0 && vmax > -3.0e38f) atomicMax(p.maxkey + b, mel_order_key(vmax));
}

// reference audio.rs:256-258: floor at (global max - 8), then (x + 4) / 4, in place on [n_batch, 80, F]
__global__ void mel_scale_kernel(float* __restrict__ x, const int* __restrict__ maxkey, int64_t per_batch, int64_t n_batch) {
    const int64_t b = blockIdx.y;
    const float floor_v = mel_key_to_float(maxkey[b]) - 8.0f;
    float* row = x + b * per_batch;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < per_batch; i += (int64_t)gridDim.x * blockDim.x)
        row[i] = (fmaxf(row[i], floor_v) + 4.0f) / 4.0f;
}

size_t mel_twiddle_packed_floats() { return (size_t)MEL_NT * MEL_K4 * 64; }

// host: DFT matrix rows 2b = hann[n] cos(2 pi b n / 400), 2b+1 = -hann[n] sin(..) (periodic Hann, audio.rs:233), rows >= 402 zero,
// in MFMA-fragment order: out[(nt * 100 + k4) * 64 + l] = tw[nt*16 + (l & 15)][k4*4 + (l >> 4)]
void mel_pack_twiddle(float* out) {
    const double two_pi = 6.283185307179586476925286766559;
    for (int nt = 0; nt < MEL_NT; ++nt)
        for (int k4 = 0; k4 < MEL_K4; ++k4)
            for (int l = 0; l < 64; ++l) {
                const int row = nt * 16 + (l & 15), n = k4 * 4 + (l >> 4);
                float v = 0.f;
                if (row < 402) {
                    const int bb = row >> 1;
                    const double win = 0.5 - 0.5 * std::cos(two_pi * n / 400.0);
                    const double ang = two_pi * (double)((bb * n) % 400) / 400.0;
                    v = (float)((row & 1) ? -win * std::sin(ang) : win * std::cos(ang));
                }
                out[((size_t)nt * MEL_K4 + k4) * 64 + l] = v;
            }
}

void launch_mel_fused(const float* audio, int64_t n_batch, int64_t n, const float* twp, const float* filt, const int* krange, float* out,
                      int* maxkey, hipStream_t s) {
    const int64_t F = n / 160;
    WRS_REQUIRE(n_batch >= 1 && n >= 201 && F >= 1 && n_batch < 65536, "mel: bad shape");
    MelFusedParams p{audio, n, F, twp, filt, krange, out, maxkey};
    const size_t lds = (size_t)std::max(MEL_SIG_LDS, MEL_FT * MEL_P_LD) * 4;
    static std::atomic<bool> attr[64];  // per device (a window pool drives several GPUs from several threads)
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    if (dev >= 0 && dev < 64 && !attr[dev].load(std::memory_order_acquire)) {
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(mel_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr[dev].store(true, std::memory_order_release);
    }
    dim3 grid((unsigned)cdiv(F, MEL_FT), (unsigned)n_batch);
    mel_fused_kernel<<<grid, 256, lds, s>>>(p);
    dim3 g2((unsigned)std::min<int64_t>(cdiv(80 * F, 256 * 4), 1024), (unsigned)n_batch);
    mel_scale_kernel<<<g2, 256, 0, s>>>(out, maxkey, 80 * F, n_batch);
}

}  // namespace wrs
