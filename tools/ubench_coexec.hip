// Micro-benchmark: do matrix-core instructions execute beside VALU instructions on gfx950?
// Times a VALU-only loop, an MFMA-only loop (bf16 16x16x32 and f32 16x16x4) and the two interleaved,
// at 4 waves per SIMD on every CU.  If t(both) ~ max(t(valu), t(mfma)) the pipes overlap;
// if t(both) ~ t(valu) + t(mfma) they share the same ALUs.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 2048;

#define V4(a) asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" \
                          : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(vb), "v"(vc))
#define V4B(a) asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" \
                          : "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(vb), "v"(vc))
#define MB(acc) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(ma), "v"(mb))
#define MF(acc) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(fa), "v"(fb))

// per iteration: NMFMA (0 or 4) matrix instructions on 4 independent accumulators, each followed by
// VPER VALU fmas on 8 independent accumulators; everything in VGPRs, nothing depends across pipes
template <int VPER, int NMFMA, bool F32M>
__global__ __launch_bounds__(256) void k(float* out, float b, float c) {
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    bf16x8 ma = {1, 2, 3, 4, 5, 6, 7, 8}, mb = {8, 7, 6, 5, 4, 3, 2, 1};
    float fa = b + threadIdx.x, fb = c, vb = b + threadIdx.x * 1e-9f, vc = c;
    for (int it = 0; it < ITERS; ++it) {
#define GROUP(acc)                                         \
        if (NMFMA) { if (F32M) MF(acc); else MB(acc); }    \
        _Pragma("unroll") for (int v = 0; v < VPER / 8; ++v) { V4(a); V4B(a); }
        GROUP(acc0) GROUP(acc1) GROUP(acc2) GROUP(acc3)
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    s += acc0[0] + acc1[1] + acc2[2] + acc3[3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// Specialised waves: 512-thread blocks; waves 0-3 issue only matrix instructions, waves 4-7 only VALU, so every
// SIMD hosts one wave of each kind per block (2 blocks per CU: two of each per SIMD).
template <bool F32M>
__global__ __launch_bounds__(512) void ksplit(float* out, float b, float c) {
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    bf16x8 ma = {1, 2, 3, 4, 5, 6, 7, 8}, mb = {8, 7, 6, 5, 4, 3, 2, 1};
    float fa = b + threadIdx.x, fb = c, vb = b + threadIdx.x * 1e-9f, vc = c;
    if (threadIdx.x >= 256) { // waves 4-7 share SIMDs 0-3 with waves 0-3
        for (int it = 0; it < ITERS; ++it) { V4(a); V4B(a); V4(a); V4B(a); V4(a); V4B(a); V4(a); V4B(a); }
    } else {
        for (int it = 0; it < ITERS; ++it) {
            if (F32M) { MF(acc0); MF(acc1); MF(acc2); MF(acc3); } else { MB(acc0); MB(acc1); MB(acc2); MB(acc3); }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    s += acc0[0] + acc1[1] + acc2[2] + acc3[3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MB32(acc) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(ma), "v"(mb))

// 32x32x16 bf16 (32 cycles each).  MODE 0: matrix only; 1: each followed by VPER VALU fmas in the same wave;
// 2: waves 0-3 matrix only, waves 4-7 VALU only (512-thread blocks)
template <int MODE, int VPER>
__global__ __launch_bounds__(512) void k32(float* out, float b, float c) {
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
    f32x16 acc0 = {0}, acc1 = {0};
    bf16x8 ma = {1, 2, 3, 4, 5, 6, 7, 8}, mb = {8, 7, 6, 5, 4, 3, 2, 1};
    float vb = b + threadIdx.x * 1e-9f, vc = c;
    const bool valu_wave = MODE == 2 && threadIdx.x >= 256, mfma_wave = MODE != 2 || threadIdx.x < 256;
    for (int it = 0; it < ITERS; ++it) {
        if (mfma_wave) MB32(acc0);
        if (MODE == 1 || valu_wave) { _Pragma("unroll") for (int v = 0; v < VPER / 8; ++v) { V4(a); V4B(a); } }
        if (mfma_wave) MB32(acc1);
        if (MODE == 1 || valu_wave) { _Pragma("unroll") for (int v = 0; v < VPER / 8; ++v) { V4(a); V4B(a); } }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    s += acc0[0] + acc1[1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
int run(const char* name, K kern, float* d_out, int cus, int threads = 256) {
    const int blocks = cus * 4 * 256 / threads;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d_out, 1.0000001f, 1e-9f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d_out, 1.0000001f, 1e-9f);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-34s %8.3f ms per launch\n", name, ms / 5);
    return 0;
}

int main() {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    float* d_out;
    CHECK(hipMalloc(&d_out, sizeof(float) * 256 * p.multiProcessorCount * 4));
    const int cus = p.multiProcessorCount;
    if (run("valu only: 4 x 16 fma", k<16, 0, false>, d_out, cus)) return 1;
    if (run("valu only: 4 x 8 fma", k<8, 0, false>, d_out, cus)) return 1;
    if (run("bf16 mfma only: 4", k<0, 4, false>, d_out, cus)) return 1;
    if (run("bf16 mfma 4 + 4 x 8 fma", k<8, 4, false>, d_out, cus)) return 1;
    if (run("bf16 mfma 4 + 4 x 16 fma", k<16, 4, false>, d_out, cus)) return 1;
    if (run("split: 2 bf16-mfma waves + 2 valu(4x8) waves per SIMD", ksplit<false>, d_out, cus, 512)) return 1;
    if (run("split: 2 f32-mfma waves + 2 valu(4x8) waves per SIMD", ksplit<true>, d_out, cus, 512)) return 1;
    if (run("32x32x16 bf16 only: 2 per iter (256-thr)", k32<0, 0>, d_out, cus)) return 1;
    if (run("32x32x16 bf16 2 + 2 x 8 fma same wave", k32<1, 8>, d_out, cus)) return 1;
    if (run("32x32x16 bf16 2 + 2 x 16 fma same wave", k32<1, 16>, d_out, cus)) return 1;
    if (run("valu only 2 x 16 fma (mode 2 valu waves alone ~)", k<8, 0, false>, d_out, cus)) return 1;
    if (run("split 32x32x16: mfma waves + valu(2x16) waves", k32<2, 16>, d_out, cus, 512)) return 1;
    if (run("split 32x32x16: mfma waves + valu(2x32) waves", k32<2, 32>, d_out, cus, 512)) return 1;
    if (run("f32 mfma only: 4", k<0, 4, true>, d_out, cus)) return 1;
    if (run("f32 mfma 4 + 4 x 8 fma", k<8, 4, true>, d_out, cus)) return 1;
    if (run("f32 mfma 4 + 4 x 16 fma", k<16, 4, true>, d_out, cus)) return 1;
    CHECK(hipFree(d_out));
    return 0;
}
