// Micro-benchmark: sustained rate of v_fma_f32 and v_pk_fma_f32 on gfx950 at 1..8 waves per SIMD,
// with VGPR and with SGPR multiplicands.  Prints TFLOP/s per variant.  Used to decide whether the
// sphere scan should be written with packed or plain f32 operations.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;

__global__ void k_fma(float* out, float b, float c) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float vb = b + threadIdx.x * 1e-9f, vc = c;
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                     "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(vb), "v"(vc));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

__global__ void k_fma_sgpr(float* out, float b, float c) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float vc = c;
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                     "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(b), "v"(vc));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

__global__ void k_pk(float* out, float b, float c) {
    f2 a0 = {(float)threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    f2 vb = {b + threadIdx.x * 1e-9f, b}, vc = {c, c};
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                     "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(vb), "v"(vc));
    }
    f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

// the sphere-scan pattern: 3 sub (SGPR src), mul, 2 fmac, 3 fma, fmac, cmp -- no loads
__global__ void k_scan(float* out, float4 s0, float ox, float oy, float oz) {
    float vx = ox + threadIdx.x * 1e-3f, vy = oy, vz = oz, dx = 0.6f, dy = 0.8f, dz = 0.f;
    float acc = 0.f;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            float cx = __builtin_amdgcn_readfirstlane(s0.x + u + i * 1e-6f);
            float ocx = cx - vx, ocy = s0.y - vy, ocz = s0.z - vz;
            float h = fmaf(ocz, dz, fmaf(ocy, dy, ocx * dx));
            float q = fmaf(-ocz, ocz, fmaf(-ocy, ocy, fmaf(-ocx, ocx, s0.w)));
            float disc = fmaf(h, h, q);
            acc = fmaxf(acc, disc);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <typename K, typename... A>
int run(const char* name, K kern, double flop_per_thread, int cus, float* d_out, A... args) {
    for (int waves_per_simd = 1; waves_per_simd <= 8; waves_per_simd *= 2) {
        int blocks = cus * waves_per_simd; // 256 threads = 4 waves = 1 wave per SIMD per block
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, args...);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        const int reps = 5;
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, args...);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        double tf = flop_per_thread * 256.0 * blocks * reps / (ms * 1e-3) / 1e12;
        printf("%-14s waves/SIMD=%d  %8.3f ms  %8.2f TFLOP/s\n", name, waves_per_simd, ms / reps, tf);
    }
    return 0;
}

int main() {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    printf("device %s, CUs %d, clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
    float* d_out;
    CHECK(hipMalloc(&d_out, sizeof(float) * 256 * p.multiProcessorCount * 8));
    int cus = p.multiProcessorCount;
    if (run("v_fma_f32", k_fma, 2.0 * 8 * ITERS, cus, d_out, 1.0000001f, 1e-9f)) return 1;
    if (run("v_fma_f32 sgpr", k_fma_sgpr, 2.0 * 8 * ITERS, cus, d_out, 1.0000001f, 1e-9f)) return 1;
    if (run("v_pk_fma_f32", k_pk, 4.0 * 8 * ITERS, cus, d_out, 1.0000001f, 1e-9f)) return 1;
    // scan: 18 algorithmic flop per sphere test, 8 tests per iteration
    if (run("scan pattern", k_scan, 18.0 * 8 * ITERS, cus, d_out, make_float4(1.f, 2.f, 3.f, 0.04f), 0.1f, 0.2f, 0.3f)) return 1;
    CHECK(hipFree(d_out));
    return 0;
}
