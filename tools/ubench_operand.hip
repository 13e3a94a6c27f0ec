// Micro-benchmark: cost of where a VALU operand comes from on gfx950 -- VGPR, SGPR, or another lane through DPP
// row_newbcast -- at 6 waves per SIMD on every CU.  Prints giga wave-instructions per second per variant; the
// ratios say how many cycles an SGPR-sourced or DPP-sourced operation costs relative to a VGPR-only one.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 4096;

#define OP8(TXT) asm volatile(TXT(%0) TXT(%1) TXT(%2) TXT(%3) TXT(%4) TXT(%5) TXT(%6) TXT(%7) \
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(vb), "s"(sb), "v"(vc))
#define SUB_V(r) "v_sub_f32 " #r ", %8, " #r "\n"
#define SUB_S(r) "v_sub_f32 " #r ", %9, " #r "\n"
#define SUB_D(r) "v_sub_f32_dpp " #r ", %8, " #r " row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
#define FMA_V(r) "v_fma_f32 " #r ", " #r ", %8, %10\n"
#define FMA_S(r) "v_fma_f32 " #r ", " #r ", %9, %10\n"
#define FMAC_D(r) "v_fmac_f32_dpp " #r ", %8, %10 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
#define MOV_D(r) "v_mov_b32_dpp " #r ", %8 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"

template <int WHICH>
__global__ __launch_bounds__(256) void k(float* out, float sb, float c) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float vb = sb + threadIdx.x * 1e-9f, vc = c * 1e-3f;
    for (int i = 0; i < ITERS; ++i) {
        if (WHICH == 0) OP8(SUB_V);
        if (WHICH == 1) OP8(SUB_S);
        if (WHICH == 2) OP8(SUB_D);
        if (WHICH == 3) OP8(FMA_V);
        if (WHICH == 4) OP8(FMA_S);
        if (WHICH == 5) OP8(FMAC_D);
        if (WHICH == 6) OP8(MOV_D);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <typename K>
int run(const char* name, K kern, float* d_out, int cus) {
    const int blocks = cus * 6;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, 1.0000001f, 1e-9f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, 1.0000001f, 1e-9f);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double winstr = 8.0 * ITERS * 4 * blocks * 5; // wave-instructions
    printf("%-30s %8.3f ms  %8.1f G wave-instr/s  (%.2f per SIMD per ns)\n", name, ms / 5, winstr / (ms * 1e-3) / 1e9,
           winstr / (ms * 1e-3) / 1e9 / (cus * 4));
    return 0;
}

int main() {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    float* d_out;
    CHECK(hipMalloc(&d_out, sizeof(float) * 256 * p.multiProcessorCount * 6));
    const int cus = p.multiProcessorCount;
    if (run("v_sub_f32 v,v,v", k<0>, d_out, cus)) return 1;
    if (run("v_sub_f32 v,s,v", k<1>, d_out, cus)) return 1;
    if (run("v_sub_f32_dpp row_newbcast", k<2>, d_out, cus)) return 1;
    if (run("v_fma_f32 v,v,v,v", k<3>, d_out, cus)) return 1;
    if (run("v_fma_f32 v,v,s,v", k<4>, d_out, cus)) return 1;
    if (run("v_fmac_f32_dpp row_newbcast", k<5>, d_out, cus)) return 1;
    if (run("v_mov_b32_dpp row_newbcast", k<6>, d_out, cus)) return 1;
    CHECK(hipFree(d_out));
    return 0;
}
