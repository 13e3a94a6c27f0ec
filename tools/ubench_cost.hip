// Micro-benchmark: what kinds of vector instruction cost on gfx950 -- selects and compares through VCC or through a pair
// of scalar registers, literal and inline constants, f64 arithmetic, 64-bit integer multiply-adds, transcendentals,
// conversions -- as wave-instructions per nanosecond per SIMD at 6, 4 and 1 waves per SIMD on every CU.  Eight
// independent chains per variant, so that neither a dependency nor a result's latency is what is measured.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 2048;

#define OP8(TXT) asm volatile(TXT(%0) TXT(%1) TXT(%2) TXT(%3) TXT(%4) TXT(%5) TXT(%6) TXT(%7) \
    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(vb), "s"(sb), "v"(vc) : "vcc", "s20", "s21")
#define OP8D(TXT) asm volatile(TXT(%0) TXT(%1) TXT(%2) TXT(%3) TXT(%4) TXT(%5) TXT(%6) TXT(%7) \
    : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(db), "s"(sb), "v"(dc) : "vcc")
#define SUB_V(r) "v_sub_f32 " #r ", %8, " #r "\n"
#define SEL_VCC(r) "v_cndmask_b32_e32 " #r ", %8, " #r ", vcc\n"
#define SEL_SGPR(r) "v_cndmask_b32_e64 " #r ", %8, " #r ", s[20:21]\n"
#define CMP_VCC(r) "v_cmp_lt_f32_e32 vcc, %8, " #r "\n"
#define CMP_SGPR(r) "v_cmp_lt_f32_e64 s[20:21], %8, " #r "\n"
#define CMPSEL_VCC(r) "v_cmp_lt_f32_e32 vcc, %8, " #r "\nv_cndmask_b32_e32 " #r ", %10, " #r ", vcc\n"
#define CMPSEL_SGPR(r) "v_cmp_lt_f32_e64 s[20:21], %8, " #r "\nv_cndmask_b32_e64 " #r ", %10, " #r ", s[20:21]\n"
#define ADD_LIT(r) "v_add_f32 " #r ", 0x3f800011, " #r "\n"
#define ADD_INL(r) "v_add_f32 " #r ", 1.0, " #r "\n"
#define FMAAK(r) "v_fmaak_f32 " #r ", " #r ", %8, 0x3a83126f\n"
#define SQRT(r) "v_sqrt_f32 " #r ", " #r "\n"
#define RCP(r) "v_rcp_f32 " #r ", " #r "\n"
#define CVT(r) "v_cvt_u32_f32 " #r ", " #r "\n"
#define MIN3(r) "v_min3_f32 " #r ", " #r ", %8, %10\n"
#define FMA64(r) "v_fma_f64 " #r ", " #r ", %8, %10\n"
#define MUL64(r) "v_mul_f64 " #r ", " #r ", %8\n"
#define ADD64(r) "v_add_f64 " #r ", " #r ", %8\n"
#define MAD64(r) "v_mad_u64_u32 " #r ", vcc, %9, v1, " #r "\n"
#define RCP64(r) "v_rcp_f64 " #r ", " #r "\n"
#define MAD64V(r) "v_mad_u64_u32 " #r ", vcc, v2, v1, " #r "\n"
#define XOR_S(r) "v_xor_b32 " #r ", %9, " #r "\n"
#define XOR_V(r) "v_xor_b32 " #r ", %8, " #r "\n"
#define MULHI(r) "v_mul_hi_u32 " #r ", %8, " #r "\n"
#define LSHLADD_S(r) "v_lshl_add_u32 " #r ", " #r ", 2, %9\n"
#define CMPU64(r) "v_cmp_lt_u64_e32 vcc, " #r ", %8\n"

template <int WHICH>
__global__ __launch_bounds__(256) void k(float* out, float sb, float c) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7;
    float vb = sb + threadIdx.x * 1e-9f, vc = c * 1e-3f;
    double db = vb, dc = vc;
    for (int i = 0; i < ITERS; ++i) {
        if (WHICH == 0) OP8(SUB_V);
        if (WHICH == 1) OP8(SEL_VCC);
        if (WHICH == 2) OP8(SEL_SGPR);
        if (WHICH == 3) OP8(CMP_VCC);
        if (WHICH == 4) OP8(CMP_SGPR);
        if (WHICH == 5) OP8(CMPSEL_VCC);
        if (WHICH == 6) OP8(CMPSEL_SGPR);
        if (WHICH == 7) OP8(ADD_LIT);
        if (WHICH == 8) OP8(ADD_INL);
        if (WHICH == 9) OP8(FMAAK);
        if (WHICH == 10) OP8(SQRT);
        if (WHICH == 11) OP8(RCP);
        if (WHICH == 12) OP8(CVT);
        if (WHICH == 13) OP8(MIN3);
        if (WHICH == 14) OP8D(FMA64);
        if (WHICH == 15) OP8D(MUL64);
        if (WHICH == 16) OP8D(ADD64);
        if (WHICH == 17) OP8D(MAD64);
        if (WHICH == 18) OP8D(RCP64);
        if (WHICH == 19) OP8D(MAD64V);
        if (WHICH == 20) OP8(XOR_S);
        if (WHICH == 21) OP8(XOR_V);
        if (WHICH == 22) OP8(MULHI);
        if (WHICH == 23) OP8(LSHLADD_S);
        if (WHICH == 24) OP8D(CMPU64);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
}

template <typename K>
int run(const char* name, K kern, float* d_out, int cus, int per_cu, double per_op) {
    const int blocks = cus * per_cu;
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
    const double winstr = per_op * 8.0 * ITERS * 4 * blocks * 5; // wave-instructions
    const double per_simd_ns = winstr / (ms * 1e-3) / 1e9 / (cus * 4);
    printf("%-34s %d waves/SIMD %8.3f ms  %.3f per SIMD per ns  (%.2f ns each)\n", name, per_cu, ms / 5, per_simd_ns, 1. / per_simd_ns);
    return 0;
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    float* d_out;
    CHECK(hipMalloc(&d_out, sizeof(float) * 256 * p.multiProcessorCount * 8));
    const int cus = p.multiProcessorCount;
    for (int per_cu : {6, 1}) {
#define RUN(name, which, n) if (run(name, k<which>, d_out, cus, per_cu, n)) return 1
        RUN("v_sub_f32 v,v,v", 0, 1);
        RUN("v_cndmask vcc", 1, 1);
        RUN("v_cndmask s[20:21]", 2, 1);
        RUN("v_cmp -> vcc", 3, 1);
        RUN("v_cmp -> s[20:21]", 4, 1);
        RUN("v_cmp + v_cndmask via vcc", 5, 2);
        RUN("v_cmp + v_cndmask via s[20:21]", 6, 2);
        RUN("v_add_f32 literal", 7, 1);
        RUN("v_add_f32 inline 1.0", 8, 1);
        RUN("v_fmaak_f32", 9, 1);
        RUN("v_sqrt_f32", 10, 1);
        RUN("v_rcp_f32", 11, 1);
        RUN("v_cvt_u32_f32", 12, 1);
        RUN("v_min3_f32", 13, 1);
        RUN("v_fma_f64", 14, 1);
        RUN("v_mul_f64", 15, 1);
        RUN("v_add_f64", 16, 1);
        RUN("v_mad_u64_u32 (s operand)", 17, 1);
        RUN("v_rcp_f64", 18, 1);
        RUN("v_mad_u64_u32 (v operands)", 19, 1);
        RUN("v_xor_b32 s operand", 20, 1);
        RUN("v_xor_b32 v operands", 21, 1);
        RUN("v_mul_hi_u32", 22, 1);
        RUN("v_lshl_add_u32 s operand", 23, 1);
        RUN("v_cmp_lt_u64 -> vcc", 24, 1);
    }
    CHECK(hipFree(d_out));
    return 0;
}
