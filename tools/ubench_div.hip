// Check on the GPU: the quotient formed from a divisor's refined reciprocal (div_by in csrc/ref_device.h) against the
// compiler's own f64 division, bit for bit, over operands across the exponent band that form is used in -- random
// mantissas, mantissas of all ones and of all zeros, divisors next to powers of two -- and its cost.  Prints the
// number of mismatches (must be 0) and the time of each form.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t z) { // splitmix64
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ bool div_band(double x) {
    const uint32_t e = ((uint32_t)__double2hiint(x) >> 20) & 0x7ffu;
    return e - 700u <= 650u;
}
__device__ __forceinline__ double refined_rcp(double s) {
    const double r0 = __builtin_amdgcn_rcp(s);
    const double e0 = __builtin_fma(-s, r0, 1.0);
    const double r1 = __builtin_fma(r0, e0, r0);
    const double e1 = __builtin_fma(-s, r1, 1.0);
    return __builtin_fma(r1, e1, r1);
}
__device__ __forceinline__ double div_by(double x, double s, double y) {
    const double q0 = x * y;
    const double e = __builtin_fma(-s, q0, x);
    return __builtin_fma(e, y, q0);
}
// an operand of the band: sign, exponent 700..1350, and a mantissa of one of a few kinds
__device__ __forceinline__ double operand(uint64_t r, int kind) {
    const uint64_t sign = (r >> 63) << 63;
    const uint64_t expo = 700ull + (mix(r ^ 0x55) % 651ull);
    uint64_t man = r & 0xFFFFFFFFFFFFFull;
    switch (kind & 7) {
        case 1: man = 0xFFFFFFFFFFFFFull; break;                 // all ones
        case 2: man = 0ull; break;                               // a power of two
        case 3: man = 0xFFFFFFFFFFFFFull - (r & 0xFFull); break; // just below a power of two
        case 4: man = r & 0xFFull; break;                        // just above a power of two
        case 5: man &= 0xFFFFFF0000000ull; break;                // 24 significant bits (a Philox uniform)
        default: break;
    }
    return __longlong_as_double((long long)(sign | (expo << 52) | man));
}
__global__ __launch_bounds__(256) void check(unsigned long long* mismatches, unsigned long long* first, int rounds, uint64_t seed, int near_exp) {
    uint64_t st = seed ^ ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 0xD6E8FEB86659FD93ull;
    unsigned long long bad = 0;
    for (int i = 0; i < rounds; ++i) {
        st = mix(st);
        const uint64_t a = st, b = mix(st ^ 0xABCDEF);
        double s = operand(a, (int)(b >> 3)), x = operand(b, (int)(b >> 7));
        if (near_exp) { // exponents close together: quotients around 1, where most divisions of the render are
            const uint64_t xs = (uint64_t)__double_as_longlong(x), ss = (uint64_t)__double_as_longlong(s);
            x = __longlong_as_double((long long)((xs & 0x800FFFFFFFFFFFFFull) | (ss & 0x7FF0000000000000ull)));
        }
        if (!(div_band(s) && div_band(x))) continue;
        const double q = x / s, f = div_by(x, s, refined_rcp(s));
        if (__double_as_longlong(q) != __double_as_longlong(f)) {
            if (bad == 0 && atomicAdd(mismatches + 1, 1ull) == 0ull) { first[0] = (unsigned long long)__double_as_longlong(x); first[1] = (unsigned long long)__double_as_longlong(s); }
            ++bad;
        }
    }
    if (bad) atomicAdd(mismatches, bad);
}
template <int WHICH>
__global__ __launch_bounds__(256) void cost(double* out, int rounds) {
    double s = 1.0 + threadIdx.x * 1e-3, x0 = 3.0 + blockIdx.x, x1 = 5.0, x2 = 7.0, acc = 0.;
    for (int i = 0; i < rounds; ++i) {
        if (WHICH == 0) { acc += x0 / s + x1 / s + x2 / s; }
        else { const double y = refined_rcp(s); acc += div_by(x0, s, y) + div_by(x1, s, y) + div_by(x2, s, y); }
        s += 1e-9; x0 += 1e-7; x1 += acc * 1e-30; x2 -= 1e-8;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
    unsigned long long *d_bad, *d_first, h_bad[2] = {0, 0}, h_first[2] = {0, 0};
    CHECK(hipMalloc(&d_bad, 16)); CHECK(hipMalloc(&d_first, 16));
    CHECK(hipMemset(d_bad, 0, 16)); CHECK(hipMemset(d_first, 0, 16));
    const int blocks = 2048, rounds = 2048; // 2^30 pairs per launch
    for (int pass = 0; pass < 4; ++pass) {
        hipLaunchKernelGGL(check, dim3(blocks), dim3(256), 0, 0, d_bad, d_first, rounds, 0x1234567ull + pass, pass & 1);
        CHECK(hipDeviceSynchronize());
    }
    CHECK(hipMemcpy(h_bad, d_bad, 16, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(h_first, d_first, 16, hipMemcpyDeviceToHost));
    printf("pairs checked: %.3g, mismatches: %llu", 4.0 * blocks * 256.0 * rounds, h_bad[0]);
    if (h_bad[0]) printf(" (first: x = %016llx, s = %016llx)", h_first[0], h_first[1]);
    printf("\n");
    double* d_out;
    CHECK(hipMalloc(&d_out, sizeof(double) * 1024 * 256));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int which = 0; which < 2; ++which) {
        float ms = 0.f;
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipEventRecord(e0, 0));
            if (which == 0) hipLaunchKernelGGL(cost<0>, dim3(1024), dim3(256), 0, 0, d_out, 4096);
            else hipLaunchKernelGGL(cost<1>, dim3(1024), dim3(256), 0, 0, d_out, 4096);
            CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("%s: %.3f ms for 1024 x 256 lanes x 4096 rounds of three quotients by one divisor\n", which == 0 ? "division" : "refined reciprocal", ms);
    }
    return h_bad[0] ? 2 : 0;
}
