// ubench_intmul.hip — gfx950 integer-multiply throughput probe for the BabyBear Montgomery product.
// SURVEY.md §7 hard part 4: the int-multiply rates are not in the local guides, so they are measured.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_intmul.hip -o tools/ubench_intmul
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../zeth_amd/csrc/fp.h"
using namespace zkh;

constexpr int CH = 8, ITERS = 2048;

template <int V> __device__ __forceinline__ uint32_t op(uint32_t a, uint32_t b) {
    if (V == 0) return a * b;                                   // v_mul_lo_u32
    if (V == 1) return __umulhi(a, b);                          // v_mul_hi_u32
    if (V == 2) { uint64_t t = (uint64_t)a * b + a; return (uint32_t)t ^ (uint32_t)(t >> 32); }   // v_mad_u64_u32
    if (V == 3) return mul_mod(a, b);                           // shipped Montgomery product
    if (V == 4) {                                               // m by shift-adds instead of v_mul_lo
        uint64_t t = (uint64_t)a * b;
        uint32_t lo = (uint32_t)t, m = lo + (lo << 27) + (lo << 31);
        uint32_t u = __umulhi(m, P), hi = (uint32_t)(t >> 32), r = hi - u, r2 = r + P;
        return r2 < r ? r2 : r;
    }
    if (V == 5) return add_mod(a, b);
    if (V == 6) return __umul24(a, b);                          // v_mul_u32_u24
    if (V == 7) { float x = __uint_as_float(a), y = __uint_as_float(b); return __float_as_uint(fmaf(x, y, x)); }
    if (V == 8) {                                               // u via the special form of P: m*P = (m<<31) - (m<<27) + m
        uint64_t t = (uint64_t)a * b;
        uint32_t lo = (uint32_t)t, m = lo + (lo << 27) + (lo << 31);
        uint64_t mp = ((uint64_t)m << 31) - ((uint64_t)m << 27) + m;
        uint32_t u = (uint32_t)(mp >> 32), hi = (uint32_t)(t >> 32), r = hi - u, r2 = r + P;
        return r2 < r ? r2 : r;
    }
    return 0;
}
template <int V> __global__ void k(uint32_t* out, uint32_t seed) {
    uint32_t x[CH], y = seed | 1;
    for (int c = 0; c < CH; c++) x[c] = (threadIdx.x * 2654435761u + c * 40503u + seed) % P;
    for (int i = 0; i < ITERS; i++) {
#pragma unroll
        for (int c = 0; c < CH; c++) x[c] = op<V>(x[c], y + c);
    }
    uint32_t s = 0;
    for (int c = 0; c < CH; c++) s ^= x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int V> void run(const char* name, uint32_t* d) {
    const int blocks = 256 * 16, threads = 256;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<V><<<blocks, threads>>>(d, 12345); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; r++) k<V><<<blocks, threads>>>(d, 12345 + r);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double ops = 5.0 * blocks * threads * (double)CH * ITERS;
    printf("%-28s %8.3f ms  %8.2f Gop/s/lane-total  (%.2f Tops/s)\n", name, ms / 5, ops / (ms * 1e-3) / 1e9, ops / (ms * 1e-3) / 1e12);
}
int main() {
    uint32_t* d; hipMalloc(&d, 256 * 16 * 256 * 4);
    run<7>("v_fma_f32 (ref)", d);
    run<5>("add_mod", d);
    run<6>("v_mul_u32_u24", d);
    run<0>("v_mul_lo_u32", d);
    run<1>("v_mul_hi_u32", d);
    run<2>("v_mad_u64_u32", d);
    run<3>("mul_mod (shipped)", d);
    run<4>("mul_mod m=shift-add", d);
    run<8>("mul_mod all shift-add", d);
    return 0;
}
