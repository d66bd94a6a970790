// ubench_valu.hip — per-instruction VALU issue rates on gfx950 (inline asm so the compiler cannot fold anything).
// Decides the arithmetic representation of the Poseidon2 kernels (int32 Montgomery vs fp64).  Build:
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o tools/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(x) x x x x x x x x
constexpr int ITERS = 512;

#define KERNEL32(NAME, ASM)                                                                   \
    __global__ void NAME(uint32_t* out) {                                                     \
        uint32_t r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7; \
        uint32_t y = blockIdx.x | 3, z = 0x78000001u;                                          \
        for (int i = 0; i < ITERS; i++) {                                                     \
            REP8(asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)          \
                              : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(y), "v"(z) : "vcc");) \
        }                                                                                     \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;  \
    }
#define A_ADD(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define A_MIN(i) "v_min_u32 %" #i ", %" #i ", %8\n"
#define A_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define A_MULHI(i) "v_mul_hi_u32 %" #i ", %" #i ", %8\n"
#define A_ADD3(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define A_MIN3(i) "v_min3_u32 %" #i ", %" #i ", %8, %9\n"
#define A_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 3, %8\n"
#define A_FMA32(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define A_MUL24(i) "v_mul_u32_u24 %" #i ", %" #i ", %8\n"
#define A_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define A_XOR(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define A_SUB(i) "v_sub_u32 %" #i ", %" #i ", %8\n"
#define A_SUBCO(i) "v_sub_co_u32 %" #i ", vcc, %" #i ", %8\n"
#define A_CND(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define A_SUBCND(i) "v_sub_co_u32 %" #i ", vcc, %" #i ", %8\nv_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"
#define A_ASHR(i) "v_ashrrev_i32 %" #i ", 31, %" #i "\n"
#define A_AND(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define A_MULHII(i) "v_mul_hi_i32 %" #i ", %" #i ", %8\n"
#define A_MAXI(i) "v_max_i32 %" #i ", %" #i ", %8\n"
#define A_LSHL(i) "v_lshlrev_b32 %" #i ", 3, %" #i "\n"
#define A_BFI(i) "v_bfi_b32 %" #i ", %" #i ", %8, %9\n"
#define A_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9\n"
KERNEL32(k_sub, A_SUB) KERNEL32(k_subco, A_SUBCO) KERNEL32(k_cnd, A_CND) KERNEL32(k_subcnd, A_SUBCND) KERNEL32(k_ashr, A_ASHR)
KERNEL32(k_and, A_AND) KERNEL32(k_mulhii, A_MULHII) KERNEL32(k_maxi, A_MAXI) KERNEL32(k_lshl, A_LSHL) KERNEL32(k_bfi, A_BFI) KERNEL32(k_andor, A_ANDOR)
KERNEL32(k_add, A_ADD) KERNEL32(k_min, A_MIN) KERNEL32(k_mullo, A_MULLO) KERNEL32(k_mulhi, A_MULHI)
KERNEL32(k_add3, A_ADD3) KERNEL32(k_min3, A_MIN3) KERNEL32(k_lshladd, A_LSHLADD) KERNEL32(k_fma32, A_FMA32)
KERNEL32(k_mul24, A_MUL24) KERNEL32(k_mad24, A_MAD24) KERNEL32(k_xor, A_XOR)

#define KERNEL64(NAME, ASM)                                                                   \
    __global__ void NAME(uint32_t* out) {                                                     \
        double r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7; \
        double y = 1.0000001 + blockIdx.x * 1e-9, z = 0.5;                                     \
        for (int i = 0; i < ITERS; i++) {                                                     \
            REP8(asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)          \
                              : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(y), "v"(z));) \
        }                                                                                     \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7); \
    }
#define D_ADD(i) "v_add_f64 %" #i ", %" #i ", %8\n"
#define D_MUL(i) "v_mul_f64 %" #i ", %" #i ", %8\n"
#define D_FMA(i) "v_fma_f64 %" #i ", %" #i ", %8, %9\n"
#define D_RND(i) "v_rndne_f64 %" #i ", %" #i "\n"
#define D_PKFMA(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define D_PKADD(i) "v_pk_add_f32 %" #i ", %" #i ", %8\n"
#define D_MAD64(i) "v_mad_u64_u32 %" #i ", vcc, %8, %8, %" #i "\n"
KERNEL64(k_dadd, D_ADD) KERNEL64(k_dmul, D_MUL) KERNEL64(k_dfma, D_FMA) KERNEL64(k_drnd, D_RND)
KERNEL64(k_pkfma, D_PKFMA) KERNEL64(k_pkadd, D_PKADD)

__global__ void k_mad64(uint32_t* out) {
    uint64_t r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
    uint32_t y = blockIdx.x | 3;
    for (int i = 0; i < ITERS; i++) {
        REP8(asm volatile("v_mad_u64_u32 %0, vcc, %8, %8, %0\nv_mad_u64_u32 %1, vcc, %8, %8, %1\nv_mad_u64_u32 %2, vcc, %8, %8, %2\n"
                          "v_mad_u64_u32 %3, vcc, %8, %8, %3\nv_mad_u64_u32 %4, vcc, %8, %8, %4\nv_mad_u64_u32 %5, vcc, %8, %8, %5\n"
                          "v_mad_u64_u32 %6, vcc, %8, %8, %6\nv_mad_u64_u32 %7, vcc, %8, %8, %7\n"
                          : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(y) : "vcc");)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7);
}

template <typename K> void run(const char* name, K kern, uint32_t* d) {
    const int blocks = 256 * 8, threads = 256;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    kern<<<blocks, threads>>>(d); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int r = 0; r < 5; r++) kern<<<blocks, threads>>>(d);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double ops = 5.0 * blocks * threads * 64.0 * ITERS;
    printf("%-16s %8.3f ms   %7.2f T lane-instr/s   (%.1f lanes/clk/SIMD @2.4GHz)\n", name, ms / 5, ops / (ms * 1e-3) / 1e12,
           ops / (ms * 1e-3) / (256.0 * 4 * 2.4e9));
}
int main() {
    uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run("v_fma_f32", k_fma32, d); run("v_pk_fma_f32", k_pkfma, d); run("v_pk_add_f32", k_pkadd, d);
    run("v_add_u32", k_add, d); run("v_min_u32", k_min, d); run("v_xor_b32", k_xor, d);
    run("v_sub_u32", k_sub, d); run("v_sub_co_u32", k_subco, d); run("v_cndmask_b32", k_cnd, d); run("sub_co+cndmask(x2)", k_subcnd, d);
    run("v_ashrrev_i32", k_ashr, d); run("v_and_b32", k_and, d); run("v_lshlrev_b32", k_lshl, d); run("v_bfi_b32", k_bfi, d); run("v_and_or_b32", k_andor, d);
    run("v_mul_hi_i32", k_mulhii, d); run("v_max_i32", k_maxi, d);
    run("v_add3_u32", k_add3, d); run("v_min3_u32", k_min3, d); run("v_lshl_add_u32", k_lshladd, d);
    run("v_mul_u32_u24", k_mul24, d); run("v_mad_u32_u24", k_mad24, d);
    run("v_mul_lo_u32", k_mullo, d); run("v_mul_hi_u32", k_mulhi, d); run("v_mad_u64_u32", k_mad64, d);
    run("v_add_f64", k_dadd, d); run("v_mul_f64", k_dmul, d); run("v_fma_f64", k_dfma, d); run("v_rndne_f64", k_drnd, d);
    return 0;
}
