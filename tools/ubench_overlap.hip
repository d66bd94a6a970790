// ubench_overlap.hip — does the dispatcher pair a VALU-bound kernel with an HBM-bound one when told which to prefer?
//
// The seal pipeline overlaps three seals on three streams: 27.6 ms of kernel time per seal when alone become 23.4 ms per seal,
// because the VALU-bound Poseidon2 row hash of one seal sometimes shares the chip with the HBM-bound NTT passes of another
// ("untried on the seal itself").  Before any stream of the prover is split by kernel class, this measures the
// ceiling on a stand-in: A = an integer multiply-add chain per lane (no memory traffic: hash_rows' shape), B = a streaming
// read-modify-write over 1 GiB (the NTT passes' shape), each sized to ~2 ms alone, K of each per stream:
//     A then B on ONE stream (no overlap)          | A and B on two streams of equal priority
//     B's stream at the HIGHEST priority           | A's stream at the highest priority
//     A with a reduced register / occupancy share (launch bounds) so that B's waves always find room
// If the paired time approaches max(A, B) only with a priority (or only with the occupancy share), the prover's lanes should run
// their NTT passes and their hashing on two streams; if equal priorities already get there, nothing is left to take.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_overlap.hip -o tools/ubench_overlap && tools/ubench_overlap
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// A: a dependent multiply-add chain (v_mad_u64_u32 + v_mul_lo_u32, the opcodes of a Montgomery product), 4 independent chains per lane
template <int BOUNDS>
__global__ __launch_bounds__(256, BOUNDS) void k_valu(uint32_t* out, uint32_t iters) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t a = i | 1u, b = i * 3u + 7u, c = i * 5u + 11u, d = i * 7u + 13u;
    const uint32_t P = 2013265921u, M = 0x88000001u;
    for (uint32_t k = 0; k < iters; k++) {
        a = (a * (uint32_t)b + (uint64_t)((uint32_t)a * M) * P) >> 32;
        b = (b * (uint32_t)c + (uint64_t)((uint32_t)b * M) * P) >> 32;
        c = (c * (uint32_t)d + (uint64_t)((uint32_t)c * M) * P) >> 32;
        d = (d * (uint32_t)a + (uint64_t)((uint32_t)d * M) * P) >> 32;
    }
    out[i] = (uint32_t)(a ^ b ^ c ^ d);
}
// B: streaming read-modify-write, 16 bytes per lane per step
__global__ __launch_bounds__(256) void k_stream(uint4* io, size_t n4) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) { uint4 v = io[i]; v.x += 1; io[i] = v; }
}

static float run(hipStream_t sa, hipStream_t sb, int K, int bounds, uint32_t* outA, uint32_t lanesA, uint32_t iters, uint4* bufB, size_t n4, bool doA, bool doB) {
    hipEvent_t e0, ea, eb;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, sa));
    if (sb != sa) CK(hipStreamWaitEvent(sb, e0, 0));
    for (int k = 0; k < K; k++) {
        if (doA) {
            if (bounds == 1) k_valu<1><<<lanesA / 256, 256, 0, sa>>>(outA, iters);
            else if (bounds == 2) k_valu<2><<<lanesA / 256, 256, 0, sa>>>(outA, iters);
            else k_valu<8><<<lanesA / 256, 256, 0, sa>>>(outA, iters);
        }
        if (doB) k_stream<<<256 * 32, 256, 0, sb>>>(bufB, n4);
    }
    CK(hipEventRecord(ea, sa));
    CK(hipEventRecord(eb, sb));
    CK(hipEventSynchronize(ea)); CK(hipEventSynchronize(eb));
    float ta = 0, tb = 0;
    CK(hipEventElapsedTime(&ta, e0, ea)); CK(hipEventElapsedTime(&tb, e0, eb));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(ea)); CK(hipEventDestroy(eb));
    return ta > tb ? ta : tb;
}

int main() {
    const size_t bytesB = (size_t)1 << 30, n4 = bytesB / 16;
    const uint32_t lanesA = 256u * 256u * 16u;            // 16 workgroups of 256 lanes per CU
    uint32_t* outA = nullptr;
    uint4* bufB = nullptr;
    CK(hipMalloc((void**)&outA, (size_t)lanesA * 4));
    CK(hipMalloc((void**)&bufB, bytesB));
    CK(hipMemset(bufB, 0, bytesB));
    int lo = 0, hi = 0;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));        // lo = lowest priority (largest number), hi = highest
    hipStream_t s_norm_a, s_norm_b, s_high, s_low;
    CK(hipStreamCreateWithFlags(&s_norm_a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s_norm_b, hipStreamNonBlocking));
    CK(hipStreamCreateWithPriority(&s_high, hipStreamNonBlocking, hi));
    CK(hipStreamCreateWithPriority(&s_low, hipStreamNonBlocking, lo));
    // size A to about B's duration
    const int K = 8;
    run(s_norm_a, s_norm_a, 2, 8, outA, lanesA, 2000, bufB, n4, true, true);                       // warm-up
    const float tb = run(s_norm_a, s_norm_a, K, 8, outA, lanesA, 0, bufB, n4, false, true) / K;
    uint32_t iters = 2000;
    float ta = run(s_norm_a, s_norm_a, K, 8, outA, lanesA, iters, bufB, n4, true, false) / K;
    iters = (uint32_t)(iters * tb / ta);
    ta = run(s_norm_a, s_norm_a, K, 8, outA, lanesA, iters, bufB, n4, true, false) / K;
    printf("{\"A_valu_ms\": %.3f, \"B_stream_ms\": %.3f, \"B_GBps\": %.0f, \"iters\": %u, \"priority_range\": [%d, %d]}\n", ta, tb, 2.0 * bytesB / tb / 1e6, iters, lo, hi);
    struct Case { const char* name; hipStream_t sa, sb; int bounds; } cases[] = {
        {"one stream (serial)", s_norm_a, s_norm_a, 8},
        {"two streams, equal priority", s_norm_a, s_norm_b, 8},
        {"B (stream) high priority", s_norm_a, s_high, 8},
        {"A (valu) high priority", s_high, s_norm_b, 8},
        {"B high, A low", s_low, s_high, 8},
        {"two streams, A at 2 workgroups per SIMD", s_norm_a, s_norm_b, 2},
        {"two streams, A at 1 workgroup per SIMD", s_norm_a, s_norm_b, 1},
        {"B high, A at 2 workgroups per SIMD", s_norm_a, s_high, 2},
    };
    for (const Case& c : cases) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; rep++) { const float t = run(c.sa, c.sb, K, c.bounds, outA, lanesA, iters, bufB, n4, true, true) / K; if (t < best) best = t; }
        printf("{\"case\": \"%s\", \"pair_ms\": %.3f, \"vs_serial_sum\": %.3f, \"vs_max\": %.3f}\n", c.name, best, best / (ta + tb), best / (ta > tb ? ta : tb));
    }
    return 0;
}
