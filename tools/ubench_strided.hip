// ubench_strided.hip — memory-side ceiling of the strided NTT pass's access pattern on gfx950, with NO arithmetic:
// every workgroup reads a tile of 2^RH rows x 2^LOGT consecutive words (row stride 2^L words) of one column into
// registers and writes it back (+1), exactly the traffic of k_ntt_high (ntt.hip), for several tile shapes with
// RH + LOGT = 14 (64 KiB of data per 1024-lane workgroup, 16 words per lane).  Tells how much of the strided pass's time
// is the access pattern itself (64-B runs at a 16-KiB stride vs 256-B runs at a 64-KiB stride) before any kernel is rewritten.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_strided.hip -o tools/ubench_strided && tools/ubench_strided
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ uint32_t xcd_remap(uint32_t b, uint32_t nblocks) {
    const uint32_t xcd = b & 7u, q = nblocks >> 3, idx = b >> 3;
    return xcd * q + idx;
}

template <int RH, int LOGT, bool REMAP, int THREADS = 1024>
__global__ __launch_bounds__(THREADS) void k_tile(uint32_t* io, uint32_t log_n, uint32_t lds_dummy) {
    extern __shared__ uint32_t lds[];
    constexpr int T = 1 << LOGT, GROUPS = THREADS >> LOGT, PER = (1 << RH) / GROUPS;
    const uint32_t L = log_n - RH;
    const uint32_t tid = threadIdx.x, t = tid & (T - 1), g = tid >> LOGT;
    const uint32_t tiles = 1u << (L - LOGT);
    const uint32_t tile = REMAP ? xcd_remap(blockIdx.x, tiles) : blockIdx.x;
    uint32_t* col = io + ((size_t)blockIdx.y << log_n) + ((size_t)tile << LOGT) + t;
    uint32_t v[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) v[k] = col[(size_t)(k * GROUPS + g) << L];
    if (lds_dummy == 12345u) lds[tid] = v[0];          // keeps the dynamic LDS allocation alive
#pragma unroll
    for (int k = 0; k < PER; k++) col[(size_t)(k * GROUPS + g) << L] = v[k] + 1;
}

__global__ void k_copy(uint4* io, size_t n4) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) { uint4 v = io[i]; v.x += 1; io[i] = v; }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int RH, int LOGT, bool REMAP, int THREADS = 1024>
void run(uint32_t* d, uint32_t log_n, uint32_t cols, size_t lds_bytes, const char* what) {
    CK(hipFuncSetAttribute((const void*)k_tile<RH, LOGT, REMAP, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    dim3 grid(1u << (log_n - RH - LOGT), cols);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    k_tile<RH, LOGT, REMAP, THREADS><<<grid, THREADS, lds_bytes>>>(d, log_n, 0);
    CK(hipDeviceSynchronize());
    const int reps = 5;
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) k_tile<RH, LOGT, REMAP, THREADS><<<grid, THREADS, lds_bytes>>>(d, log_n, 0);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= reps;
    const double bytes = 8.0 * cols * (double)((size_t)1 << log_n);
    printf("{\"tile_rows\": %d, \"run_bytes\": %d, \"lds_KiB\": %zu, \"xcd_remap\": %s, \"ms\": %.3f, \"TBps\": %.2f, \"what\": \"%s\"}\n",
           1 << RH, 4 << LOGT, lds_bytes / 1024, REMAP ? "true" : "false", ms, bytes / ms / 1e9, what);
}

int main(int argc, char** argv) {
    const uint32_t log_n = 22, cols = argc > 1 ? atoi(argv[1]) : 208;
    const size_t words = (size_t)cols << log_n;
    uint32_t* d; CK(hipMalloc((void**)&d, words * 4)); CK(hipMemset(d, 1, words * 4));
    {
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        k_copy<<<256 * 16, 256>>>((uint4*)d, words / 4); CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        for (int i = 0; i < 5; i++) k_copy<<<256 * 16, 256>>>((uint4*)d, words / 4);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
        printf("{\"what\": \"contiguous read+write (16 B per lane)\", \"ms\": %.3f, \"TBps\": %.2f}\n", ms, 8.0 * words / ms / 1e9);
    }
    run<10, 4, true>(d, log_n, cols, 64 << 10, "k_ntt_high<10> shape: 1024 rows x 64 B, 2 workgroups per CU");
    run<10, 4, false>(d, log_n, cols, 64 << 10, "same without the XCD remap");
    run<10, 4, true>(d, log_n, cols, 0, "same, no LDS (occupancy limited by waves only)");
    run<9, 5, true>(d, log_n, cols, 64 << 10, "512 rows x 128 B");
    run<8, 6, true>(d, log_n, cols, 64 << 10, "256 rows x 256 B (a 14 + 8 split of 2^22)");
    run<8, 6, false>(d, log_n, cols, 64 << 10, "256 rows x 256 B without the XCD remap");
    run<7, 7, true>(d, log_n, cols, 64 << 10, "128 rows x 512 B");
    run<6, 8, true>(d, log_n, cols, 64 << 10, "64 rows x 1 KiB");
    run<8, 6, true>(d, log_n, cols, 0, "256 rows x 256 B, no LDS");
    run<10, 3, true, 512>(d, log_n, cols, 32 << 10, "1024 rows x 32 B, 512 lanes, 32 KiB: four workgroups per CU");
    run<10, 2, true, 256>(d, log_n, cols, 16 << 10, "1024 rows x 16 B, 256 lanes, 16 KiB: eight workgroups per CU");
    run<10, 4, true, 512>(d, log_n, cols, 64 << 10, "1024 rows x 64 B, 512 lanes x 32 words");
    return 0;
}
