// Microbenchmark: the tick's access shape (27 u64 loads, 17 of those columns rewritten in place, one lane per group) with
// plain / non-temporal loads and stores, over engine sizes from inside the Infinity Cache to far beyond it. What the
// all-streamed regime of the tick (profiles/r04_nt_state.txt) can expect from the memory system alone, and whether its
// upper end (plain accesses faster again from 16 M groups) is a property of the kernel or of the machine.
// build: hipcc -O3 --offload-arch=gfx950 nt_shape.hip -o nt_shape ; run: ./nt_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint64_t u64;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <bool NT> __device__ __forceinline__ u64 ld(const u64 *p) { if constexpr (NT) return __builtin_nontemporal_load(p); else return *p; }
template <bool NT> __device__ __forceinline__ void st(u64 *p, u64 v) { if constexpr (NT) __builtin_nontemporal_store(v, p); else *p = v; }

template <int R, int W, bool NTL, bool NTS> __global__ __launch_bounds__(256) void k_inplace(u64 *io, u64 N) {
    const u64 g = (u64)blockIdx.x * 256 + threadIdx.x;
    if (g >= N) return;
    u64 v[R];
#pragma unroll
    for (int c = 0; c < R; c++) v[c] = ld<NTL>(io + (u64)c * N + g);
    u64 acc = 0;
#pragma unroll
    for (int c = 0; c < R; c++) acc += v[c];
#pragma unroll
    for (int c = 0; c < W; c++) st<NTS>(io + (u64)c * N + g, acc + v[c]);
}

template <typename F> float time_it(F f, int iters) {
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    f(); f(); f();
    (void)hipEventRecord(a);
    for (int i = 0; i < iters; i++) f();
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return ms / iters;
}

int main() {
    constexpr int R = 27, W = 17;
    const u64 sizes[] = {1u << 20, 2u << 20, 3u << 20, 4u << 20, 8u << 20, 12u << 20, 16u << 20, 24u << 20, 32u << 20};
    const u64 NMAX = 32u << 20;
    u64 *io;
    CHECK(hipMalloc(&io, NMAX * R * 8));
    CHECK(hipMemset(io, 1, NMAX * R * 8));
    printf("27 loads + 17 in-place stores of 8 B per group; GB/s by those bytes (state touched = 216 B per group)\n");
    printf("%10s %10s | %12s %12s %12s %12s\n", "groups", "MB touched", "plain/plain", "ntL/plain", "plain/ntS", "ntL/ntS");
    for (u64 N : sizes) {
        const double bytes = (double)N * (R + W) * 8;
        const int it = N <= (4u << 20) ? 40 : 12;
        dim3 grid((unsigned)((N + 255) / 256)), block(256);
        float a = time_it([&] { hipLaunchKernelGGL((k_inplace<R, W, false, false>), grid, block, 0, 0, io, N); }, it);
        float b = time_it([&] { hipLaunchKernelGGL((k_inplace<R, W, true, false>), grid, block, 0, 0, io, N); }, it);
        float c = time_it([&] { hipLaunchKernelGGL((k_inplace<R, W, false, true>), grid, block, 0, 0, io, N); }, it);
        float d = time_it([&] { hipLaunchKernelGGL((k_inplace<R, W, true, true>), grid, block, 0, 0, io, N); }, it);
        printf("%10llu %10.0f | %12.0f %12.0f %12.0f %12.0f\n", (unsigned long long)N, (double)N * R * 8 / 1048576.0, bytes / a / 1e6, bytes / b / 1e6,
               bytes / c / 1e6, bytes / d / 1e6);
    }
    return 0;
}
