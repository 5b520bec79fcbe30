// Microbenchmark (round 3): does the out-of-cache throughput of the tick's access shape depend on the NUMBER of
// concurrent column streams?  Constant bytes per launch (~2.95 GB, far beyond the 256 MB Infinity Cache), the same
// read : write ratio as the tick kernel (27 : 17), one lane = one group; what varies is how many separate column
// streams carry those bytes and how wide a lane's access to each is:
//     R27/W17 x  8 B cells  (the engine's layout: one u64 column per field and slot)
//     R14/W9  x 16 B cells  (fields paired: match|pr_commit, m_index|m_commit as u64x2 columns)
//     R7/W5   x 32 B cells  (four fields per cell, two dwordx4 accesses per lane and stream)
// plus the same stream counts at 8 B (fewer bytes per group, more groups) to separate "streams" from "cell width",
// in-place variants (the tick rewrites columns it has read), and a plain two-stream copy as the chip's ceiling.
// build: hipcc -O3 --offload-arch=gfx950 stream_sweep.hip -o stream_sweep ; run: ./stream_sweep
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef uint64_t u64;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int W64> struct Cell { u64 v[W64]; };
template <int W64> __device__ __forceinline__ Cell<W64> ld(const u64 *p) {
    Cell<W64> c;
    if constexpr (W64 == 1) c.v[0] = *p;
    else {
        typedef u64 u64x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int i = 0; i < W64; i += 2) {
            const u64x2 t = *reinterpret_cast<const u64x2 *>(p + i);
            c.v[i] = t.x;
            c.v[i + 1] = t.y;
        }
    }
    return c;
}
template <int W64> __device__ __forceinline__ void st(u64 *p, const Cell<W64> &c) {
    if constexpr (W64 == 1) *p = c.v[0];
    else {
        typedef u64 u64x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int i = 0; i < W64; i += 2) {
            u64x2 t;
            t.x = c.v[i];
            t.y = c.v[i + 1];
            *reinterpret_cast<u64x2 *>(p + i) = t;
        }
    }
}

// R read streams from `in`, W write streams to `out` (or, INPLACE, the first W of the R streams are rewritten);
// stream c is a column [N] of W64-word cells
template <int R, int W, int W64, int BLOCK, bool INPLACE>
__global__ __launch_bounds__(BLOCK) void k_streams(u64 *in, u64 *out, u64 N) {
    const u64 g = (u64)blockIdx.x * BLOCK + threadIdx.x;
    if (g >= N) return;
    Cell<W64> v[R];
#pragma unroll
    for (int c = 0; c < R; c++) v[c] = ld<W64>(in + ((u64)c * N + g) * W64);
    u64 acc = 0;
#pragma unroll
    for (int c = 0; c < R; c++)
#pragma unroll
        for (int i = 0; i < W64; i++) acc += v[c].v[i];
    u64 *dst = INPLACE ? in : out;
#pragma unroll
    for (int c = 0; c < W; c++) {
        Cell<W64> o = v[c % R];
        o.v[0] += acc;
        st<W64>(dst + ((u64)c * N + g) * W64, o);
    }
}

template <typename F> float time_it(F f, int iters) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    f();
    f();
    hipEventRecord(a);
    for (int i = 0; i < iters; i++) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms / iters;
}

static u64 *g_in, *g_out;
static const u64 TOTAL = (u64)8 << 20; // groups at R27/W17 x 8 B: 8 Mi x 44 x 8 B = 2.95 GB per launch

template <int R, int W, int W64, int BLOCK, bool INPLACE> void run(const char *what) {
    // same bytes per launch for every shape: N x (R + W) x 8 W64 = TOTAL x 44 x 8
    const u64 N = (TOTAL * 44 / ((u64)(R + W) * W64)) / BLOCK * BLOCK;
    const double bytes = (double)N * (R + W) * W64 * 8;
    float ms = time_it([&] { hipLaunchKernelGGL((k_streams<R, W, W64, BLOCK, INPLACE>), dim3((unsigned)(N / BLOCK)), dim3(BLOCK), 0, 0, g_in, g_out, N); }, 8);
    printf("%-26s R%-2d W%-2d cell %2d B  block %3d  %s  %9llu groups  %8.1f us  %7.1f GB/s\n", what, R, W, W64 * 8, BLOCK,
           INPLACE ? "in-place " : "out-of-pl", (unsigned long long)N, ms * 1e3, bytes / ms / 1e6);
}

int main() {
    const size_t bytes = (size_t)TOTAL * 27 * 8 + (1 << 20);
    CHECK(hipMalloc(&g_in, bytes));
    CHECK(hipMalloc(&g_out, bytes));
    CHECK(hipMemset(g_in, 1, bytes));
    CHECK(hipMemset(g_out, 0, bytes));
    CHECK(hipDeviceSynchronize());
    printf("# constant bytes per launch (2.95 GB), read : write = 27 : 17, one lane per group\n");
    run<27, 17, 1, 64, false>("engine layout");
    run<27, 17, 1, 256, false>("engine layout");
    run<14, 9, 2, 64, false>("paired fields");
    run<14, 9, 2, 256, false>("paired fields");
    run<7, 5, 4, 64, false>("four fields per cell");
    run<7, 5, 4, 256, false>("four fields per cell");
    run<14, 9, 1, 64, false>("fewer streams, 8 B");
    run<7, 5, 1, 64, false>("fewer streams, 8 B");
    run<7, 5, 1, 256, false>("fewer streams, 8 B");
    run<3, 2, 1, 256, false>("fewer streams, 8 B");
    run<3, 2, 2, 256, false>("few streams, 16 B");
    run<27, 17, 1, 64, true>("engine layout");
    run<14, 9, 2, 64, true>("paired fields");
    run<7, 5, 4, 64, true>("four fields per cell");
    run<1, 1, 2, 256, false>("plain copy 16 B");
    run<1, 1, 4, 256, false>("plain copy 32 B");
    run<2, 2, 2, 256, false>("2+2 copy 16 B");
    {
        const size_t cb = (size_t)TOTAL * 22 * 8;
        float ms = time_it([&] { hipMemcpyAsync(g_out, g_in, cb, hipMemcpyDeviceToDevice, 0); }, 5);
        printf("%-26s %8.1f us  %7.1f GB/s (read + write)\n", "hipMemcpy D2D", ms * 1e3, 2.0 * cb / ms / 1e6);
    }
    return 0;
}
