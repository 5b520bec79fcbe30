// Microbenchmark: how does HBM throughput on MI355X depend on the NUMBER of concurrent column
// streams and on the per-lane access width, for the access shape of the tick kernel
// (R u64 loads + W u64 stores per group)?  Layouts: column-major [col][N] ("soa") vs tiles of T
// groups with all columns of a tile adjacent ([tile][col][T], "tiled").
// build: hipcc -O3 --offload-arch=gfx950 streams.hip -o streams ; run: ./streams
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint64_t u64;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int R, int W, int BLOCK> __global__ __launch_bounds__(BLOCK) void k_soa(const u64 *in, u64 *out, u64 N) {
    const u64 g = (u64)blockIdx.x * BLOCK + threadIdx.x;
    if (g >= N) return;
    u64 v[R];
#pragma unroll
    for (int c = 0; c < R; c++) v[c] = in[(u64)c * N + g];
    u64 acc = 0;
#pragma unroll
    for (int c = 0; c < R; c++) acc += v[c];
#pragma unroll
    for (int c = 0; c < W; c++) out[(u64)c * N + g] = acc + v[c % R];
}
template <int R, int W, int BLOCK> __global__ __launch_bounds__(BLOCK) void k_soa2(const u64 *in, u64 *out, u64 N) {
    const u64 g = ((u64)blockIdx.x * BLOCK + threadIdx.x) * 2;
    if (g >= N) return;
    u64x2 v[R];
#pragma unroll
    for (int c = 0; c < R; c++) v[c] = *(const u64x2 *)(in + (u64)c * N + g);
    u64x2 acc = 0;
#pragma unroll
    for (int c = 0; c < R; c++) acc += v[c];
#pragma unroll
    for (int c = 0; c < W; c++) *(u64x2 *)(out + (u64)c * N + g) = acc + v[c % R];
}
// tiled: tile t holds groups [t*T, (t+1)*T): in-tile layout [col][T]; reads from `in` tiles of R cols,
// writes to `out` tiles of W cols
template <int R, int W, int BLOCK> __global__ __launch_bounds__(BLOCK) void k_tiled(const u64 *in, u64 *out, u64 N) {
    const u64 t = blockIdx.x;
    const u64 g = t * BLOCK + threadIdx.x;
    if (g >= N) return;
    const u64 *ib = in + t * (u64)R * BLOCK + threadIdx.x;
    u64 *ob = out + t * (u64)W * BLOCK + threadIdx.x;
    u64 v[R];
#pragma unroll
    for (int c = 0; c < R; c++) v[c] = ib[c * BLOCK];
    u64 acc = 0;
#pragma unroll
    for (int c = 0; c < R; c++) acc += v[c];
#pragma unroll
    for (int c = 0; c < W; c++) ob[c * BLOCK] = acc + v[c % R];
}
// in-place variant of soa: W of the R columns are rewritten in place (like the tick kernel)
template <int R, int W, int BLOCK> __global__ __launch_bounds__(BLOCK) void k_soa_inplace(u64 *io, u64 N) {
    const u64 g = (u64)blockIdx.x * BLOCK + threadIdx.x;
    if (g >= N) return;
    u64 v[R];
#pragma unroll
    for (int c = 0; c < R; c++) v[c] = io[(u64)c * N + g];
    u64 acc = 0;
#pragma unroll
    for (int c = 0; c < R; c++) acc += v[c];
#pragma unroll
    for (int c = 0; c < W; c++) io[(u64)c * N + g] = acc + v[c];
}

template <typename F> float time_it(F f, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); f();
    hipEventRecord(a);
    for (int i = 0; i < iters; i++) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / iters;
}

int main() {
    const u64 N = 8u << 20; // groups
    constexpr int R = 27, W = 17;
    u64 *in, *out;
    CHECK(hipMalloc(&in, N * R * 8)); CHECK(hipMalloc(&out, N * R * 8));
    CHECK(hipMemset(in, 1, N * R * 8)); CHECK(hipMemset(out, 0, N * R * 8));
    const double bytes = (double)N * (R + W) * 8;
    auto rep = [&](const char *name, float ms) { printf("%-40s %8.1f us  %7.1f GB/s\n", name, ms * 1e3, bytes / ms / 1e6); };
    rep("soa  8B/lane block256 R27 W17", time_it([&] { hipLaunchKernelGGL((k_soa<R, W, 256>), dim3(N / 256), dim3(256), 0, 0, in, out, N); }, 10));
    rep("soa  8B/lane block64  R27 W17", time_it([&] { hipLaunchKernelGGL((k_soa<R, W, 64>), dim3(N / 64), dim3(64), 0, 0, in, out, N); }, 10));
    rep("soa 16B/lane block256 R27 W17", time_it([&] { hipLaunchKernelGGL((k_soa2<R, W, 256>), dim3(N / 512), dim3(256), 0, 0, in, out, N); }, 10));
    rep("soa 16B/lane block64  R27 W17", time_it([&] { hipLaunchKernelGGL((k_soa2<R, W, 64>), dim3(N / 128), dim3(64), 0, 0, in, out, N); }, 10));
    rep("tiled 8B/lane T=256 R27 W17", time_it([&] { hipLaunchKernelGGL((k_tiled<R, W, 256>), dim3(N / 256), dim3(256), 0, 0, in, out, N); }, 10));
    rep("tiled 8B/lane T=64  R27 W17", time_it([&] { hipLaunchKernelGGL((k_tiled<R, W, 64>), dim3(N / 64), dim3(64), 0, 0, in, out, N); }, 10));
    rep("soa in-place 8B block256 R27 W17", time_it([&] { hipLaunchKernelGGL((k_soa_inplace<R, W, 256>), dim3(N / 256), dim3(256), 0, 0, in, N); }, 10));
    rep("soa in-place 8B block64  R27 W17", time_it([&] { hipLaunchKernelGGL((k_soa_inplace<R, W, 64>), dim3(N / 64), dim3(64), 0, 0, in, N); }, 10));
    // reference point: plain 2-stream copy, 16 B/lane
    {
        const double cb = (double)N * R * 8 * 2;
        float ms = time_it([&] { hipMemcpyAsync(out, in, N * R * 8, hipMemcpyDeviceToDevice, 0); }, 5);
        printf("%-40s %8.1f us  %7.1f GB/s\n", "hipMemcpy D2D (read+write)", ms * 1e3, cb / ms / 1e6);
    }
    return 0;
}
