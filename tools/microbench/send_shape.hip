// Microbenchmark (round 3): what does a TRIVIAL kernel reach with the access shape of the send stage -- the number of u64 and
// u32 column streams k_send_dense and k_tick_send read and write per group, one lane per group, 64-thread workgroups, the
// state streams rewritten in place -- at 8 M groups (HBM regime) and 1 M groups?  The ceiling the stage's achieved bytes/s
// should be compared with (profiles/r03_tick_send.txt), as stream_sweep.hip is for the tick.
//   k_send_dense (P = 5, 4 peers in the work set): reads out, cfg (u32) pflags, last_index, first_index (u64) + per peer
//     meta (u32), head, tail, next (u64) = 15 u64 + 6 u32; writes pflags + per peer meta (u32), head, tail, next + items
//     n (5 u32), prev, last (4 x 2 u64) = 21 u64 + 9 u32
//   k_tick_send: the tick's 27 u64 reads + per peer meta, head, tail + first_index = 36 u64 + 4 u32 reads; the tick's 17 u64
//     writes + per peer meta, head, tail + items = 33 u64 + 9 u32 writes
//   k_tick_lane for reference: 27 u64 reads, 17 u64 writes
// build: hipcc -O3 --offload-arch=gfx950 send_shape.hip -o send_shape ; run: ./send_shape
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef uint64_t u64;
typedef uint32_t u32;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// streams: R8 u64 columns + R4 u32 columns read; W8 u64 + W4 u32 written; the first min(R8, W8) u64 and min(R4, W4) u32
// streams are rewritten in place, the rest of the writes go to columns of their own
// MASK: one lane in MASK (hash-selected per group and stream) skips its store -- lane-masked partial lines, what a stage
// that only rewrites "the cells of the work set" produces; MASKLD: the same lanes skip the load as well
template <int R8, int R4, int W8, int W4, int BLOCK, int MASK = 0, bool MASKLD = false>
__global__ __launch_bounds__(BLOCK) void k_shape(u64 *c8, u32 *c4, u64 N) {
    const u64 g = (u64)blockIdx.x * BLOCK + threadIdx.x;
    if (g >= N) return;
    u64 v8[R8 > 0 ? R8 : 1];
    u32 v4[R4 > 0 ? R4 : 1];
    const u32 h = (u32)g * 2654435761u;
#pragma unroll
    for (int c = 0; c < R8; c++) {
        v8[c] = 0;
        if (!(MASK && MASKLD) || ((h >> (c & 15)) % (MASK ? MASK : 1)) != 0) v8[c] = c8[(u64)c * N + g];
    }
#pragma unroll
    for (int c = 0; c < R4; c++) {
        v4[c] = 0;
        if (!(MASK && MASKLD) || ((h >> (c & 15)) % (MASK ? MASK : 1)) != 0) v4[c] = c4[(u64)c * N + g];
    }
    u64 acc = 0;
#pragma unroll
    for (int c = 0; c < R8; c++) acc += v8[c];
#pragma unroll
    for (int c = 0; c < R4; c++) acc += v4[c];
    // (writes beyond the read streams land in the columns behind them)
#pragma unroll
    for (int c = 0; c < W8; c++)
        if (!MASK || ((h >> (c & 15)) % (MASK ? MASK : 1)) != 0) c8[(u64)c * N + g] = (c < R8 ? v8[c] : 0) + acc;
#pragma unroll
    for (int c = 0; c < W4; c++)
        if (!MASK || ((h >> (c & 15)) % (MASK ? MASK : 1)) != 0) c4[(u64)c * N + g] = (c < R4 ? v4[c] : 0) + (u32)acc;
}

template <typename F> float time_it(F f, int iters) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    f();
    f();
    (void)hipEventRecord(a);
    for (int i = 0; i < iters; i++) f();
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms / iters;
}

static u64 *g8;
static u32 *g4;

template <int R8, int R4, int W8, int W4, int MASK = 0, bool MASKLD = false> void run(const char *what, u64 N) {
    const double bytes = (double)N * ((R8 + W8) * 8 + (R4 + W4) * 4);
    float ms = time_it([&] { hipLaunchKernelGGL((k_shape<R8, R4, W8, W4, 64, MASK, MASKLD>), dim3((unsigned)((N + 63) / 64)), dim3(64), 0, 0, g8, g4, N); }, 10);
    printf("%-44s %9llu groups  reads %2d x 8 B + %d x 4 B, writes %2d x 8 B + %d x 4 B (%3.0f%% written)  %8.1f us  %7.1f GB/s\n", what,
           (unsigned long long)N, R8, R4, W8, W4, 100.0 * (W8 * 8 + W4 * 4) / ((R8 + W8) * 8 + (R4 + W4) * 4), ms * 1e3, bytes / ms / 1e6);
}

int main() {
    const u64 NMAX = (u64)8 << 20;
    CHECK(hipMalloc(&g8, NMAX * 40 * 8));
    CHECK(hipMalloc(&g4, NMAX * 12 * 4));
    CHECK(hipMemset(g8, 1, NMAX * 40 * 8));
    CHECK(hipMemset(g4, 1, NMAX * 12 * 4));
    CHECK(hipDeviceSynchronize());
    // (column c starts c x N x 8 B behind column 0: 2^23 groups put every stream at a multiple of 64 MiB -- does the chip's
    // address interleaving mind? 8 000 000 is what the engine's 8 M-group runs use, 8 388 608 + 4 352 a near-by odd multiple of 256)
    for (u64 N : {(u64)8000000, NMAX - 65536 + 4352}) {
        run<27, 0, 17, 0>("k_tick_lane shape", N);
        run<15, 6, 21, 9>("k_send_dense shape", N);
        run<36, 4, 33, 9>("k_tick_send shape", N);
    }
    for (u64 N : {NMAX, (u64)1 << 20}) {
        run<27, 0, 17, 0>("k_tick_lane shape", N);
        run<15, 6, 21, 9>("k_send_dense shape", N);
        run<36, 4, 33, 9>("k_tick_send shape", N);
        run<36, 0, 33, 0>("k_tick_send shape without the u32 streams", N);
        run<18, 2, 17, 5>("half the streams of k_tick_send", N);
        run<22, 0, 22, 0>("22 + 22 u64 (50 % written)", N);
        run<33, 0, 11, 0>("33 + 11 u64 (25 % written)", N);
        // (GB/s of the masked rows are by the UNMASKED byte count: what matters is the time)
        run<15, 6, 21, 9, 20>("k_send_dense shape, 1 lane in 20 skips its store", N);
        run<15, 6, 21, 9, 20, true>("k_send_dense shape, ... and its load", N);
        run<15, 6, 21, 9, 8>("k_send_dense shape, 1 lane in 8 skips its store", N);
        run<36, 4, 33, 9, 20>("k_tick_send shape, 1 lane in 20 skips its store", N);
        run<27, 0, 17, 0, 20>("k_tick_lane shape, 1 lane in 20 skips its store", N);
    }
    return 0;
}
