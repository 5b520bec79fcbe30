// Microbenchmark: what it costs the ENGINE's stream to tell a side stream "tick k is complete" after every tick (the commit
// publication, DESIGN §6: the side stream all-gathers the slice tick k wrote while tick k + 1 runs).
//   0  no side stream at all: back-to-back ticks (the floor)
//   1  hipEventRecord(ev, engine stream) behind every tick + hipStreamWaitEvent(side, ev)        (what rounds 2-6 do)
//   2  the tick is launched with hipExtLaunchKernelGGL(..., stopEvent = ev): the event rides on the dispatch packet's own
//      completion signal, no packet of its own in the engine's queue
//   3  no event: tick k + 1's first workgroup stores k into a flag word as it starts (an in-order queue starts k + 1 only when k
//      is complete), the side stream waits for flag >= k with hipStreamWaitValue64
//   4 / 5 / 6  = 1 / 2 / 3 with the rest of what the engine's side stream does per publication: the copy as a hipMemcpyAsync
//      (RCCL's all-gather at world size 1), a hipMemsetAsync of the slice, an event recorded on the side stream, and the HOST
//      waiting for the side event of three publications ago before it goes on (the slice re-use gate)
// "tick" = a streaming kernel of about the headline tick's length that also fills one of four rotating 1 MB slices with its tick
// number; side work = copy that slice out (checked at the end: every copy must hold its tick's number in every word).
// build: hipcc -O3 --offload-arch=gfx950 pub_signal.hip -o pub_signal ; run: ./pub_signal
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint64_t u64;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr u64 N = 1u << 20;         // "groups"
constexpr int COLS = 20;            // 20 u64 columns read + written: ~335 MB per launch
constexpr u64 SLICE = 1u << 17;     // 1 MB of u64

__global__ __launch_bounds__(64) void k_tick(u64 *io, u64 *slice, u64 tick, u64 *flag) {
    const u64 g = (u64)blockIdx.x * 64 + threadIdx.x;
    if (flag && g == 0) __hip_atomic_store(flag, tick, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); // "tick - 1 is complete"
    u64 v[COLS], acc = 0;
#pragma unroll
    for (int c = 0; c < COLS; c++) v[c] = io[(u64)c * N + g];
#pragma unroll
    for (int c = 0; c < COLS; c++) acc += v[c];
#pragma unroll
    for (int c = 0; c < COLS; c++) io[(u64)c * N + g] = acc + v[c];
    if (g < SLICE) slice[g] = tick;
}
__global__ void k_copy(const u64 *src, u64 *dst) { const u64 i = (u64)blockIdx.x * 256 + threadIdx.x; if (i < SLICE) dst[i] = src[i]; }
__global__ void k_flag(u64 *flag, u64 v) { __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

int main() {
    const int T = 200;
    u64 *io, *slices, *out, *flag;
    CHECK(hipMalloc(&io, COLS * N * 8));
    CHECK(hipMemset(io, 1, COLS * N * 8));
    CHECK(hipMalloc(&slices, 4 * SLICE * 8));
    CHECK(hipMalloc(&out, (u64)T * SLICE * 8));
    hipError_t fe = hipExtMallocWithFlags((void **)&flag, 8, hipMallocSignalMemory);
    if (fe != hipSuccess) { printf("hipMallocSignalMemory: %s -- variant 3 uses plain device memory\n", hipGetErrorString(fe)); CHECK(hipMalloc(&flag, 8)); }
    hipStream_t s1, s2;
    CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    std::vector<hipEvent_t> ev(T);
    for (auto &evt : ev) CHECK(hipEventCreateWithFlags(&evt, hipEventDisableTiming | hipEventDisableSystemFence));
    hipEvent_t a, b, sev[4];
    for (auto &evt : sev) CHECK(hipEventCreateWithFlags(&evt, hipEventDisableTiming));
    u64 *scratch;
    CHECK(hipMalloc(&scratch, SLICE * 8));
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    std::vector<u64> host((u64)T * SLICE);
    for (int rep = 0; rep < 2; rep++)
        for (int variant = 0; variant < 7; variant++) {
            const int base = variant >= 4 ? variant - 3 : variant; const bool rich = variant >= 4;
            CHECK(hipMemset(out, 0xff, (u64)T * SLICE * 8));
            CHECK(hipMemset(flag, 0, 8));
            CHECK(hipDeviceSynchronize());
            for (int w = 0; w < 5; w++) hipLaunchKernelGGL(k_tick, dim3(N / 64), dim3(64), 0, s1, io, slices, (u64)0, (u64 *)nullptr);
            CHECK(hipEventRecord(a, s1));
            for (int t = 1; t <= T; t++) {
                u64 *sl = slices + (u64)(t % 4) * SLICE;
                if (base == 2) hipExtLaunchKernelGGL(k_tick, dim3(N / 64), dim3(64), 0, s1, nullptr, ev[t - 1], 0, io, sl, (u64)t, (u64 *)nullptr);
                else hipLaunchKernelGGL(k_tick, dim3(N / 64), dim3(64), 0, s1, io, sl, (u64)t, base == 3 ? flag : (u64 *)nullptr);
                if (rich && t > 3) CHECK(hipEventSynchronize(sev[(t - 4) % 4]));
                if (base == 1) CHECK(hipEventRecord(ev[t - 1], s1));
                if (base == 1 || base == 2) CHECK(hipStreamWaitEvent(s2, ev[t - 1], 0));
                // variant 3: tick t is complete once tick t + 1 has started (flag >= t + 1); the last one is released below
                if (base == 3) CHECK(hipStreamWaitValue64(s2, flag, (u64)t + 1, hipStreamWaitValueGte, ~0ull));
                if (variant != 0 && !rich) hipLaunchKernelGGL(k_copy, dim3(SLICE / 256), dim3(256), 0, s2, sl, out + (u64)(t - 1) * SLICE);
                if (rich) {
                    CHECK(hipMemcpyAsync(out + (u64)(t - 1) * SLICE, sl, SLICE * 8, hipMemcpyDeviceToDevice, s2));
                    CHECK(hipMemsetAsync(scratch, 0, SLICE * 8, s2));
                    CHECK(hipEventRecord(sev[(t - 1) % 4], s2));
                }
                if (base == 3 && t == T) hipLaunchKernelGGL(k_flag, dim3(1), dim3(1), 0, s1, flag, (u64)T + 1); // (what rg_publish_sync would do)
            }
            CHECK(hipEventRecord(b, s1));
            CHECK(hipEventSynchronize(b));
            CHECK(hipStreamSynchronize(s2));
            float ms;
            CHECK(hipEventElapsedTime(&ms, a, b));
            u64 bad = 0;
            if (variant != 0) {
                CHECK(hipMemcpy(host.data(), out, (u64)T * SLICE * 8, hipMemcpyDeviceToHost));
                for (int t = 1; t <= T; t++)
                    for (u64 i = 0; i < SLICE; i++) bad += host[(u64)(t - 1) * SLICE + i] != (u64)t;
            }
            printf("variant %d: %.2f us per tick (engine stream), %llu wrong words in the %d copied slices\n", variant, ms * 1e3 / T, (unsigned long long)bad, variant ? T : 0);
        }
    return 0;
}
