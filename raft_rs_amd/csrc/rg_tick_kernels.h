// rg_tick_kernels.h -- the tick kernels (one lane per raft group; LDS-staged variant) and their
// launcher. Instantiated once per slot count in tick_inst.hip (-DRG_P=n) so the eight
// specialisations compile in parallel; the ABI units (abi_*.hip) only see the extern template declarations.
#pragma once

#include "rg_group.h"
#include "rg_send.h"

#ifndef RG_SEND_NT_ITEMS /* 1: the work-item columns (written once per stage, read by the message builder) as non-temporal stores:
                            they stop evicting the tick's state from the Infinity Cache (profiles/r03_tick_send.txt, call r) */
#define RG_SEND_NT_ITEMS 1
#endif
#ifndef RG_TS_ORDER /* k_tick_send: 1 = the tick's stores are issued before the stage's loads (experiment) */
#define RG_TS_ORDER 0
#endif
#ifndef RG_TS_SPEC /* k_tick_send: 1 = the window columns of every slot are requested with the group's own loads, into registers
                      (rg_send_prefetch); 2 = the same, straight into LDS (gfx950 LDS-DMA: no register is held across the tick) */
#define RG_TS_SPEC 0
#endif

// Build-time tuning knobs (python -m raft_rs_amd.build --opt N); the default is what measured best
// on MI355X (profiles/).  bit0: non-temporal loads of the read-once message columns;
// bit1: unconditional stores of match/next/pr_commit for slots that have a Progress (full 128-B
// lines instead of lane-masked partial lines); bit2: 64-thread workgroups.
#ifndef RG_OPT
#define RG_OPT 6 /* measured best on MI355X: gpurun sweep in profiles/r01_tuning_sweep.txt */
#endif
#define RG_OPT_NT_MSG (RG_OPT & 1)
#ifndef RG_LANE_NX /* how the lane / list kernels get at `next` and the rare-path operands (rg_group.h: RG_NX_*) */
#define RG_LANE_NX RG_NX_PREFETCH
#endif
#define RG_OPT_UNCOND_ST (RG_OPT & 2)
// bit3: non-temporal stores of the `next` column and the result word -- written every tick, (almost) never read back by
// the next one, so they need not displace the columns that are; bit4: non-temporal stores of every state column
#define RG_OPT_NT_NEXT (RG_OPT & 8)
#define RG_OPT_NT_ALL (RG_OPT & 16)
// bit5: the lane kernels rewrite a slot's `matched` / `committed_index` cell in EVERY lane of a wave as soon as one lane
// changed it (both are in registers anyway): whole 128-B lines instead of lane-masked ones (tools/microbench/send_shape.hip)
#define RG_OPT_WAVE_ST (RG_OPT & 32)
// (NT is a TEMPLATE argument on purpose: as a function argument the two stores of `nt ? nontemporal : plain` are merged into one
// plain store when this function is simplified on its own, before it is inlined and the flag folds -- which of the two happens
// first depends on what else the translation unit holds; round 4 found the kernels of tick_inst.hip without a single `nt` store)
template <bool NT, typename T> RG_HD void rg_st(T &dst, T v) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (NT) __builtin_nontemporal_store(v, &dst);
    else dst = v;
#else
    dst = v;
#endif
}
#ifdef RG_BLOCK_SIZE /* experiment override */
#define RG_BLOCK RG_BLOCK_SIZE
#elif RG_OPT & 4
#define RG_BLOCK 64
#else
#define RG_BLOCK 256
#endif
// The 32-bit cell index of the single-tick lane kernels, by slot count: u32, or -- from RG_U32O_FROM slots on -- rg_u32o
// (opaque offsets, rg_common.h), which keeps `base + offset` out of the registers between a cell's load and its store:
// 6 slots 142 -> 120 VGPRs (3 -> 4 waves per SIMD), 8 slots 182 -> 147 (2 -> 3), 7 slots with the compressed hints and the
// early stores 162 -> 133 (128 with RG_MIN_WAVES=4). MEASURED (round 5, profiles/r05_c5_occupancy.txt): the extra wave buys
// NOTHING -- 1 M x 6 at four waves 65.4 us against 64.2 at three, config 5 in one launch at four waves 87.8 against 87.5 at
// three -- because these kernels are bound by instruction issue (1 100 - 2 100 VALU and 800 - 1 150 SALU per wave), not by
// latency an extra wave could hide; the opaque offsets themselves cost 0-8 %. So every slot count stays on u32 (99 = never);
// the knobs remain for the comparison. -DRG_LANE_IX32=<type>: one type for every slot count (experiment builds).
#ifndef RG_U32O_FROM
#define RG_U32O_FROM 99
#endif
#ifdef RG_LANE_IX32
template <int P> struct RgLaneIx { typedef RG_LANE_IX32 type; };
#else
template <int P> struct RgLaneIx { typedef typename std::conditional<(P >= RG_U32O_FROM), rg_u32o, u32>::type type; };
#endif
// Minimum waves per SIMD the register allocator must leave room for in the lane kernels of RG_MIN_WAVES_FROM slots and more
// (experiment builds; the default asks for nothing).
#ifndef RG_MIN_WAVES
#define RG_MIN_WAVES 1
#endif
#ifndef RG_MIN_WAVES_FROM
#define RG_MIN_WAVES_FROM 1
#endif
#define RG_LANE_BOUNDS(P) __launch_bounds__(RG_BLOCK, ((P) >= RG_MIN_WAVES_FROM ? RG_MIN_WAVES : 1))
#define RG_TICK_BOUNDS __launch_bounds__(RG_BLOCK)

static inline unsigned rg_grid_for(u64 n, unsigned per_block) { return (unsigned)((n + per_block - 1) / per_block); }

// A dense tick's launch can carry an EVENT on its own dispatch packet (hipExtLaunchKernelGGL's stopEvent: the packet's completion
// signal) instead of a hipEventRecord behind it, which is a barrier packet of its own in the engine's queue and costs the NEXT
// tick 2.5 us of idle queue (tools/microbench/pub_signal.hip: 51.3 vs 50.1 us per tick, 48.8 without any event). The commit
// publication uses it (abi_tick.hip sets the thread's pending event around the launch of the lane / class / split kernels;
// rg_publish_commit then only makes its side stream wait for it).
inline thread_local hipEvent_t rg_tls_stop_event = nullptr;
#define RG_LAUNCH_TICK(kernel, grid, block, stream, ...)                                                                          \
    do {                                                                                                                          \
        if (rg_tls_stop_event) {                                                                                                  \
            hipExtLaunchKernelGGL(kernel, grid, block, 0, stream, nullptr, rg_tls_stop_event, 0, __VA_ARGS__);                    \
            rg_tls_stop_event = nullptr; /* (consumed: the caller sees that the event went out with a launch) */                  \
        }                                                                                                                         \
        else hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__);                                                     \
    } while (0)

// ------------------------------------------------------------------------------------------------
// kernels: the tick (RG_VARIANT_LANE)
// ------------------------------------------------------------------------------------------------
// NTM: non-temporal. Chosen per launch (rg_launch_tick_t): where an engine's state and one tick of messages together
// straddle the 256 MB Infinity Cache, streaming the messages keeps the state resident (1.5 M x 5: 86 -> 82 us, config 5 in one
// 7-slot engine: 117 -> 109); where everything fits anyway (1 M x 5) it costs 0.8 % (profiles/r03_nt_messages.txt).
template <bool NTM = (RG_OPT_NT_MSG != 0), typename T> RG_HD T rg_ld_stream(const T *p) { // read-once data: keep it out of L2/MALL
    if (NTM) return __builtin_nontemporal_load(p);
    return *p;
}

// The flag rows and the cfg word go first: rg_prefetch_rare decides from them alone, so its loads can be issued
// while the bulk loads below are still in flight (memory returns a wave's loads in order).
// NTS: the STATE columns streamed as well (loads here, stores in rg_store_group) -- the third memory regime, for engines whose
// state alone is far beyond the Infinity Cache: nothing a launch touches is touched again before the cache has turned over,
// so nothing should be allocated there. Only together do the two halves pay (8 M x 5, one box: loads alone 525 -> 616 us,
// stores alone 534, both 485: profiles/r04_nt_state.txt).
template <int P, int NXM, typename IX, bool NTM = (RG_OPT_NT_MSG != 0), bool NTS = false, typename ES = RgNoEarlyStores>
RG_HD void rg_load_group(RgGroup<P> &r, const RgState &st, const RgMsgs &ms, IX g) {
    constexpr bool LOAD_NX = NXM == RG_NX_LOADED;
    r.mf = rg_ld_stream<NTM>(&rg_at(ms.mflags, g));
    r.pf = rg_ld_stream<NTS>(&rg_at(st.pflags, g));
    r.cfg = rg_ld_stream<NTS>(&rg_at(st.cfg, g));
    r.commit = rg_ld_stream<NTS>(&rg_at(st.commit, g));
    r.lo = rg_ld_stream<NTS>(&rg_at(st.lo, g));
    r.hi = rg_ld_stream<NTS>(&rg_at(st.hi, g));
    r.adv = rg_pub_load(st, g);
#pragma unroll
    for (int p = 0; p < P; p++) {
        const IX o = (IX)p * (IX)st.stride + g;
        r.mt[p] = rg_ld_stream<NTS>(&rg_at(st.match, o));
        if (LOAD_NX) r.nx[p] = rg_at(st.next, o);
        if (!ES::late_pc) r.pc[p] = rg_ld_stream<NTS>(&rg_at(st.prc, o));
        r.mi[p] = rg_ld_stream<NTM>(&rg_at(ms.mi, o));
        if (!ES::late_pc) r.mc[p] = rg_ld_stream<NTM>(&rg_at(ms.mc, o)); // (late loads: RgTick::run requests the two columns itself)
    }
    static_assert(!ES::late_pc || NXM == RG_NX_PREFETCH, "late loads ride with the rare-path prefetch batch");
    if (NXM == RG_NX_PREFETCH) rg_prefetch_rare<P, IX, ES>(r, st, ms, g);
}

// WHICH: bit 0 = everything but `next` and the flag row, bit 1 = those two (k_tick_send stores them behind its send
// stage, which changes both; every other caller stores the group in one go).
// WAVE_ST (kernels whose lanes hold CONSECUTIVE groups, all lanes of the wave arriving here together): RG_OPT_WAVE_ST
// EARLY: the tick ran with RgEarlyStores -- `next` and the followers' committed_index are in memory already; the one
// committed_index cell still due is the leader's own (RgGroup::pc_self, which RgTick::self_committed may have raised).
template <int P, typename IX, int WHICH = 3, bool WAVE_ST = false, bool NTS = false, bool EARLY = false>
RG_HD void rg_store_group(const RgGroup<P> &r, const RgState &st, IX g) {
    u32 d = r.dirty;
#if RG_OPT_UNCOND_ST
    {   // rewrite every cell of a slot that has a Progress and any event this tick: whole lines
        const u32 ev = r.evm; // slots with a Progress that had an event (RgTick)
        d |= EARLY ? ev | ((ev & (1u << RG_CFG_SELF(r.cfg))) << 16) : ev | (ev << 8) | (ev << 16);
    }
#endif
#if defined(__HIP_DEVICE_COMPILE__)
    if (WAVE_ST && RG_OPT_WAVE_ST && (WHICH & 1)) {
#pragma unroll
        for (int p = 0; p < P; p++) {
            if (__builtin_amdgcn_ballot_w64((d >> p) & 1u) != 0) d |= 1u << p;
            // (EARLY: only the lanes whose leader sits in slot p still hold a value for that cell)
            if (__builtin_amdgcn_ballot_w64((d >> (16 + p)) & 1u) != 0 && (!EARLY || (RG_CFG_SELF(r.cfg) == (u32)p && ((RG_CFG_PRESENT(r.cfg) >> p) & 1u)))) d |= 1u << (16 + p);
        }
    }
#endif
#pragma unroll
    for (int p = 0; p < P; p++) {
        const IX o = (IX)p * (IX)st.stride + g;
        if ((WHICH & 1) && (d & (1u << p))) rg_st<(RG_OPT_NT_ALL != 0 || NTS)>(rg_at(st.match, o), r.mt[p]);
        if (!EARLY && (WHICH & 2) && (d & (1u << (8 + p)))) rg_st<((RG_OPT_NT_ALL | RG_OPT_NT_NEXT) != 0 || NTS)>(rg_at(st.next, o), r.nx[p]);
        if ((WHICH & 1) && (d & (1u << (16 + p)))) rg_st<(RG_OPT_NT_ALL != 0 || NTS)>(rg_at(st.prc, o), EARLY ? r.pc_self : r.pc[p]);
    }
    if ((WHICH & 2) && (d & RG_DIRTY_PF)) rg_at(st.pflags, g) = r.pf;
    if (WHICH & 1) {
        if (d & RG_DIRTY_COMMIT) {
            if (st.pub) rg_pub_store(st, g, r.adv, r.commit); // commit publication: one byte per advanced group
            rg_at(st.commit, g) = r.commit;
        }
        if (d & RG_DIRTY_HI) rg_at(st.hi, g) = r.hi;
        rg_st<((RG_OPT_NT_ALL | RG_OPT_NT_NEXT) != 0 || NTS)>(rg_at(st.out, g), r.out);
    }
    // (an election's own stores go with part 2: k_tick_send runs its stage -- on the registers, it reads neither term_lo nor the
    // cfg word nor the table from memory -- between the two parts, where the kernel is at the limit of its scalar registers)
    if ((WHICH & 2) && (d & (RG_DIRTY_LO | RG_DIRTY_CFG))) { // an election (rare)
        if (d & RG_DIRTY_LO) rg_at(st.lo, g) = r.lo;
        if (d & RG_DIRTY_CFG) rg_at(st.cfg, g) = r.cfg;
    }
}

// RgTick's early stores (rg_group.h: ES) for kernels whose lanes hold consecutive groups and walk the tick together: the same
// cells rg_store_group would write at the end -- the unconditional / whole-line rules included -- issued as soon as the value
// is final. The dirty bits are cleared so that nothing is stored twice.
// MEASURED (round 5, profiles/r05_c5_occupancy.txt): it frees the registers it promises (P = 7: 161 -> 151 VGPRs with u32 offsets) and
// it LOSES -- 1 M x 7 steady 74.8 -> 84.0 us, 1 M x 8 87.1 -> 94.2 --: stores issued in the middle of the tick queue up in front of
// nothing useful and take issue slots from the arithmetic. Off by default (99 = never); kept, host-checked, as the comparison.
#ifndef RG_EARLY_ST_FROM /* slot counts from which the lane bodies store early (99 = never) */
#define RG_EARLY_ST_FROM 99
#endif
template <bool NTS> struct RgEarlyStores {
    static constexpr bool on = true;
    template <int P, typename IX> RG_HD static void store_pc(RgGroup<P> &r, const RgState &st, IX g, u32 self) {
        u32 d = r.dirty;
#if RG_OPT_UNCOND_ST
        d |= r.evm << 16;
#endif
        d &= ~(1u << (16 + self)); // (the leader's own cell goes with the group: the commit phase may still raise it)
#pragma unroll
        for (int p = 0; p < P; p++) {
#if defined(__HIP_DEVICE_COMPILE__)
            if (RG_OPT_WAVE_ST && __builtin_amdgcn_ballot_w64((d >> (16 + p)) & 1u) != 0) d |= 1u << (16 + p);
#endif
            if (d & (1u << (16 + p))) rg_st<(RG_OPT_NT_ALL != 0 || NTS)>(rg_at(st.prc, (IX)p * (IX)st.stride + g), r.pc[p]);
        }
        r.dirty &= ~(0xffu << 16) | (1u << (16 + self));
    }
    template <int S, int P, typename IX> RG_HD static void store_next(RgGroup<P> &r, const RgState &st, IX g) {
        u32 d = r.dirty >> (8 + S);
#if RG_OPT_UNCOND_ST
        d |= r.evm >> S;
#endif
        if (d & 1u) rg_st<((RG_OPT_NT_ALL | RG_OPT_NT_NEXT) != 0 || NTS)>(rg_at(st.next, (IX)S * (IX)st.stride + g), r.nx[S]);
        r.dirty &= ~(1u << (8 + S));
    }
};
// RgTick's late loads (rg_group.h: ES::late_pc): committed_index and Message.commit of the followers requested behind the slots,
// the own slot's two cells with the rare-path batch. NTM / NTS: the launch's streaming choice for message / state columns.
// MEASURED (round 5, profiles/r05_c5_occupancy.txt, second table): requested BEHIND the commit phase, with opaque 32-bit offsets,
// the 7-slot body needs 112 VGPRs (162 before) -- four waves per SIMD without scratch, without a dependent fetch in the rare paths
// and without mid-tick stores -- at the price of one exposed round trip at the end of every wave. Where the state sits in the
// Infinity Cache that round trip is short: config 5 in one launch 84.8 -> 80.8 us (0.52 -> 0.55 on that box), 1 M x 8 87.9 -> 84.9;
// where it comes from HBM it is not: 8 M x 7 all-streamed 639 -> 668 us. A steady 7-slot shard gains nothing (74.6 -> 75.5).
// So: the bodies of k_tick_classes with RG_LATE_PC_FROM slots or more, in the cached regimes (NTS = false) -- the launch whose
// occupancy the 7-slot body sets for every class. RG_LATE_PC_LANE_FROM: the same for the plain lane kernel (experiment, 99 = never).
#ifndef RG_LATE_PC_FROM
#define RG_LATE_PC_FROM 7
#endif
#ifndef RG_LATE_PC_LANE_FROM
#define RG_LATE_PC_LANE_FROM 99
#endif
template <bool NTM, bool NTS> struct RgLatePc : RgNoEarlyStores {
    static constexpr bool late_pc = true;
    template <int P, typename IX> RG_HD static void load_self(RgGroup<P> &r, const RgState &st, const RgMsgs &ms, IX g, u32 self, bool has_event) {
        r.pc_self = 0;
        r.mc_self_v = 0;
        if (self < (u32)P) {
            const IX o = (IX)self * (IX)st.stride + g;
            r.pc_self = rg_ld_stream<NTS>(&rg_at(st.prc, o));
            if (has_event) r.mc_self_v = rg_ld_stream<NTM>(&rg_at(ms.mc, o));
        }
    }
    template <int P, typename IX> RG_HD static void load_pc_mc(RgGroup<P> &r, const RgState &st, const RgMsgs &ms, IX g) {
#pragma unroll
        for (int p = 0; p < P; p++) {
            const IX o = (IX)p * (IX)st.stride + g;
            r.pc[p] = rg_ld_stream<NTS>(&rg_at(st.prc, o));
            r.mc[p] = rg_ld_stream<NTM>(&rg_at(ms.mc, o));
        }
    }
};
template <int P, bool NTS, bool NTM = true, bool CLS = false> struct RgLaneStores {
    static constexpr bool late = !NTS && P >= (CLS ? RG_LATE_PC_FROM : RG_LATE_PC_LANE_FROM);
    typedef typename std::conditional<late, RgLatePc<NTM, NTS>,
                                      typename std::conditional<(P >= RG_EARLY_ST_FROM), RgEarlyStores<NTS>, RgNoEarlyStores>::type>::type type;
};
// The cell index of a body of k_tick_classes: opaque offsets where the body loads late (they are what keeps the address pairs of
// the cells still to be stored out of the registers while the two late columns are in flight: 146 VGPRs with plain u32, 112 with)
template <int Q, int NTM> struct RgClassIx {
    typedef typename std::conditional<RgLaneStores<Q, NTM == 2, true, true>::late, rg_u32o, typename RgLaneIx<Q>::type>::type type;
};

template <int P, typename F> RG_D void rg_cpt_fields(RgGroup<P> &r, u64 &g64, u64 &cfg_adv, F &&f);
// IX = u32 when every cell of the engine's columns lies within 4 GiB of its column's start (rg_launch_tick_t decides).
// NTM: 1 = the read-once message columns as non-temporal loads (rg_ld_stream), 2 = the state columns as well, loads and stores
// (rg_load_group: NTS); decided per launch from the engine's footprint (rg_create)
template <int P, bool GC, typename IX, int NTM = 0> __global__ RG_LANE_BOUNDS(P) void k_tick_lane(RgState st, RgMsgs ms) {
    const u64 g64 = (u64)blockIdx.x * RG_BLOCK + threadIdx.x;
    if (g64 >= st.G) return;
    const IX g = (IX)g64;
#if defined(RG_LANE_LDS_PAD) && defined(__HIP_DEVICE_COMPILE__) /* experiment: steer the scheduler's occupancy target with an LDS allocation */
    __shared__ u64 occ_pad[RG_LANE_LDS_PAD / 8];
    asm volatile("" ::"v"((u32)(uintptr_t)&occ_pad[threadIdx.x]));
#endif
    RgGroup<P> r;
    typedef typename RgLaneStores<P, NTM == 2, NTM != 0 || (RG_OPT_NT_MSG != 0)>::type ES;
    rg_load_group<P, RG_LANE_NX, IX, NTM != 0 || (RG_OPT_NT_MSG != 0), NTM == 2, ES>(r, st, ms, g);
#if defined(RG_LANE_FENCE) && defined(__HIP_DEVICE_COMPILE__) /* experiment: keep the compiler from interleaving the tick with the loads */
#if RG_LANE_FENCE == 1
    asm volatile("" ::: "memory");
#elif RG_LANE_FENCE == 3
    {
        u64 g64b = g64, ca = (u64)r.cfg | ((u64)r.adv << 32);
        rg_cpt_fields<P>(r, g64b, ca, [&](int, u64 &x) { RG_OPAQUE64(x); });
        r.cfg = (u32)ca;
        r.adv = (u32)(ca >> 32);
    }
#else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#endif
    rg_group_tick<P, GC, RG_LANE_NX, false, IX, ES>(r, st, ms, g);
    rg_store_group<P, IX, 3, true, NTM == 2, ES::on>(r, st, g);
}

// ------------------------------------------------------------------------------------------------
// kernels: the tick over an engine whose groups are placed by SIZE CLASS (BASELINE config 5: replica sets of 3 / 5 / 7)
// ------------------------------------------------------------------------------------------------
// Peers a group does not have still occupy cells of an engine with P slots. Where sizes interleave every line of a column holds
// some group that has the peer, so every line travels (config 5 in one 7-slot engine: 533 MB of HBM traffic for 354 MB of
// algorithmic bytes, profiles/traffic.json). Where the host places groups of one size in contiguous ranges, whole blocks of 64
// groups use only the first Q < P slots -- the engine derives that from the cfg words by itself (rg_refresh_classes in
// abi_tick.hip (rg_refresh_classes): per block the highest slot any cfg word of the block names, one byte per block in device memory; a wave reads
// its byte with ONE scalar load before its first vector load. A table of ranges in the kernel arguments was tried first: its
// 17 SGPRs, live at the top of the kernel together with every column pointer, pushed 16 of those pointers into spill lanes
// for good -- 1 566 v_readlane / v_writelane in the code, ~250 executed per wave -- and it capped the layout at 8 ranges,
// which one conf change in the middle of a class breaks) -- and such a block runs the tick
// INSTANTIATED FOR Q SLOTS over the P-slot columns: no load, no store and no instruction for the absent peers (the quorum
// matrix alone is 9 / 25 / 49 compares). One launch for the whole shard instead of one engine, stream and launch per size
// class: the three 333 k-group launches of the size-class layout were tail-bound (154 us of kernel time overlapped into
// 100 us, profiles/r03_c5_kernel_stats.csv). The register allocation is the largest body's (P = 7: 3 waves per SIMD), code
// size the sum of the bodies -- what the three kernels running side by side occupied as well.
// Valid for a block iff every slot its groups' cfg words name (present, voters, self, transferee) is below Q: then the
// Q-slot tick and the P-slot tick are the same function of the group (slots >= Q carry no Progress and no event is applied
// to a slot without one). Bodies exist for Q in {3, 5, 7} below P, and P itself.
struct RgClasses {
    // One word per WORKGROUP, in launch order: the block of RG_BLOCK groups it runs (bits 0-27) and the slots that block's
    // groups name at most, rounded up to a body -- 3, 5, 7 or P -- (bits 28-31). nullptr on the host side: not class-placed,
    // the plain kernel runs. A wave reads its word with one scalar load before its first vector load.
    // Launch order: the ranges of equal blocks are dealt out PROPORTIONALLY (block i of a range of n sorts by (i + 1/2) / n), so
    // that at any moment the resident waves are a cross-section of the shard. Placed by class and launched in block order the
    // shard would be swept class by class: first every CU holds 3-slot waves (short, bandwidth-hungry), at the end every CU
    // holds 7-slot waves (three per SIMD, long, instruction- and latency-bound under rollover): 98 us instead of 85-90 for
    // config 5. (A fixed `(w % ways) * per_way + w / ways` deal was the first form: as good while the parts coincide with the
    // classes -- 3, 9, 12, 30 parts of three equal classes: 85 us --, worse than no deal at all when they do not -- 4 parts:
    // 107 us; profiles/calls/gpu_r04_f.sh, gpu_r04_i.sh.)
    const u32 *order;
};
// The kernel's arguments as ONE struct, so that a body can find them in the kernarg segment by itself (below).
struct RgClassArgs {
    RgState st;
    RgMsgs ms;
    RgClasses cls;
};
// Every body reads the column pointers from the kernarg segment ITSELF, through its own opaque copy of the segment pointer.
// Taken from the kernel's parameters instead, all of them are loaded at the top of the kernel and stay live through
// whichever body runs; what does not fit the scalar registers there is spilled to VGPR lanes for good and read back at every
// use -- 1 500 v_readlane / v_writelane in the code, ~250 executed per wave, in bodies (3 and 5 slots) that compiled on their
// own spill nothing. A scalar load that hits the constant cache costs the VALU nothing.
template <int Q, typename IX, int NTM, bool CLS = false> RG_D void rg_lane_body(IX g) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(4))) RgClassArgs *KArgs;
    KArgs ka = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka)); // (opaque per body: nothing of this can be hoisted above the dispatch or shared between bodies)
    const RgState st = ka->st;
    const RgMsgs ms = ka->ms;
    RgGroup<Q> r;
    typedef typename RgLaneStores<Q, NTM == 2, NTM != 0 || (RG_OPT_NT_MSG != 0), CLS>::type ES;
    rg_load_group<Q, RG_LANE_NX, IX, NTM != 0 || (RG_OPT_NT_MSG != 0), NTM == 2, ES>(r, st, ms, g);
    rg_group_tick<Q, false, RG_LANE_NX, false, IX, ES>(r, st, ms, g);
    rg_store_group<Q, IX, 3, true, NTM == 2, ES::on>(r, st, g);
#endif
}
template <int P, typename IX, int NTM> __global__ RG_LANE_BOUNDS(P) void k_tick_classes(RgClassArgs a) {
    const u32 e = a.cls.order[blockIdx.x]; // (scalar: blockIdx is uniform -- one s_load_dword)
    const u32 blk = e & 0x0fffffffu, np = e >> 28;
    const u64 g64 = (u64)blk * RG_BLOCK + threadIdx.x;
    if (g64 >= a.st.G) return;
    // (each body with the index type and the load policy of its own slot count: RgClassIx, RgLaneStores<.., CLS = true>)
    if (P > 3 && np <= 3) rg_lane_body<3, typename RgClassIx<3, NTM>::type, NTM, true>((typename RgClassIx<3, NTM>::type)g64);
    else if (P > 5 && np <= 5) rg_lane_body<5, typename RgClassIx<5, NTM>::type, NTM, true>((typename RgClassIx<5, NTM>::type)g64);
    else if (P > 7 && np <= 7) rg_lane_body<7, typename RgClassIx<7, NTM>::type, NTM, true>((typename RgClassIx<7, NTM>::type)g64);
    else rg_lane_body<P, typename RgClassIx<P, NTM>::type, NTM, true>((typename RgClassIx<P, NTM>::type)g64);
}

// The lane kernel over an engine whose state is PARTLY RESIDENT in the Infinity Cache (round 4). Beyond the cache a launch
// either allocates every state line there (and finds none of them again: LRU over more than the cache holds) or streams
// them all (k_tick_lane<.., 2>: the whole state from HBM, every tick). This kernel does both, by group range: the first
// `resident_blocks` workgroups run the body that goes through the cache (messages streamed), the rest the all-streamed body.
// What the resident groups re-read is then always a cache hit, whatever the size of the engine, and nothing else competes
// for those lines: every other access of the launch carries the non-temporal hint. rg_create sizes the range
// (profiles/r04_resident.txt). One launch, two bodies of the same register footprint (the hint is a bit of the instruction).
struct RgSplitArgs {
    RgState st;
    RgMsgs ms; // (st and ms where RgClassArgs has them: the bodies find them in the kernarg segment by themselves)
    u64 resident_blocks;
};
static_assert(offsetof(RgSplitArgs, st) == offsetof(RgClassArgs, st) && offsetof(RgSplitArgs, ms) == offsetof(RgClassArgs, ms) &&
                  sizeof(RgSplitArgs) >= sizeof(RgClassArgs),
              "rg_lane_body reads st and ms through an RgClassArgs view of the kernarg segment");
template <int P, typename IX> __global__ RG_LANE_BOUNDS(P) void k_tick_split(RgSplitArgs a) {
    const u64 g64 = (u64)blockIdx.x * RG_BLOCK + threadIdx.x;
    if (g64 >= a.st.G) return;
    const IX g = (IX)g64;
    if ((u64)blockIdx.x < a.resident_blocks) rg_lane_body<P, IX, 1>(g);
    else rg_lane_body<P, IX, 2>(g);
}

// ------------------------------------------------------------------------------------------------
// kernels: the tick AND its send stage in one launch (rg_tick_device_send; engines with device Inflights)
// ------------------------------------------------------------------------------------------------
// rg_tick_device + rg_send_appends as two launches move every word the stage needs twice: the tick stores the result
// word, the flag row, `next`, `matched`, last_index -- and the stage, one kernel boundary later, reads them back (88 B per
// group at P = 5) and rewrites `next` and the flag row (46 B). Here the stage runs on the registers the tick leaves:
// what it still reads from memory is what the tick never touched (the window columns of the peers in the work set,
// first_index, the `next` cell of a peer a broadcast reaches although it had no event in this tick), and the group is
// stored once. Order inside a lane: tick -> the stage's loads are REQUESTED -> the tick's own stores (everything but
// `next` and the flag row) -> the stage -> `next`, the flag row, the window columns, the work items. The stores sit
// between the request and the first use so that the wave waits for the loads only (vmcnt counts in issue order).
// One group of that launch, shared with the host twin of the tests (tests/host_check):
// The window columns of a workgroup's 64 groups as the LDS-DMA of k_tick_send (RG_TS_SPEC == 2) leaves them
template <int P> struct RgSendWin {
    u64 ht[P][2][64];             // [slot][0 = oldest inflight (head), 1 = newest (tail)][group of the block]
    u32 meta[(P + 3) / 4 * 4][64]; // [slot][group]: Inflights.start | count << 16 (rows beyond P: padding of the last DMA)
};

// WAVE: the lanes of the calling wave hold consecutive groups and arrive together (k_tick_send): whole-line accesses
// (rg_wave_any); the small-batch flush, whose lanes hold unrelated groups, passes false.
// Where the column pointers come from. A: every phase of the launch asks for the structs it needs when it starts --
// RgTsDirect hands out the caller's (the small-batch flush, the host twin of the tests); k_tick_send's provider re-reads them
// from the kernarg segment (RgTsKernarg, below), so that the stage's pointers are not live during the tick nor the messages'
// during the stage: the kernel names ~40 columns, twice what the scalar registers hold, and kept 428 v_readlane / v_writelane
// (9 % of its vector instructions) busy moving them in and out of spill lanes.
struct RgTsDirect {
    const RgState &s;
    const RgMsgs &m;
    const RgIns &i;
    RG_HD RgState st() const { return s; }
    RG_HD RgMsgs ms() const { return m; }
    RG_HD RgIns ins() const { return i; }
};
template <int P, bool GC, typename IX, bool WAVE, typename A, bool NTS = false>
RG_HD void rg_group_tick_send_a(RgGroup<P> &r, const A &a, IX g, u64 max_entries, u32 flags, RgSendRegs<P> &it,
                                const RgSendWin<P> *win = nullptr, u32 lane = 0) {
    RgSendOps<P> q;
    constexpr bool PRE = RG_TS_SPEC != 0;
    constexpr bool TSW = WAVE; // (on the host rg_wave_any is the lane's own answer)
    {   // ---- phase 1: the tick ----
        const RgState st = a.st();
        const RgMsgs ms = a.ms();
        if (RG_TS_SPEC == 1 || (RG_TS_SPEC == 2 && !win)) {
            const RgIns ins = a.ins();
            rg_send_prefetch<P, IX>(st, ins, g, q); // (behind the group's own loads, which the caller has issued)
        }
        if (RG_TS_SPEC == 2 && win) q.first_index = rg_at(st.dummy_idx, g) + 1;
        rg_group_tick<P, GC, RG_LANE_NX, false, IX>(r, st, ms, g);
    }
    // ---- phase 2: the stage, on the registers the tick leaves ----
    const RgState st = a.st();
    const RgIns ins = a.ins();
    // A reject of this group waits for the host's log (RG_OUT_HOST_HINT, rg_resolve_host_hints): its send_append belongs BEFORE
    // the group's other sends of the tick, so the group's send REQUESTS wait with it -- that call serves them. The stage below
    // applies the tick's Inflights effects only (free_to, free_first_one, the window resets): no work item; the real word goes
    // to RG_COL_OUT as always.
    const u32 sout = r.out;
    const bool hold = (r.out & RG_OUT_HOST_HINT) != 0;
#if defined(__HIP_DEVICE_COMPILE__) && RG_TS_SPEC == 2
    if (win) {
        // the DMA the kernel issued before the tick is a pending LDS write on the VM counter
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int s = 0; s < P; s++) {
            q.meta_v[s] = win->meta[s][lane];
            q.head_v[s] = win->ht[s][0][lane];
            q.tail_v[s] = win->ht[s][1][lane];
        }
    }
#endif
    // the `next` cells the tick holds: every slot with an event was fetched (or is overwritten by its SENT event), an
    // election rewrites all of them (rg_prefetch_rare / RgTick::set_next)
    const u32 nxv = r.evm | ((r.dirty >> 8) & 0xffu);
#if RG_TS_ORDER == 1 /* experiment: the tick's stores first (their registers are free before the stage's operands arrive) */
    rg_store_group<P, IX, 1, WAVE, NTS>(r, st, g);
    rg_send_request<P, IX, false, true, PRE, TSW>(st, ins, g, sout, flags, q, &r, nxv, hold);
#else
    rg_send_request<P, IX, false, true, PRE, TSW>(st, ins, g, sout, flags, q, &r, nxv, hold);
    rg_store_group<P, IX, 1, WAVE, NTS>(r, st, g);
#endif
    rg_send_serve<P, IX, true, TSW>(st, ins, g, sout, max_entries, flags, q, it, &r, nxv);
    rg_store_group<P, IX, 2, false, NTS>(r, st, g);
}
template <int P, bool GC, typename IX, bool WAVE = (RG_SEND_WAVE_LINES != 0)>
RG_HD void rg_group_tick_send(RgGroup<P> &r, const RgState &st, const RgMsgs &ms, const RgIns &ins, IX g, u64 max_entries,
                              u32 flags, RgSendRegs<P> &it, const RgSendWin<P> *win = nullptr, u32 lane = 0) {
    const RgTsDirect a = {st, ms, ins};
    rg_group_tick_send_a<P, GC, IX, WAVE>(r, a, g, max_entries, flags, it, win, lane);
}

// The work items of a dense stage into their peer-major columns (k_send_dense, k_tick_send).
// Round 5: an item whose last_index IS the window's new newest inflight (every MsgAppend with entries to a Replicate peer: the
// steady case) does not store it a second time -- bit 31 of the n / kind word says so and the consumer reads the window's tail
// column, which the stage has just written (8 B less per item; the compact list rg_send_items materialises is unchanged).
#ifndef RG_SEND_DROP_LAST /* 0 = every item stores its `last` cell (rounds 1-4; A/B builds) */
#define RG_SEND_DROP_LAST 1
#endif
template <int P, typename IX> RG_HD void rg_store_send_items(const RgSendRegs<P> &it, const RgSendCols &oc, u64 stride, IX g) {
#pragma unroll
    for (int s = 0; s < P; s++) {
        const IX o = (IX)s * (IX)stride + g;
        const u32 nk0 = rg_send_nk<P>(it, s);
        const bool in_tail = RG_SEND_DROP_LAST && nk0 != 0 && ((it.tailm >> s) & 1u);
        // (an empty MsgAppend -- send_append to a peer that has everything, raft.rs:773-819 with no entries left -- carries
        // last_index == prev_index: the second most common item of a steady stream, and as long as ONE lane of a wave needs its
        // `last` cell the whole line is written)
        const bool is_prev = RG_SEND_DROP_LAST && nk0 != 0 && !in_tail && (nk0 >> 16) == RG_SEND_APPEND && it.last[s] == it.prev[s];
        const u32 nk = nk0 | (in_tail ? RG_SEND_NK_LAST_IS_TAIL : 0u) | (is_prev ? RG_SEND_NK_LAST_IS_PREV : 0u);
        rg_st<(RG_SEND_NT_ITEMS != 0)>(rg_at(oc.n, o), nk); // every cell, every stage: 0 = nothing for this peer
        // whole lines where the wave stores at all: a slot some lane of the wave has an item for is written by every lane
        // (zeros where there is none); a slot nobody sends to -- the leaders' own, mostly -- is not touched; and the `last`
        // column of a slot is written (by every lane) only if some lane's item needs it there
#if defined(__HIP_DEVICE_COMPILE__)
        const bool any = RG_SEND_WHOLE_LINES ? __builtin_amdgcn_ballot_w64(nk != 0) != 0 : nk != 0;
        const bool any_last = RG_SEND_WHOLE_LINES ? __builtin_amdgcn_ballot_w64(nk != 0 && !in_tail && !is_prev) != 0 : (nk != 0 && !in_tail && !is_prev);
#else
        const bool any = true, any_last = true; // (the host twin writes every cell: zeros where the device leaves a line alone)
#endif
        if (any) rg_st<(RG_SEND_NT_ITEMS != 0)>(rg_at(oc.prev, o), nk ? it.prev[s] : (u64)0);
        if (any && any_last) rg_st<(RG_SEND_NT_ITEMS != 0)>(rg_at(oc.last, o), nk ? it.last[s] : (u64)0);
    }
}

#ifndef RG_TS_WAVES
#define RG_TS_WAVES 1 /* minimum waves per SIMD k_tick_send is compiled for (experiment knob) */
#endif
struct RgTickSendArgs { // k_tick_send's arguments as ONE struct: a phase finds them in the kernarg segment by itself
    RgState st;
    RgMsgs ms;
    RgIns ins;
    u64 max_entries;
    u32 flags;
    RgSendCols oc;
};
#if defined(__HIP_DEVICE_COMPILE__)
struct RgTsKernarg { // (each call is opaque to the others: a phase's scalar loads cannot be hoisted into an earlier phase)
    typedef const __attribute__((address_space(4))) RgTickSendArgs *KA;
    RG_D static KA ptr() {
        KA ka = (KA)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(ka));
        return ka;
    }
    RG_D RgState st() const { return ptr()->st; }
    RG_D RgMsgs ms() const { return ptr()->ms; }
    RG_D RgIns ins() const { return ptr()->ins; }
};
#else
struct RgTsKernarg { // (never used on the host)
    RgState st() const { return RgState(); }
    RgMsgs ms() const { return RgMsgs(); }
    RgIns ins() const { return RgIns(); }
};
#endif
// NTS: the tick's STATE columns streamed as well, loads and stores (rg_load_group / rg_store_group: the third memory regime) --
// for an engine with device Inflights whose state is far beyond the Infinity Cache (rg_config.cache_policy =
// RG_CACHE_STREAM_ALL; the window and work-item columns are streamed in every form).
template <int P, bool GC, typename IX, bool NTS = false>
__global__ __launch_bounds__(RG_BLOCK, RG_TS_WAVES) void k_tick_send(RgTickSendArgs ka_) {
    const RgState &st = ka_.st; // (the prologue below; the phases go through RgTsKernarg)
    const RgIns &ins = ka_.ins;
    const u64 g64 = (u64)blockIdx.x * RG_BLOCK + threadIdx.x;
#if RG_TS_SPEC == 2
    // The stage's window columns (meta 4 B, oldest / newest inflight 8 B each, per slot and group) travel straight from
    // global memory into LDS while the tick runs: global_load_lds_dwordx4 moves 16 B per lane to `base + lane x 16`, the
    // lanes' global addresses are free -- lanes 0-31 fetch the wave's 64 `head` cells of a slot, lanes 32-63 its `tail` cells
    // (one instruction per slot), 16 lanes fetch a slot's 64 `meta` cells (one instruction per four slots). Issued by the
    // WHOLE wave before the range check (a lane moves other lanes' groups; the columns are padded to the stride, a multiple
    // of 256, so the last block stays in bounds). No register is held across the tick: one memory round trip instead of
    // two, at the tick's occupancy.
    static_assert(RG_BLOCK % 64 == 0, "one window block per wave");
    __shared__ RgSendWin<P> win_all[RG_BLOCK / 64];
    RgSendWin<P> &win = win_all[threadIdx.x >> 6];
    const u32 lane = threadIdx.x & 63u;
    {
        const u64 w0 = g64 - lane; // first group of this wave
#pragma unroll
        for (int s = 0; s < P; s++) {
            const u64 *col = lane < 32 ? ins.head : ins.tail;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(col + (u64)s * st.stride + w0 + 2 * (lane & 31u)),
                                             (__attribute__((address_space(3))) void *)&win.ht[s][0][0], 16, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < (P + 3) / 4; k++) {
            const u32 slot = 4 * k + (lane >> 4);
            const u32 sc = slot < (u32)P ? slot : (u32)P - 1u; // (lanes beyond the last slot refetch it into the padding rows)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ins.meta + (u64)sc * st.stride + w0 + 4 * (lane & 15u)),
                                             (__attribute__((address_space(3))) void *)&win.meta[4 * k][0], 16, 0, 0);
        }
    }
#endif
    (void)ins;
    if (g64 >= st.G) return;
    const IX g = (IX)g64;
    const RgTsKernarg a;
    RgGroup<P> r;
    {
        const RgState st1 = a.st();
        const RgMsgs ms1 = a.ms();
        rg_load_group<P, RG_LANE_NX, IX, true, NTS>(r, st1, ms1, g); // (message columns streamed: an engine with device Inflights is past the cache at any size that matters)
    }
    RgSendRegs<P> it;
#if RG_TS_SPEC == 2
    rg_group_tick_send_a<P, GC, IX, (RG_SEND_WAVE_LINES != 0), RgTsKernarg, NTS>(r, a, g, ka_.max_entries, ka_.flags, it, &win, lane);
#else
    rg_group_tick_send_a<P, GC, IX, (RG_SEND_WAVE_LINES != 0), RgTsKernarg, NTS>(r, a, g, ka_.max_entries, ka_.flags, it);
#endif
#if defined(__HIP_DEVICE_COMPILE__)
    const RgSendCols oc = RgTsKernarg::ptr()->oc;
    const u64 stride = RgTsKernarg::ptr()->st.stride;
    rg_store_send_items<P, IX>(it, oc, stride, g);
#endif
}

// ------------------------------------------------------------------------------------------------
// kernels: the tick with rare-lane compaction (RG_VARIANT_COMPACT) -- for streams with leader-term rollover
// ------------------------------------------------------------------------------------------------
// Under BASELINE config 5 (elections, rejected probes, Probe -> Replicate) ~12 % of the groups leave the steady path in
// every tick, spread evenly: practically every 64-lane wave of k_tick_lane holds such a lane and executes the UNION of
// the rare paths (2 089 instead of 1 095 VALU per wave at P = 7, profiles/r02_pmc_sq_c5_one_engine.txt) -- the kernel
// is instruction-bound there. This variant runs 256-thread workgroups and, after the loads, trades the rare groups of
// three of the block's four waves for steady groups of the fourth (the "designated" wave, rotating with the block
// index so that the long waves spread over a CU's four SIMDs): the loaded REGISTERS of both travel through LDS -- the
// operands go with the group, nothing is re-read from memory -- so that one wave per block executes the rare paths and
// the other three only the steady one. Stores are addressed by the group a lane ends up holding. The classification
// (rg_is_rare) is a performance hint only: every lane runs the same complete tick code on whatever group it holds, so
// results do not depend on it (the parity suite runs this variant like the others).
#define RG_CPT_BLOCK 256
#ifndef RG_CPT_CAP
#define RG_CPT_CAP 48 /* swap pairs per block (LDS: 2 x (6 P + 8) x 8 B each); more rare lanes stay where they are */
#endif

// Can this group leave the steady path in this tick? Decided from the two flag rows and the cfg word, byte-parallel:
// an election; a heartbeat response; an AppendResponse that is a reject, comes from a peer outside Replicate (Probe /
// Snapshot transitions) or from one whose window is full (`old_paused`: the exact replay of commit_phase).
RG_HD bool rg_is_rare(u64 mf, u64 pf, u32 cfg, u32 n_slots) {
    const u64 L = 0x0101010101010101ULL;
    const u64 valid = mf & L, reject = (mf >> 1) & L, hb = (mf >> 6) & L;
    const u64 full = ((mf >> 3) | (pf >> 4)) & L;
    const u64 nonrepl = ((pf & L) ^ L) | ((pf >> 1) & L); // state != Replicate (1)
    return ((hb | (valid & (reject | nonrepl | full))) != 0) || rg_has_election(mf, cfg, n_slots);
}

// Every input of a tick, as rg_load_group leaves it, in a fixed field order: f(i, x) for field i.
template <int P, typename F> RG_D void rg_cpt_fields(RgGroup<P> &r, u64 &g64, u64 &cfg_adv, F &&f) {
    int i = 0;
#pragma unroll
    for (int p = 0; p < P; p++) {
        f(i++, r.mt[p]);
        f(i++, r.pc[p]);
        f(i++, r.mi[p]);
        f(i++, r.mc[p]);
        f(i++, r.nx[p]);
    }
#pragma unroll
    for (int h = 0; h < RgGroup<P>::HN; h++) f(i++, r.hint[h]);
    f(i++, r.mf);
    f(i++, r.pf);
    f(i++, r.commit);
    f(i++, r.lo);
    f(i++, r.hi);
    f(i++, r.el_old);
    {
        u64 n64 = (u64)r.el_n | ((u64)r.hsel << 32);
        f(i++, n64);
        r.el_n = (u32)n64;
        r.hsel = (u32)(n64 >> 32);
    }
    f(i++, cfg_adv);
    f(i++, g64);
}

// LDS-only barrier: the global loads of the block stay in flight across it (a __syncthreads() fences global memory
// too, i.e. waits for every outstanding load of the wave first).
RG_D void rg_lds_barrier() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

#ifdef RG_CPT_WAVES /* experiment: the occupancy the register allocator aims for (default: what the LDS footprint allows) */
#define RG_CPT_BOUNDS __launch_bounds__(RG_CPT_BLOCK) __attribute__((amdgpu_waves_per_eu(1, RG_CPT_WAVES)))
#else
#define RG_CPT_BOUNDS __launch_bounds__(RG_CPT_BLOCK)
#endif
template <int P, bool GC, typename IX>
__global__ RG_CPT_BOUNDS void k_tick_compact(RgState st, RgMsgs ms) {
    constexpr int NF = 5 * P + RgGroup<P>::HN + 9;
    __shared__ u64 x_rare[NF][RG_CPT_CAP];  // rare groups handed over by the natural waves
    __shared__ u64 x_disp[NF][RG_CPT_CAP];  // steady groups the designated wave gives away in exchange
    __shared__ u32 cnt[2];                  // [0] rare lanes of the natural waves, [1] steady lanes of the designated wave
    const u32 tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
#ifdef RG_CPT_FIXED_WD /* experiment: the designated wave does not rotate */
    const u32 wd = 0;
#else
    const u32 wd = blockIdx.x & 3u;
#endif
    u64 g64 = (u64)blockIdx.x * RG_CPT_BLOCK + tid;
    const bool valid = g64 < st.G; // (whole groups travel, so a lane that is valid stays valid)
    if (tid < 2) cnt[tid] = 0;
    RgGroup<P> r;
    r.mf = 0;
    r.pf = 0;
    r.cfg = 0;
    if (valid) rg_load_group<P, RG_NX_PREFETCH, IX>(r, st, ms, (IX)g64);
    rg_lds_barrier(); // the counters are zero
    const bool rare = valid && rg_is_rare(r.mf, r.pf, r.cfg, P);
    const bool cand = wave == wd ? (valid && !rare) : rare;
#if defined(__HIP_DEVICE_COMPILE__)
    const u64 bal = __ballot(cand);
    const u32 rank = __builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u));
    const u32 n_w = (u32)__builtin_popcountll(bal);
    u32 base = 0;
    if (wave == wd) {
        if (lane == 0) cnt[1] = n_w;
    } else if (n_w) {
        if (lane == 0) base = atomicAdd(&cnt[0], n_w);
        base = (u32)__builtin_amdgcn_readfirstlane((int)base);
    }
    const u32 k = base + rank;
    rg_lds_barrier();
    u32 n_swap = cnt[0] < cnt[1] ? cnt[0] : cnt[1];
    if (n_swap > RG_CPT_CAP) n_swap = RG_CPT_CAP;
    if (n_swap) { // (uniform over the block)
        const bool sw = cand && k < n_swap;
        u64 cfg_adv = (u64)r.cfg | ((u64)r.adv << 32);
        u64(*mine)[RG_CPT_CAP] = wave == wd ? x_disp : x_rare;
        u64(*theirs)[RG_CPT_CAP] = wave == wd ? x_rare : x_disp;
        if (sw) rg_cpt_fields<P>(r, g64, cfg_adv, [&](int i, u64 &x) { mine[i][k] = x; });
        rg_lds_barrier();
        if (sw) {
            rg_cpt_fields<P>(r, g64, cfg_adv, [&](int i, u64 &x) { x = theirs[i][k]; });
            r.cfg = (u32)cfg_adv;
            r.adv = (u32)(cfg_adv >> 32);
        }
    }
#else
    (void)x_rare; (void)x_disp; (void)cnt; (void)wave; (void)lane; (void)wd; (void)cand;
#endif
    if (!valid) return;
    const IX g = (IX)g64;
    rg_group_tick<P, GC, RG_NX_PREFETCH, false, IX>(r, st, ms, g);
    rg_store_group<P, IX>(r, st, g);
}

// Sparse variant: lane i owns group list[i] (the groups an ingest touched). Same arithmetic; accesses
// are gathers instead of streams, and the consumed event row is cleared so the message columns are
// all-zero again outside the touched set.
// The results of the listed groups are gathered on the way out (compact arrays for later device consumers and,
// when `packed` is given, 24-byte records behind a {groups, duplicates} header for ONE D2H copy).
struct RgListOut {
    u64 *rl, *rc; // [n] group, commit
    u32 *ro;      // [n] result word
    char *packed; // nullable: u32 n, u32 n_duplicates, 8 B pad, then {u64 group, u64 commit, u32 out, u32 pad}[n]
};

// entry i of the tick list: group g
// IX: index type of the column accesses -- u32 wherever the engine allows it (rg_fits_u32_offsets: fewer address registers,
// one more wave per SIMD for these gather-bound kernels), decided per launch like k_tick_lane's.
template <int P, bool GC, typename IX = u64>
RG_D void rg_tick_listed(const RgState &st, const RgMsgs &ms, u64 g64, u64 i, u64 *mflags_rw, const RgListOut &lo) {
    const IX g = (IX)g64;
    RgGroup<P> r;
    rg_load_group<P, RG_LANE_NX, IX>(r, st, ms, g);
    rg_group_tick<P, GC, RG_LANE_NX, false, IX>(r, st, ms, g);
    rg_store_group<P, IX>(r, st, g);
    rg_at(mflags_rw, g) = 0;
    lo.rl[i] = g;
    lo.rc[i] = r.commit;
    lo.ro[i] = r.out;
    if (lo.packed) {
        u64 *rec = reinterpret_cast<u64 *>(lo.packed + 16 + i * 24);
        rec[0] = g;
        rec[1] = r.commit;
        rec[2] = (u64)r.out;
    }
}

// The send stage inside the one-launch small flush (rg_flush_send on engines with device Inflights): where its work items go
struct RgSmallSend {
    RgIns ins;
    u64 max_entries;
    u32 flags;
    rg_send_item *items; // the device-side list (rg_send_items_ptr) ...
    u32 *counter;        // ... and its length
    char *pin;           // the same in pinned host memory for the caller: u32 count at byte 0, the items from byte 16
};

// entry i of the tick list with the send stage of its group behind the tick, on the tick's registers (rg_group_tick_send);
// the group's work items are appended through `lds_cnt`, a counter in the workgroup's LDS
template <int P, bool GC, typename IX = u64>
RG_D void rg_tick_send_listed(const RgState &st, const RgMsgs &ms, u64 g64, u64 i, u64 *mflags_rw, const RgListOut &lo,
                              const RgSmallSend &ss, u32 *lds_cnt) {
    const IX g = (IX)g64;
    RgGroup<P> r;
    rg_load_group<P, RG_LANE_NX, IX>(r, st, ms, g);
    RgSendRegs<P> it;
    rg_group_tick_send<P, GC, IX, false>(r, st, ms, ss.ins, g, ss.max_entries, ss.flags, it);
    rg_at(mflags_rw, g) = 0;
    lo.rl[i] = g;
    lo.rc[i] = r.commit;
    lo.ro[i] = r.out;
    if (lo.packed) {
        u64 *rec = reinterpret_cast<u64 *>(lo.packed + 16 + i * 24);
        rec[0] = g;
        rec[1] = r.commit;
        rec[2] = (u64)r.out;
    }
#pragma unroll
    for (int s = 0; s < P; s++) {
        const u32 nk = rg_send_nk<P>(it, s);
        if (!nk) continue;
        rg_send_item rec;
        rec.group = g64;
        rec.prev_index = it.prev[s];
        rec.last_index = it.last[s];
        rec.slot = (u32)s;
        rec.n_msgs = (uint16_t)(nk & 0xffffu);
        rec.kind = (uint16_t)(nk >> 16);
        const u32 k = atomicAdd(lds_cnt, 1u);
        ss.items[k] = rec;
        reinterpret_cast<rg_send_item *>(ss.pin + 16)[k] = rec;
    }
}

template <int P, bool GC, typename IX>
__global__ RG_TICK_BOUNDS void k_tick_list(RgState st, RgMsgs ms, const u64 *list, const u32 *n_ptr, u64 *mflags_rw,
                                           RgListOut lo) {
    const u64 i = (u64)blockIdx.x * RG_BLOCK + threadIdx.x;
    if (lo.packed && i == 0) {
        reinterpret_cast<u32 *>(lo.packed)[0] = n_ptr[0];
        reinterpret_cast<u32 *>(lo.packed)[1] = n_ptr[1];
    }
    if (i >= *n_ptr) return;
    rg_tick_listed<P, GC, IX>(st, ms, list[i], i, mflags_rw, lo);
}

// ------------------------------------------------------------------------------------------------
// ingest: wire-order AoS records -> the slot matrix (the sparse path's first step)
// ------------------------------------------------------------------------------------------------
#define RG_INGEST_BLOCK 256
struct RgClear { // housekeeping the ingest kernel does on the side, so the small-batch flush needs no extra command:
    const u64 *list; // result words of the PREVIOUS sparse tick to zero first (n of them, through `out`)
    u32 *out;
    u32 n;
    u32 *zero_ctr;   // the counter pair the NEXT sparse tick will use (the two pairs alternate): reset it
};
struct RgIngest {
    const rg_wire_msg *rec;
    u64 n, G, stride;
    u32 P;
    u64 *mi, *mc, *mh, *mrs, *mlt; // the engine-owned message columns
    u32 *mflags32;
    u32 *gmark;
    u32 epoch;
    u64 *list;
    u32 *counters; // [0] groups touched so far in this window, [1] records dropped
    RgClear clr;
};

RG_D void rg_ingest_housekeeping(const RgClear &clr) {
    for (u32 k = blockIdx.x * RG_INGEST_BLOCK + threadIdx.x; k < clr.n; k += gridDim.x * RG_INGEST_BLOCK)
        clr.out[clr.list[k]] = 0;
    if (blockIdx.x == 0 && threadIdx.x < 2) clr.zero_ctr[threadIdx.x] = 0;
}

// A workgroup stages 256 records (16 KiB) through LDS with fully coalesced 16-B loads, then lane t
// decodes record t and scatters its fields to the peer-major message columns. The event byte of the
// cell is claimed with a CAS on its 32-bit word; a second record for the same cell is dropped and
// counted. The first record that touches a group appends it to the tick list.
RG_D void rg_ingest_block(const RgIngest &a, uint4 *stage) {
    const u64 base = (u64)blockIdx.x * RG_INGEST_BLOCK;
    const u32 nrec = base >= a.n ? 0u : (u32)((a.n - base) < RG_INGEST_BLOCK ? (a.n - base) : RG_INGEST_BLOCK);
    const uint4 *src = reinterpret_cast<const uint4 *>(a.rec + base);
    for (u32 k = threadIdx.x; k < 4 * nrec; k += RG_INGEST_BLOCK) stage[k] = src[k];
    __syncthreads();
    const u32 t = threadIdx.x;
    if (t >= nrec) return;
    // record t occupies 4 x 16 B of the staged block
    const uint4 ra = stage[4 * t], rb = stage[4 * t + 1], rc = stage[4 * t + 2], rd = stage[4 * t + 3];
    const u64 group = (u64)ra.x | ((u64)ra.y << 32), index = (u64)ra.z | ((u64)ra.w << 32);
    const u64 commit = (u64)rb.x | ((u64)rb.y << 32), hint = (u64)rb.z | ((u64)rb.w << 32);
    const u64 rs = (u64)rc.x | ((u64)rc.y << 32), log_term = (u64)rc.z | ((u64)rc.w << 32);
    const u32 slot = rd.x, flags = rd.y & 0xffu;
    if (group >= a.G || slot >= a.P || flags == 0) { // malformed record: counted with the duplicates
        atomicAdd(&a.counters[1], 1u);
        return;
    }
    u32 *word = a.mflags32 + group * 2 + (slot >> 2);
    const u32 shift = 8u * (slot & 3u);
    u32 old = *word;
    for (;;) {
        if ((old >> shift) & 0xffu) { // the cell already holds an event of this tick
            atomicAdd(&a.counters[1], 1u);
            return;
        }
        const u32 seen = atomicCAS(word, old, old | (flags << shift));
        if (seen == old) break;
        old = seen;
    }
    const u64 o = (u64)slot * a.stride + group;
    a.mi[o] = index;
    a.mc[o] = commit;
    if (flags & RG_MF_REJECT) a.mh[o] = hint;
    if (flags & RG_MF_HAS_RS) a.mrs[o] = rs;
    if (flags & RG_MF_HAS_LOGTERM) a.mlt[o] = log_term;
    if (atomicExch(&a.gmark[group], a.epoch) != a.epoch) a.list[atomicAdd(&a.counters[0], 1u)] = group;
}

// The whole small-batch flush in ONE launch (<= 256 records, one workgroup): housekeeping, ingest, hint resolution,
// the tick of the touched groups and the packed results -- what k_ingest, k_resolve_hints_list and k_tick_list do
// in three. A host that waits for one RawNode::step's worth of results pays one launch latency instead of two or three.
// SEND: every touched group's send stage runs behind its tick (ss, lds_cnt: rg_tick_send_listed)
template <int P, bool GC, bool SEND = false>
RG_D void rg_flush_small_body(const RgState &st, const RgMsgs &ms, const RgIngest &a, u64 *rh, u64 *mflags_rw,
                              const RgListOut &lo, uint4 *stage, const RgSmallSend *ss = nullptr, u32 *lds_cnt = nullptr) {
    if (SEND && threadIdx.x == 0) *lds_cnt = 0; // (published by the barrier behind the ingest)
    rg_ingest_housekeeping(a.clr);
    rg_ingest_block(a, stage);
    // the message cells, the tick list and the counters were written by this workgroup: make them visible to all of it
    __threadfence();
    __syncthreads();
    const u32 n_groups = atomicAdd(&a.counters[0], 0u), n_dropped = atomicAdd(&a.counters[1], 0u);
    if (ms.mhr == rh) { // some record may carry Message.log_term: find_conflict_by_term first (k_resolve_hints_list)
        for (u32 i = threadIdx.x; i < n_groups; i += RG_INGEST_BLOCK) rg_resolve_hints(st, ms, a.list[i], P, rh);
        __threadfence();
        __syncthreads();
    }
    if (lo.packed && threadIdx.x == 0) {
        reinterpret_cast<u32 *>(lo.packed)[0] = n_groups;
        reinterpret_cast<u32 *>(lo.packed)[1] = n_dropped;
    }
    for (u32 i = threadIdx.x; i < n_groups; i += RG_INGEST_BLOCK) {
        if (SEND) rg_tick_send_listed<P, GC>(st, ms, a.list[i], i, mflags_rw, lo, *ss, lds_cnt);
        else rg_tick_listed<P, GC>(st, ms, a.list[i], i, mflags_rw, lo);
    }
    if (SEND) {
        __syncthreads();
        if (threadIdx.x == 0) {
            const u32 cnt = *lds_cnt;
            *ss->counter = cnt;
            *reinterpret_cast<u32 *>(ss->pin) = cnt;
        }
    }
}

template <int P, bool GC>
__global__ __launch_bounds__(RG_INGEST_BLOCK) void k_flush_small(RgState st, RgMsgs ms, RgIngest a, u64 *rh, u64 *mflags_rw,
                                                                 RgListOut lo) {
    __shared__ uint4 stage[RG_INGEST_BLOCK * 4];
    rg_flush_small_body<P, GC>(st, ms, a, rh, mflags_rw, lo, stage);
}

// ... and with the send stage of the touched groups in the same launch (rg_flush_send, engines with device Inflights)
template <int P, bool GC>
__global__ __launch_bounds__(RG_INGEST_BLOCK) void k_flush_small_send(RgState st, RgMsgs ms, RgIngest a, u64 *rh, u64 *mflags_rw,
                                                                      RgListOut lo, RgSmallSend ss) {
    __shared__ uint4 stage[RG_INGEST_BLOCK * 4];
    __shared__ u32 lds_cnt;
    rg_flush_small_body<P, GC, true>(st, ms, a, rh, mflags_rw, lo, stage, &ss, &lds_cnt);
}

// The resident small-batch path ("mailbox", rg_mailbox_start): ONE workgroup stays on the device and serves small flushes
// out of pinned host memory -- the host writes the records and a request block, bumps a sequence word, and spins on the
// answer; no launch, no stream synchronisation, no wake-up of the HIP runtime (that is ~15 us of the 21 us a one-launch
// flush costs). Each request runs exactly k_flush_small's body. The workgroup leaves when told to (any other entry point
// of the engine stops it first), when it has been idle for `idle_ticks`, or after `max_ticks` (a bound on how long one
// launch can occupy the device whatever the host does); the host relaunches it on the next small flush.
struct RgMbox {
    // host -> device: 32 bytes the resident workgroup fetches with four independent 8-byte reads over PCIe per poll (in
    // flight together: one round trip). Every request word is SELF-VALIDATING: its upper half carries the request number,
    // its lower half a payload, and the host writes each word with ONE 8-byte store -- a word can never be torn, and a
    // request is accepted only when all four words carry the same new number, whatever order the four reads were
    // sampled in. (Round 2 bracketed plain fields with a head and a tail sequence word; the head is written first and
    // read first, the tail written last and read last, so a poll could see head = tail = s around fields of request
    // s - 1.)
    u64 w[4];  // w[0] = n (16 bits) | ctr_sel << 16 | any_logterm << 17 | send << 18 | send flags << 19, w[1] = epoch,
               // w[2] = clr_n, w[3] = the send stage's max_entries_per_msg (0xffffffff = NO_LIMIT); each | seq << 32
    u32 stop;  // written on its own by rg_mailbox_quiesce: not part of a request
    u32 pad_stop;
    u32 pad0[6];
    // device -> host
    u32 seq_done;
    u32 alive;
    u32 pad1[14];
};
RG_HD u64 rg_mbox_word(u32 seq, u32 payload) { return ((u64)seq << 32) | payload; }

template <int P, bool GC>
__global__ __launch_bounds__(RG_INGEST_BLOCK) void k_mailbox(RgState st, RgMsgs ms, RgIngest a0, u32 *ctr_base, u64 *rh,
                                                             u64 *mflags_rw, RgListOut lo, RgMbox *mb, u64 idle_ticks,
                                                             u64 max_ticks, RgSmallSend ss0) {
    __shared__ uint4 stage[RG_INGEST_BLOCK * 4];
    __shared__ u32 req[12];
    __shared__ u32 lds_cnt;
    const u64 t_start = wall_clock64(); // constant 100 MHz
    u64 t_last = t_start;
    u32 seq = 0;
    if (threadIdx.x == 0) {
        seq = __hip_atomic_load(&mb->seq_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); // the last request served
        __hip_atomic_store(&mb->alive, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    for (;;) {
        if (threadIdx.x == 0) {
            u32 leave = 0;
            u64 w0, w1, w2, w3, w4;
            u64 *q = reinterpret_cast<u64 *>(mb);
            for (;;) {
                // four relaxed system-scope loads: independent, so they are in flight together (one PCIe round trip);
                // `volatile` accesses would be waited for one by one
                w0 = __hip_atomic_load(q + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                w1 = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                w2 = __hip_atomic_load(q + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                w3 = __hip_atomic_load(q + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                w4 = __hip_atomic_load(q + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                const u32 s0 = (u32)(w0 >> 32);
                if (s0 != seq && (u32)(w1 >> 32) == s0 && (u32)(w2 >> 32) == s0 && (u32)(w3 >> 32) == s0) break; // all four words of ONE new request
                const u64 now = wall_clock64();
                if ((u32)w4 /* stop */ || now - t_last > idle_ticks || now - t_start > max_ticks) {
                    leave = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
            req[0] = (u32)(w0 >> 32);          // request number
            req[1] = leave;
            req[2] = (u32)w0 & 0xffffu;        // n
            req[3] = (u32)w1;                  // epoch
            req[4] = (u32)w2;                  // clr_n
            req[5] = ((u32)w0 >> 16) & 1u;     // ctr_sel
            req[6] = ((u32)w0 >> 17) & 1u;     // any_logterm
            req[7] = ((u32)w0 >> 18) & 1u;     // the send stage rides along (rg_flush_send)
            req[8] = ((u32)w0 >> 19) & 3u;     // its RG_SEND_* flags
            req[9] = (u32)w3;                  // its max_entries_per_msg
        }
        __syncthreads();
        if (req[1]) break;
        RgIngest a = a0;
        a.n = req[2];
        a.epoch = req[3];
        a.clr.n = req[4];
        a.counters = ctr_base + 2 * (req[5] & 1u);
        a.clr.zero_ctr = ctr_base + 2 * ((req[5] & 1u) ^ 1u);
        RgMsgs m = ms;
        m.mhr = req[6] ? rh : ms.mh;
        const u32 s = req[0];
        if (req[7]) {
            RgSmallSend ss = ss0;
            ss.flags = req[8];
            ss.max_entries = req[9] == 0xffffffffu ? ~0ULL : (u64)req[9];
            rg_flush_small_body<P, GC, true>(st, m, a, rh, mflags_rw, lo, stage, &ss, &lds_cnt);
        } else {
            rg_flush_small_body<P, GC>(st, m, a, rh, mflags_rw, lo, stage);
        }
        __threadfence_system(); // the packed results (pinned host memory) before the sequence word
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_store(&mb->seq_done, s, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            seq = s;
            t_last = wall_clock64();
        }
    }
    if (threadIdx.x == 0) __hip_atomic_store(&mb->alive, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Temporal fusion: T consecutive ticks of a group in ONE launch. A group's tick t+1 depends only on
// its own tick t, so a lane keeps its group's Progress/commit state in registers, streams the T
// message sets through, and writes the state back once: per tick the traffic is the message columns
// plus 1/T of the state read+write. Every tick's result word (and, optionally, commit index) is still
// produced. Bit-identical to T launches of k_tick_lane (tests). For backlogs / replay, not for latency.
#define RG_MAX_FUSE 8
#ifndef RG_FUSED_NT /* 1: the T message sets (read once) and the per-tick result columns (written once) go past the Infinity
                       Cache, so that what stays there from launch to launch is the state. profiles/r04_fused_nt.txt: 8 ticks per
                       launch at 1 M x 5 29.6 -> 28.1 us per tick, 4 ticks 33.5 -> 30.9; never slower (2 M, 8 M, 7 slots, config 5) */
#define RG_FUSED_NT 1
#endif
struct RgFused {
    RgMsgs m[RG_MAX_FUSE];
    u32 *out_t;    // [T][G] result word of every tick
    u64 *commit_t; // [T][G] commit index after every tick, or nullptr
    u32 n_ticks;
};

// IX: index type of the column accesses (rg_common.h: rg_at), u32 where the engine allows it -- like k_tick_lane.
template <int P, bool GC, typename IX> __global__ RG_TICK_BOUNDS void k_tick_fused(RgState st, RgFused fm) {
    const u64 g64 = (u64)blockIdx.x * RG_BLOCK + threadIdx.x;
    if (g64 >= st.G) return;
    const IX g = (IX)g64;
    RgGroup<P> r;
    r.pf = rg_at(st.pflags, g);
    r.cfg = rg_at(st.cfg, g);
    r.commit = rg_at(st.commit, g);
    r.lo = rg_at(st.lo, g);
    r.hi = rg_at(st.hi, g);
#pragma unroll
    for (int p = 0; p < P; p++) {
        const IX o = (IX)p * (IX)st.stride + g;
        r.mt[p] = rg_at(st.match, o);
        r.pc[p] = rg_at(st.prc, o);
        r.nx[p] = 0;
    }
    r.dirty = 0;
    r.evm = 0;
    r.adv = rg_pub_load(st, g); // commit publication: the launch's total advance lands in the group's byte (rg_store_group)
    for (u32 t = 0; t < fm.n_ticks; t++) {
        const RgMsgs &ms = fm.m[t];
        r.mf = rg_ld_stream<(RG_FUSED_NT != 0)>(&rg_at(ms.mflags, g));
#pragma unroll
        for (int p = 0; p < P; p++) {
            const IX o = (IX)p * (IX)st.stride + g;
            r.mi[p] = rg_ld_stream<(RG_FUSED_NT != 0)>(&rg_at(ms.mi, o));
            r.mc[p] = rg_ld_stream<(RG_FUSED_NT != 0)>(&rg_at(ms.mc, o));
        }
        rg_group_tick<P, GC, RG_NX_LAZY, true, IX>(r, st, ms, g);
        rg_st<(RG_FUSED_NT != 0)>(fm.out_t[(u64)t * st.G + g64], r.out);
        if (fm.commit_t) rg_st<(RG_FUSED_NT != 0)>(fm.commit_t[(u64)t * st.G + g64], r.commit);
    }
    // `next` cells that were only fetched (never needed a write) are harmlessly rewritten with their value
    rg_store_group<P, IX>(r, st, g);
}

// ------------------------------------------------------------------------------------------------
// kernels: the tick (RG_VARIANT_LDS) -- one wave per 128-group batch, columns staged through LDS.
// Global side: lane l moves groups {2l, 2l+1} of the batch with 16-B loads/stores (1 KiB per wave
// instruction). Compute side: lane l owns groups l and l+64 of the batch in turn. LDS holds the
// batch's 5 hot columns as [col][P][128] u64; a lane's ds_read_b64 at stride 8 B is conflict-free.
// ------------------------------------------------------------------------------------------------
#define RG_LDS_WAVES 1
#define RG_LDS_BATCH 128

// DMA = true (RG_VARIANT_LDS_DMA): the stage-in uses gfx950's LDS-DMA -- global_load_lds_dwordx4, 16 B per lane
// straight from global memory into LDS (destination = wave-uniform base + lane x 16 B, exactly this layout) without
// passing through VGPRs. Bytes in flight are then bounded by LDS, not registers: 5 P x 128 x 8 B = 25.6 KB per wave at
// P = 5, i.e. 6 waves per CU (160 KB) = 154 KB in flight per CU -- against ~254 KB for the lane kernel, whose 16 waves
// per CU each hold 31 x 512 B of loads in VGPRs. The trial is kept as a measured variant (profiles/r02_lds_dma.txt).
template <int P, bool GC, bool DMA>
__global__ __launch_bounds__(64 * RG_LDS_WAVES) void k_tick_lds(RgState st, RgMsgs ms) {
    __shared__ u64 lds[RG_LDS_WAVES][5][P][RG_LDS_BATCH];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u64 b0 = ((u64)blockIdx.x * RG_LDS_WAVES + wave) * RG_LDS_BATCH;
    if (b0 >= st.G) return; // whole wave out of range (stride is a multiple of 256, so loads below stay in bounds)
    // LDS operations of one wave execute in issue order, so a wave-level fence (a compiler ordering
    // point; no s_barrier needed for a single-wave batch) is all the staging needs.
    typedef u64 u64x2 __attribute__((ext_vector_type(2)));
    u64(*L)[P][RG_LDS_BATCH] = lds[wave];
    const u64 *cols[5] = {st.match, st.next, st.prc, ms.mi, ms.mc};
    if (DMA) {
        // stage in: 5*P LDS-DMA pieces of 1 KiB per wave, no VGPR in between
#pragma unroll
        for (int c = 0; c < 5; c++)
#pragma unroll
            for (int p = 0; p < P; p++)
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)(cols[c] + (u64)p * st.stride + b0 + 2 * lane),
                    (__attribute__((address_space(3))) void *)&L[c][p][0], 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // an LDS-DMA is a pending LDS write on the VM counter
    } else {
        // stage in: 5*P coalesced 16-B loads per lane, all issued before the first LDS write
        u64x2 tmp[5][P];
#pragma unroll
        for (int c = 0; c < 5; c++)
#pragma unroll
            for (int p = 0; p < P; p++)
                tmp[c][p] = *reinterpret_cast<const u64x2 *>(cols[c] + (u64)p * st.stride + b0 + 2 * lane);
#pragma unroll
        for (int c = 0; c < 5; c++)
#pragma unroll
            for (int p = 0; p < P; p++)
                *reinterpret_cast<u64x2 *>(&L[c][p][2 * lane]) = tmp[c][p];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int half = 0; half < 2; half++) {
        const int li = lane + 64 * half;
        const u64 g = b0 + li;
        if (g < st.G) {
            RgGroup<P> r;
            r.mf = ms.mflags[g];
            r.pf = st.pflags[g];
            r.cfg = st.cfg[g];
            r.commit = st.commit[g];
            r.lo = st.lo[g];
            r.hi = st.hi[g];
            r.adv = rg_pub_load(st, g);
#pragma unroll
            for (int p = 0; p < P; p++) {
                r.mt[p] = L[0][p][li];
                r.nx[p] = L[1][p][li];
                r.pc[p] = L[2][p][li];
                r.mi[p] = L[3][p][li];
                r.mc[p] = L[4][p][li];
            }
            rg_group_tick<P, GC, RG_NX_LOADED>(r, st, ms, g);
            const u32 d = r.dirty;
#pragma unroll
            for (int p = 0; p < P; p++) {
                if (d & (1u << p)) L[0][p][li] = r.mt[p];
                if (d & (1u << (8 + p))) L[1][p][li] = r.nx[p];
                if (d & (1u << (16 + p))) L[2][p][li] = r.pc[p];
            }
            if (d & RG_DIRTY_PF) st.pflags[g] = r.pf;
            if (d & RG_DIRTY_COMMIT) {
                if (st.pub) rg_pub_store(st, g, r.adv, r.commit);
                st.commit[g] = r.commit;
            }
            if (d & RG_DIRTY_HI) st.hi[g] = r.hi;
            if (d & RG_DIRTY_LO) st.lo[g] = r.lo;
            if (d & RG_DIRTY_CFG) st.cfg[g] = r.cfg;
            st.out[g] = r.out;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // stage out: the three state columns, 16-B stores (whole rows; unchanged cells rewrite their value)
    u64 *ocols[3] = {st.match, st.next, st.prc};
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int p = 0; p < P; p++) {
            const u64x2 v = *reinterpret_cast<const u64x2 *>(&L[c][p][2 * lane]);
            if (b0 + 2 * lane < st.G) // never write padding past G (keeps padding zero)
                *reinterpret_cast<u64x2 *>(ocols[c] + (u64)p * st.stride + b0 + 2 * lane) = v;
        }
}


// ------------------------------------------------------------------------------------------------
// launcher
// ------------------------------------------------------------------------------------------------
#define RG_VARIANT_NT_MSGS 0x100u /* engine-internal flag on the variant word: stream the message columns (k_tick_lane<.., NTM = 1>) */
#define RG_VARIANT_NT_ALL 0x200u  /* ... and the state columns, loads and stores (NTM = 2): engines far beyond the Infinity Cache */
template <int P> void rg_launch_tick_t(hipStream_t stream, const RgState &st, const RgMsgs &ms, u32 variant, bool gc);
// the lane kernel over a class-placed engine (no group commit, 32-bit cell offsets: the caller checks both); P >= 4
template <int P> void rg_launch_tick_classes_t(hipStream_t stream, const RgState &st, const RgMsgs &ms, int ntm, const RgClasses &cls);
// the lane kernel with the first resident_blocks workgroups' state kept in the Infinity Cache (no group commit, 32-bit cell offsets)
template <int P> void rg_launch_tick_split_t(hipStream_t stream, const RgState &st, const RgMsgs &ms, u64 resident_blocks);
template <int P>
void rg_launch_tick_list_t(hipStream_t stream, const RgState &st, const RgMsgs &ms, bool gc, const u64 *list,
                           const u32 *n_ptr, u64 n_upper, u64 *mflags_rw, const RgListOut &lo);
template <int P> void rg_launch_tick_fused_t(hipStream_t stream, const RgState &st, const RgFused &fm, bool gc);
template <int P>
void rg_launch_tick_send_t(hipStream_t stream, const RgState &st, const RgMsgs &ms, bool gc, const RgIns &ins, u64 max_entries,
                           u32 flags, const RgSendCols &oc, bool nts);
template <int P>
void rg_launch_flush_small_t(hipStream_t stream, const RgState &st, const RgMsgs &ms, bool gc, const RgIngest &a, u64 *rh,
                             u64 *mflags_rw, const RgListOut &lo);

template <int P>
void rg_launch_mailbox_t(hipStream_t stream, const RgState &st, const RgMsgs &ms, bool gc, const RgIngest &a0, u32 *ctr_base,
                         u64 *rh, u64 *mflags_rw, const RgListOut &lo, RgMbox *mb, u64 idle_ticks, u64 max_ticks,
                         const RgSmallSend &ss0);
template <int P>
void rg_launch_flush_small_send_t(hipStream_t stream, const RgState &st, const RgMsgs &ms, bool gc, const RgIngest &a, u64 *rh,
                                  u64 *mflags_rw, const RgListOut &lo, const RgSmallSend &ss);

#ifdef RG_TICK_INSTANTIATE
template <int P>
void rg_launch_flush_small_send_t(hipStream_t stream, const RgState &st, const RgMsgs &ms, bool gc, const RgIngest &a, u64 *rh,
                                  u64 *mflags_rw, const RgListOut &lo, const RgSmallSend &ss) {
    if (gc) hipLaunchKernelGGL((k_flush_small_send<P, true>), dim3(1), dim3(RG_INGEST_BLOCK), 0, stream, st, ms, a, rh, mflags_rw, lo, ss);
    else hipLaunchKernelGGL((k_flush_small_send<P, false>), dim3(1), dim3(RG_INGEST_BLOCK), 0, stream, st, ms, a, rh, mflags_rw, lo, ss);
}
template <int P>
void rg_launch_mailbox_t(hipStream_t stream, const RgState &st, const RgMsgs &ms, bool gc, const RgIngest &a0, u32 *ctr_base,
                         u64 *rh, u64 *mflags_rw, const RgListOut &lo, RgMbox *mb, u64 idle_ticks, u64 max_ticks,
                         const RgSmallSend &ss0) {
    if (gc) hipLaunchKernelGGL((k_mailbox<P, true>), dim3(1), dim3(RG_INGEST_BLOCK), 0, stream, st, ms, a0, ctr_base, rh, mflags_rw, lo, mb, idle_ticks, max_ticks, ss0);
    else hipLaunchKernelGGL((k_mailbox<P, false>), dim3(1), dim3(RG_INGEST_BLOCK), 0, stream, st, ms, a0, ctr_base, rh, mflags_rw, lo, mb, idle_ticks, max_ticks, ss0);
}
template <int P>
void rg_launch_flush_small_t(hipStream_t stream, const RgState &st, const RgMsgs &ms, bool gc, const RgIngest &a, u64 *rh,
                             u64 *mflags_rw, const RgListOut &lo) {
    if (gc) hipLaunchKernelGGL((k_flush_small<P, true>), dim3(1), dim3(RG_INGEST_BLOCK), 0, stream, st, ms, a, rh, mflags_rw, lo);
    else hipLaunchKernelGGL((k_flush_small<P, false>), dim3(1), dim3(RG_INGEST_BLOCK), 0, stream, st, ms, a, rh, mflags_rw, lo);
}
// (the group-commit instantiation has no streaming twin: it is the rare kernel and twice the code)
#define RG_LAUNCH_LANE(IXT)                                                                                                       \
    do {                                                                                                                          \
        if (ntm == 2 && !GC)                                                                                                      \
            RG_LAUNCH_TICK((k_tick_lane<P, GC, IXT, GC ? 0 : 2>), dim3(rg_grid_for(st.G, RG_BLOCK)), dim3(RG_BLOCK), stream, st, ms); \
        else if (ntm && !GC)                                                                                                      \
            RG_LAUNCH_TICK((k_tick_lane<P, GC, IXT, GC ? 0 : 1>), dim3(rg_grid_for(st.G, RG_BLOCK)), dim3(RG_BLOCK), stream, st, ms); \
        else                                                                                                                      \
            RG_LAUNCH_TICK((k_tick_lane<P, GC, IXT, 0>), dim3(rg_grid_for(st.G, RG_BLOCK)), dim3(RG_BLOCK), stream, st, ms); \
    } while (0)
template <int P, bool GC> static void rg_launch_tick_gc(hipStream_t stream, const RgState &st, const RgMsgs &ms, u32 variant) {
    const int ntm = (variant & RG_VARIANT_NT_ALL) ? 2 : (variant & RG_VARIANT_NT_MSGS) ? 1 : 0;
    variant &= ~(RG_VARIANT_NT_MSGS | RG_VARIANT_NT_ALL);
    if (variant == RG_VARIANT_LDS) {
        hipLaunchKernelGGL((k_tick_lds<P, GC, false>), dim3(rg_grid_for(st.G, RG_LDS_BATCH * RG_LDS_WAVES)),
                           dim3(64 * RG_LDS_WAVES), 0, stream, st, ms);
    } else if (variant == RG_VARIANT_LDS_DMA) {
        hipLaunchKernelGGL((k_tick_lds<P, GC, true>), dim3(rg_grid_for(st.G, RG_LDS_BATCH * RG_LDS_WAVES)),
                           dim3(64 * RG_LDS_WAVES), 0, stream, st, ms);
    } else if (variant == RG_VARIANT_COMPACT) {
        const dim3 grid(rg_grid_for(st.G, RG_CPT_BLOCK)), block(RG_CPT_BLOCK);
        if (rg_ix32(st, P)) hipLaunchKernelGGL((k_tick_compact<P, GC, u32>), grid, block, 0, stream, st, ms);
        else hipLaunchKernelGGL((k_tick_compact<P, GC, u64>), grid, block, 0, stream, st, ms);
    } else {
        // 32-bit cell offsets when every cell a lane addresses is below 4 GiB from its column's start
        if (rg_ix32(st, P))
            RG_LAUNCH_LANE(typename RgLaneIx<P>::type);
        else
            RG_LAUNCH_LANE(u64);
    }
}
template <int P> void rg_launch_tick_t(hipStream_t stream, const RgState &st, const RgMsgs &ms, u32 variant, bool gc) {
    if (gc) rg_launch_tick_gc<P, true>(stream, st, ms, variant);
    else rg_launch_tick_gc<P, false>(stream, st, ms, variant);
}
template <int P> void rg_launch_tick_classes_t(hipStream_t stream, const RgState &st, const RgMsgs &ms, int ntm, const RgClasses &cls) {
    if constexpr (P >= 4) {
        const dim3 grid(rg_grid_for(st.G, RG_BLOCK)), block(RG_BLOCK);
        RgClassArgs a;
        a.st = st;
        a.ms = ms;
        a.cls = cls;
        typedef typename RgLaneIx<P>::type IXP;
        if (ntm == 2) RG_LAUNCH_TICK((k_tick_classes<P, IXP, 2>), grid, block, stream, a);
        else if (ntm) RG_LAUNCH_TICK((k_tick_classes<P, IXP, 1>), grid, block, stream, a);
        else RG_LAUNCH_TICK((k_tick_classes<P, IXP, 0>), grid, block, stream, a);
    }
}
template <int P> void rg_launch_tick_split_t(hipStream_t stream, const RgState &st, const RgMsgs &ms, u64 resident_blocks) {
    RgSplitArgs a;
    a.st = st;
    a.ms = ms;
    a.resident_blocks = resident_blocks;
    RG_LAUNCH_TICK((k_tick_split<P, typename RgLaneIx<P>::type>), dim3(rg_grid_for(st.G, RG_BLOCK)), dim3(RG_BLOCK), stream, a);
}
template <int P>
void rg_launch_tick_list_t(hipStream_t stream, const RgState &st, const RgMsgs &ms, bool gc, const u64 *list,
                           const u32 *n_ptr, u64 n_upper, u64 *mflags_rw, const RgListOut &lo) {
    const dim3 grid(rg_grid_for(n_upper, RG_BLOCK)), block(RG_BLOCK);
    const bool ix32 = rg_ix32(st, P);
    if (gc) {
        if (ix32) hipLaunchKernelGGL((k_tick_list<P, true, u32>), grid, block, 0, stream, st, ms, list, n_ptr, mflags_rw, lo);
        else hipLaunchKernelGGL((k_tick_list<P, true, u64>), grid, block, 0, stream, st, ms, list, n_ptr, mflags_rw, lo);
    } else {
        if (ix32) hipLaunchKernelGGL((k_tick_list<P, false, u32>), grid, block, 0, stream, st, ms, list, n_ptr, mflags_rw, lo);
        else hipLaunchKernelGGL((k_tick_list<P, false, u64>), grid, block, 0, stream, st, ms, list, n_ptr, mflags_rw, lo);
    }
}
template <int P> void rg_launch_tick_fused_t(hipStream_t stream, const RgState &st, const RgFused &fm, bool gc) {
    const dim3 grid(rg_grid_for(st.G, RG_BLOCK)), block(RG_BLOCK);
    const bool ix32 = rg_ix32(st, P); // 32-bit cell offsets (rg_launch_tick_t)
    if (gc) {
        if (ix32) hipLaunchKernelGGL((k_tick_fused<P, true, rg_u32o>), grid, block, 0, stream, st, fm);
        else hipLaunchKernelGGL((k_tick_fused<P, true, u64>), grid, block, 0, stream, st, fm);
    } else {
        if (ix32) hipLaunchKernelGGL((k_tick_fused<P, false, rg_u32o>), grid, block, 0, stream, st, fm);
        else hipLaunchKernelGGL((k_tick_fused<P, false, u64>), grid, block, 0, stream, st, fm);
    }
}
template <int P>
void rg_launch_tick_send_t(hipStream_t stream, const RgState &st, const RgMsgs &ms, bool gc, const RgIns &ins, u64 max_entries,
                           u32 flags, const RgSendCols &oc, bool nts) {
    const dim3 grid(rg_grid_for(st.G, RG_BLOCK)), block(RG_BLOCK);
    const bool ix32 = rg_ix32(st, P); // 32-bit cell offsets (rg_launch_tick_t)
    RgTickSendArgs ta;
    ta.st = st;
    ta.ms = ms;
    ta.ins = ins;
    ta.max_entries = max_entries;
    ta.flags = flags;
    ta.oc = oc;
    if (gc) {
        if (ix32) hipLaunchKernelGGL((k_tick_send<P, true, u32>), grid, block, 0, stream, ta);
        else hipLaunchKernelGGL((k_tick_send<P, true, u64>), grid, block, 0, stream, ta);
    } else {
        if (ix32 && nts) hipLaunchKernelGGL((k_tick_send<P, false, u32, true>), grid, block, 0, stream, ta); // (the one streamed instantiation)
        else if (ix32) hipLaunchKernelGGL((k_tick_send<P, false, u32>), grid, block, 0, stream, ta);
        else hipLaunchKernelGGL((k_tick_send<P, false, u64>), grid, block, 0, stream, ta);
    }
}
#else
extern template void rg_launch_tick_t<1>(hipStream_t, const RgState &, const RgMsgs &, u32, bool);
extern template void rg_launch_tick_classes_t<1>(hipStream_t, const RgState &, const RgMsgs &, int, const RgClasses &);
extern template void rg_launch_tick_split_t<1>(hipStream_t, const RgState &, const RgMsgs &, u64);
extern template void rg_launch_tick_list_t<1>(hipStream_t, const RgState &, const RgMsgs &, bool, const u64 *, const u32 *, u64, u64 *, const RgListOut &);
extern template void rg_launch_tick_fused_t<1>(hipStream_t, const RgState &, const RgFused &, bool);
extern template void rg_launch_tick_send_t<1>(hipStream_t, const RgState &, const RgMsgs &, bool, const RgIns &, u64, u32, const RgSendCols &, bool);
extern template void rg_launch_flush_small_t<1>(hipStream_t, const RgState &, const RgMsgs &, bool, const RgIngest &, u64 *, u64 *, const RgListOut &);
extern template void rg_launch_flush_small_send_t<1>(hipStream_t, const RgState &, const RgMsgs &, bool, const RgIngest &, u64 *, u64 *, const RgListOut &, const RgSmallSend &);
extern template void rg_launch_mailbox_t<1>(hipStream_t, const RgState &, const RgMsgs &, bool, const RgIngest &, u32 *, u64 *, u64 *, const RgListOut &, RgMbox *, u64, u64, const RgSmallSend &);
extern template void rg_launch_tick_t<2>(hipStream_t, const RgState &, const RgMsgs &, u32, bool);
extern template void rg_launch_tick_classes_t<2>(hipStream_t, const RgState &, const RgMsgs &, int, const RgClasses &);
extern template void rg_launch_tick_split_t<2>(hipStream_t, const RgState &, const RgMsgs &, u64);
extern template void rg_launch_tick_list_t<2>(hipStream_t, const RgState &, const RgMsgs &, bool, const u64 *, const u32 *, u64, u64 *, const RgListOut &);
extern template void rg_launch_tick_fused_t<2>(hipStream_t, const RgState &, const RgFused &, bool);
extern template void rg_launch_tick_send_t<2>(hipStream_t, const RgState &, const RgMsgs &, bool, const RgIns &, u64, u32, const RgSendCols &, bool);
extern template void rg_launch_flush_small_t<2>(hipStream_t, const RgState &, const RgMsgs &, bool, const RgIngest &, u64 *, u64 *, const RgListOut &);
extern template void rg_launch_flush_small_send_t<2>(hipStream_t, const RgState &, const RgMsgs &, bool, const RgIngest &, u64 *, u64 *, const RgListOut &, const RgSmallSend &);
extern template void rg_launch_mailbox_t<2>(hipStream_t, const RgState &, const RgMsgs &, bool, const RgIngest &, u32 *, u64 *, u64 *, const RgListOut &, RgMbox *, u64, u64, const RgSmallSend &);
extern template void rg_launch_tick_t<3>(hipStream_t, const RgState &, const RgMsgs &, u32, bool);
extern template void rg_launch_tick_classes_t<3>(hipStream_t, const RgState &, const RgMsgs &, int, const RgClasses &);
extern template void rg_launch_tick_split_t<3>(hipStream_t, const RgState &, const RgMsgs &, u64);
extern template void rg_launch_tick_list_t<3>(hipStream_t, const RgState &, const RgMsgs &, bool, const u64 *, const u32 *, u64, u64 *, const RgListOut &);
extern template void rg_launch_tick_fused_t<3>(hipStream_t, const RgState &, const RgFused &, bool);
extern template void rg_launch_tick_send_t<3>(hipStream_t, const RgState &, const RgMsgs &, bool, const RgIns &, u64, u32, const RgSendCols &, bool);
extern template void rg_launch_flush_small_t<3>(hipStream_t, const RgState &, const RgMsgs &, bool, const RgIngest &, u64 *, u64 *, const RgListOut &);
extern template void rg_launch_flush_small_send_t<3>(hipStream_t, const RgState &, const RgMsgs &, bool, const RgIngest &, u64 *, u64 *, const RgListOut &, const RgSmallSend &);
extern template void rg_launch_mailbox_t<3>(hipStream_t, const RgState &, const RgMsgs &, bool, const RgIngest &, u32 *, u64 *, u64 *, const RgListOut &, RgMbox *, u64, u64, const RgSmallSend &);
extern template void rg_launch_tick_t<4>(hipStream_t, const RgState &, const RgMsgs &, u32, bool);
extern template void rg_launch_tick_classes_t<4>(hipStream_t, const RgState &, const RgMsgs &, int, const RgClasses &);
extern template void rg_launch_tick_split_t<4>(hipStream_t, const RgState &, const RgMsgs &, u64);
extern template void rg_launch_tick_list_t<4>(hipStream_t, const RgState &, const RgMsgs &, bool, const u64 *, const u32 *, u64, u64 *, const RgListOut &);
extern template void rg_launch_tick_fused_t<4>(hipStream_t, const RgState &, const RgFused &, bool);
extern template void rg_launch_tick_send_t<4>(hipStream_t, const RgState &, const RgMsgs &, bool, const RgIns &, u64, u32, const RgSendCols &, bool);
extern template void rg_launch_flush_small_t<4>(hipStream_t, const RgState &, const RgMsgs &, bool, const RgIngest &, u64 *, u64 *, const RgListOut &);
extern template void rg_launch_flush_small_send_t<4>(hipStream_t, const RgState &, const RgMsgs &, bool, const RgIngest &, u64 *, u64 *, const RgListOut &, const RgSmallSend &);
extern template void rg_launch_mailbox_t<4>(hipStream_t, const RgState &, const RgMsgs &, bool, const RgIngest &, u32 *, u64 *, u64 *, const RgListOut &, RgMbox *, u64, u64, const RgSmallSend &);
extern template void rg_launch_tick_t<5>(hipStream_t, const RgState &, const RgMsgs &, u32, bool);
extern template void rg_launch_tick_classes_t<5>(hipStream_t, const RgState &, const RgMsgs &, int, const RgClasses &);
extern template void rg_launch_tick_split_t<5>(hipStream_t, const RgState &, const RgMsgs &, u64);
extern template void rg_launch_tick_list_t<5>(hipStream_t, const RgState &, const RgMsgs &, bool, const u64 *, const u32 *, u64, u64 *, const RgListOut &);
extern template void rg_launch_tick_fused_t<5>(hipStream_t, const RgState &, const RgFused &, bool);
extern template void rg_launch_tick_send_t<5>(hipStream_t, const RgState &, const RgMsgs &, bool, const RgIns &, u64, u32, const RgSendCols &, bool);
extern template void rg_launch_flush_small_t<5>(hipStream_t, const RgState &, const RgMsgs &, bool, const RgIngest &, u64 *, u64 *, const RgListOut &);
extern template void rg_launch_flush_small_send_t<5>(hipStream_t, const RgState &, const RgMsgs &, bool, const RgIngest &, u64 *, u64 *, const RgListOut &, const RgSmallSend &);
extern template void rg_launch_mailbox_t<5>(hipStream_t, const RgState &, const RgMsgs &, bool, const RgIngest &, u32 *, u64 *, u64 *, const RgListOut &, RgMbox *, u64, u64, const RgSmallSend &);
extern template void rg_launch_tick_t<6>(hipStream_t, const RgState &, const RgMsgs &, u32, bool);
extern template void rg_launch_tick_classes_t<6>(hipStream_t, const RgState &, const RgMsgs &, int, const RgClasses &);
extern template void rg_launch_tick_split_t<6>(hipStream_t, const RgState &, const RgMsgs &, u64);
extern template void rg_launch_tick_list_t<6>(hipStream_t, const RgState &, const RgMsgs &, bool, const u64 *, const u32 *, u64, u64 *, const RgListOut &);
extern template void rg_launch_tick_fused_t<6>(hipStream_t, const RgState &, const RgFused &, bool);
extern template void rg_launch_tick_send_t<6>(hipStream_t, const RgState &, const RgMsgs &, bool, const RgIns &, u64, u32, const RgSendCols &, bool);
extern template void rg_launch_flush_small_t<6>(hipStream_t, const RgState &, const RgMsgs &, bool, const RgIngest &, u64 *, u64 *, const RgListOut &);
extern template void rg_launch_flush_small_send_t<6>(hipStream_t, const RgState &, const RgMsgs &, bool, const RgIngest &, u64 *, u64 *, const RgListOut &, const RgSmallSend &);
extern template void rg_launch_mailbox_t<6>(hipStream_t, const RgState &, const RgMsgs &, bool, const RgIngest &, u32 *, u64 *, u64 *, const RgListOut &, RgMbox *, u64, u64, const RgSmallSend &);
extern template void rg_launch_tick_t<7>(hipStream_t, const RgState &, const RgMsgs &, u32, bool);
extern template void rg_launch_tick_classes_t<7>(hipStream_t, const RgState &, const RgMsgs &, int, const RgClasses &);
extern template void rg_launch_tick_split_t<7>(hipStream_t, const RgState &, const RgMsgs &, u64);
extern template void rg_launch_tick_list_t<7>(hipStream_t, const RgState &, const RgMsgs &, bool, const u64 *, const u32 *, u64, u64 *, const RgListOut &);
extern template void rg_launch_tick_fused_t<7>(hipStream_t, const RgState &, const RgFused &, bool);
extern template void rg_launch_tick_send_t<7>(hipStream_t, const RgState &, const RgMsgs &, bool, const RgIns &, u64, u32, const RgSendCols &, bool);
extern template void rg_launch_flush_small_t<7>(hipStream_t, const RgState &, const RgMsgs &, bool, const RgIngest &, u64 *, u64 *, const RgListOut &);
extern template void rg_launch_flush_small_send_t<7>(hipStream_t, const RgState &, const RgMsgs &, bool, const RgIngest &, u64 *, u64 *, const RgListOut &, const RgSmallSend &);
extern template void rg_launch_mailbox_t<7>(hipStream_t, const RgState &, const RgMsgs &, bool, const RgIngest &, u32 *, u64 *, u64 *, const RgListOut &, RgMbox *, u64, u64, const RgSmallSend &);
extern template void rg_launch_tick_t<8>(hipStream_t, const RgState &, const RgMsgs &, u32, bool);
extern template void rg_launch_tick_classes_t<8>(hipStream_t, const RgState &, const RgMsgs &, int, const RgClasses &);
extern template void rg_launch_tick_split_t<8>(hipStream_t, const RgState &, const RgMsgs &, u64);
extern template void rg_launch_tick_list_t<8>(hipStream_t, const RgState &, const RgMsgs &, bool, const u64 *, const u32 *, u64, u64 *, const RgListOut &);
extern template void rg_launch_tick_fused_t<8>(hipStream_t, const RgState &, const RgFused &, bool);
extern template void rg_launch_tick_send_t<8>(hipStream_t, const RgState &, const RgMsgs &, bool, const RgIns &, u64, u32, const RgSendCols &, bool);
extern template void rg_launch_flush_small_t<8>(hipStream_t, const RgState &, const RgMsgs &, bool, const RgIngest &, u64 *, u64 *, const RgListOut &);
extern template void rg_launch_flush_small_send_t<8>(hipStream_t, const RgState &, const RgMsgs &, bool, const RgIngest &, u64 *, u64 *, const RgListOut &, const RgSmallSend &);
extern template void rg_launch_mailbox_t<8>(hipStream_t, const RgState &, const RgMsgs &, bool, const RgIngest &, u32 *, u64 *, u64 *, const RgListOut &, RgMbox *, u64, u64, const RgSmallSend &);
#endif
