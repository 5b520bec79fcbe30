// host_tick.hip -- TEST INFRASTRUCTURE. Compiles the engine's per-group arithmetic header
// (raft_rs_amd/csrc/rg_group.h, the exact code the HIP kernels inline) for the HOST, so CPU-only
// tests can diff it against the oracle. This file is never part of libraftgroups.so: the product has
// no CPU path. Build: hipcc -O2 -shared -fPIC --offload-arch=gfx950 host_tick.hip -o libhost_tick.so
#include "../../raft_rs_amd/csrc/rg_tick_kernels.h"

template <int P> static void host_tick(const RgState &st, const RgMsgs &ms, bool gc, u64 g0, u64 g1) {
    for (u64 g = g0; g < g1; g++) {
        RgGroup<P> r;
        rg_load_group<P, !RG_LAZY_NEXT>(r, st, ms, g); // exactly what k_tick_lane does
        if (gc) rg_group_tick<P, true, RG_LAZY_NEXT>(r, st, ms, g);
        else rg_group_tick<P, false, RG_LAZY_NEXT>(r, st, ms, g);
        rg_store_group<P>(r, st, g);
    }
}

extern "C" int rg_host_check_tick(unsigned P, unsigned long G, unsigned long stride, u64 *match, u64 *next, u64 *prc,
                                  u64 *psnap, u64 *prs, u64 *gid, u64 *pflags, u64 *commit, u64 *lo, u64 *hi, u32 *cfg,
                                  u32 *out, const u64 *mi, const u64 *mc, const u64 *mh, const u64 *mrs,
                                  const u64 *mflags, int group_commit_kernel, unsigned long g_begin,
                                  unsigned long g_end) {
    RgState st;
    st.match = match; st.next = next; st.prc = prc; st.psnap = psnap; st.prs = prs; st.gid = gid;
    st.pflags = pflags; st.commit = commit; st.lo = lo; st.hi = hi; st.cfg = cfg; st.out = out;
    st.G = G; st.stride = stride;
    RgMsgs ms;
    ms.mi = mi; ms.mc = mc; ms.mh = mh; ms.mrs = mrs; ms.mflags = mflags;
    const bool gc = group_commit_kernel != 0;
    const u64 g0 = g_begin, g1 = g_end < G ? g_end : G;
    switch (P) {
    case 1: host_tick<1>(st, ms, gc, g0, g1); break;
    case 2: host_tick<2>(st, ms, gc, g0, g1); break;
    case 3: host_tick<3>(st, ms, gc, g0, g1); break;
    case 4: host_tick<4>(st, ms, gc, g0, g1); break;
    case 5: host_tick<5>(st, ms, gc, g0, g1); break;
    case 6: host_tick<6>(st, ms, gc, g0, g1); break;
    case 7: host_tick<7>(st, ms, gc, g0, g1); break;
    case 8: host_tick<8>(st, ms, gc, g0, g1); break;
    default: return -1;
    }
    return 0;
}

// Fused replay on the host: the same sequence k_tick_fused runs per lane (state in "registers" across ticks).
template <int P> static void host_fused(const RgState &st, const RgMsgs *ms, u32 T, u32 *out_t, u64 *commit_t, bool gc) {
    for (u64 g = 0; g < st.G; g++) {
        RgGroup<P> r;
        r.pf = st.pflags[g]; r.cfg = st.cfg[g]; r.commit = st.commit[g]; r.lo = st.lo[g]; r.hi = st.hi[g];
        for (int p = 0; p < P; p++) {
            const u64 o = (u64)p * st.stride + g;
            r.mt[p] = st.match[o]; r.pc[p] = st.prc[o]; r.nx[p] = 0;
        }
        r.dirty = 0; r.evm = 0;
        for (u32 t = 0; t < T; t++) {
            r.mf = ms[t].mflags[g];
            for (int p = 0; p < P; p++) {
                const u64 o = (u64)p * st.stride + g;
                r.mi[p] = ms[t].mi[o]; r.mc[p] = ms[t].mc[o];
            }
            if (gc) rg_group_tick<P, true, true, true>(r, st, ms[t], g);
            else rg_group_tick<P, false, true, true>(r, st, ms[t], g);
            out_t[(u64)t * st.G + g] = r.out;
            if (commit_t) commit_t[(u64)t * st.G + g] = r.commit;
        }
        rg_store_group<P>(r, st, g);
    }
}

extern "C" int rg_host_check_fused(unsigned P, unsigned long G, unsigned long stride, u64 *match, u64 *next, u64 *prc,
                                   u64 *psnap, u64 *prs, u64 *gid, u64 *pflags, u64 *commit, u64 *lo, u64 *hi, u32 *cfg,
                                   u32 *out, unsigned T, const u64 *const *mi, const u64 *const *mc, const u64 *const *mh,
                                   const u64 *const *mrs, const u64 *const *mflags, u32 *out_t, u64 *commit_t,
                                   int group_commit_kernel) {
    RgState st;
    st.match = match; st.next = next; st.prc = prc; st.psnap = psnap; st.prs = prs; st.gid = gid;
    st.pflags = pflags; st.commit = commit; st.lo = lo; st.hi = hi; st.cfg = cfg; st.out = out;
    st.G = G; st.stride = stride;
    RgMsgs ms[RG_MAX_FUSE];
    if (T == 0 || T > RG_MAX_FUSE) return -1;
    for (unsigned t = 0; t < T; t++) {
        ms[t].mi = mi[t]; ms[t].mc = mc[t]; ms[t].mh = mh[t]; ms[t].mrs = mrs[t]; ms[t].mflags = mflags[t];
    }
    const bool gc = group_commit_kernel != 0;
    switch (P) {
    case 1: host_fused<1>(st, ms, T, out_t, commit_t, gc); break;
    case 2: host_fused<2>(st, ms, T, out_t, commit_t, gc); break;
    case 3: host_fused<3>(st, ms, T, out_t, commit_t, gc); break;
    case 4: host_fused<4>(st, ms, T, out_t, commit_t, gc); break;
    case 5: host_fused<5>(st, ms, T, out_t, commit_t, gc); break;
    case 6: host_fused<6>(st, ms, T, out_t, commit_t, gc); break;
    case 7: host_fused<7>(st, ms, T, out_t, commit_t, gc); break;
    case 8: host_fused<8>(st, ms, T, out_t, commit_t, gc); break;
    default: return -1;
    }
    return 0;
}
