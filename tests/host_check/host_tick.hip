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
