// rg_publish.h -- publication of commit indices across ranks: the delta encoding a tick produces,
// and the kernels that keep every rank's replica of all commit indices.
//
// What crosses xGMI is what RawNode::advance_append surfaces per group (src/raw_node.rs:643-651), fed by
// Raft::maybe_commit (src/raft.rs:893-904): the new commit index. It never decreases (RaftLog::commit_to,
// src/raft_log.rs:286-300), so a publication carries, per group, how far the index ADVANCED since the previous
// publication: one byte. The column of u64 indices (8 B/group: 8 MB per rank and tick at 1 M groups, 64 MB
// gathered by each of 8 ranks -- infeasible per tick on 7 x ~64 GB/s of xGMI ingress) shrinks to ~1 B/group.
//
// Encoding of one rank's slice (what ncclAllGather moves), all additive so receivers need no ordering:
//   [ RgPubHdr | RgPubOvf list[cap] | u8 delta[Gpad] ]
//   advance of group g since the last publication = delta[g] + sum of list entries {g, extra}.
// The tick kernels ACCUMULATE into delta[] (saturating at 255, the excess goes to the list), so any publication
// cadence works; a full list sets RG_PUB_LOST in the header and all ranks fall back to one full u64 snapshot
// (they all see the same headers, so the decision needs no extra collective).
//
// Receivers keep the gathered slices of the last R publications in a ring and fold them into the replica
// [world][Gpad] u64 only when the ring is full or somebody reads it: the per-tick cost of a publication is the
// all-gather itself (on a side stream) plus ~(R + 16)/R bytes per group and tick of HBM traffic.
//
// Host+device code: tests/ runs the same pack / apply arithmetic on the CPU under gloo (rg_pub_*_host).
#pragma once

#include "rg_common.h"

struct RgPubHdr {
    u32 n_overflow; // list entries appended (may exceed the capacity: then RG_PUB_LOST is implied)
    u32 flags;      // RG_PUB_*
    u64 seq;        // publication number of the sender (diagnostics)
};
#define RG_PUB_LOST 0x1u /* this rank's replica is no longer exact (list overflow / rollback): resynchronise */

struct RgPubOvf {
    u64 group, extra;
};

struct RgPubLayout {
    u64 G, Gpad;      // groups per rank; padded to 256
    u32 cap;          // list capacity
    u64 off_list, off_delta, bytes_per_rank;
};

static inline RgPubLayout rg_pub_layout(u64 G, u32 cap) {
    RgPubLayout l;
    l.G = G;
    l.Gpad = (G + 255) / 256 * 256;
    l.cap = cap;
    l.off_list = sizeof(RgPubHdr);
    l.off_delta = (l.off_list + (u64)cap * sizeof(RgPubOvf) + 255) / 256 * 256;
    l.bytes_per_rank = l.off_delta + l.Gpad;
    return l;
}

// The accumulate step of one group whose commit index just moved from `old_commit` to `new_commit`:
// `acc` = the byte accumulated so far in this publication interval. Returns the new byte; a saturated step
// appends its excess to the list.
RG_HD u32 rg_pub_accumulate(u32 acc, u64 old_commit, u64 new_commit, u64 g, RgPubHdr *hdr, RgPubOvf *list, u32 cap) {
    const u64 total = (u64)acc + (new_commit - old_commit);
    if (total <= 255) return (u32)total;
#if defined(__HIP_DEVICE_COMPILE__)
    const u32 k = atomicAdd(&hdr->n_overflow, 1u);
#else
    const u32 k = hdr->n_overflow++;
#endif
    if (k < cap) {
        list[k].group = g;
        list[k].extra = total - 255;
    } else {
#if defined(__HIP_DEVICE_COMPILE__)
        atomicOr(&hdr->flags, RG_PUB_LOST);
#else
        hdr->flags |= RG_PUB_LOST;
#endif
    }
    return 255u;
}

// Fold `n_slots` gathered publications into the replica, for 8 consecutive groups of one rank.
// slices[s] = the gathered buffer of publication s ([world] x bytes_per_rank).
#define RG_PUB_MAX_RING 64
struct RgPubSlots {
    const char *slice[RG_PUB_MAX_RING];
    u32 n;
};

RG_HD void rg_pub_apply8(u64 *replica, const RgPubSlots &sl, const RgPubLayout &l, u32 rank, u64 g8) {
    u64 add[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (u32 s = 0; s < sl.n; s++) {
        const u64 w = *reinterpret_cast<const u64 *>(sl.slice[s] + (u64)rank * l.bytes_per_rank + l.off_delta + g8);
#pragma unroll
        for (int k = 0; k < 8; k++) add[k] += (w >> (8 * k)) & 0xffu;
    }
    u64 *dst = replica + (u64)rank * l.Gpad + g8;
#pragma unroll
    for (int k = 0; k < 8; k++)
        if (add[k]) dst[k] += add[k];
}

