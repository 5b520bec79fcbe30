// rg_wire.h -- the proto3 wire format of eraftpb::Message (proto/proto/eraftpb.proto:23-44, :49-92, :118-132), host code only:
// the decoder behind rg_decode_message / rg_step_bytes (include/raftgroups.h). Kept in a header of its own so that the
// sanitiser harness of the tests (tests/host_check/wire_asan.cpp: -fsanitize=address,undefined over mutated byte strings)
// compiles exactly this code without the HIP runtime.
#pragma once

#include <stdint.h>
#include <string.h>

#include "../../include/raftgroups.h"

typedef uint64_t rg_wire_u64;

// varint: 7 bits per byte, least significant group first, at most 10 bytes for a rg_wire_u64
static inline bool rg_pb_varint(const uint8_t *&p, const uint8_t *end, rg_wire_u64 &v) {
    v = 0;
    for (int shift = 0; shift < 70 && p < end; shift += 7) {
        const uint8_t b = *p++;
        if (shift < 64) v |= (rg_wire_u64)(b & 0x7f) << shift;
        if (!(b & 0x80)) return true;
    }
    return false;
}

// The message-typed and packed fields of eraftpb (proto/proto/eraftpb.proto:23-44, :71-92, :118-132): a real parser
// descends into them, so bytes that are malformed INSIDE an entry or a snapshot fail the whole Message::parse_from_bytes.
enum RgPbSchema { RG_PB_MESSAGE, RG_PB_ENTRY, RG_PB_SNAPSHOT, RG_PB_SNAPSHOT_META, RG_PB_CONF_STATE, RG_PB_OPAQUE };
static inline int rg_pb_child(int schema, rg_wire_u64 field) { // >= 0: length-delimited `field` is a message of that schema; -1: bytes;
    switch (schema) {                           // -2: a packed run of varints (repeated uint64)
    case RG_PB_MESSAGE: return field == 7 ? RG_PB_ENTRY : field == 9 ? RG_PB_SNAPSHOT : -1;
    case RG_PB_SNAPSHOT: return field == 2 ? RG_PB_SNAPSHOT_META : -1;
    case RG_PB_SNAPSHOT_META: return field == 1 ? RG_PB_CONF_STATE : -1;
    case RG_PB_CONF_STATE: return field >= 1 && field <= 4 ? -2 : -1;
    default: return -1;
    }
}
// Walk one message of `schema` in [p, end): structure only, except for the top-level Message, whose fields land in `out`.
// `group` != 0: we are inside an unknown GROUP of that field number and stop at its END_GROUP tag.
static inline bool rg_pb_walk(const uint8_t *&p, const uint8_t *end, int schema, rg_decoded_message *out, rg_wire_u64 group, int depth) {
    if (depth > 64) return false;
    while (p < end) {
        rg_wire_u64 key, v;
        if (!rg_pb_varint(p, end, key)) return false;
        if (key > 0xffffffffULL) return false; // a tag is 32 bits: field numbers end at 2^29 - 1
        const rg_wire_u64 field = key >> 3;
        const uint32_t wt = (uint32_t)(key & 7);
        if (field == 0) return false;
        switch (wt) {
        case 0:
            if (!rg_pb_varint(p, end, v)) return false;
            if (out && schema == RG_PB_MESSAGE && !group) {
                switch (field) {
                case 1: out->msg_type = (uint32_t)v; break;
                case 2: out->to = v; break;
                case 3: out->from = v; break;
                case 4: out->term = v; break;
                case 5: out->log_term = v; break;
                case 6: out->index = v; break;
                case 8: out->commit = v; break;
                case 10: out->reject = v != 0; break;
                case 11: out->reject_hint = v; break;
                case 13: out->request_snapshot = v; break;
                case 14: out->priority = v; break;
                case 15: out->commit_term = v; break;
                default: break; // unknown varint field (or a known one of another wire type): skipped, like protobuf does
                }
            }
            break;
        case 1:
            if (end - p < 8) return false;
            p += 8;
            break;
        case 2: {
            if (!rg_pb_varint(p, end, v) || v > (rg_wire_u64)(end - p)) return false;
            const uint8_t *q = p, *qe = p + v;
            const int child = group ? -1 : rg_pb_child(schema, field);
            if (child >= 0) {
                if (!rg_pb_walk(q, qe, child, nullptr, 0, depth + 1)) return false;
            } else if (child == -2) {
                while (q < qe) {
                    rg_wire_u64 x;
                    if (!rg_pb_varint(q, qe, x)) return false;
                }
            }
            if (out && schema == RG_PB_MESSAGE && !group) {
                if (field == 7) out->n_entries++;
                else if (field == 9) out->has_snapshot = 1;
                else if (field == 12) out->context_len = (uint32_t)v;
            }
            p = qe;
            break;
        }
        case 3: // an unknown GROUP (deprecated; a conforming parser skips a well-formed one, nested groups included)
            if (!rg_pb_walk(p, end, RG_PB_OPAQUE, nullptr, field, depth + 1)) return false;
            break;
        case 4: return group != 0 && field == group; // END_GROUP: it has to close THE group we are in
        case 5:
            if (end - p < 4) return false;
            p += 4;
            break;
        default: return false;
        }
    }
    return group == 0; // (inside a group: ran off the end)
}


// Decode one message; false = not a protobuf-encoded eraftpb::Message (*bad_at = offset where the walk stopped).
static inline bool rg_wire_decode(const uint8_t *bytes, rg_wire_u64 len, rg_decoded_message *out, rg_wire_u64 *bad_at) {
    memset(out, 0, sizeof(*out));
    const uint8_t *p = bytes;
    const bool ok = rg_pb_walk(p, bytes + len, RG_PB_MESSAGE, out, 0, 0);
    if (bad_at) *bad_at = (rg_wire_u64)(p - bytes);
    return ok;
}
