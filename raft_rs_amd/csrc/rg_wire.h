// rg_wire.h -- the proto3 wire format of eraftpb::Message (proto/proto/eraftpb.proto:23-44, :49-92, :118-132), host code only:
// the decoder behind rg_decode_message / rg_step_bytes and the encoder behind rg_encode_message / rg_entry_size /
// rg_limit_size (include/raftgroups.h): bytes in on one side of the path, bytes out on the other. Kept in a header of its
// own so that the sanitiser harness of the tests (tests/host_check/wire_asan.cpp: -fsanitize=address,undefined over mutated
// byte strings, exact-size output buffers) compiles exactly this code without the HIP runtime.
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
// The wire type the schema declares for `field`: 0 varint (integers, bools, enums), 2 length-delimited (bytes, messages),
// 9 = a repeated uint64, which parsers take packed (2) or one varint at a time (0); -1 = not a field of the schema (skipped
// whatever its wire type). A KNOWN field that arrives with another wire type is a parse error in both codecs the reference
// can be built with -- rust-protobuf 2 raises WireError::UnexpectedWireType from the generated merge_from, prost fails
// check_wire_type -- so Message::parse_from_bytes never hands such a message to RawNode::step (a mistyped `term` would
// otherwise decode as 0 and walk past the term gate).
static inline int rg_pb_declared(int schema, rg_wire_u64 field) {
    switch (schema) {
    case RG_PB_MESSAGE: return field == 7 || field == 9 || field == 12 ? 2 : field >= 1 && field <= 15 ? 0 : -1;
    case RG_PB_ENTRY: return field == 4 || field == 6 ? 2 : field >= 1 && field <= 6 ? 0 : -1;
    case RG_PB_SNAPSHOT: return field == 1 || field == 2 ? 2 : -1;
    case RG_PB_SNAPSHOT_META: return field == 1 ? 2 : field == 2 || field == 3 ? 0 : -1;
    case RG_PB_CONF_STATE: return field >= 1 && field <= 4 ? 9 : field == 5 ? 0 : -1;
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
        if (!group) {
            const int want = rg_pb_declared(schema, field);
            if (want >= 0 && !(want == 9 ? (wt == 0 || wt == 2) : wt == (uint32_t)want)) return false;
        }
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
                default: break; // unknown varint field: skipped, like protobuf does
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


// ---- the other direction: what the path SENDS (Raft::send, src/raft.rs:602-662; prepare_send_entries :714-731) ----
// Canonical proto3 serialisation: fields in field-number order, default values omitted, enums and bools as varints --
// byte for byte what the protobuf runtime writes for the same message (tests/test_wire_format.py), and what
// Message::parse_from_bytes of either Rust codec reads back field for field (parsers do not depend on the order).
static inline unsigned rg_pb_varint_len(rg_wire_u64 v) {
    unsigned n = 1;
    while (v >= 0x80) {
        v >>= 7;
        n++;
    }
    return n;
}
static inline uint8_t *rg_pb_put_varint(uint8_t *p, rg_wire_u64 v) {
    while (v >= 0x80) {
        *p++ = (uint8_t)(v | 0x80);
        v >>= 7;
    }
    *p++ = (uint8_t)v;
    return p;
}
// a varint field (tags of eraftpb are one byte: every field number is below 16)
static inline rg_wire_u64 rg_pb_size_u64(rg_wire_u64 v) { return v ? 1u + rg_pb_varint_len(v) : 0u; }
static inline rg_wire_u64 rg_pb_size_bytes(rg_wire_u64 len) { return len ? 1u + rg_pb_varint_len(len) + len : 0u; }
static inline uint8_t *rg_pb_put_u64(uint8_t *p, unsigned field, rg_wire_u64 v) {
    if (!v) return p;
    *p++ = (uint8_t)(field << 3);
    return rg_pb_put_varint(p, v);
}
static inline uint8_t *rg_pb_put_bytes(uint8_t *p, unsigned field, const uint8_t *b, rg_wire_u64 len) {
    if (!len) return p;
    *p++ = (uint8_t)(field << 3 | 2);
    p = rg_pb_put_varint(p, len);
    memcpy(p, b, (size_t)len);
    return p + len;
}

// Entry::compute_size() (eraftpb.proto:23-31): what util::limit_size adds up (src/util.rs:52-76) and what
// rg_log_sizes_write wants accumulated for RG_SEND_BYTES. Entry::default() is 0 bytes.
static inline rg_wire_u64 rg_wire_entry_size(const rg_entry *e) {
    return rg_pb_size_u64(e->entry_type) + rg_pb_size_u64(e->term) + rg_pb_size_u64(e->index) + rg_pb_size_bytes(e->data_len) +
           rg_pb_size_u64(e->sync_log ? 1u : 0u) + rg_pb_size_bytes(e->context_len);
}
static inline uint8_t *rg_wire_put_entry(uint8_t *p, const rg_entry *e) {
    p = rg_pb_put_u64(p, 1, e->entry_type);
    p = rg_pb_put_u64(p, 2, e->term);
    p = rg_pb_put_u64(p, 3, e->index);
    p = rg_pb_put_bytes(p, 4, e->data, e->data_len);
    p = rg_pb_put_u64(p, 5, e->sync_log ? 1u : 0u);
    return rg_pb_put_bytes(p, 6, e->context, e->context_len);
}

// util::limit_size (src/util.rs:52-76) over n entries: how many a message keeps. <= 1 entries and NO_LIMIT (UINT64_MAX) keep
// everything; otherwise entries are taken while the running total stays <= max, EXCEPT that while the total is still 0 the
// next entry is taken unconditionally (so the first always is, and so is whatever follows a run of empty entries).
static inline rg_wire_u64 rg_wire_limit_size(const rg_entry *entries, rg_wire_u64 n, rg_wire_u64 max) {
    if (n <= 1 || max == ~(rg_wire_u64)0) return n;
    rg_wire_u64 size = 0, k = 0;
    for (; k < n; k++) {
        const bool first = size == 0;
        size += rg_wire_entry_size(&entries[k]);
        if (!first && size > max) break;
    }
    return k;
}

// false: a length without its pointer, or an entry / the message beyond what a protobuf message may hold (2 GiB - 1)
static inline bool rg_wire_message_size(const rg_message *m, rg_wire_u64 *len) {
    const rg_wire_u64 cap = 0x7fffffffULL;
    if ((m->n_entries && !m->entries) || (m->snapshot_len && !m->snapshot) || (m->context_len && !m->context)) return false;
    if (m->snapshot_len > cap || m->context_len > cap) return false;
    rg_wire_u64 n = rg_pb_size_u64(m->msg_type) + rg_pb_size_u64(m->to) + rg_pb_size_u64(m->from) + rg_pb_size_u64(m->term) +
                    rg_pb_size_u64(m->log_term) + rg_pb_size_u64(m->index) + rg_pb_size_u64(m->commit) +
                    rg_pb_size_u64(m->reject ? 1u : 0u) + rg_pb_size_u64(m->reject_hint) + rg_pb_size_bytes(m->context_len) +
                    rg_pb_size_u64(m->request_snapshot) + rg_pb_size_u64(m->priority) + rg_pb_size_u64(m->commit_term);
    // Snapshot snapshot = 9: a message field is written whenever it is present, empty or not (has_snapshot)
    if (m->snapshot) n += 1u + rg_pb_varint_len(m->snapshot_len) + m->snapshot_len;
    for (rg_wire_u64 i = 0; i < m->n_entries; i++) {
        const rg_entry *e = &m->entries[i];
        if ((e->data_len && !e->data) || (e->context_len && !e->context) || e->data_len > cap || e->context_len > cap) return false;
        const rg_wire_u64 es = rg_wire_entry_size(e);
        n += 1u + rg_pb_varint_len(es) + es; // a repeated message element is written even when it is Entry::default()
        if (n > cap) return false;
    }
    if (n > cap) return false;
    *len = n;
    return true;
}
// `buf` holds at least rg_wire_message_size() bytes; returns the end of what was written
static inline uint8_t *rg_wire_encode(const rg_message *m, uint8_t *buf) {
    uint8_t *p = buf;
    p = rg_pb_put_u64(p, 1, m->msg_type);
    p = rg_pb_put_u64(p, 2, m->to);
    p = rg_pb_put_u64(p, 3, m->from);
    p = rg_pb_put_u64(p, 4, m->term);
    p = rg_pb_put_u64(p, 5, m->log_term);
    p = rg_pb_put_u64(p, 6, m->index);
    for (rg_wire_u64 i = 0; i < m->n_entries; i++) {
        const rg_entry *e = &m->entries[i];
        *p++ = (uint8_t)(7u << 3 | 2);
        p = rg_pb_put_varint(p, rg_wire_entry_size(e));
        p = rg_wire_put_entry(p, e);
    }
    p = rg_pb_put_u64(p, 8, m->commit);
    if (m->snapshot) {
        *p++ = (uint8_t)(9u << 3 | 2);
        p = rg_pb_put_varint(p, m->snapshot_len);
        if (m->snapshot_len) memcpy(p, m->snapshot, (size_t)m->snapshot_len);
        p += m->snapshot_len;
    }
    p = rg_pb_put_u64(p, 10, m->reject ? 1u : 0u);
    p = rg_pb_put_u64(p, 11, m->reject_hint);
    p = rg_pb_put_bytes(p, 12, m->context, m->context_len);
    p = rg_pb_put_u64(p, 13, m->request_snapshot);
    p = rg_pb_put_u64(p, 14, m->priority);
    return rg_pb_put_u64(p, 15, m->commit_term);
}
