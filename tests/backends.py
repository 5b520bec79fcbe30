"""Two interchangeable single-group "leader" backends for scenario tests that read like the
reference's own tests (harness/tests/integration_cases/*.rs): the CPU oracle and the HIP engine.

Peer id i lives in slot i-1. Test infrastructure.
"""
import numpy as np

import oracle_lib as O

PROBE, REPLICATE, SNAPSHOT = 0, 1, 2


class OracleLeader:
    name = "oracle"

    def __init__(self, self_id, term, voters, outgoing=(), learners=(), log=(), committed=0, dummy=(0, 0),
                 next_idx=1, max_inflight=256, max_entries=0, max_bytes=None, entry_bytes=None):
        self.cl = O.Cluster(1).config(0, self_id, term, voters, outgoing, learners, next_idx=next_idx,
                                      max_inflight=max_inflight)
        self.cl.set_log(0, list(log), committed=committed, dummy=dummy)
        self.term = term
        self._voters, self._self_id = list(voters), self_id
        self.max_inflight, self.max_entries = max_inflight, max_entries
        self.skip_bcast_commit = False
        # Config::max_size_per_msg in bytes: entry_bytes(index) = Entry::compute_size() of that entry
        self.entry_bytes = entry_bytes
        if max_bytes is not None:
            self.max_entries = max_bytes
            self.cl.set_limit_bytes(True)
            self._sized = dummy[0]
            self._feed_sizes()

    def _feed_sizes(self):
        last = self.cl.last_index(0)
        if last > self._sized:
            self.cl.append_entry_sizes(0, self._sized + 1, [self.entry_bytes(i) for i in range(self._sized + 1, last + 1)])
            self._sized = last

    # ---- flow control: the Progress's own Inflights + the send decisions (test_raft_flow_control.rs) ----
    def _send(self, out_word):
        """Serve the send requests of one step the way the reference does right after it."""
        m = self.cl.send_stage_soa(np.array([out_word], dtype=np.uint32), self.max_entries, capacity=1 << 16,
                                   skip_bcast_commit=self.skip_bcast_commit)
        return [(int(x["to"]), int(x["kind"]), int(x["index"]), int(x["n_entries"])) for x in m]

    def propose(self, n=1):
        """step(MsgPropose): append_entry + bcast_append (raft.rs:2044-2053). Returns the messages sent."""
        self.append(n)
        return self._send(0x8)

    def ack(self, from_, index, commit=0):
        """step(MsgAppendResponse) followed by its sends. Returns the messages sent."""
        o = self.cl.step(0, from_, index, commit, ins_full=-1)
        s = from_ - 1
        word = (1 if o.commit_changed else 0) | (int(o.send_append) << (8 + s)) | (int(o.send_more) << (16 + s))
        return self._send(word)

    def reject(self, from_, index, reject_hint=0, commit=0, request_snapshot=0):
        """step(MsgAppendResponse{reject}) followed by its sends."""
        o = self.cl.step(0, from_, index, commit, True, reject_hint, request_snapshot, ins_full=-1)
        return self._send(int(o.send_append) << (8 + from_ - 1))

    def become_snapshot(self, pid, snapshot_index):
        """The host's half of a snapshot send: Progress::become_snapshot (progress.rs:117-121)."""
        import ctypes as C
        O.lib().ro_progress_become_snapshot(C.byref(self.cl.pr(0, pid)), snapshot_index)

    def heartbeat_response(self, from_, commit=0):
        o = O.Out()
        O.lib().ro_handle_heartbeat_response(self.cl.h, 0, from_, commit, -1, o)
        return self._send(int(o.send_append) << (8 + from_ - 1))

    def set_pending_conf(self, on):
        O.lib().ro_group_set_pending_conf(self.cl.h, 0, on)

    def ins_full(self, pid):
        return bool(O.lib().ro_ins_full(self.cl.pr(0, pid).ins))

    def inflights(self, pid):
        return self.cl.ins_contents(0, pid)

    def set_progress(self, pid, **kw):
        p = self.cl.pr(0, pid)
        for k, v in kw.items():
            setattr(p, {"match": "matched", "next": "next_idx"}.get(k, k), v)

    def progress(self, pid):
        p = self.cl.pr(0, pid)
        return {"match": p.matched, "next": p.next_idx, "state": p.state, "paused": bool(p.paused),
                "pending_snapshot": p.pending_snapshot, "pending_request_snapshot": p.pending_request_snapshot,
                "recent_active": bool(p.recent_active), "committed_index": p.committed_index}

    def committed(self):
        return self.cl.committed(0)

    def remove_node(self, pid):
        # rebuild the group without `pid`, keeping every other Progress and the log (apply_conf Remove)
        L = O.lib()
        g = self.cl
        keep = [i for i in self._voters if i != pid]
        saved = {i: self.progress(i) for i in keep}
        committed, last = g.committed(0), g.last_index(0)
        entries = [(L.ro_log_term(g.h, 0, i), i) for i in range(1, last + 1)]
        g.config(0, self._self_id, self.term, keep)
        g.set_log(0, entries, committed=committed)
        for i, pr in saved.items():
            self.set_progress(i, match=pr["match"], next=pr["next"], state=pr["state"], paused=pr["paused"],
                              committed_index=pr["committed_index"], recent_active=pr["recent_active"])
        self._voters = keep

    def enable_group_commit(self, on):
        O.lib().ro_group_set_group_commit(self.cl.h, 0, on)

    def set_transferee(self, pid):
        O.lib().ro_group_set_transferee(self.cl.h, 0, pid)

    def maybe_commit(self):
        return self.cl.maybe_commit(0)

    def mci(self):
        return self.cl.mci(0)

    def append(self, n):
        O.lib().ro_group_append(self.cl.h, 0, n)
        if self.entry_bytes:
            self._feed_sizes()

    def become_leader(self, term):
        """become_candidate + become_leader (raft.rs:1113-1202): Raft::reset(term), the leader's own Progress to
        Replicate, the new leader's empty entry; then bcast_append (raft.rs:2190-2191). Returns the messages sent."""
        assert O.lib().ro_group_become_leader(self.cl.h, 0, term) == 0
        self.term = term

    def become_leader_and_bcast(self, term):
        self.become_leader(term)
        return self._send(0x18)

    def persisted(self, index):
        return O.lib().ro_on_persist_entries(self.cl.h, 0, index)

    def sent(self, pid):
        return O.lib().ro_progress_update_state(self.cl.pr(0, pid), self.cl.last_index(0))

    def heartbeat_commit(self, to):
        return O.lib().ro_heartbeat_commit(self.cl.h, 0, to)

    def report_unreachable(self, pid):
        """RawNode::report_unreachable (raw_node.rs:692-698): step(MsgUnreachable) -> handle_unreachable."""
        return O.lib().ro_handle_unreachable(self.cl.h, 0, pid)

    def report_snapshot(self, pid, failure):
        """RawNode::report_snapshot (raw_node.rs:701-709): step(MsgSnapStatus { reject: failure })."""
        return O.lib().ro_handle_snapshot_status(self.cl.h, 0, pid, failure)

    def step_heartbeat_response(self, from_, commit=0, ins_full=False):
        o = O.Out()
        O.lib().ro_handle_heartbeat_response(self.cl.h, 0, from_, commit, 1 if ins_full else 0, o)
        return {"send_append": bool(o.send_append), "free_first_one": bool(o.free_to)}

    def log_term(self, idx):
        return O.lib().ro_log_term(self.cl.h, 0, idx)

    def step(self, from_, index, reject=False, reject_hint=0, commit=0, request_snapshot=0, ins_full=False,
             log_term=0):
        o = self.cl.step(0, from_, index, commit, reject, reject_hint, request_snapshot, 1 if ins_full else 0,
                         log_term)
        return {"send_append": bool(o.send_append), "send_more": bool(o.send_more),
                "changed": bool(o.commit_changed), "free_to": bool(o.free_to), "timeout_now": bool(o.timeout_now)}


class EngineLeader:
    """Same surface over the HIP engine: one group, one message per tick through the C ABI."""
    name = "engine"

    def __init__(self, self_id, term, voters, outgoing=(), learners=(), log=(), committed=0, dummy=(0, 0),
                 next_idx=1, n_slots=None, max_inflight=0, max_entries=0, max_bytes=None, entry_bytes=None):
        import raft_rs_amd as rg
        self.rg = rg
        ids = sorted(set(voters) | set(outgoing) | set(learners))
        self.P = n_slots or max(ids)
        self.eng = rg.Engine(1, self.P, max_inflight=max_inflight)
        self.max_inflight, self.max_entries = max_inflight, max_entries
        self.skip_bcast_commit = False
        self.term = term
        self.self_id = self_id
        mask = lambda s: sum(1 << (i - 1) for i in s)
        self.cfg = dict(incoming=mask(voters), outgoing=mask(outgoing), self_slot=self_id - 1,
                        group_commit=False, transferee_plus1=0, present=mask(ids))
        st = O.alloc_state(1, self.P, stride=self.eng.stride)
        for i in ids:
            st["next"][i - 1, 0] = next_idx
        # current-term range from the log (entries of `term` are the contiguous tail)
        log = list(log)
        last = log[-1][1] if log else dummy[0]
        # the leader's own run = the contiguous TAIL of entries at `term` (some reference tests build logs whose
        # older entries reuse the leader's term number; only the tail run is "the leader's entries")
        cur = []
        for t, idx in reversed(log):
            if t != term:
                break
            cur.insert(0, idx)
        if not cur:  # test_commit-style logs: entries of `term` that are not the tail
            cur = [idx for (t, idx) in log if t == term]
        # For a real leader the entries of its term are the log tail and term_hi == last_index. A few
        # reference tests (test_commit) build logs whose tail has a HIGHER term than the leader's; the
        # gate term(mci) == term is then still the contiguous range [cur[0], cur[-1]].
        lo, hi = (cur[0], cur[-1]) if cur else (last + 1, last)
        st["term_lo"][0], st["term_hi"][0], st["commit"][0] = lo, hi, committed
        # compact term-run table of the entries below the leader's own run (find_conflict_by_term on device)
        O.add_term_table(st)
        self.log = {idx: t for t, idx in log}
        self.dummy = dummy
        older = [(t, idx) for t, idx in log if idx < lo]
        runs = []
        for t, idx in older:
            if not runs or runs[-1][1] != t:
                runs.append((idx, t))
        self.table_ok = len(runs) <= 4 and (not cur or cur[-1] == last)
        if self.table_ok:
            for k, (f, t) in enumerate(runs):
                st["run_first"][k, 0], st["run_term"][k, 0] = f, t
        st["dummy_index"][0], st["dummy_term"][0], st["cur_term"][0] = dummy[0], dummy[1], term
        st["cfg"][0] = rg.cfg_make(**self.cfg)
        self.eng.load_state(st)
        self.msgs = rg.MsgBuffers(1, self.P, self.eng.stride)
        # Config::max_size_per_msg in bytes (RG_SEND_BYTES): the host writes the cumulative size of every entry it appends
        self.max_bytes, self.entry_bytes = max_bytes, entry_bytes
        if max_bytes is not None:
            self.eng.log_sizes_enable(64)
            self._cum = {dummy[0]: 0}
            self._write_sizes(hi)

    def _write_sizes(self, last):
        new = [i for i in range(max(self._cum) + 1, last + 1)]
        for i in new:
            self._cum[i] = self._cum[i - 1] + self.entry_bytes(i)
        recs = np.zeros(len(new) + 1, dtype=self.rg.engine.LOG_SIZE_DTYPE)
        for k, i in enumerate([min(self._cum)] + new):  # (the base of the sums rides along: harmless)
            recs[k] = (0, i, self._cum[i])
        self.eng.log_sizes_write(recs)

    def _push_cfg(self):
        self.eng.set_config(0, self.rg.cfg_make(**self.cfg))

    def remove_node(self, pid):
        """apply_conf_change(RemoveNode) for a simple (non-joint) config: tracker.rs:380-397."""
        bit = 1 << (pid - 1)
        self.cfg["incoming"] &= ~bit
        self.cfg["present"] &= ~bit
        self._push_cfg()

    def set_progress(self, pid, **kw):
        cell = {"group": 0, "slot": pid - 1}
        names = {"match": "match", "next": "next", "pending_snapshot": "pend_snap",
                 "pending_request_snapshot": "pend_rs", "commit_group_id": "gid", "committed_index": "pr_commit"}
        flags = None
        for k, v in kw.items():
            if k in names:
                cell[names[k]] = v
            elif k in ("state", "paused", "recent_active"):
                if flags is None:
                    flags = int(self.eng.read_column(self.rg.COL.PFLAGS)[0, pid - 1])
                if k == "state":
                    flags = (flags & ~3) | v
                elif k == "paused":
                    flags = (flags & ~4) | (4 if v else 0)
                else:
                    flags = (flags & ~8) | (8 if v else 0)
            else:
                raise KeyError(k)
        if flags is not None:
            cell["pflags"] = flags
        self.eng.write_cells([cell])

    def progress(self, pid):
        st = self.eng.read_state()
        f = int(st["pflags"][0, pid - 1])
        s = pid - 1
        return {"match": int(st["match"][s, 0]), "next": int(st["next"][s, 0]), "state": f & 3,
                "paused": bool(f & 4), "pending_snapshot": int(st["pend_snap"][s, 0]),
                "pending_request_snapshot": int(st["pend_rs"][s, 0]), "recent_active": bool(f & 8),
                "committed_index": int(st["pr_commit"][s, 0])}

    def committed(self):
        return int(self.eng.read_column(self.rg.COL.COMMIT)[0])

    def enable_group_commit(self, on):
        self.cfg["group_commit"] = on
        self._push_cfg()

    def set_transferee(self, pid):
        self.cfg["transferee_plus1"] = pid
        self._push_cfg()

    def maybe_commit(self):
        self.eng.recompute()
        _, out = self.eng.results()
        return bool(out[0] & 1)

    def mci(self):
        m, f = self.eng.maximal_committed_index(with_flag=True)
        return int(m[0]), bool(f[0])

    def _tick(self):
        self.eng.tick(self.msgs)
        self.msgs.clear()
        _, out = self.eng.results()
        return int(out[0])

    # ---- flow control: Inflights on the device + the send stage (needs max_inflight > 0) ----
    def _send(self):
        """rg_send_appends for the tick that just ran; the per-peer items expanded to single messages."""
        if self.max_bytes is not None:
            self._write_sizes(int(self.eng.read_column(self.rg.COL.TERM_HI)[0]))
        self.eng.send_appends(self.max_entries, skip_bcast_commit=self.skip_bcast_commit, max_bytes=self.max_bytes)
        msgs = []
        for it in self.eng.send_items():
            to, kind, prev, last, n = int(it["slot"]) + 1, int(it["kind"]), int(it["prev_index"]), int(it["last_index"]), int(it["n_msgs"])
            if kind == self.rg.engine.SEND_SNAPSHOT:
                msgs.append((to, kind, prev, 0))
                continue
            E = self.max_entries
            for k in range(n):
                if self.max_bytes is not None:
                    # what the host does when it builds the messages of an item: util::limit_size over its own entries
                    cnt, size = 0, 0
                    for i in range(prev + 1, last + 1):
                        if size == 0 or size + self.entry_bytes(i) <= self.max_bytes:
                            size += self.entry_bytes(i)
                            cnt += 1
                        else:
                            break
                else:
                    cnt = (last - prev) if not E else min(E, last - prev)
                msgs.append((to, kind, prev, cnt))
                prev += cnt
            assert prev == last
        return sorted(msgs)

    def propose(self, n=1):
        hi = int(self.eng.read_column(self.rg.COL.TERM_HI)[0])
        s = self.self_id - 1
        self.msgs.m_commit[s, 0] = hi + n
        self.msgs.m_flags[0, s] = self.rg.MF.APPEND
        out = self._tick()
        assert out & self.rg.OUT.APPENDED
        return self._send()

    def ack(self, from_, index, commit=0):
        s = from_ - 1
        self.msgs.m_index[s, 0] = index
        self.msgs.m_commit[s, 0] = commit
        self.msgs.m_flags[0, s] = self.rg.MF.VALID
        self._tick()
        return self._send()

    def reject(self, from_, index, reject_hint=0, commit=0, request_snapshot=0):
        s = from_ - 1
        self.msgs.m_index[s, 0] = index
        self.msgs.m_commit[s, 0] = commit
        self.msgs.m_hint[s, 0] = reject_hint
        self.msgs.m_rs[s, 0] = request_snapshot
        self.msgs.m_flags[0, s] = self.rg.MF.VALID | self.rg.MF.REJECT | (self.rg.MF.HAS_RS if request_snapshot else 0)
        self._tick()
        return self._send()

    def become_snapshot(self, pid, snapshot_index):
        f = int(self.eng.read_column(self.rg.COL.PFLAGS)[0, pid - 1])
        self.eng.write_cells([{"group": 0, "slot": pid - 1, "pend_snap": snapshot_index,
                               "pflags": (f & ~0x7) | SNAPSHOT}])

    def heartbeat_response(self, from_, commit=0):
        s = from_ - 1
        self.msgs.m_commit[s, 0] = commit
        self.msgs.m_flags[0, s] = self.rg.MF.HEARTBEAT
        self._tick()
        return self._send()

    def set_pending_conf(self, on):
        """Raft::has_pending_conf(): flag bit RG_PF_PENDING_CONF on the leader's own slot."""
        s = self.self_id - 1
        f = int(self.eng.read_column(self.rg.COL.PFLAGS)[0, s])
        f = (f | self.rg.PF.PENDING_CONF) if on else (f & ~self.rg.PF.PENDING_CONF)
        self.eng.write_cells([{"group": 0, "slot": s, "pflags": f}])

    def ins_full(self, pid):
        return bool(int(self.eng.read_column(self.rg.COL.PFLAGS)[0, pid - 1]) & self.rg.PF.INS_FULL)

    def inflights(self, pid):
        return self.eng.inflights(0, pid - 1)

    def append(self, n):
        hi = int(self.eng.read_column(self.rg.COL.TERM_HI)[0])
        s = self.self_id - 1
        self.msgs.m_commit[s, 0] = hi + n
        self.msgs.m_flags[0, s] = self.rg.MF.APPEND
        self._tick()
        for k in range(hi + 1, hi + n + 1):
            self.log[k] = self.term

    def become_leader(self, term):
        """The RG_MF_BECOME_LEADER event on the leader's own slot (new term in m_hint), then the send stage when the
        engine holds the Inflights."""
        s = self.self_id - 1
        self.msgs.m_hint[s, 0] = term
        self.msgs.m_flags[0, s] = self.rg.MF.BECOME_LEADER
        out = self._tick()
        assert out & self.rg.OUT.BECAME_LEADER and out & self.rg.OUT.APPENDED and not out & self.rg.OUT.FAULT, hex(out)
        self.term = term
        self.log[int(self.eng.read_column(self.rg.COL.TERM_HI)[0])] = term  # the new leader's empty entry

    def become_leader_and_bcast(self, term):
        self.become_leader(term)
        return self._send()

    def persisted(self, index):
        s = self.self_id - 1
        self.msgs.m_index[s, 0] = index
        self.msgs.m_flags[0, s] = self.rg.MF.VALID
        return bool(self._tick() & 1)

    def sent(self, pid):
        self.msgs.m_flags[0, pid - 1] = self.rg.MF.SENT
        return -1 if (self._tick() & 2) else 0

    def report_unreachable(self, pid):
        self.eng.progress_events([(0, pid - 1, self.rg.engine.EV_UNREACHABLE)])

    def report_snapshot(self, pid, failure):
        E = self.rg.engine
        self.eng.progress_events([(0, pid - 1, E.EV_SNAPSHOT_FAILURE if failure else E.EV_SNAPSHOT_FINISH)])

    def heartbeat_commit(self, to):
        return int(self.eng.heartbeat_commits()[to - 1, 0])

    def step_heartbeat_response(self, from_, commit=0, ins_full=False):
        MF = self.rg.MF
        s = from_ - 1
        self.msgs.m_commit[s, 0] = commit
        self.msgs.m_flags[0, s] = MF.HEARTBEAT | (MF.INS_FULL if ins_full else 0)
        out = self._tick()
        return {"send_append": bool((out >> (8 + s)) & 1), "free_first_one": bool((out >> (24 + s)) & 1)}

    def log_term(self, idx):
        if idx == self.dummy[0]:
            return self.dummy[1]
        return self.log.get(idx, 0)

    def _find_conflict_on_host(self, index, term):
        """What a host whose log has more term runs than the device table does itself (raft_log.rs:209-235)."""
        last = max(self.log) if self.log else self.dummy[0]
        if index > last:
            return index
        ci = index
        while True:
            t = 0 if (ci < self.dummy[0] or ci > last) else self.log_term(ci)
            if t > term:
                ci -= 1
            else:
                return ci

    def step(self, from_, index, reject=False, reject_hint=0, commit=0, request_snapshot=0, ins_full=False,
             log_term=0):
        MF = self.rg.MF
        s = from_ - 1
        if reject and log_term and not self.table_ok:
            reject_hint, log_term = self._find_conflict_on_host(reject_hint, log_term), 0
        self.msgs.m_index[s, 0] = index
        self.msgs.m_commit[s, 0] = commit
        self.msgs.m_hint[s, 0] = reject_hint
        self.msgs.m_rs[s, 0] = request_snapshot
        self.msgs.m_logterm[s, 0] = log_term
        self.msgs.m_flags[0, s] = (MF.VALID | (MF.REJECT if reject else 0) | (MF.HAS_RS if request_snapshot else 0) |
                                   (MF.INS_FULL if ins_full else 0) | (MF.HAS_LOGTERM if (reject and log_term) else 0))
        out = self._tick()
        return {"send_append": bool((out >> (8 + s)) & 1), "send_more": bool((out >> (16 + s)) & 1),
                "changed": bool(out & 1), "free_to": bool((out >> (24 + s)) & 1), "timeout_now": bool(out & 4)}
