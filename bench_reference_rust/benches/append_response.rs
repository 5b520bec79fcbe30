// UNRUN / UNCOMPILED (no Rust toolchain in the build image) -- see ../README.md.
//
// BASELINE.json configs[0]: 1 000 raft groups x 3 peers, MemStorage, synthetic AppendResponse stream on the
// reference Rust CPU path. The stream is the RG_WL_MAJORITY generator of raft_rs_amd/csrc/rg_workload.h.
use criterion::{criterion_group, criterion_main, Criterion, Throughput};
use raft::eraftpb::{Entry, Message, MessageType};
use raft::storage::MemStorage;
use raft::{Config, RawNode, StateRole};

const SEED: u64 = 0x5EED_5EED;

fn splitmix64(mut x: u64) -> u64 {
    x = x.wrapping_add(0x9E37_79B9_7F4A_7C15);
    x = (x ^ (x >> 30)).wrapping_mul(0xBF58_476D_1CE4_E5B9);
    x = (x ^ (x >> 27)).wrapping_mul(0x94D0_49BB_1331_11EB);
    x ^ (x >> 31)
}

fn hash(tick: u64, group: u64, slot: u64) -> u64 {
    splitmix64(SEED ^ (tick << 40) ^ (group << 3) ^ slot)
}

fn new_leader(logger: &slog::Logger) -> RawNode<MemStorage> {
    let storage = MemStorage::new_with_conf_state((vec![1, 2, 3], vec![]));
    let cfg = Config { id: 1, election_tick: 10, heartbeat_tick: 1, max_inflight_msgs: 256, ..Default::default() };
    let mut node = RawNode::new(&cfg, storage, logger).unwrap();
    node.campaign().unwrap();
    // one granted vote makes a majority of 3
    let mut vote = Message::default();
    vote.set_msg_type(MessageType::MsgRequestVoteResponse);
    vote.from = 2;
    vote.to = 1;
    vote.term = node.raft.term;
    node.step(vote).unwrap();
    assert_eq!(node.raft.state, StateRole::Leader);
    drain(&mut node);
    node
}

// Persist what the leader appended and discard the messages it wants to send (the transport is out of
// scope; what is measured is RawNode::step(MsgAppendResponse) -> handle_append_response -> maybe_commit).
fn drain(node: &mut RawNode<MemStorage>) {
    while node.has_ready() {
        let mut rd = node.ready();
        if !rd.entries().is_empty() {
            node.mut_store().wl().append(rd.entries()).unwrap();
        }
        if let Some(hs) = rd.hs() {
            node.mut_store().wl().set_hardstate(hs.clone());
        }
        let _ = rd.take_messages();
        let _ = rd.take_persisted_messages();
        let mut light = node.advance(rd);
        let _ = light.take_messages();
        node.advance_apply();
    }
}

fn tick_group(node: &mut RawNode<MemStorage>, group: u64, tick: u64) {
    // leader appends d = h & 7 entries
    let d = hash(tick + 1, group, 0) & 7;
    for _ in 0..d {
        node.propose(vec![], vec![0u8; 8]).unwrap();
    }
    drain(node); // persists them: on_persist_entries -> maybe_commit
    let last = node.raft.raft_log.last_index();
    let committed = node.raft.raft_log.committed;
    for peer in 2..=3u64 {
        let r = hash(tick + 1, group, peer - 1);
        let matched = node.raft.prs().get(peer).unwrap().matched;
        let u = r % 100;
        let index = if u < 90 {
            std::cmp::min(last, matched + ((r >> 8) & 15))
        } else if u < 95 {
            matched - std::cmp::min(matched, (r >> 8) & 3)
        } else {
            continue;
        };
        let mut m = Message::default();
        m.set_msg_type(MessageType::MsgAppendResponse);
        m.from = peer;
        m.to = 1;
        m.term = node.raft.term;
        m.index = index;
        m.commit = std::cmp::min(committed, index);
        node.step(m).unwrap();
    }
    drain(node);
}

fn bench_append_response(c: &mut Criterion) {
    let logger = raft::default_logger();
    let n_groups = 1000u64;
    let mut nodes: Vec<_> = (0..n_groups).map(|_| new_leader(&logger)).collect();
    let mut tick = 0u64;
    let mut group = c.benchmark_group("RawNode::step(MsgAppendResponse)");
    group.throughput(Throughput::Elements(n_groups));
    group.bench_function("1000 groups x 3 peers, one tick", |b| {
        b.iter(|| {
            for (g, node) in nodes.iter_mut().enumerate() {
                tick_group(node, g as u64, tick);
            }
            tick += 1;
        })
    });
    group.finish();
    let _ = Entry::default();
}

criterion_group!(benches, bench_append_response);
criterion_main!(benches);
