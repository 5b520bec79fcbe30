"""The C++ host side above the C ABI (include/raftgroups.hpp: RawNode::step / ready surface, Progress, Error under the
reference's names): examples/cpp_reference_tests.cpp restates reference tests against it -- test_leader_append_response
(test_raft.rs:2611-2675), test_leader_acknowledge_commit (test_raft_paper.rs:499-534), test_msg_app_flow_control_full
(test_raft_flow_control.rs:24-58), RawNode::step's error behaviour (raw_node.rs:402-411) -- and runs them on the GPU.
On a CPU-only host the program must build warning-free and fail loudly. Round 3 adds the send side: test_leader_start_replication
(test_raft_paper.rs:425-457) and test_progress_flow_control (test_raft.rs:369-435, max_size_per_msg in bytes) checked on the
MESSAGES MultiRaft::messages builds out of a Storage, and examples/cpp_message_builder.cpp -- the host half alone (Storage,
limit_size, build_messages, Message::write_to_bytes against test_storage_entries, test_slice and a restated send loop), which
needs no device and runs in the CPU suite. Also restated for the GPU run: test_recv_msg_unreachable (test_raft.rs:2913-2933),
test_snapshot_failure / test_snapshot_succeed (test_raft_snap.rs:68-109) through MultiRaft::report_unreachable / report_snapshot,
test_bcast_beat (test_raft.rs:2680-2752) through MultiRaft::bcast_heartbeat. Round 5: three rows of test_fast_log_rejection
(test_raft.rs:5573-5839) through MultiRaft::ready(Storage &) -- the rejection's find_conflict_by_term answered from the HOST's log
where the device hands it back (RG_OUT_HOST_HINT), with the Inflights on the host and on the device; ready() without a Storage
refuses such a batch."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(tmp_path, rg, name="cpp_reference_tests"):
    exe = str(tmp_path / name)
    libdir = os.path.dirname(rg.LIB_PATH)
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", name + ".cpp"), "-o", exe, "-L", libdir, "-lraftgroups",
           "-Wl,-rpath," + libdir]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    return exe


def test_cpp_host_side_builds_and_fails_loudly_without_a_gpu(tmp_path, rg):
    exe = build(tmp_path, rg)
    if rg.load_library().rg_device_count() == 0:
        r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 2 and "no CPU fallback" in r.stdout, r.stdout


def test_message_builder_on_the_host(tmp_path, rg):
    """From send items to the reference's Messages and their bytes: pure host code around the engine (no device)."""
    exe = build(tmp_path, rg, "cpp_message_builder")
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "CPP_MESSAGE_BUILDER_OK" in r.stdout, r.stdout


@pytest.mark.gpu
def test_reference_tests_through_the_cpp_host_side(tmp_path, rg):
    exe = build(tmp_path, rg)
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0 and "CPP_REFERENCE_TESTS_OK" in r.stdout, r.stdout
