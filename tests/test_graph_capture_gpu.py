"""GPU: the tick path makes no synchronising call, so a caller can capture a stream of rg_tick_device launches into a
hipGraph (here through torch.cuda.graph, with the engine on the capture stream) and replay it -- tools/probe_graph.py checks
the replay against the eager run and times both."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("groups", [1000, 60_000])
def test_ticks_captured_into_a_hipgraph_replay_identically(groups):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "probe_graph.py"), str(groups), "16"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and "GRAPH_OK" in r.stdout, r.stdout


@pytest.mark.gpu
def test_class_placed_ticks_captured_into_a_hipgraph_replay_identically():
    """The one-launch kernel of a shard placed by size class (k_tick_classes) under capture: its launch-order table and class
    bytes are device memory the engine owns, nothing in the launch synchronises."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "probe_graph.py"), "30000", "12", "classes"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and "GRAPH_OK" in r.stdout, r.stdout


@pytest.mark.gpu
def test_fused_calls_with_log_term_ticks_captured_into_a_hipgraph_replay_identically():
    """rg_tick_device_fused whose ticks carry a log-term column, under capture: the call's question to the pre-pass ("did you leave
    a reject to the host?", round 6) is a host wait in eager mode and must not be asked of a stream that is being captured."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "probe_graph.py"), "20000", "16", "fused-logterm"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and "GRAPH_OK" in r.stdout, r.stdout
