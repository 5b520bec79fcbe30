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
