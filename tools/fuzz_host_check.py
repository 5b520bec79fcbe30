#!/usr/bin/env python3
"""Long-running seeded fuzz of the kernels' arithmetic (rg_group.h compiled for the host: tests/host_check) against the oracle --
the body of tests/test_host_check.py::test_device_arithmetic_on_host_matches_oracle with FRESH seeds, for as long as asked:

    python tools/fuzz_host_check.py [--seconds 600] [--seed0 N] [--send]

--send: the tick AND its send stage (device Inflights, compact / ring windows, work items) in both forms -- two launches'
code and the one-launch lane code -- against the oracle with its own Inflights (test_send_stage_on_host_matches_oracle's body).

P = 1..8, joint configurations, group commit on / off, slots without a Progress, small and full-range index values, malformed
events on some ticks. CPU only (not part of the suite: minutes). Prints one line per 50 rounds and the totals; exits 1 on the
first mismatch with the seed that reproduces it."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import fuzz  # noqa: E402
import oracle_lib as O  # noqa: E402
import test_host_check as H  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=600)
    ap.add_argument("--seed0", type=int, default=int(time.time()))
    ap.add_argument("--send", action="store_true")
    args = ap.parse_args()
    if args.send:
        return send_mode(args)
    tick = H.host_tick.__wrapped__() if hasattr(H.host_tick, "__wrapped__") else None
    if tick is None:
        raise SystemExit("cannot reach the host_tick fixture body")
    t0, rounds, group_ticks = time.time(), 0, 0
    seed = args.seed0
    while time.time() - t0 < args.seconds:
        seed += 1
        rng = np.random.default_rng(seed)
        P = int(rng.integers(1, 9))
        gc = bool(rng.integers(0, 2))
        G = int(rng.integers(200, 6000))
        small = bool(rng.integers(0, 4))  # (one round in four with full-range values)
        st = O.alloc_state(G, P)
        st["cfg"][:] = fuzz.random_cfg(rng, G, P, missing_progress_frac=float(rng.choice([0.0, 0.05, 0.3])),
                                       group_commit_frac=0.5 if gc else 0.0)
        fuzz.random_state(rng, st, small_values=small, with_gids=gc)
        cl = O.Cluster(G)
        cl.load_soa(st, term=6)
        eng = H.copy_state(st)
        msgs = O.alloc_msgs(G, P)
        gout = np.zeros(G, dtype=np.uint32)
        out = np.zeros(G, dtype=np.uint32)
        T = int(rng.integers(3, 9))
        for t in range(T):
            cl.store_soa(st)
            fuzz.random_msgs(rng, st, msgs, malformed_p=0.03 if rng.integers(0, 3) == 0 else 0.0)
            tick(eng, msgs, out, gc)
            cl.tick_soa(msgs, gout)
            cl.store_soa(st)
            diffs = fuzz.diff_states(st, eng, G, P)
            if diffs or not (out == gout).all():
                print(f"MISMATCH seed {seed} P {P} gc {gc} G {G} small {small} tick {t}: {diffs[:6]} "
                      f"out {np.nonzero(out != gout)[0][:5]}", flush=True)
                sys.exit(1)
            group_ticks += G
        rounds += 1
        if rounds % 50 == 0:
            print(f"{rounds} rounds, {group_ticks} group-ticks, {time.time() - t0:.0f} s, last seed {seed}", flush=True)
    print(f"clean: {rounds} rounds, {group_ticks} group-ticks, seeds {args.seed0 + 1}..{seed}")


def _fixture(f):
    return f.__wrapped__()


def send_mode(args):
    tick, send, tick_send = _fixture(H.host_tick), _fixture(H.host_send), _fixture(H.host_tick_send)
    t0, rounds, group_ticks, seed = time.time(), 0, 0, args.seed0
    tot = {}
    while time.time() - t0 < args.seconds:
        seed += 1
        rng = np.random.default_rng(seed)
        P = int(rng.integers(1, 9))
        cap = int(rng.choice([1, 2, 3, 4, 5, 6, 8, 16, 256]))
        max_entries = int(rng.choice([0, 0, 1, 2, 3, 7]))
        fused = bool(rng.integers(0, 2))
        G, ticks = (int(rng.integers(100, 400)), int(rng.integers(20, 70))) if rng.integers(0, 2) else (int(rng.integers(500, 2000)), int(rng.integers(5, 14)))
        try:
            seen = H.send_stage_round(rng, tick, send, tick_send, fused, P, cap, max_entries, G, ticks)
        except AssertionError as e:
            print(f"MISMATCH seed {seed} P {P} cap {cap} max_entries {max_entries} fused {fused} G {G} ticks {ticks}: {str(e)[:600]}", flush=True)
            sys.exit(1)
        for k, v in seen.items():
            tot[k] = tot.get(k, 0) + v
        group_ticks += G * ticks
        rounds += 1
        if rounds % 20 == 0:
            print(f"{rounds} rounds, {group_ticks} group-ticks, {time.time() - t0:.0f} s, last seed {seed}, {tot}", flush=True)
    print(f"clean: {rounds} rounds, {group_ticks} group-ticks, seeds {args.seed0 + 1}..{seed}, {tot}")


if __name__ == "__main__":
    main()
