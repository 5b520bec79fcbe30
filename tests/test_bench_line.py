"""The ONE line bench.py writes to stdout must fit the driver's record: BENCH_r05.json had `parsed: null` because the line
had grown to 25 KB (the driver keeps an ~8 KB tail of stdout). These tests run complete nested results -- the round-5 line
itself, with more side configurations added, and an N > 1 result with its publication statistics -- through the very
function main() emits with."""
import copy
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


def _full_result():
    """profiles/r05_bench_n1.json is a complete nested result (it WAS the line, 25 KB of it)."""
    return json.load(open(os.path.join(ROOT, "profiles", "r05_bench_n1.json")))


def _check(line_text, bench):
    assert line_text.endswith("\n") and line_text.count("\n") == 1
    assert len(line_text.encode()) <= bench.LINE_LIMIT <= 6000, len(line_text)
    d = json.loads(line_text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "full"):
        assert k in d, k
    for name in ("config", "roofline", "cpu_baseline", "send_stage", "latency_us"):
        for k, v in (d.get(name) or {}).items():
            assert not isinstance(v, (dict, list)), (name, k)  # scalars only: what the driver's record keeps of a sub-object
    for k, v in d.items():
        assert not isinstance(v, list) and (not isinstance(v, dict) or k in ("config", "roofline", "cpu_baseline", "send_stage", "latency_us")), k
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_us"):
        assert k in r, k
    return d


def test_the_line_of_a_full_default_run_fits_and_parses():
    bench = _bench()
    full = _full_result()
    oc = full["other_configs"]
    # round 6 adds side configurations: group commit at 1 M x 5 and config 5 loaded interleaved, then placed
    oc["configs[1] group commit"] = copy.deepcopy(oc["configs[2] joint"])
    oc["configs[4] interleaved, after rg_permute_groups"] = copy.deepcopy(oc["configs[4] one launch, class-sorted"])
    full["roofline"]["by_config"] = bench.by_config_summary(full)
    full["roofline"].update(bench.flat_config_keys(full["roofline"]["by_config"]))
    d = _check(bench.line_text(full), bench)
    r = d["roofline"]
    for n in ("c2_hbm_8M", "c2_resident_2_4M", "c3_joint", "c4_shard", "c5_one_launch", "c5_one_launch_hbm_8M", "c5_size_class",
              "c5_interleaved", "c5_placed", "c2_group_commit", "recompute", "recompute_hbm_8M", "send_two_launch", "send_one_launch"):
        f, us, mb = r[f"frac_{n}"], r[f"us_{n}"], r[f"mb_{n}"]
        assert abs(f - mb * 1e6 / (us * 1e-6) / 8e12) < 2e-3, (n, f, mb, us)  # recomputable from the line alone
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["config1_value_1core"] > 0
    assert abs(d["value"] - d["config"]["groups_per_gpu"] / (d["ms_per_step"] * 1e-3)) < 1e-4 * d["value"]


def test_an_oversized_result_drops_detail_not_the_contract():
    bench = _bench()
    full = _full_result()
    for i in range(10):  # more side configurations than any run has
        for p in ("frac_", "us_", "us_min_", "us_max_", "mb_", "traffic_mb_"):
            full["roofline"][f"{p}made_up_configuration_{i}"] = 0.123456
    d = _check(bench.line_text(full), bench)
    assert "latency_us" not in d and not any(k.startswith("us_min_") for k in d["roofline"])
    assert "frac_made_up_configuration_9" in d["roofline"] and "frac_c2_hbm_8M" in d["roofline"]
    for i in range(400):
        full["roofline"][f"frac_another_{i}"] = 0.5
    try:
        bench.line_text(full)
    except SystemExit as e:
        assert "limit" in str(e)
    else:
        raise AssertionError("a line that cannot be made to fit must not be printed")


def test_the_line_of_a_multi_gpu_run_fits_and_parses():
    bench = _bench()
    full = _full_result()
    for k in ("other_configs", "out_of_cache", "between_regimes", "recompute_only", "recompute_only_out_of_cache", "small_batch_latency"):
        full.pop(k)
    full["roofline"] = {k: v for k, v in full["roofline"].items() if not k.startswith(bench._FLAT_PREFIXES) and k != "by_config"}
    full["n_gpus"], full["cpu_baseline"] = 8, None
    c = full["config"]
    c["workload"] = ("8000000 groups x 7 peers sharded over 8 GPUs (1000000 per GPU), commit indices published every tick "
                     "(BASELINE configs[3]: 8 M x 7 over 8 GPUs)")
    c["sharding"] = ("8 disjoint group ranges, commit indices published every 1 tick(s) through rg_publish_commit: ncclAllGather (RCCL) "
                     "of 1000448 B/rank delta slices (full column: 8000000 B)")
    c["publication"] = {"publications": 55, "full": 1, "delta": 54, "bytes_per_rank_delta": 1000448, "bytes_per_rank_full": 8000000,
                        "escapes": 0, "exchange_us_avg": 41.5, "apply_us_avg": 12.25}
    c["publication_mode"] = "delta slices (~1 B/group)"
    c["publication_compare"] = {"mode": "raw 8 B/group column every tick (RG_PUBLISH_FULL)", "ms_per_step": 0.41, "value": 1.9e10,
                                "bytes_per_rank_per_publication": 8000000}
    c["publish_every"], c["publish_every_auto"] = 1, {"publish_every": 1, "tick_us": 75.1, "exchange_us": 44.0, "rule": "x" * 300}
    c["rccl_ranks"], c["rccl_rank"], c["transport"] = 8, 0, "ncclAllGather (RCCL)"
    d = _check(bench.line_text(full), bench)
    assert d["cpu_baseline"] is None and d["config"]["rccl_ranks"] == 8 and d["config"]["pub_bytes_per_rank_delta"] == 1000448
    assert d["config"]["pub_compare_value"] == 1.9e10 and d["config"]["publish_auto_exchange_us"] == 44.0
