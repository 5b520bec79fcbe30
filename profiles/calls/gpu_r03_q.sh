cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_sendstage_gpu.py tests/test_api_sequences_gpu.py tests/test_scenarios.py tests/test_votes_and_mirror_gpu.py tests/test_cpp_host.py tests/test_sparse_path_gpu.py -m gpu -x -q 2>&1 | tail -12
python tools/bench_flush_latency.py 2>&1 | tail -30
