set -e
timeout -k 10 800 python -m pytest tests/test_gpu_search.py tests/test_gpu_edges.py tests/test_gpu_lazy_prm.py tests/test_cpp_shim.py -m gpu -x -q 2>&1 | tail -15
PROBE_MODES=auto,auto,auto TENDON_HIP_SEARCH_STATS=1 timeout -k 10 300 python profiles/probe_search_quick.py 2>&1 | grep -E "search state|search graph|_ms" | cut -c1-400
