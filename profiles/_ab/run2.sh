timeout -k 10 600 python -m pytest tests/test_gpu_search.py tests/test_gpu_lazy_prm.py -x -q -m gpu 2>&1 | tail -5
PROBE_MODES=host TENDON_HIP_SEARCH_HIST=1 timeout -k 10 300 python profiles/probe_search_quick.py 2>&1 | grep -E "found|not found" | head -4
PROBE_MODES=auto,auto,auto,auto,auto,auto TENDON_HIP_SEARCH_STATS=1 timeout -k 10 300 python profiles/probe_search_quick.py 2>&1 | grep -E "round|eager_auto_ms" | head -40
