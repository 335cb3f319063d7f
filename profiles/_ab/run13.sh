timeout -k 10 900 python -m pytest tests/test_gpu_search.py tests/test_gpu_lazy_prm.py -x -q -m gpu 2>&1 | tail -5
PROBE_MODES=device,auto,auto,auto,auto TENDON_HIP_SEARCH_STATS=1 timeout -k 10 300 python profiles/probe_search_quick.py 2>&1 | grep -E "round 1|eager_auto_ms" | cut -c1-1300 | tail -5
PROBE_MODES=device TENDON_HIP_LIB=profiles/_ab/libtendon_hip_clocks.so TENDON_HIP_SEARCH_STATS=1 timeout -k 10 300 python profiles/probe_search_quick.py 2>&1 | grep -E "search steps|inside arcs|search clocks|searches ended" | head -4 | cut -c1-700
