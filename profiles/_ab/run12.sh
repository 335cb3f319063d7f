timeout -k 10 700 python -m pytest tests/test_gpu_retraction.py tests/test_gpu_voxelize.py -x -q -m gpu 2>&1 | tail -3
echo "pool 1024,128,8"; TENDON_HIP_SEARCH_POOL=1024,128,8 PROBE_MODES=auto,auto,auto TENDON_HIP_SEARCH_STATS=1 timeout -k 10 300 python profiles/probe_search_quick.py 2>&1 | grep -E "search state|eager_auto_ms" | cut -c1-700
echo "default pool"; PROBE_MODES=auto,auto,auto TENDON_HIP_SEARCH_STATS=1 timeout -k 10 300 python profiles/probe_search_quick.py 2>&1 | grep -E "search state|eager_auto_ms" | cut -c1-700
