for slots in 2048 3072 4096; do for b in 4000 6000; do echo "slots=$slots budget=$b"; TENDON_HIP_SEARCH_SLOTS=$slots TENDON_HIP_SEARCH_BUDGET=$b PROBE_MODES=auto,auto,auto TENDON_HIP_SEARCH_STATS=1 timeout -k 10 300 python profiles/probe_search_quick.py 2>&1 | grep -E "eager_auto_ms" | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  eager', d['eager_auto_ms'], 'lazy', d['lazy_auto_ms'], d['eager_auto_where']['handed_back'])
"; done; done
echo "device-only clocks at 3072 slots"; TENDON_HIP_SEARCH_SLOTS=3072 PROBE_MODES=device TENDON_HIP_LIB=profiles/_ab/libtendon_hip_clocks.so TENDON_HIP_SEARCH_STATS=1 timeout -k 10 300 python profiles/probe_search_quick.py 2>&1 | grep -E "search steps|search clocks" | head -2
echo "codec tests"; timeout -k 10 600 python -m pytest tests/test_gpu_sampling.py -x -q -m gpu -k "wire" 2>&1 | tail -3
