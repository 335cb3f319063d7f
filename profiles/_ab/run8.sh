timeout -k 10 600 python -m pytest tests/test_gpu_sampling.py -x -q -m gpu -k "wire" 2>&1 | tail -3
PROBE_MODES=device TENDON_HIP_LIB=profiles/_ab/libtendon_hip_clocks.so TENDON_HIP_SEARCH_STATS=1 timeout -k 10 300 python profiles/probe_search_quick.py 2>&1 | grep -E "search steps|inside arcs|search clocks" | head -3
timeout -k 10 500 python bench.py --workload config4 --emulate-world 8 --steps 2 --warmup 1 2>/dev/null | python3 -c "
import sys, json
d=json.loads(sys.stdin.read()); print(json.dumps(d.get('signature_wire')))"
