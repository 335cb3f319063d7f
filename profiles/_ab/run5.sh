for sh in 0.25 0.5 1; do for b in 5500 6500 8000; do echo "share=$sh budget=$b"; TENDON_HIP_SEARCH_HOST_SHARE=$sh TENDON_HIP_SEARCH_BUDGET=$b PROBE_MODES=auto,auto,auto,auto TENDON_HIP_SEARCH_STATS=1 timeout -k 10 300 python profiles/probe_search_quick.py 2>&1 | grep -E "eager_auto_ms" | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  eager', d['eager_auto_ms'], 'lazy', d['lazy_auto_ms'], d['eager_auto_where']['handed_back'], d['eager_auto_where']['host_meanwhile'])
"; done; done
