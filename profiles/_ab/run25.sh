PROBE_MODES=host,host,auto,auto,auto TENDON_HIP_SEARCH_STATS=1 timeout -k 10 500 python profiles/probe_search_quick.py 2>&1 | grep -E "round 1:|_ms" | cut -c1-330
