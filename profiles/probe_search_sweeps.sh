#!/bin/bash
# config 5's 10 000 queries (validity known) under the switches of the device searches: slots, pool, budget, host share -- one process per
# setting (the switches are read when a roadmap's first large round sets the searches up), three shared-schedule solves each, milliseconds
cd ${GRAFT_REPO_ROOT:-.}
run() { echo "== $*"; env "$@" PROBE_MODES=auto,auto,auto timeout -k 10 200 python3 profiles/probe_search_quick.py 2>&1 | grep -o '"eager_auto_ms": \[[0-9., ]*\]\|"handed_back": [0-9]*, "host_meanwhile": [0-9]*' | head -2 | tr '\n' ' '; echo; }
run TENDON_HIP_SEARCH_SLOTS=3072
run TENDON_HIP_SEARCH_SLOTS=2048
run TENDON_HIP_SEARCH_SLOTS=1024
run TENDON_HIP_SEARCH_POOL=2048,256,8
run TENDON_HIP_SEARCH_POOL=1024,128,8
run TENDON_HIP_SEARCH_BUDGET=4000
run TENDON_HIP_SEARCH_BUDGET=5000
run TENDON_HIP_SEARCH_BUDGET=8000
run TENDON_HIP_SEARCH_HOST_SHARE=0.25
run TENDON_HIP_SEARCH_HOST_SHARE=0.5
run TENDON_HIP_SEARCH_HOST_SHARE=2
run TENDON_HIP_SEARCH_K=8
run TENDON_HIP_SEARCH_K=4
run TENDON_HIP_SEARCH_K=1
