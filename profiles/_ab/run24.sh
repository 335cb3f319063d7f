set -e
timeout -k 10 600 python -m pytest tests/test_gpu_edges.py -m gpu -x -q 2>&1 | tail -3
echo "== classes on"; TENDON_HIP_EDGE_TIMING=1 timeout -k 10 300 python profiles/probe_edge_vs_verdict.py 2>&1 | grep -E "^fk_verdict|^edge queue,|launch" | tail -18
echo "== classes off"; TENDON_HIP_EDGE_SEED_CLASSES=0 TENDON_HIP_EDGE_TIMING=1 timeout -k 10 300 python profiles/probe_edge_vs_verdict.py 2>&1 | grep -E "^edge queue,|launch" | tail -16
echo "== trace"; TENDON_HIP_LIB=profiles/_ab/libtendon_hip_eqtrace.so TENDON_HIP_EDGE_TIMING=1 timeout -k 10 300 python profiles/probe_edge_vs_verdict.py 2>&1 | grep -E "^edge queue,|launch|wave +[0-9]+, rounds" | head -44 | tail -11 | cut -c1-900
