TENDON_HIP_EDGE_TIMING=1 timeout -k 10 500 python bench.py --workload config4 --emulate-world 8 --steps 2 --warmup 1 > gpurun_out/r05_c4proj_v1.json 2> gpurun_out/r05_c4proj_v1.err
echo rc=$?
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r05_c4proj_v1.json'))
print(json.dumps(d['world_1_ms']), json.dumps(d.get('signature_wire')))
w=d['emulated_worlds']['8']
print(json.dumps({k:w[k] for k in ('phases_ms_max_over_ranks','replicated_ms_on_every_rank','compute_critical_path_ms','allgather_bytes_per_rank','allgather_model_ms_ring','projected_build_ms','projected_speedup_over_world_1')}))
PY
grep -E "edge queue\]|edges_indexed\]" gpurun_out/r05_c4proj_v1.err | tail -28
