timeout -k 10 900 python -m pytest tests/test_gpu_search.py tests/test_gpu_lazy_prm.py tests/test_gpu_distributed.py tests/test_cpp_shim.py -x -q -m gpu 2>&1 | tail -5
timeout -k 10 600 python bench.py --steps 10 --warmup 2 > gpurun_out/r05_bench_v1.json 2> gpurun_out/r05_bench_v1.err; echo bench rc=$?
python3 - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05_bench_v1.json') if l.startswith('{')][0])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
e=d.get('extras',{})
print(json.dumps(e.get('config3_100k_vertices_k10',{}).get('edge_kernel_roofline')))
print(json.dumps(e.get('config5_10k_queries',{}).get('rooflines')))
c5=e.get('config5_10k_queries',{})
print({k:c5.get(k) for k in ('queries_per_s','queries_per_s_lazy','rounds','queries_per_s_eager_incl_revalidation')})
PY
