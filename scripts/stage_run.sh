mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_graph_capture_gpu.py -m gpu -q -x -k "zmp" 2>&1 | tail -4
timeout 600 python bench.py --workload zmp100 2>&1 | tail -1 > gpurun_out/r06_bench_zmp100.json; python - <<'PY'
import json
d=json.load(open('gpurun_out/r06_bench_zmp100.json'))
print({k:d[k] for k in ('value','ms_per_step','p50_ms','unsolved')}, d['roofline']['kernel_avg_ms'], d['roofline']['workspace'], d.get('cpu_baseline',{}).get('value'), d.get('parity'))
PY
