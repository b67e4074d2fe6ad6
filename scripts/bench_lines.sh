#!/bin/bash
# Every workload's bench line with its CPU baseline (through gpurun, from the repo root):
#   bash scripts/bench_lines.sh > gpurun_out/bench_lines.jsonl      -> profiles/<round>_bench_lines.jsonl
# and the driver's default command (headline + `secondary`) into gpurun_out/bench_default_line.json
python bench.py --steps 200 --warmup 20 --no-secondary --no-live-counters 2>/dev/null | tail -1
for w in xy ddp srb walk multi xywalk ism z ddpzmp; do python bench.py --workload $w --no-live-counters 2>/dev/null | tail -1; done
python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_default_line.json
