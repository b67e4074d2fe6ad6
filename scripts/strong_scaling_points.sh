#!/bin/bash
# Per-rank operating points of a STRONG-scaling run (BASELINE: 65536 LinearMpcZmp instances in total over 8 GPUs = 8192
# per GPU; configs 4 / 5 likewise), measured on ONE MI355X: what each rank of the 8-GPU job computes per step.
# usage (through gpurun, from the repo root): bash scripts/strong_scaling_points.sh > gpurun_out/strong_points.jsonl
# (4096 = BASELINE config 2 as quoted: one GPU)
for b in 4096 8192 16384 32768 65536; do python bench.py --batch $b --steps 50 --warmup 10 --no-cpu-baseline --no-live-counters 2>/dev/null | tail -1; done
python bench.py --workload xy --batch 8192 --steps 10 --warmup 3 --no-cpu-baseline --no-live-counters 2>/dev/null | tail -1
python bench.py --workload srb --batch 4096 --steps 5 --warmup 2 --no-cpu-baseline --no-live-counters 2>/dev/null | tail -1
python bench.py --workload ddp --batch 512 --steps 5 --warmup 2 --no-cpu-baseline --no-live-counters 2>/dev/null | tail -1
