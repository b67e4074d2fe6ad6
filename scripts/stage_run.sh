mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_zmp_gpu.py -m gpu -x -q -k "stage_kernel" 2>&1 | tail -15
for s in 0 1; do CCC_ZMP_DEBUG=1 CCC_ZMP_STAGE=$s CCC_ZMP_BENCH_CHECK=1 timeout 300 python scripts/zmp_n_bench.py 100 32768 3 2>&1 | tail -6; done
for it in 16 24 32 64; do CCC_ZMP_STAGE_ITERS=$it CCC_ZMP_DEBUG=1 timeout 300 python scripts/zmp_n_bench.py 100 32768 3 2>&1 | tail -3; done
