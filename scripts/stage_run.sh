mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_zmp_gpu.py tests/test_graph_capture_gpu.py -m gpu -q -s -k "not closed_loop_cpp" 2>&1 | grep -v "^zmp \|amdgpu.ids" | tail -12
