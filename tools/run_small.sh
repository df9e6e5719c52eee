mkdir -p gpurun_out/r2
python -X faulthandler -m pytest tests/test_gpu_kernels.py tests/test_gpu_configs.py -q -x -k "not c5" > gpurun_out/r2/k9.log 2>&1; tail -2 gpurun_out/r2/k9.log
grep -q " passed" gpurun_out/r2/k9.log && for v in 8 0 8 0; do F3D_PAIR8_TY=$v python bench.py --no-extra 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read()); print('TY', $v, b['value'], b['ms_per_step'], b['roofline']['avg_launch_us'], b['parity']['match'])"; done > gpurun_out/r2/ty_bench.log 2>&1; cat gpurun_out/r2/ty_bench.log
