mkdir -p gpurun_out/r2
for s in 24 40 70 100 128 180 256 384 512; do python tools/mbench.py --size $s; done > gpurun_out/r2/mb4.log 2>&1; cat gpurun_out/r2/mb4.log
python -X faulthandler -m pytest tests/test_gpu_kernels.py -q -x -k median > gpurun_out/r2/m4.log 2>&1; tail -2 gpurun_out/r2/m4.log
python bench.py > gpurun_out/r2/bench6.json 2> gpurun_out/r2/bench6.err; cut -c1-300 gpurun_out/r2/bench6.json
