mkdir -p gpurun_out/r2
python -X faulthandler -m pytest tests/test_gpu_kernels.py tests/test_gpu_slab.py tests/test_gpu_pipeline.py -q -x -k "median or slab or pipeline or golden or registration" > gpurun_out/r2/m1.log 2>&1; tail -3 gpurun_out/r2/m1.log
grep -q " passed" gpurun_out/r2/m1.log && for s in 128 256 512; do for p in 0 1; do F3D_MEDIAN_PAIR=$p python tools/mbench.py --size $s; done; done > gpurun_out/r2/mb1.log 2>&1; cat gpurun_out/r2/mb1.log
F3D_MEDIAN_PAIR=1 python tools/mbench.py --size 256 --radius 3; F3D_MEDIAN_PAIR=0 python tools/mbench.py --size 256 --radius 3
