set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | tail -2
python tools/kbench.py --size 512 --reps 10 | tee gpurun_out/kb5.log
F3D_SOLVER_VARIANT=2 python tools/kbench.py --size 512 --reps 10 --kernel phi
python tools/kbench.py --size 256 --reps 20
python tools/kbench.py --size 128 --reps 20
python tools/kbench.py --dims 70 70 70 --reps 20
python tools/kbench.py --dims 584 388 5 --reps 20
python bench.py --steps 1 --warmup 1 --no-cpu 2>/dev/null | tee gpurun_out/b512_2.json
