mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python tools/kbench.py --size 512 --reps 10 | tee gpurun_out/kb12.log
F3D_SWEEP4=1 python tools/kbench.py --size 512 --reps 10 --kernel phi
python tools/kbench.py --size 256 --reps 20
python tools/kbench.py --dims 584 388 5 --reps 20
python bench.py --steps 2 --warmup 1 --no-cpu 2>/dev/null | tee gpurun_out/b512_3.json
