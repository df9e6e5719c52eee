python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "median" 2>&1 | tail -2
python tools/mbench.py --size 256 --reps 10
python tools/mbench.py --size 512 --reps 5
python tools/mbench.py --size 128 --reps 20
python tools/mbench.py --size 256 --reps 10 --radius 3
