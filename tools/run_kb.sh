set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "phi_ksi or slab" 2>&1 | tail -3
for v in "3 1 0" "3 1 32" "3 1 64" "3 1 128"; do
  set -- $v
  echo "variant=$1 xcd=$2 zchunk=$3"
  F3D_SOLVER_VARIANT=$1 F3D_XCD_REMAP=$2 F3D_ZCHUNK=$3 python tools/kbench.py --size 512 --reps 10 --kernel sweep
done 2>&1 | tee gpurun_out/kb3.log
F3D_ZCHUNK=64 python tools/kbench.py --dims 584 388 5 --reps 10 --kernel sweep
python tools/kbench.py --size 128 --reps 20 --kernel sweep
python tools/kbench.py --size 256 --reps 20 --kernel sweep
