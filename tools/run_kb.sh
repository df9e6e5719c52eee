set -e
mkdir -p gpurun_out
for dims in "70 70 70" "128 128 128" "200 200 200" "256 256 256" "360 360 360"; do
  for zc in 0 8 16 32 64; do
    echo -n "dims=$dims zchunk=$zc : "
    F3D_ZCHUNK=$zc python tools/kbench.py --dims $dims --reps 30 --kernel sweep | tail -1
  done
done 2>&1 | tee gpurun_out/kb7.log
