mkdir -p gpurun_out/r2
for v in 100 110 120 128 140 155 100 110 120 128 140 155; do F3D_PAIR8_STEP12=$v python bench.py --no-extra 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read()); print('STEP12', $v, b['value'], b['ms_per_step'], b['roofline']['avg_launch_us'], b['parity']['match'])"; done > gpurun_out/r2/step12.log 2>&1; cat gpurun_out/r2/step12.log
