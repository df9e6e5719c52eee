mkdir -p gpurun_out/r2
python bench.py --no-extra > gpurun_out/r2/bench7.json 2> gpurun_out/r2/bench7.err; python - <<'PY'
import json; b=json.load(open('gpurun_out/r2/bench7.json')); print(b['value'], b['parity'], b['roofline'].get('traffic'), b['roofline'].get('hbm_frac'))
PY
F3D_COMM_BACKEND=shm timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 1 --warmup 0 --size 512 > gpurun_out/r2/bench_shm2.json 2> gpurun_out/r2/bench_shm2.err; tail -n 3 gpurun_out/r2/bench_shm2.err; cut -c1-700 gpurun_out/r2/bench_shm2.json
