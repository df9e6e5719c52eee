mkdir -p gpurun_out/r2
for i in 1 2 3; do for v in old new; do for s in 256 512; do for k in sweep2 sweeppk; do echo -n "$v "; F3D_LIBDIR=$GRAFT_REPO_ROOT/ab_$v python tools/kbench.py --size $s --reps 100 --kernel $k 2>&1 | tail -1; done; done; done; done > gpurun_out/r2/ab1.log 2>&1; cat gpurun_out/r2/ab1.log
