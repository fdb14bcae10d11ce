#!/bin/bash
# timings of the x / y / z passes of the preconditioner lattice over the lattices of the BASELINE configurations
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
for shape in "768 768 194" "384 384 354" "1024 1024 60" "192 192 182" "256 256 250" "512 512 400" "256 256 700" "192 192 1025"; do
  timeout -k 10 120 python tools/dst_bench.py $shape 0 20 || exit 1
done 2>&1 | grep -v amdgpu.ids | tee $O/dst_shapes.txt
