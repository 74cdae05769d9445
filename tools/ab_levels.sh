#!/bin/bash
# Same-box A/B of two library builds (MMGP_LIBDIR): per-level sweep / residual tables of the 2-D 7-level and the
# 3-D 4-level hierarchies.  usage: tools/ab_levels.sh <tag> [libdir_A]   (B = the in-tree build)
tag=${1:?tag}; A=${2:-$PWD/ablib}
for side in A B; do
  if [ $side = A ]; then export MMGP_LIBDIR=$A; else unset MMGP_LIBDIR; fi
  python bench_vcycle.py --cloud gmsh --sides 15,30,59,117,233,466,931 --cycles 20 --per-level gpurun_out/${tag}_${side}_levels.md > gpurun_out/${tag}_${side}_v2d7.json 2> gpurun_out/${tag}_${side}_v2d7.err || exit 1
  python bench_vcycle.py --dim 3 --nside 216 --levels 4 --polydeg 3 --cycles 10 --per-level gpurun_out/${tag}_${side}_levels.md > gpurun_out/${tag}_${side}_v3d.json 2> gpurun_out/${tag}_${side}_v3d.err || exit 1
  python tools/neumann3d_timing.py 54 3 20 > gpurun_out/${tag}_${side}_n3d.log 2>&1 || exit 1
done
echo ab done
