# round-3 profiles: rocprofv3 stats + PMC passes of the default bench (profiles/collect.sh r3), per-level V-cycle tables
mkdir -p gpurun_out
bash profiles/collect.sh r3 > gpurun_out/r3g_collect.log 2>&1; echo "collect rc=$?"; tail -5 gpurun_out/r3g_collect.log
rm -f gpurun_out/r3_vcycle_levels.md
python bench_vcycle.py --cloud gmsh --sides 15,30,59,117,233,466,931 --cycles 20 --per-level gpurun_out/r3_vcycle_levels.md > gpurun_out/r3_vcycle2d_7.json 2>>gpurun_out/r3g_err.log
python bench_vcycle.py --cloud gmsh --sides 59,117,233,466,931 --cycles 20 --per-level gpurun_out/r3_vcycle_levels.md > gpurun_out/r3_vcycle2d_5.json 2>>gpurun_out/r3g_err.log
python bench_vcycle.py --cloud gmsh --sides 59,117,233,466,931 --point-colouring 1 --cycles 20 --per-level gpurun_out/r3_vcycle_levels.md > gpurun_out/r3_vcycle2d_5_colour.json 2>>gpurun_out/r3g_err.log
python bench_vcycle.py --dim 3 --nside 216 --levels 4 --polydeg 3 --cycles 10 --per-level gpurun_out/r3_vcycle_levels.md > gpurun_out/r3_vcycle3d_216.json 2>>gpurun_out/r3g_err.log
ls -la gpurun_out/profiles_r3 2>/dev/null | head
