# final round-3 measurement: default bench line, profiles (stats + PMC), per-level tables, N>1 rehearsal with one rank
mkdir -p gpurun_out
timeout -k 10 900 python bench.py > gpurun_out/r3w_bench.json 2> gpurun_out/r3w_bench.err; echo "bench rc=$?"
bash profiles/collect.sh r3 > gpurun_out/r3w_collect.log 2>&1; echo "collect rc=$?"; tail -8 gpurun_out/r3w_collect.log
rm -f gpurun_out/r3_vcycle_levels.md
python bench_vcycle.py --cloud gmsh --sides 15,30,59,117,233,466,931 --cycles 20 --per-level gpurun_out/r3_vcycle_levels.md > gpurun_out/r3_vcycle2d_7.json 2>>gpurun_out/r3w_err.log
python bench_vcycle.py --cloud gmsh --sides 59,117,233,466,931 --cycles 20 --per-level gpurun_out/r3_vcycle_levels.md > gpurun_out/r3_vcycle2d_5.json 2>>gpurun_out/r3w_err.log
python bench_vcycle.py --cloud gmsh --sides 59,117,233,466,931 --point-colouring 1 --cycles 20 --per-level gpurun_out/r3_vcycle_levels.md > gpurun_out/r3_vcycle2d_5_colour.json 2>>gpurun_out/r3w_err.log
python bench_vcycle.py --dim 3 --nside 216 --levels 4 --polydeg 3 --cycles 10 --per-level gpurun_out/r3_vcycle_levels.md > gpurun_out/r3_vcycle3d_216.json 2>>gpurun_out/r3w_err.log
python bench_vcycle.py --cloud gmsh --sides 13,25,49,97 --polydeg 6 --neumann 1 --cycles 20 --per-level gpurun_out/r3_vcycle_levels.md > gpurun_out/r3_vcycle2d_neumann_L6.json 2>>gpurun_out/r3w_err.log
python bench_vcycle.py --cloud gmsh --sides 13,25,49,97 --polydeg 6 --neumann 1 --ordering rcm --cycles 20 > gpurun_out/r3_vcycle2d_neumann_L6_rcm.json 2>>gpurun_out/r3w_err.log
MMG_NUM_THREADS=2 timeout -k 10 700 python bench.py --force-dd --steps 10 --warmup 2 --no-cpu > gpurun_out/r3w_bench_dd.json 2> gpurun_out/r3w_bench_dd.err; echo "dd rc=$?"
ls gpurun_out/profiles_r3
