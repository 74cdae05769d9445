set -x
python -m pytest tests -q -m gpu -x > gpurun_out/r3c_tests.log 2>&1; tail -3 gpurun_out/r3c_tests.log
S5=59,117,233,466,931
S7=15,30,59,117,233,466,931
for pc in 1 2; do
  python bench_vcycle.py --cloud gmsh --sides $S5 --point-colouring $pc --cycles 20 > gpurun_out/r3c_2d5_pc$pc.json 2>gpurun_out/r3c_err.log
  python bench_vcycle.py --cloud gmsh --sides $S7 --point-colouring $pc --cycles 20 > gpurun_out/r3c_2d7_pc$pc.json 2>>gpurun_out/r3c_err.log
done
for t in 256 1024; do
  python bench_vcycle.py --cloud gmsh --sides $S7 --point-colouring 2 --tile $t --cycles 20 > gpurun_out/r3c_2d7_pc2_t$t.json 2>>gpurun_out/r3c_err.log
done
python bench_vcycle.py --cloud gmsh --sides $S7 --point-colouring 2 --tile-order 1 --cycles 20 > gpurun_out/r3c_2d7_pc2_to1.json 2>>gpurun_out/r3c_err.log
python bench_vcycle.py --cloud gmsh --sides $S7 --point-colouring 2 --waves 1 --cycles 20 > gpurun_out/r3c_2d7_pc2_w1.json 2>>gpurun_out/r3c_err.log
python bench_vcycle.py --cloud gmsh --sides $S7 --point-colouring 2 --cycles 20 --per-level gpurun_out/r3c_levels_pc2.md > /dev/null 2>>gpurun_out/r3c_err.log
for pc in 1 2; do
  python bench_vcycle.py --dim 3 --sides 14,27,54,108 --polydeg 3 --point-colouring $pc --cycles 10 > gpurun_out/r3c_3d_pc$pc.json 2>>gpurun_out/r3c_err.log
done
for f in gpurun_out/r3c_*.json; do echo $f; python - "$f" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(d["workload"]); print("  ms/cycle", round(d["device_ms_per_vcycle"],3), "contraction", d["contraction_per_cycle"], "res", ["%.2e"%r for r in d["residuals"][:5]])
PY
done
