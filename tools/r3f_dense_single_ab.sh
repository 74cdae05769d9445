# A/B of the one-wavefront dense layout on sweep-ordered 2-D levels (same box), parity subset, dd rehearsal with the frac-step leg
mkdir -p gpurun_out
S5=59,117,233,466,931
S7=15,30,59,117,233,466,931
for ds in 0 1 0 1; do
  python bench_vcycle.py --cloud gmsh --sides $S7 --dense-single $ds --cycles 20 > gpurun_out/r3f_2d7_ds${ds}_$RANDOM.json 2>>gpurun_out/r3f_err.log
done
python bench_vcycle.py --cloud gmsh --sides $S5 --dense-single 1 --cycles 20 > gpurun_out/r3f_2d5_ds1.json 2>>gpurun_out/r3f_err.log
python bench_vcycle.py --cloud gmsh --sides $S7 --dense-single 1 --cycles 20 --per-level gpurun_out/r3f_levels_ds1.md > /dev/null 2>>gpurun_out/r3f_err.log
python bench_vcycle.py --cloud gmsh --sides $S7 --dense-single 1 --graph 0 --cycles 20 > gpurun_out/r3f_2d7_ds1_nograph.json 2>>gpurun_out/r3f_err.log
for f in gpurun_out/r3f_2d*.json; do python - "$f" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[1], "ms/cycle", round(d["device_ms_per_vcycle"],3), "contraction", d["contraction_per_cycle"])
PY
done
python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_live_params.py tests/test_gpu_host.py -q -m gpu > gpurun_out/r3f_tests.log 2>&1; tail -3 gpurun_out/r3f_tests.log
MMG_NUM_THREADS=2 timeout -k 10 600 python bench.py --force-dd --steps 10 --warmup 2 --no-cpu --no-vcycle --nside 128 > gpurun_out/r3f_bench_dd.json 2> gpurun_out/r3f_bench_dd.err; echo "dd rc=$?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3f_bench_dd.json")); print(json.dumps(d["multi_gpu"].get("fracstep"))[:900])
PY
