# round-3 check on the GPU box: full GPU suite, one-rank rehearsal of the N>1 bench path, default bench line
mkdir -p gpurun_out
python -m pytest tests -q -m gpu > gpurun_out/r3e_tests.log 2>&1; tail -4 gpurun_out/r3e_tests.log
MMG_NUM_THREADS=2 timeout -k 10 900 python bench.py --force-dd --steps 10 --warmup 2 --no-cpu --dd-vcycle-nside 96 > gpurun_out/r3e_bench_dd.json 2> gpurun_out/r3e_bench_dd.err; echo "dd rc=$?"; tail -c 1500 gpurun_out/r3e_bench_dd.json
timeout -k 10 900 python bench.py > gpurun_out/r3e_bench.json 2> gpurun_out/r3e_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3e_bench.json"))
print("value", d["value"], "ms/step", d["ms_per_step"], "roofline", d["roofline"]["frac"], "in_vcycle", d["roofline"].get("frac_in_vcycle"))
for v in d.get("vcycle", []):
    print(" vcycle:", v.get("workload", "")[:90], "| ms", v.get("ms_per_vcycle"), "frac", v.get("frac"), "contraction", v.get("contraction_per_cycle_timed_cycles"), v.get("error"))
print(" spmv", d["spmv"]["frac"], "fracstep", d.get("fracstep"))
print(" setup", d["config"]["setup_seconds"], d["config"]["hierarchy_setup_seconds"], "cpu", d.get("cpu_baseline", {}).get("value"))
PY
