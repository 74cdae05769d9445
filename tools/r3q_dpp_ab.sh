mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_live_params.py tests/test_gpu_configs.py -q -m gpu 2>&1 | tail -2
for i in 1 2; do
  python bench_vcycle.py --cloud gmsh --sides 15,30,59,117,233,466,931 --cycles 20 2>>gpurun_out/r3q_err.log | python -c "import sys,json; d=json.load(sys.stdin); print('new 2-D 7-level', round(d['device_ms_per_vcycle'],3))"
  python bench_vcycle.py --dim 3 --nside 216 --levels 4 --polydeg 3 --cycles 10 2>>gpurun_out/r3q_err.log | python -c "import sys,json; d=json.load(sys.stdin); print('new 3-D 216', round(d['device_ms_per_vcycle'],3))"
  MMGP_LIBDIR=$PWD/abl python bench_vcycle.py --cloud gmsh --sides 15,30,59,117,233,466,931 --cycles 20 2>>gpurun_out/r3q_err.log | python -c "import sys,json; d=json.load(sys.stdin); print('old(HEAD~1) 2-D 7-level', round(d['device_ms_per_vcycle'],3))"
  MMGP_LIBDIR=$PWD/abl python bench_vcycle.py --dim 3 --nside 216 --levels 4 --polydeg 3 --cycles 10 2>>gpurun_out/r3q_err.log | python -c "import sys,json; d=json.load(sys.stdin); print('old(HEAD~1) 3-D 216', round(d['device_ms_per_vcycle'],3))"
done
python bench.py --no-vcycle --no-cpu --no-fracstep --steps 40 2>>gpurun_out/r3q_err.log | python -c "import sys,json; d=json.load(sys.stdin); print('new bench', d['value'], d['roofline']['frac'], d['roofline']['frac_in_vcycle'], d['spmv']['frac'])"
MMGP_LIBDIR=$PWD/abl python bench.py --no-vcycle --no-cpu --no-fracstep --steps 40 2>>gpurun_out/r3q_err.log | python -c "import sys,json; d=json.load(sys.stdin); print('old bench', d['value'], d['roofline']['frac'], d['roofline']['frac_in_vcycle'], d['spmv']['frac'])"
