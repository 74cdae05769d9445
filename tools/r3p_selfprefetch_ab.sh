mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_live_params.py tests/test_gpu_configs.py tests/test_gpu_fracstep.py -q -m gpu 2>&1 | tail -2
for i in 1 2; do
  for lib in old new; do
    if [ $lib = old ]; then export MMGP_LIBDIR=$PWD/abl; else unset MMGP_LIBDIR; fi
    python bench_vcycle.py --cloud gmsh --sides 15,30,59,117,233,466,931 --cycles 20 2>>gpurun_out/r3p_err.log | python -c "import sys,json; d=json.load(sys.stdin); print('$lib 2-D 7-level', round(d['device_ms_per_vcycle'],3))"
    python bench_vcycle.py --dim 3 --nside 216 --levels 4 --polydeg 3 --cycles 10 2>>gpurun_out/r3p_err.log | python -c "import sys,json; d=json.load(sys.stdin); print('$lib 3-D 216', round(d['device_ms_per_vcycle'],3))"
  done
done
unset MMGP_LIBDIR
MMGP_LIBDIR=$PWD/abl python tools/scan_levels2d.py 233 3 0 2>>gpurun_out/r3p_err.log | sed 's/^/old /'
python tools/scan_levels2d.py 233 3 0 2>>gpurun_out/r3p_err.log | sed 's/^/new /'
MMGP_LIBDIR=$PWD/abl python tools/scan_levels3d.py 54 0 2>>gpurun_out/r3p_err.log | head -1 | sed 's/^/old /'
python tools/scan_levels3d.py 54 0 2>>gpurun_out/r3p_err.log | head -1 | sed 's/^/new /'
