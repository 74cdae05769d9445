mkdir -p gpurun_out
for i in 1 2; do
  MMGP_LIBDIR=$PWD/abl python tools/scan_levels2d.py 931 4 0 256 2>>gpurun_out/r3o_err.log | sed 's/^/depth4 /'
  python tools/scan_levels2d.py 931 4 0 256 2>>gpurun_out/r3o_err.log | sed 's/^/depth8 /'
done
MMGP_LIBDIR=$PWD/abl python tools/scan_levels2d.py 466 3 0 128 2>>gpurun_out/r3o_err.log | sed 's/^/depth4 /'
python tools/scan_levels2d.py 466 3 0 128 2>>gpurun_out/r3o_err.log | sed 's/^/depth8 /'
MMGP_LIBDIR=$PWD/abl python bench_vcycle.py --cloud gmsh --sides 15,30,59,117,233,466,931 --cycles 20 2>>gpurun_out/r3o_err.log | python -c "import sys,json; d=json.load(sys.stdin); print('depth4 7-level', d['device_ms_per_vcycle'])"
python bench_vcycle.py --cloud gmsh --sides 15,30,59,117,233,466,931 --cycles 20 2>>gpurun_out/r3o_err.log | python -c "import sys,json; d=json.load(sys.stdin); print('depth8 7-level', d['device_ms_per_vcycle'])"
python -m pytest tests/test_gpu_parity.py tests/test_gpu_live_params.py -q -m gpu 2>&1 | tail -2
