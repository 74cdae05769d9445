mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_host.py tests/test_gpu_fracstep.py -q -m gpu > gpurun_out/r3h_tests.log 2>&1; tail -3 gpurun_out/r3h_tests.log
for ns in 108 150 171; do python tools/scan_levels3d.py $ns 0 6 12 1 2>>gpurun_out/r3h_err.log | tee -a gpurun_out/r3h_scan.jsonl; done
for x in 1 0 1 0; do python bench_vcycle.py --dim 3 --nside 216 --levels 4 --polydeg 3 --cycles 10 --dense-xtra $x 2>>gpurun_out/r3h_err.log | python -c "import sys,json; d=json.load(sys.stdin); print('xtra $x', d['device_ms_per_vcycle'], d['contraction_per_cycle'])"; done
