mkdir -p gpurun_out
for t in 1280 1536 2048; do TILE=$t python tools/scan_levels3d.py 171 6 8 2>>gpurun_out/r3j_err.log | grep '"xtra": 2' | tee -a gpurun_out/r3j_scan.jsonl; done
for t in 1536; do TILE=$t python tools/scan_levels3d.py 150 6 8 2>>gpurun_out/r3j_err.log | grep '"xtra": 2' | tee -a gpurun_out/r3j_scan.jsonl; done
for t in 1536; do TILE=$t python tools/scan_levels3d.py 216 6 8 2>>gpurun_out/r3j_err.log | grep '"xtra": 2' | tee -a gpurun_out/r3j_scan.jsonl; done
tail -3 gpurun_out/r3j_err.log
