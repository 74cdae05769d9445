mkdir -p gpurun_out
for ns in 171 190; do
  for t in 768 1024; do TILE=$t python tools/scan_levels3d.py $ns 6 12 2>>gpurun_out/r3i_err.log | tee -a gpurun_out/r3i_scan.jsonl; done
  python tools/scan_levels3d.py $ns 1 2>>gpurun_out/r3i_err.log | head -1 | tee -a gpurun_out/r3i_scan.jsonl
done
