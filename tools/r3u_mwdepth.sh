mkdir -p gpurun_out
for i in 1 2; do
for d in default 6 8; do
  if [ $d = default ]; then unset MMGP_LIBDIR; else export MMGP_LIBDIR=$PWD/abl$d; fi
  python tools/scan_levels3d.py 108 0 2>>gpurun_out/r3u_err.log | head -1 | sed "s/^/depth-$d /"
  python tools/scan_levels3d.py 54 0 2>>gpurun_out/r3u_err.log | head -1 | sed "s/^/depth-$d /"
  python tools/scan_levels3d.py 150 0 2>>gpurun_out/r3u_err.log | head -1 | sed "s/^/depth-$d /"
done
done
