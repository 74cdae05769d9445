mkdir -p gpurun_out
for i in 1 2; do
for d in default 8 12; do
  if [ $d = default ]; then unset MMGP_LIBDIR; else export MMGP_LIBDIR=$PWD/abl$d; fi
  python tools/scan_levels2d.py 931 4 0 2>>gpurun_out/r3r_err.log | sed "s/^/depth-$d /"
  python tools/scan_levels2d.py 466 3 0 2>>gpurun_out/r3r_err.log | sed "s/^/depth-$d /"
done
done
