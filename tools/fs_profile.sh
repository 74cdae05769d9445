set -e
root=$(pwd)
export TMPDIR=/tmp
export PYTHONPATH=$root
cd /tmp
# one time step of the bench's fractional-step leg (54^3 / 108^3, 60 coarse sweeps), plain launches (no graph: kernel names stay visible), 40 V-cycles
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fsprof -- python3 $root/tools/fracstep_timing.py 54,108 3 1 40 60 > $root/gpurun_out/fs_prof_run.log 2>&1
f=$(find /tmp/fsprof -name "*kernel_stats.csv" | head -1)
cp "$f" $root/gpurun_out/fs_kernel_stats.csv
head -25 $root/gpurun_out/fs_kernel_stats.csv
tail -3 $root/gpurun_out/fs_prof_run.log
