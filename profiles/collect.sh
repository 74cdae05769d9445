#!/bin/bash
# profiles/collect.sh <tag> -- run ON THE GPU BOX (through gpurun) from the repo root:
#   gpurun --timeout 1200 -- 'bash profiles/collect.sh r1e'
# Three separate rocprofv3 passes over the SAME bench command (MI355X_MICROARCH.md "HBM":
# counters in their own runs, never combined with --kernel-trace/--stats), every launch of the
# sweep kernel carrying 16 fused sweeps; then profiles/summarize.py condenses them into
# profiles/<tag>_{kernel_stats.csv,summary.md}, profiles/<tag>_profiled_run_bench.json and
# profiles/pmc_traffic.json (read by bench.py for roofline.traffic).
set -eo pipefail
tag=${1:?tag}
root=$(pwd)
out=$root/gpurun_out
export TMPDIR=/tmp
# (the run includes the two whole-V-cycle legs: the kernel statistics then also show the dense multi-wavefront
#  kernels of the small levels, the residual and the transfer kernels)
# (without the fractional-step leg: its ~1e5 small launches would make the kernel trace too large to bring back)
cmd="$root/bench.py --steps 16 --warmup 16 --no-cpu --no-fracstep"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -- python3 $cmd > $out/${tag}_profiled_run_bench.json 2> $out/${tag}_stats.log
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${tag}_fetch -- python3 $cmd > /dev/null 2> $out/${tag}_fetch.log
echo "FETCH_SIZE pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${tag}_write -- python3 $cmd > /dev/null 2> $out/${tag}_write.log
echo "WRITE_SIZE pass done"
cd $root
python3 profiles/summarize.py $tag $out/${tag}_stats $out/${tag}_fetch $out/${tag}_write $out/${tag}_profiled_run_bench.json
cp $out/${tag}_profiled_run_bench.json profiles/ 2>/dev/null || true
mkdir -p $out/profiles_$tag && cp profiles/${tag}_* profiles/pmc_traffic.json $out/profiles_$tag/
# the raw traces stay on the box: gpurun brings back at most 64 MiB
rm -rf $out/${tag}_stats $out/${tag}_fetch $out/${tag}_write
