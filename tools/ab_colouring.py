"""A/B of the point colouring inside tiles (0 greedy in tile order, 1 smallest-last + iterated greedy): V-cycle time of
both headline hierarchies; development aid."""
import json
import subprocess
import sys

for pc in (0, 1):
    for args in (["--nside", "1000", "--levels", "5", "--polydeg", "4", "--cycles", "20"],
                 ["--dim", "3", "--nside", "216", "--levels", "4", "--polydeg", "3", "--cycles", "10"]):
        out = subprocess.run([sys.executable, "bench_vcycle.py", "--point-colouring", str(pc)] + args, capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            print(f"colouring {pc}: {d['workload'][:40]} -> {d['device_ms_per_vcycle']:.3f} ms per V-cycle, residuals {d['residuals'][:4]}", flush=True)
        except Exception as e:  # noqa: BLE001
            print("failed", pc, args, e, out.stderr[-500:], flush=True)
