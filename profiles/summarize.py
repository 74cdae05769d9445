"""Condense rocprofv3 output directories (gpurun_out/prof_*) into the small CSV/MD
summaries committed under profiles/.  Usage:
    python profiles/summarize.py <tag> <stats_dir> <fetch_dir> <write_dir> [bench.json]
HBM traffic follows MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB and
collected in separate --pmc passes; on gfx950 FETCH_SIZE counts 128-B requests as 64 B
for wide coalesced streaming reads, so read bytes = 2 * FETCH_SIZE * 1024 (this kernel
streams 8 B/lane = 512 B per wave-instruction; the factor is the guide's calibration
for 16 B/lane and is the only correction applied)."""
import collections
import csv
import glob
import json
import os
import sys


def agg_counter(d, counter):
    f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            out[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return out


def main():
    tag, stats_dir, fetch_dir, write_dir = sys.argv[1:5]
    bench = json.load(open(sys.argv[5])) if len(sys.argv) > 5 else None
    here = os.path.dirname(os.path.abspath(__file__))
    ks = glob.glob(os.path.join(stats_dir, "**", "*_kernel_stats.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(ks)))
    with open(os.path.join(here, f"{tag}_kernel_stats.csv"), "w") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys())
        w.writeheader()
        w.writerows(rows)
    fetch = agg_counter(fetch_dir, "FETCH_SIZE")
    write = agg_counter(write_dir, "WRITE_SIZE")
    lines = [f"# {tag}: rocprofv3 summary", "",
             "| kernel | calls | avg us | total ms | % |", "|---|---|---|---|---|"]
    for r in rows[:16]:
        lines.append(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | "
                     f"{float(r['TotalDurationNs'])/1e6:.2f} | {float(r['Percentage']):.2f} |")
    lines += ["", "| kernel | FETCH_SIZE KiB/launch | WRITE_SIZE KiB/launch | HBM MB/launch (2*F+W) |", "|---|---|---|---|"]
    traffic = None
    for k in fetch:
        if k not in write:
            continue
        # a kernel launched with different amounts of work (the sweep kernel carries 5 sweeps per launch inside the
        # V-cycle legs and 16 in the timed loop): keep the launches of the largest kind only
        fsel = [v for v in fetch[k] if v >= 0.9 * max(fetch[k])]
        wsel = [v for v in write[k] if v >= 0.9 * max(write[k])]
        fm = sum(fsel) / len(fsel)
        wm = sum(wsel) / len(wsel)
        mb = (2 * fm + wm) * 1024 / 1e6
        if ("tile_kernel" in k or "sweep_persistent_kernel" in k) and (traffic is None or mb > traffic):
            traffic = mb
        lines.append(f"| `{k[:70]}` | {fm:.0f} | {wm:.0f} | {mb:.1f} |")
    if bench:
        rl = bench["roofline"]
        cfg = bench["config"]
        alg_sweep = cfg["interior_points_per_gpu"] * rl["algorithmic_bytes_per_row"] / 1e6
        lines += ["", f"bench line: {bench['value']:.0f} {bench['unit']}, {bench['ms_per_step']:.3f} ms/sweep, "
                      f"roofline {rl['achieved']:.0f} GB/s = {rl['frac']*100:.1f} % of {rl['peak']:.0f} GB/s; "
                      f"HIP events: {rl.get('us_per_sweep', rl['avg_launch_us']):.1f} us per sweep "
                      f"({rl['launches']} launches carrying {rl.get('sweeps_in_launches', rl['launches'])} sweeps)"]
        sweep_rows = [r for r in rows if "sweep_persistent" in r["Name"] or "tile_kernel" in r["Name"]]
        if sweep_rows and cfg.get("sweeps_executed"):
            tot = sum(float(r["TotalDurationNs"]) for r in sweep_rows if "sweep_persistent" in r["Name"] or ", 0," in r["Name"])
            lines.append(f"rocprofv3: sweep kernels total {tot/1e6:.2f} ms over the sweeps of the profiled run")
        if traffic:
            spl = rl.get("sweeps_in_launches", rl["launches"]) / rl["launches"]
            lines.append(f"algorithmic bytes per sweep: {alg_sweep:.1f} MB; PMC traffic {traffic:.1f} MB per launch of "
                         f"{spl:.0f} sweeps = {traffic/spl:.1f} MB per sweep (ratio {traffic/spl/alg_sweep:.2f})")
    # ---- the dominant kernel split by sweeps per launch (kernel trace, one row per dispatch) ----
    # The fine-level sweep kernel is launched with 16 fused sweeps by the timed loop and with `iters` = 5 by every
    # smoothing call of the V-cycle legs; the aggregate statistics above mix the two.  Each dispatch is classified by
    # round(duration / us_per_sweep of the bench line); per class: launches, average duration, us per sweep and the
    # roofline fraction that goes with it -- `roofline.frac` (16) and `roofline.frac_in_vcycle` (5) of the bench line.
    tr = glob.glob(os.path.join(stats_dir, "**", "*_kernel_trace.csv"), recursive=True)
    if bench and tr:
        rl = bench["roofline"]
        cfg = bench["config"]
        per_sweep_us = rl.get("us_per_sweep", rl["avg_launch_us"])
        alg_sweep_b = cfg["interior_points_per_gpu"] * rl["algorithmic_bytes_per_row"]
        groups = collections.defaultdict(list)
        for r in csv.DictReader(open(tr[0])):
            if "sweep_persistent_kernel" not in r["Kernel_Name"]:
                continue
            us = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
            k = int(round(us / per_sweep_us))
            for e in (16, 5):                            # the two kinds the bench run issues: a cold 5-sweep launch
                if abs(us / (e * per_sweep_us) - 1.0) < 0.25:   # 12 % slow is still a 5-sweep launch, not a "6-sweep" one
                    k = e
                    break
            if k >= 1 and us > 0.5 * per_sweep_us:      # (launches of the same template on coarser levels are shorter)
                groups[k].append(us)
        if groups:
            lines += ["", "| `sweep_persistent_kernel` launches carrying | launches | avg us per launch | us per sweep | % of 8 TB/s |",
                      "|---|---|---|---|---|"]
            for k in sorted(groups, reverse=True):
                v = groups[k]
                if len(v) < 2:
                    continue
                avg = sum(v) / len(v)
                lines.append(f"| {k} sweeps | {len(v)} | {avg:.1f} | {avg / k:.1f} | {alg_sweep_b * k / (avg * 1e-6) / 8e12 * 100:.1f} |")
    if bench and traffic:
        # what bench.py reports as roofline.traffic (only for a run of the same shape)
        json.dump({"source": f"profiles/{tag}_summary.md", "kernel": rl["kernel"], "bytes_per_launch": traffic * 1e6,
                   "sweeps_per_launch": rl.get("sweeps_in_launches", rl["launches"]) / rl["launches"],
                   "tiles": cfg["tiles"], "tile_points": cfg["tile_points"], "interior_points_per_gpu": cfg["interior_points_per_gpu"],
                   "stencil": cfg["stencil"], "collection": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes; "
                   "bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB (gfx950 counts 128-B read requests as 64 B)"},
                  open(os.path.join(here, "pmc_traffic.json"), "w"), indent=1)
    open(os.path.join(here, f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
