"""Summarise rocprofv3 output of tools/profile_round.sh into the small text/CSV files kept under profiles/.
usage: python tools/pmc_summary.py <tag> <kernel substring> [<kernel substring> ...]"""
import csv, glob, os, sys, collections

tag, kernels = sys.argv[1], sys.argv[2:]
out = os.path.join("gpurun_out")
stats = glob.glob(os.path.join(out, f"prof_{tag}_stats", "**", "*kernel_stats.csv"), recursive=True)
os.makedirs("profiles", exist_ok=True)
if stats:
    rows = list(csv.reader(open(stats[0])))
    with open(f"profiles/{tag}_kernel_stats.csv", "w") as f:
        w = csv.writer(f)
        for r in rows[:12]:
            w.writerow(r)
    print("kernel stats ->", f"profiles/{tag}_kernel_stats.csv")
for kern in kernels:
    acc = collections.defaultdict(list)
    for path in sorted(glob.glob(os.path.join(out, f"prof_{tag}_pmc*", "**", "*counter_collection.csv"), recursive=True)):
        per_dispatch = collections.defaultdict(float)
        for r in csv.DictReader(open(path)):
            if kern in r["Kernel_Name"]:
                per_dispatch[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
        for (d, name), v in per_dispatch.items():
            acc[name].append(v)
    if not acc:
        continue
    safe = kern.replace("<", "").replace(">", "").replace(":", "_")
    with open(f"profiles/{tag}_pmc_{safe}.txt", "w") as f:
        f.write(f"# rocprofv3 --pmc <group> -- python3 bench.py (tools/profile_round.sh {tag}); kernel {kern}; one pass per group\n")
        f.write("# mean per launch.  FETCH_SIZE / WRITE_SIZE in KB as reported (see the calibration lines for the byte factor)\n")
        for name in sorted(acc):
            v = acc[name]
            f.write("%-28s launches=%4d mean=%.6g\n" % (name, len(v), sum(v) / len(v)))
    print("pmc ->", f"profiles/{tag}_pmc_{safe}.txt")
cal = glob.glob(os.path.join(out, f"prof_{tag}_calib", "**", "*counter_collection.csv"), recursive=True)
if cal:
    with open(f"profiles/{tag}_fetch_calibration.txt", "w") as f:
        f.write("# tools/calib_fetch: each kernel reads 1 GiB (1048576 KB) once; FETCH_SIZE as reported by rocprofv3 (KB)\n")
        for r in csv.DictReader(open(cal[0])):
            if "read_dword" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
                v = float(r["Counter_Value"])
                f.write("%-40s FETCH_SIZE=%.6g KB  bytes/reported=%.3f\n" % (r["Kernel_Name"][:40], v, 1048576.0 / v if v else 0))
    print("calibration ->", f"profiles/{tag}_fetch_calibration.txt")
