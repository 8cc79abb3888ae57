"""profiles/<tag>_pmc_*.txt (tools/pmc_summary.py) -> profiles/<tag>_traffic.json: memory-side bytes of one step of the dynamics
pipeline = sum over its kernels of fetch_factor x FETCH_SIZE + WRITE_SIZE (KB as reported by rocprofv3, FETCH_SIZE corrected by the
calibration run of the same profile: tools/calib_fetch).   usage: python tools/traffic_json.py <tag> <self_collision 0|1> [envs]"""
import glob, json, os, re, sys
tag, selfcol = sys.argv[1], int(sys.argv[2])
envs = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
factor = 2.0
cal = os.path.join(P, f"{tag}_fetch_calibration.txt")
if os.path.exists(cal):
    f = [float(m) for m in re.findall(r"bytes/reported=([0-9.]+)", open(cal).read())]
    if f:
        factor = sum(f) / len(f)
per, fetch, write = {}, 0.0, 0.0
for path in sorted(glob.glob(os.path.join(P, f"{tag}_pmc_k_*.txt"))):
    name = os.path.basename(path)[len(tag) + 5:-4]
    txt = open(path).read()
    fs = float(re.search(r"FETCH_SIZE\s+launches=\s*\d+ mean=([0-9.e+]+)", txt).group(1))
    ws = float(re.search(r"WRITE_SIZE\s+launches=\s*\d+ mean=([0-9.e+]+)", txt).group(1))
    per[name] = int((factor * fs + ws) * 1024)
    fetch += fs; write += ws
out = {"kernels": sorted(per), "envs_per_launch": envs, "self_collision": selfcol, "fetch_size_kb_reported": round(fetch, 2),
       "write_size_kb": round(write, 2), "fetch_factor": factor, "traffic_bytes_per_launch": int((factor * fetch + write) * 1024),
       "per_kernel_bytes": per,
       "source": f"profiles/{tag}_pmc_k_*.txt + profiles/{tag}_fetch_calibration.txt (tools/profile_round.sh {tag}): per step, summed over "
                 "the pipeline's kernels, fetch_factor x FETCH_SIZE + WRITE_SIZE; memory-side counters, Infinity-Cache hits included"}
json.dump(out, open(os.path.join(P, f"{tag}_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
