#!/bin/bash
# Run ON THE GPU BOX: counters of the policy forward, one form per run (200 launches each), --pmc passes only.
#   bash tools/policy_pmc.sh <tag>     -> gpurun_out/summary/<tag>_pmc_policy_<form>.txt
set -u
TAG=${1:-r4v}; R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
mkdir -p $O/summary
for form in "16 both" "16 actor" "32 both"; do
  name=$(echo $form | tr ' ' '_')
  i=0
  for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VALU" \
             "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD"; do
    i=$((i + 1))
    timeout -k 10 120 rocprofv3 --pmc $grp --output-format csv -d $O/prof_${TAG}_pol_${name}_pmc$i -o run -- python3 $R/tools/policy_tiles.py one $form 4096 > $O/prof_${TAG}_pol_${name}_pmc$i.log 2>&1 || echo "pass $i of $form failed"
  done
  python3 - "$O" "$TAG" "$name" <<'P'
import csv, glob, os, sys, collections
O, tag, name = sys.argv[1:4]
acc = collections.defaultdict(list)
for path in sorted(glob.glob(os.path.join(O, f"prof_{tag}_pol_{name}_pmc*", "**", "*counter_collection.csv"), recursive=True)):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(path)):
        if "k_policy_forward" in r["Kernel_Name"]:
            per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, c), v in per.items():
        acc[c].append(v)
with open(os.path.join(O, "summary", f"{tag}_pmc_policy_{name}.txt"), "w") as f:
    f.write(f"# rocprofv3 --pmc <group> -- python3 tools/policy_tiles.py one {name.replace('_', ' ')} 4096 (tools/policy_pmc.sh {tag}); mean per launch over 200 launches; one pass per group\n")
    for c in sorted(acc):
        f.write("%-28s launches=%4d mean=%.6g\n" % (c, len(acc[c]), sum(acc[c]) / len(acc[c])))
print(open(os.path.join(O, "summary", f"{tag}_pmc_policy_{name}.txt")).read())
P
  for d in $O/prof_${TAG}_pol_${name}_pmc*; do find $d -name '*counter_collection.csv' -delete 2>/dev/null; done
done
echo policy_pmc done
