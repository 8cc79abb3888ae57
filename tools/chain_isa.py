"""ISA critical-path count of one hinge visit (the unit of the root body's chain in k_sweeps_g).

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -DEVM_ISA_PROBE -x hip -c evomotion_amd/csrc/env_kernels.hip --save-temps
  python tools/chain_isa.py env_kernels-hip-amdgcn-amd-amdhsa-gfx950.s > profiles/r3_chain_isa.txt

-DEVM_ISA_PROBE adds k_probe_hinge_rows: hinge_rows() — the six rows of a hinge on a register-resident record and body pair, the
same inlined code the chain entry runs — between plain loads and stores.  This script builds the register dependency graph of that
kernel's instruction stream (true dependencies only; in-order issue is accounted for separately) and reports
  * the VALU instruction count between the first and the last arithmetic instruction (issue bound: 4 cycles per wave64 VALU),
  * the longest dependent path through them, in instructions and in cycles with the latencies tools/lat_probe.hip measured on
    MI355X (dependent v_pk_*_f32: 8.25 cycles, other dependent VALU: 4.84; an independent instruction issues every 4.3)."""
import re, sys

LAT_PK, LAT_VALU, ISSUE = 8.25, 4.84, 4.3
src = open(sys.argv[1]).read().split("\n")
kernel = sys.argv[2] if len(sys.argv) > 2 else "k_probe_hinge_rows"
i0 = next(i for i, l in enumerate(src) if re.match(r"^_ZN3evm\d+%s" % kernel, l))
body = []
for l in src[i0 + 1:]:
    t = l.split(";")[0].strip()
    if not t or t.startswith(".") or t.endswith(":"):
        continue
    body.append(t)
    if t.startswith("s_endpgm"):
        break

def regs(tok):
    tok = tok.strip().strip("|").lstrip("-").strip("|")
    m = re.match(r"^([vsa])\[(\d+):(\d+)\]$", tok)
    if m:
        return [m.group(1) + str(k) for k in range(int(m.group(2)), int(m.group(3)) + 1)]
    if re.match(r"^[vsa]\d+$", tok):
        return [tok]
    if tok in ("vcc", "exec"):
        return [tok]
    return []

ready = {}          # register -> (depth in instructions, depth in cycles, index of producer)
nodes = []          # (text, depth_n, depth_c, pred)
for idx, t in enumerate(body):
    op, _, rest = t.partition(" ")
    ops = [o for o in re.split(r",\s*", rest) if o and not o.startswith("op_sel") and not o.startswith("neg_")]
    ops = [re.sub(r"\s+(op_sel|op_sel_hi|neg_lo|neg_hi|clamp|mul:\d|div:\d).*$", "", o) for o in ops]
    is_valu = op.startswith("v_") and not re.search(r"_(u32|u64|i32|b64|co_u32)|lshl|addc|ashr|mad_u|bfe", op)  # float path only: address arithmetic feeds loads/stores
    if not is_valu:
        if op.startswith("s_and_saveexec") or op.startswith("s_or_b64") or op.startswith("s_andn2"):
            pass
        if op.startswith("global_load") or op.startswith("ds_read"):
            for r in regs(ops[0]):
                ready[r] = (0, 0.0, None)
        continue
    dst = regs(ops[0]) if ops else []
    srcs = []
    for o in ops[1:]:
        srcs += regs(o)
    if op.startswith("v_cmp") and op.endswith("_e32"):
        srcs += regs(ops[0]); dst = ["vcc"]
    if op.startswith("v_cndmask") and op.endswith("_e32"):
        srcs.append("vcc")
    if op in ("v_fmac_f32_e32", "v_pk_fmac_f32"):
        srcs += dst
    lat = LAT_PK if op.startswith("v_pk_") else LAT_VALU
    dn, dc, pred = 0, 0.0, None
    for r in srcs:
        if r in ready and ready[r][2] is not None and ready[r][1] >= dc:
            dn, dc, pred = max(dn, ready[r][0]), ready[r][1], ready[r][2]
    node = (t, dn + 1, dc + lat, pred)
    nodes.append(node)
    for r in dst:
        ready[r] = (node[1], node[2], len(nodes) - 1)

n_valu = len(nodes)
n_pk = sum(1 for n in nodes if n[0].startswith("v_pk_"))
end = max(range(n_valu), key=lambda k: nodes[k][2])
path = []
k = end
while k is not None:
    path.append(nodes[k][0]); k = nodes[k][3]
path.reverse()
print("kernel %s: %d VALU instructions (%d packed fp32) for one hinge visit = 6 rows (%.1f per row)" % (kernel, n_valu, n_pk, n_valu / 6.0))
print("issue bound     : %d x %.1f = %.0f cycles (one wave per SIMD, in-order issue)" % (n_valu, ISSUE, n_valu * ISSUE))
print("dependency bound: longest dependent path %d instructions, %.0f cycles at the measured dependent latencies" % (len(path), nodes[end][2]))
print("                  (%d packed on the path at %.2f, %d others at %.2f)" % (sum(p.startswith("v_pk_") for p in path), LAT_PK, sum(not p.startswith("v_pk_") for p in path), LAT_VALU))
print("\nlongest dependent path:")
for p in path:
    print("   ", p)
