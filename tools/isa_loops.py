"""Static view of a kernel's ISA (hipcc -S): innermost loops (backward branches) with their instruction mix.
Usage: python tools/isa_loops.py file.s kernel-name-prefix"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
st = [i for i, l in enumerate(lines) if l.startswith(sys.argv[2])][0]
body = []
for l in lines[st:]:
    body.append(l)
    if "s_endpgm" in l:
        break
label_at = {}
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        label_at[m.group(1)] = i
loops = []
for i, l in enumerate(body):
    m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", l)
    if m:
        tgt = m.group(1) or m.group(2)
        if tgt in label_at and label_at[tgt] < i:
            loops.append((label_at[tgt], i))
def is_instr(l):
    s = l.strip()
    return bool(s) and not s.startswith((";", ".")) and not s.endswith(":")
def mix(a, b):
    c = dict(valu=0, f64=0, salu=0, lds=0, vmem=0, readlane=0, dpp=0, nop=0, wait=0, trans=0, total=0)
    for l in body[a:b + 1]:
        if not is_instr(l):
            continue
        op = l.strip().split()[0]
        c["total"] += 1
        if op.startswith("v_"):
            c["valu"] += 1
            if "_f64" in op: c["f64"] += 1
            if "readlane" in op or "readfirstlane" in op: c["readlane"] += 1
            if "dpp" in l: c["dpp"] += 1
            if op.startswith(("v_rsq", "v_rcp", "v_sqrt", "v_sin", "v_cos", "v_exp", "v_log")): c["trans"] += 1
        elif op.startswith("s_nop"): c["nop"] += 1
        elif op.startswith("s_waitcnt"): c["wait"] += 1
        elif op.startswith("s_"): c["salu"] += 1
        elif op.startswith("ds_"): c["lds"] += 1
        elif op.startswith(("global_", "scratch_", "buffer_", "flat_")): c["vmem"] += 1
    return c
# innermost: loops that contain no other loop
inner = [lp for lp in loops if not any(o != lp and lp[0] <= o[0] and o[1] <= lp[1] for o in loops)]
print(f"kernel lines {len(body)}, total {mix(0, len(body) - 1)}")
for a, b in sorted(set(loops)):
    tag = "inner" if (a, b) in inner else "outer"
    print(f"{tag} loop lines {a}-{b} ({b - a}): {mix(a, b)}")
