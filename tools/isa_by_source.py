"""VALU instructions of a kernel's ISA attributed to source lines (compile with -S -gline-tables-only):  which source lines of the interior-point
iteration cost how many wave instructions.  Static counts; lines inside a loop are listed with the loop's label so that they can be weighted by
hand.  Usage: python tools/isa_by_source.py file.s kernel-prefix [file-filter] [first-line last-line]"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', l)
    if m:
        files[int(m.group(1))] = m.group(2).split("/")[-1]
st = [i for i, l in enumerate(lines) if l.startswith(sys.argv[2])][0]
flt = sys.argv[3] if len(sys.argv) > 3 else ""
lo, hi = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (0, 10 ** 9)
cur = ("?", 0)
cnt = collections.Counter(); f64 = collections.Counter()
for l in lines[st:]:
    if "s_endpgm" in l:
        break
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
        continue
    s = l.strip()
    if not s or s.startswith((";", ".")) or s.endswith(":"):
        continue
    op = s.split()[0]
    if op.startswith("v_"):
        cnt[cur] += 1
        if "_f64" in op:
            f64[cur] += 1
tot = sum(cnt.values())
print("total VALU", tot)
rows = [(k, v) for k, v in cnt.items() if flt in k[0] and lo <= k[1] <= hi]
for (f, ln), v in sorted(rows, key=lambda kv: -kv[1])[:60]:
    print(f"{f}:{ln:<5d} valu {v:5d}  f64 {f64[(f, ln)]:5d}")
