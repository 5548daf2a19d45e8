"""Summarise hipcc's -Rpass-analysis=kernel-resource-usage remarks (stderr of a build) as one line per kernel.
Usage: python tools/kernel_resources.py build.log [name-filter]"""
import re
import subprocess
import sys


def parse(text):
    out, cur = [], None
    for line in text.splitlines():
        m = re.search(r"remark: (.*?) \[-Rpass", line)
        if not m:
            continue
        body = m.group(1).strip()
        body = re.sub(r"^\S+:\d+:\d+:\s*", "", body)      # "file:line:col: " prefix (present when the source path is relative)
        if body.startswith("Function Name:"):
            cur = {"name": body.split(":", 1)[1].strip()}
            out.append(cur)
        elif cur is not None and ":" in body:
            k, v = body.split(":", 1)
            cur[k.strip()] = v.strip()
    return out


def demangle(n):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip().split("(")[0]
    except OSError:
        return n


if __name__ == "__main__":
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    for k in parse(open(sys.argv[1]).read()):
        name = demangle(k["name"])
        if flt and flt not in name:
            continue
        print(f"{name:70s} VGPR {k.get('VGPRs', '?'):>4} AGPR {k.get('AGPRs', '?'):>4} scratch {k.get('ScratchSize [bytes/lane]', '?'):>5} "
              f"LDS {k.get('LDS Size [bytes/block]', '?'):>6} occ {k.get('Occupancy [waves/SIMD]', '?')}")
