"""Which source a kernel's counter figures were taken from: sha256 of every file under minicom_amd/csrc and, per kernel name,
the file that defines it.  profiles/pmc_constants.json records these at profiling time; bench.py recomputes them and refuses
(null + "stale") the figures of a kernel whose file, or a shared header, has changed since.
   python tools/source_sha.py            prints the JSON"""
import glob
import hashlib
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "minicom_amd", "csrc")


def file_shas():
    return {os.path.basename(p): hashlib.sha256(open(p, "rb").read()).hexdigest()[:16]
            for p in sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.hpp")))}


def kernel_files():
    """{kernel base name: file} for every __global__ function of the library"""
    out = {}
    for p in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
        txt = open(p, errors="ignore").read()
        for m in re.finditer(r"__global__", txt):
            k = re.search(r"\b(k_[A-Za-z0-9_]+)\s*\(", txt[m.end():m.end() + 400])
            if k:
                out.setdefault(k.group(1), os.path.basename(p))
    return out


def kernel_sha(kernel, shas=None, kfiles=None):
    """what a kernel's code depends on: its own file + the shared headers; None for a kernel this tree does not define"""
    shas = shas or file_shas(); kfiles = kfiles or kernel_files()
    base = kernel.split("<")[0].strip()
    f = kfiles.get(base)
    if not f:
        return None
    return hashlib.sha256("|".join([shas[f]] + [shas[h] for h in sorted(shas) if h.endswith(".hpp")]).encode()).hexdigest()[:16]


def norm(name):
    """kernel names compared without blanks ("k_x<5, true>" == "k_x<5,true>"), 'void ' and argument lists dropped"""
    n = name.replace("(anonymous namespace)::", "")
    if n.startswith("void "):
        n = n[5:]
    depth = 0
    for i, ch in enumerate(n):                  # cut at the argument list: the first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            n = n[:i]
            break
    return n.replace(" ", "")


if __name__ == "__main__":
    print(json.dumps({"files": file_shas(), "kernels": kernel_files()}, indent=1, sort_keys=True))
