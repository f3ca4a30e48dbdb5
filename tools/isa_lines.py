"""Which source lines a kernel's instructions come from: compiles a generated model source with line tables
(-gline-tables-only does not change the code), disassembles one kernel and prints instruction counts per source line.

    python tools/isa_lines.py pycollo_amd/_cache/model_<digest>_<stamp>_n5_5_5_5.hip pc_bulk_all_r_w2 [top]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
src, kernel = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
tmp = tempfile.mkdtemp(prefix="isa_lines_")
hsaco, co = os.path.join(tmp, "k.hsaco"), os.path.join(tmp, "k.co")
extra = [f"-D{d}" for d in os.environ.get("PYCOLLO_AMD_DEFINES", "").split()]
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "--genco", "-O3", "-std=c++17", "-ffp-contract=off", "-mllvm",
                "-amdgpu-kernarg-preload-count=10", "-gline-tables-only", f"-I{ROOT}/pycollo_amd/csrc", "-o", hsaco, src] + extra,
               check=True)
subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                f"--input={hsaco}", f"--output={co}"], check=True)
text = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "-l", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
lines = text.split("\n")
start = next(i for i, l in enumerate(lines) if f"<{kernel}>:" in l)
end = next((i for i in range(start + 1, len(lines)) if re.match(r"^[0-9a-f]+ <", lines[i])), len(lines))
cur, hist, ops = None, collections.Counter(), collections.defaultdict(collections.Counter)
for l in lines[start + 1:end]:
    m = re.match(r"^; (.*):(\d+)$", l.strip())
    if m:
        cur = (os.path.basename(m.group(1)), int(m.group(2)))
        continue
    tok = l.split()
    if tok and tok[0].startswith(("s_", "v_", "ds_", "global_", "buffer_", "flat_", "scratch_")):
        hist[cur] += 1
        ops[cur][tok[0]] += 1
total = sum(hist.values())
print(f"{kernel}: {total} instructions")
byfile = collections.Counter()
for (f, _), c in hist.items():
    byfile[f] += c
print("by file:", dict(byfile.most_common(5)))
cache = {}
for (f, ln), c in hist.most_common(top):
    path = {"pc_kernels.hpp": f"{ROOT}/pycollo_amd/csrc/pc_kernels.hpp"}.get(f, src if f == os.path.basename(src) else None)
    text = ""
    if path:
        if path not in cache:
            cache[path] = open(path).read().split("\n")
        text = cache[path][ln - 1].strip()[:110] if ln - 1 < len(cache[path]) else ""
    print(f"{c:6d} {100 * c / total:5.1f}%  {f}:{ln:<5d} {dict(ops[(f, ln)].most_common(3))}  {text}")
