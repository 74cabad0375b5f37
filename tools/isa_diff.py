"""Which kernels of two builds of libtrayhip.so differ, and how: per kernel the instruction counts, whether the OPCODE streams are the same
(operands such as kernel-argument offsets and registers aside), and for those that differ the number of changed vector / scalar instructions.
    python tools/isa_diff.py old.so new.so [name-prefix ...]"""
import difflib
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def disassemble(lib):
    t = tempfile.mkdtemp()
    subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, f"{t}/f.bin"], check=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={t}/f.bin", f"--output={t}/k.co", "--unbundle"], check=True)
    text = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", f"{t}/k.co"], check=True, capture_output=True, text=True).stdout
    kernels, cur = {}, None
    for line in text.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            cur = m.group(1); kernels[cur] = []
        elif cur and line[:1] in " \t" and line.strip():
            ins = line.split("//")[0].strip()
            if ins:
                kernels[cur].append(ins)
    return kernels


def short(name):
    out = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    return re.sub(r"\(.*", "", out)[:64]


a, b = disassemble(sys.argv[1]), disassemble(sys.argv[2])
prefixes = sys.argv[3:]
same_stream = changed = 0
for k in a:
    if k not in b or (prefixes and not any(short(k).startswith(p) or p in short(k) for p in prefixes)):
        continue
    oa, ob = [x.split()[0] for x in a[k]], [x.split()[0] for x in b[k]]
    if oa == ob:
        same_stream += 1
        n_operand = sum(1 for x, y in zip(a[k], b[k]) if x != y)
        print(f"{short(k):66s} {len(oa):6d} instructions  opcode stream identical  ({n_operand} operand-only differences)")
        continue
    changed += 1
    sm = difflib.SequenceMatcher(None, oa, ob, autojunk=False)
    dv = ds = 0
    for tag, i1, i2, j1, j2 in sm.get_opcodes():
        if tag == "equal":
            continue
        for op in oa[i1:i2] + ob[j1:j2]:
            if op.startswith("s_"):
                ds += 1
            else:
                dv += 1
    print(f"{short(k):66s} {len(oa):6d} -> {len(ob):6d} instructions  differs: {dv} vector / memory, {ds} scalar instructions inserted or deleted")
print(f"{same_stream} kernels with identical opcode streams, {changed} changed; only in the new build: {sorted(short(k) for k in b if k not in a)}")
