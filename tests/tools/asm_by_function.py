"""Diagnostic: static instruction counts of inflate_kernel<11> per source function (needs build/asm/<tag>.s made by
`tests/tools/asm11.sh <tag> -gline-tables-only`).  Usage: python tests/tools/asm_by_function.py <tag> [function ...]"""
import re, sys, os, collections
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag = sys.argv[1]
want = set(sys.argv[2:])
files = {}
funcs = {}  # file -> sorted list of (line, name)
def load(fn):
    if fn in funcs: return
    out = []
    try:
        for i, l in enumerate(open(fn), 1):
            m = re.match(r"\s*(?:template <[^>]*>\s*)?(?:static\s+)?PZG_FN\s+[\w:<> \*&]+?\s+(\w+)\s*\(", l)
            if m: out.append((i, m.group(1)))
    except OSError:
        pass
    funcs[fn] = out
def func_of(fn, line):
    load(fn)
    name = "?"
    for l, n in funcs[fn]:
        if l <= line: name = n
        else: break
    return os.path.basename(fn) + ":" + name
cnt = collections.defaultdict(lambda: collections.Counter())
cur = "?"
lines = collections.defaultdict(lambda: collections.Counter())
incl = collections.defaultdict(lambda: collections.Counter())
csrc = os.path.join(root, "pure_zlib_amd", "csrc")
chain = []
for l in open(os.path.join(root, "build", "asm", tag + ".s")):
    m = re.match(r"\s*\.loc\s+\d+\s+(\d+)\s+\d+[^;]*;\s*(.*)", l)
    if m:
        if int(m.group(1)) == 0: continue  # compiler-generated: stays with the previous line
        frames = re.findall(r"([\w\./\-]+):(\d+):\d+", m.group(2))
        ours = [(os.path.join(csrc, os.path.basename(f)), int(n)) for f, n in frames if os.path.basename(f) in ("inflate_core.h", "wave.h", "pzg_kernels.hip")]
        if ours:
            # innermost frame in inflate_core.h / pzg_kernels.hip (wave.h helpers count for their caller)
            inner = [x for x in ours if not x[0].endswith("wave.h")] or ours
            cur = func_of(*inner[0]); curline = inner[0][1]
            chain = list(dict.fromkeys(func_of(*x) for x in inner))
        continue
    m = re.match(r"\s+([vs]_\w+|ds_\w+|global_\w+|buffer_\w+|flat_\w+)", l)
    if not m: continue
    op = m.group(1)
    kind = ("WAIT" if op.startswith("s_waitcnt") or op.startswith("s_nop") else "BR" if op.startswith("s_cbranch") or op.startswith("s_branch") else
            "SALU" if op.startswith("s_") else "VALU" if op.startswith("v_") else "LDS" if op.startswith("ds_") else "VMEM")
    cnt[cur][kind] += 1
    for f in chain: incl[f][kind] += 1
    if cur.split(":")[1] in want: lines[(cur, curline)][kind] += 1
tot = collections.Counter()
print(f"{'function':44s} VALU SALU   BR WAIT  LDS VMEM")
for f, c in sorted(cnt.items(), key=lambda kv: -sum(kv[1].values())):
    print(f"{f:44s} {c['VALU']:4d} {c['SALU']:4d} {c['BR']:4d} {c['WAIT']:4d} {c['LDS']:4d} {c['VMEM']:4d}")
    tot.update(c)
print("inclusive (with inlined callees):")
for f, c in sorted(incl.items(), key=lambda kv: -sum(kv[1].values())):
    print(f"{f:44s} {c['VALU']:4d} {c['SALU']:4d} {c['BR']:4d} {c['WAIT']:4d} {c['LDS']:4d} {c['VMEM']:4d}")
print(f"{'total':44s} {tot['VALU']:4d} {tot['SALU']:4d} {tot['BR']:4d} {tot['WAIT']:4d} {tot['LDS']:4d} {tot['VMEM']:4d}")
for (f, ln), c in sorted(lines.items()):
    print(f"  {f}:{ln}  " + " ".join(f"{k}={v}" for k, v in c.items()))
