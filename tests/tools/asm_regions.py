"""Diagnostic: instruction counts between the PZG_MARK comments of build/asm/<tag>.s (tests/tools/asm11.sh <tag> -DPZG_MARKS),
in layout order.  Usage: python tests/tools/asm_regions.py <tag>"""
import re, sys, os, collections
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cur, c = "(start)", collections.Counter()
def flush():
    if sum(c.values()):
        print(f"{cur:14s} VALU {c['VALU']:4d} SALU {c['SALU']:4d} BR {c['BR']:3d} WAIT {c['WAIT']:3d} LDS {c['LDS']:3d} VMEM {c['VMEM']:3d} labels {c['LBL']:3d}")
for l in open(os.path.join(root, "build", "asm", sys.argv[1] + ".s")):
    m = re.search(r"##MARK (\S+)", l)
    if m:
        flush(); cur, c = m.group(1), collections.Counter(); continue
    if re.match(r"\.LBB", l): c["LBL"] += 1
    m = re.match(r"\s+([vs]_\w+|ds_\w+|global_\w+|buffer_\w+|flat_\w+)", l)
    if not m: continue
    op = m.group(1)
    c["WAIT" if op.startswith(("s_waitcnt", "s_nop")) else "BR" if op.startswith(("s_cbranch", "s_branch")) else
      "SALU" if op.startswith("s_") else "VALU" if op.startswith("v_") else "LDS" if op.startswith("ds_") else "VMEM"] += 1
flush()
