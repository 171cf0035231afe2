#!/bin/bash
# Lab: what SQ_ACTIVE_INST_VALU counts per instruction of each issue class (the unit behind "the vector port is N % busy"):
# rocprofv3 counters over tests/tools/op_cost.hip's kernels for a full-rate, a half-rate and a scalar-operand opcode.
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/vc; timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/vc -o p -- $root/build/op_cost 2000 16 v_add_u32_e32,v_alignbit_b32,v_and_b32_sgpr,v_lshlrev_b32_c,v_readlane,s_add_u32 > /tmp/vc.log 2>&1
cat /tmp/vc.log | tail -8
python3 - "$(find /tmp/vc -name '*counter_collection.csv' | head -1)" <<'PY'
import csv, sys, collections
rows = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    rows.setdefault((r["Dispatch_Id"], r["Kernel_Name"][:40], r["Grid_Size"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
for (d, k, g), c in rows.items():
    iv = c.get("SQ_INSTS_VALU", 0)
    print(f"dispatch {d:>3s} {k:40s} grid {g:>8s}  INSTS_VALU {iv:.4g}  ACTIVE_INST_VALU {c.get('SQ_ACTIVE_INST_VALU', 0):.4g} "
          f"(x4 / inst = {4 * c.get('SQ_ACTIVE_INST_VALU', 0) / max(iv, 1):.2f} cycles)  GRBM/8 {c.get('GRBM_GUI_ACTIVE', 0) / 8:.4g}  "
          f"busy = ACTIVE x 4 / 1024 / (GRBM / 8) = {4 * c.get('SQ_ACTIVE_INST_VALU', 0) / 1024 / max(c.get('GRBM_GUI_ACTIVE', 1) / 8, 1):.3f}")
PY
