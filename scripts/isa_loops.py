#!/usr/bin/env python
"""Static instruction mix of ONE kernel of a `hipcc -S --cuda-device-only` listing, basic block by basic block (with the
loop-depth notes the compiler leaves on the labels): where a pivot trip / a stage of a sweep spends its instructions.
usage: isa_loops.py file.s <mangled-name substring> [min instructions per block, default 24]
  e.g. hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S csrc/zmp.hip -o /tmp/zmp.s
       isa_loops.py /tmp/zmp.s zmp_plan_kernelILi32ELi2E > profiles/r06_zmp_isa_k1.txt"""
import collections
import re
import sys

src, want = sys.argv[1], sys.argv[2]
floor = int(sys.argv[3]) if len(sys.argv) > 3 else 24
text = open(src).read()
m = re.search(r"^(_Z\w*%s\w*):" % re.escape(want), text, re.M)
if not m:
    raise SystemExit("no kernel matching %s" % want)
body = text[m.start():text.index("s_endpgm", m.start())].splitlines()


def klass(op):
    if op.startswith(("v_fma_f64", "v_fmac_f64")):
        return "fma64_dpp" if "dpp" in op else "fma64"
    if op.startswith(("v_mul_f64", "v_add_f64", "v_max_f64", "v_min_f64", "v_rcp_f64", "v_div_", "v_cmp_", "v_cmpx_", "v_cvt_", "v_rsq_", "v_sqrt_", "v_ldexp", "v_frexp", "v_trig")):
        return "fp/cmp"
    if op.startswith("v_cndmask"):
        return "select"
    if "dpp" in op or op.startswith(("v_permlane", "v_readlane", "v_writelane", "v_readfirstlane", "ds_bpermute", "ds_swizzle")):
        return "cross-lane"
    if op.startswith("v_accvgpr"):
        return "agpr move"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("v_"):
        return "valu other"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier")):
        return "wait/nop"
    return "salu"


order = ["fma64", "fma64_dpp", "fp/cmp", "select", "cross-lane", "agpr move", "valu other", "lds", "vmem", "salu", "branch", "wait/nop"]
blocks, cur = [], None
for l in body:
    lab = re.match(r"^(\.LBB\d+_\d+):\s*;?\s*(.*)$", l)
    if lab:
        cur = [lab.group(1), lab.group(2).strip(), collections.Counter()]
        blocks.append(cur)
        continue
    t = l.strip()
    if l.startswith("\t") and t and not t.startswith((".", ";")):
        if cur is None:
            cur = ["entry", "", collections.Counter()]
            blocks.append(cur)
        cur[2][klass(t.split()[0])] += 1
tot = collections.Counter()
for b in blocks:
    tot.update(b[2])
print("kernel %s: %d instructions in %d basic blocks" % (m.group(1), sum(tot.values()), len(blocks)))
print("%-14s %6s  %s" % ("block", "instrs", "  ".join("%s" % k for k in order)))
print("%-14s %6d  %s   (whole kernel)" % ("total", sum(tot.values()), "  ".join("%*d" % (len(k), tot[k]) for k in order)))
small = collections.Counter()
for name, note, c in blocks:
    n = sum(c.values())
    if n < floor:
        small.update(c)
        continue
    print("%-14s %6d  %s   %s" % (name, n, "  ".join("%*d" % (len(k), c[k]) for k in order), note))
print("%-14s %6d  %s   (the blocks below %d instructions)" % ("(small)", sum(small.values()), "  ".join("%*d" % (len(k), small[k]) for k in order), floor))
