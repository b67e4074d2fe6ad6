"""Development aid: static instruction mix of the kernels in a hipcc -S listing.  usage: isa_stats.py file.s [substr]"""
import collections, re, sys
lines = open(sys.argv[1]).read().splitlines()
want = sys.argv[2] if len(sys.argv) > 2 else ""
cur, ops = None, None
def flush():
    if cur and want in cur:
        tot = sum(ops.values())
        pick = lambda pre: sum(v for k, v in ops.items() if k.startswith(pre))
        print(cur[:60], "instrs", tot, "| scratch", pick("scratch_"), "| readlane", ops["v_readlane_b32"], "writelane", ops["v_writelane_b32"],
              "| s_waitcnt", ops["s_waitcnt"], "| ds", pick("ds_"), "| global", pick("global_"), "| fma64", ops["v_fma_f64"], "mul64", ops["v_mul_f64"], "add64", ops["v_add_f64"],
              "| cndmask", pick("v_cndmask"), "| dpp", pick("v_mov_b32_dpp"), "| v_mov", ops["v_mov_b32_e32"], "| salu", pick("s_"), "| branch", pick("s_cbranch"))
for l in lines:
    m = re.match(r"^(_Z\w+):", l)
    if m:
        flush()
        cur, ops = m.group(1), collections.Counter()
        continue
    if cur and l.startswith("\t") and not l.strip().startswith((".", ";")):
        ops[l.split()[0]] += 1
        if l.split()[0] == "s_endpgm":
            flush()
            cur = None
