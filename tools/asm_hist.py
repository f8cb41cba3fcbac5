"""Instruction-class histogram per basic block of one kernel in a hipcc -S listing:
python tools/asm_hist.py file.s <kernel-name-substring>   (blocks containing MFMAs only)"""
import re, sys, collections
src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and key in l and l.rstrip().endswith(("args", ":")) or (l.startswith("_ZN") and key in l and ":" in l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
blocks, cur = collections.OrderedDict(), "entry"
blocks[cur] = []
for l in lines[start + 1:end]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        cur = m.group(1); blocks[cur] = []; continue
    t = l.strip()
    if not t or t.startswith((";", ".")): continue
    blocks[cur].append(t.split()[0])
def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith(("v_exp", "v_rcp", "v_log", "v_rsq", "v_sqrt")): return "trans"
    if op.startswith("v_pk_"): return "valu_pk"
    if op.startswith("v_cvt"): return "valu_cvt"
    if op.startswith(("v_permlane", "v_readfirstlane", "v_readlane", "v_mov_b32_dpp")): return "valu_x"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_read") or op.startswith("ds_load"): return "lds_rd"
    if op.startswith("ds_"): return "lds_wr"
    if op.startswith(("global_load", "buffer_load", "flat_load")): return "vmem_ld"
    if op.startswith(("global_store", "buffer_store", "flat_store")): return "vmem_st"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_load"): return "smem"
    if op.startswith("s_"): return "salu"
    return "other"
for name, ops in blocks.items():
    h = collections.Counter(cls(o) for o in ops)
    if h["mfma"] or ("-a" in sys.argv and len(ops) >= 8):
        print(f"{name:10s} n={len(ops):4d} ", dict(sorted(h.items())))
        if "-v" in sys.argv:
            print("    ", collections.Counter(o for o in ops if cls(o) == "valu").most_common(14))
