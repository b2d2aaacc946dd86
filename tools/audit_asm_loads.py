"""Audit a hipcc .s for compiler accesses to the destination registers of hand-waited inline-asm loads
(development aid for the weight prefetch of conv_wino_kernel / conv_wino16_kernel): between an asm `global_load_dwordx4` and the
hand-placed `s_waitcnt vmcnt(N)` that retires it, no other instruction may read or write its destination."""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
keep = int(sys.argv[2]) if len(sys.argv) > 2 else 8
inflight = []          # list of (lo, hi, line_no) in issue order
in_asm = False
bad = 0
def regs(txt):
    out = []
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", txt):
        if m.group(1): out.append((int(m.group(1)), int(m.group(2))))
        else: out.append((int(m.group(3)), int(m.group(3))))
    return out
for i, l in enumerate(lines):
    t = l.strip()
    if t.startswith(";;#ASMSTART"): in_asm = True; continue
    if t.startswith(";;#ASMEND"): in_asm = False; continue
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"): continue
    if in_asm and t.startswith("global_load_lds_dwordx4"):
        inflight.append((-1, -1, i + 1)); continue          # an LDS-DMA occupies a slot of the in-order queue, no register
    if in_asm and t.startswith("global_load_dwordx4"):
        r = regs(t.split(",")[0])[0]
        inflight.append((r[0], r[1], i + 1)); continue
    if in_asm and t.startswith("s_waitcnt vmcnt("):
        n = int(re.search(r"vmcnt\((\d+)\)", t).group(1))
        inflight = inflight[len(inflight) - n:] if n < len(inflight) else inflight
        if n == 0: inflight = []
        continue
    if t.startswith("s_waitcnt") and "vmcnt(0)" in t:
        inflight = []; continue
    for (a, b) in regs(t):
        for (lo, hi, ln) in inflight:
            if a <= hi and b >= lo:
                print(f"line {i+1}: '{t[:70]}' touches v[{lo}:{hi}] loaded at line {ln} and not yet waited"); bad += 1
print("violations:", bad)
