"""Static check of the relay kernel's ISA (hipcc -save-temps .s): between an inline-asm `global_load_dword vX` (the bias values) and the next
`s_waitcnt vmcnt(..)`, no instruction may read or write vX -- the compiler does not know the register is still in flight and a copy
it places there would move stale bits.  Also lists every compiler-made `s_waitcnt vmcnt(0)` (outside ASMSTART/ASMEND) and scratch use."""
import re, sys

def regs_of(line):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", line):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", line):
        out.add(int(m.group(1)))
    return out

text = open(sys.argv[1]).read().split("\n")
kern = None
bad = 0
inasm = False
pending = {}
for i, l in enumerate(text):
    m = re.match(r"^(_ZN2xl22gemm_bf16_relay_kernel\w+):", l)
    if m:
        kern = m.group(1); pending = {}
    if kern is None:
        continue
    if "ASMSTART" in l: inasm = True; continue
    if "ASMEND" in l: inasm = False; continue
    ins = l.strip()
    if not ins or ins.startswith(";") or ins.startswith("."):
        continue
    if ins.startswith("s_endpgm"):
        kern = None; continue
    if "scratch_" in ins:
        print(f"{kern}:{i+1}: scratch access: {ins}"); bad += 1
    if ins.startswith("s_waitcnt") and "vmcnt" in ins:
        if not inasm and "vmcnt(0)" in ins:
            print(f"{kern}:{i+1}: compiler-made {ins}"); bad += 1
        pending = {}
        continue
    touched = regs_of(ins)
    for r, where in list(pending.items()):
        if r in touched:
            print(f"{kern}:{i+1}: v{r} (loaded at line {where}, not yet waited for) used by: {ins}"); bad += 1
    if inasm and ins.startswith("global_load_dword "):
        d = re.match(r"global_load_dword v(\d+),", ins)
        pending[int(d.group(1))] = i + 1
print("relay ISA check:", "OK" if bad == 0 else f"{bad} finding(s)")
sys.exit(1 if bad else 0)
