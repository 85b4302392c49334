"""Register / spill / LDS figures per kernel from a hipcc -S listing (amdgpu metadata).  python tools/asm_regs.py build/asm/x.s [name filter]"""
import re, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in re.split(r"\n  - \.agpr_count", txt)[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = g("name")
    if flt in name:
        print(f"{name[:70]:70s} vgpr {g('vgpr_count'):>4} sgpr {g('sgpr_count'):>4} sspill {g('sgpr_spill_count'):>3} vspill {g('vgpr_spill_count'):>3} lds {g('group_segment_fixed_size'):>6} scratch {g('private_segment_fixed_size')}")
