# usage: kernel_resources.py file.s [filter]   -- VGPRs, spills, LDS and scratch of every kernel in a hipcc -S output
import re, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", txt, re.S):
    blk = m.group(0)
    g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk).group(1)
    name = g("name")
    if flt in name:
        print(f"{name[:70]:70s} vgpr {g('vgpr_count'):>4s} spill {g('vgpr_spill_count'):>3s} sgpr_spill {g('sgpr_spill_count'):>3s} lds {g('group_segment_fixed_size'):>6s} scratch {g('private_segment_fixed_size'):>4s}")
