"""Register use of the kernels in a device-only assembly file (hipcc --cuda-device-only -S): VGPRs, AGPRs, spills, scratch.
usage: python3 tools/isa_regs.py file.s [substring of the mangled kernel name]"""
import re, sys
s = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r'\.name:\s+(\S+)\n(.*?)\.wavefront_size', s, re.S):
    name, body = m.group(1), m.group(2)
    if want not in name:
        continue
    g = lambda k: (re.search(r'\.' + k + r':\s+(\d+)', body) or [None, None])[1]
    print(f"{name[:90]:90s} vgpr {g('vgpr_count')} agpr {g('agpr_count')} spill {g('vgpr_spill_count')} "
          f"scratch {g('private_segment_fixed_size')} sgpr {g('sgpr_count')} lds {g('group_segment_fixed_size')}")
