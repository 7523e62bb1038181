"""Register / LDS / scratch footprint of the kernels in a `hipcc -S --cuda-device-only` listing whose name contains a substring.

usage: kernel_regs.py LISTING.s SUBSTRING
"""
import re
import sys

s = open(sys.argv[1]).read()
meta = s[s.index("amdhsa.kernels:"):]
for blk in meta.split("  - .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", blk).group(1)
    if sys.argv[2] not in name:
        continue
    f = lambda k: re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)
    print("%-90s vgpr %3s agpr %3s sgpr %3s scratch %s spill %s" % (
        name[:90], f("vgpr_count"), blk.split()[0], f("sgpr_count"), f("private_segment_fixed_size"), f("vgpr_spill_count")))
