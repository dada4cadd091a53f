"""Register / scratch use of every kernel in a hipcc -S --cuda-device-only listing: python tools/kernel_regs.py file.s"""
import re, subprocess, sys
s = open(sys.argv[1]).read()
pat = re.compile(r'\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.sgpr_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)')
for m in pat.finditer(s):
    try:
        name = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
    except OSError:
        name = m.group(1)
    name = re.sub(r'\(.*', '', name).replace('void matinv::', '')
    print(f"{name:60s} scratch {m.group(2):>5s}  sgpr {m.group(3):>3s}  vgpr {m.group(4):>3s}  spilled {m.group(5):>3s}")
