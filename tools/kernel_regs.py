#!/usr/bin/env python3
"""Print register use / spills per kernel from hipcc's --save-temps .s files (amdhsa metadata)."""
import re
import subprocess
import sys

for path in sys.argv[1:]:
    txt = open(path).read()
    for blk in txt.split('  - .agpr_count:')[1:]:
        blk = '.agpr_count:' + blk
        f = dict(re.findall(r'\.(\w+):\s+(\S+)', blk))
        name = subprocess.run(['c++filt', f.get('name', '?')], capture_output=True, text=True).stdout.strip()
        name = name.replace('(anonymous namespace)::', '').replace('void ', '')
        print('%-70s vgpr %3s agpr %3s sgpr %3s  spill v %3s s %3s  lds %s' % (name[:70], f.get('vgpr_count'), f.get('agpr_count'),
              f.get('sgpr_count'), f.get('vgpr_spill_count'), f.get('sgpr_spill_count'), f.get('group_segment_fixed_size')))
