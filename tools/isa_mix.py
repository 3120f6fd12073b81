#!/usr/bin/env python3
"""Static instruction mix per kernel from the device assembly (hipcc -S --cuda-device-only):
   tools/isa_mix.py <file.s> <substring of the mangled kernel name> [...]"""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
parts = re.split(r'\n(_ZN4mals\w+):[^\n]*\n', txt)
for name, body in zip(parts[1::2], parts[2::2]):
    if not any(p in name for p in sys.argv[2:]):
        continue
    body = body.split('.Lfunc_end')[0]
    cnt = collections.Counter()
    for line in body.split('\n'):
        line = line.strip()
        m = re.match(r'([a-z_0-9]+)', line)
        if not m or line.startswith((';', '.')):
            continue
        op = m.group(1)
        if op.startswith('v_mfma'):
            cnt[op] += 1
        elif 'dpp' in line or 'row_newbcast' in line or 'quad_perm' in line or 'row_ror' in line:
            cnt['dpp'] += 1
        elif op.startswith('ds_bpermute'):
            cnt['bperm'] += 1
        elif op.startswith('ds_'):
            cnt['lds'] += 1
        elif op.startswith('global_load'):
            cnt['gload'] += 1
        elif op.startswith('global_store') or op.startswith('global_atomic'):
            cnt['gstore'] += 1
        elif op.startswith('scratch'):
            cnt['scratch'] += 1
        elif op.startswith('v_cvt') or op.startswith('v_fma_mix'):
            cnt['cvt/mix'] += 1
        elif op.startswith('v_readlane') or op.startswith('v_readfirstlane'):
            cnt['readlane'] += 1
        elif op.startswith('v_rcp') or op.startswith('v_rsq') or op.startswith('v_sqrt'):
            cnt['trans'] += 1
        elif op.startswith('v_cndmask'):
            cnt['cndmask'] += 1
        elif op.startswith('v_accvgpr'):
            cnt['accvgpr'] += 1
        elif op.startswith('v_'):
            cnt['valu'] += 1
        elif op == 's_waitcnt':
            cnt['waitcnt'] += 1
        elif op == 's_nop':
            cnt['nop'] += 1
        elif op.startswith('s_'):
            cnt['salu'] += 1
    print(name[9:70], sum(cnt.values()), dict(sorted(cnt.items())))
