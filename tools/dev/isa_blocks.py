#!/usr/bin/env python3
"""dev: basic blocks of one kernel's ISA listing (the file tools/dev/isa_mix.sh leaves in /tmp/isa_mix_k.s):
per block its line range, instruction mix and successors; back edges marked.  With --weights FILE (lines
`label count`) the blocks' VALU counts are weighted into a dynamic total.
usage: tools/dev/isa_blocks.py [/tmp/isa_mix_k.s] [--min-valu N]"""
import re
import sys


def classify(op):
    if op.startswith('v_readlane') or op.startswith('v_readfirstlane'):
        return 'rdl'
    if op.startswith('v_writelane'):
        return 'wrl'
    if op.endswith('_dpp') or '_dpp' in op:
        return 'dpp'
    if op.startswith('v_mov_b32') or op.startswith('v_mov_b64') or op.startswith('v_accvgpr'):
        return 'mov'
    if re.match(r'v_(fma|fmac|add|mul|max|min|div_scale|div_fmas|div_fixup|rcp|rsq|sqrt|trig_preop|ldexp|frexp|fract|floor|ceil|rndne|cvt)\w*_f64', op) or op.startswith('v_cvt_f64') or op.startswith('v_cvt_i32_f64'):
        return 'f64'
    if op.startswith('v_cndmask'):
        return 'cnd'
    if op.startswith('v_cmp') or op.startswith('v_cmpx'):
        return 'cmp'
    if op.startswith('v_'):
        return 'vother'
    if op.startswith('ds_'):
        return 'lds'
    if op.startswith('global_') or op.startswith('scratch_') or op.startswith('buffer_') or op.startswith('flat_'):
        return 'vmem'
    if op.startswith('s_waitcnt'):
        return 'wait'
    if op.startswith('s_nop'):
        return 'nop'
    if op.startswith('s_cbranch') or op.startswith('s_branch'):
        return 'br'
    if op.startswith('s_load') or op.startswith('s_buffer_load'):
        return 'smem'
    if op.startswith('s_'):
        return 'salu'
    return 'other'


def main():
    path = '/tmp/isa_mix_k.s'
    min_valu = 0
    args = sys.argv[1:]
    while args:
        a = args.pop(0)
        if a == '--min-valu':
            min_valu = int(args.pop(0))
        else:
            path = a
    blocks = []
    cur = {'label': 'entry', 'start': 1, 'ins': [], 'succ': [], 'depth': 0, 'hdr': ''}
    for i, line in enumerate(open(path), 1):
        s = line.strip()
        m = re.match(r'^(\.LBB\d+_\d+):', s)
        if not m and (not s or s.startswith(';') or s.startswith('.')):
            continue
        if m:
            cur['end'] = i - 1
            blocks.append(cur)
            cur = {'label': m.group(1), 'start': i, 'ins': [], 'succ': []}
            d = re.search(r'Depth=(\d+)', s)
            h = re.search(r'Header=(\w+)', s)
            inner = re.search(r'Inner Loop Header', s)
            cur['depth'] = int(d.group(1)) if d else 0
            cur['hdr'] = h.group(1) if h else ('*' if inner or 'Loop Header' in s else '')
            continue
        op = s.split()[0]
        cur['ins'].append((op, s))
        if op.startswith('s_cbranch') or op.startswith('s_branch'):
            cur['succ'].append(s.split()[-1])
    cur['end'] = i
    blocks.append(cur)
    order = {b['label']: k for k, b in enumerate(blocks)}
    cats = ['f64', 'rdl', 'wrl', 'mov', 'dpp', 'cnd', 'cmp', 'vother', 'lds', 'vmem', 'smem', 'salu', 'wait', 'nop', 'br']
    print('%-12s %11s %2s %-9s %5s | %s | succ' % ('block', 'lines', 'd', 'loop', 'valu', ' '.join('%4s' % c[:4] for c in cats)))
    by_depth = {}
    tot = dict.fromkeys(cats, 0)
    for k, b in enumerate(blocks):
        cnt = dict.fromkeys(cats + ['other'], 0)
        for op, _ in b['ins']:
            cnt[classify(op)] += 1
        valu = sum(cnt[c] for c in ('f64', 'rdl', 'wrl', 'mov', 'dpp', 'cnd', 'cmp', 'vother'))
        for c in cats:
            tot[c] += cnt[c]
        bd = by_depth.setdefault((b['depth'], b['hdr']), dict.fromkeys(cats + ['valu', 'all'], 0))
        bd['valu'] += valu
        bd['all'] += len(b['ins'])
        for c in cats:
            bd[c] += cnt[c]
        if valu < min_valu:
            continue
        succ = []
        for t in b['succ']:
            succ.append(t + ('^' if t in order and order[t] <= k else ''))
        print('%-12s %5d-%-5d %2d %-9s %5d | %s | %s' % (b['label'], b['start'], b['end'], b['depth'], b['hdr'], valu,
                                               ' '.join('%4d' % cnt[c] for c in cats), ' '.join(succ)))
    print('total', tot)
    print('by loop (depth, header): all instructions, valu, then the mix')
    for k in sorted(by_depth):
        v = by_depth[k]
        print('  depth %d %-10s all %5d valu %5d | %s' % (k[0], k[1], v['all'], v['valu'], ' '.join('%s %d' % (c, v[c]) for c in cats if v[c])))


if __name__ == '__main__':
    main()
