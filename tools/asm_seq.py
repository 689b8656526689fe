"""One-line schematic of a kernel's hot loop from a hipcc -S dump: M mfma, r ds_read, w ds_write, G global load,
[..] s_waitcnt, |BAR| barrier, nK s_nop K, . anything else.   python tools/asm_seq.py file.s mangled_substring"""
import sys
s = open(sys.argv[1]).read()
i = s.index(sys.argv[2]); i = s.index(':', i); j = s.index('.Lfunc_end', i)
body = s[i:j].split('\n')
for loop in [n for n, l in enumerate(body) if 'Loop Header' in l]:
    n = loop + 1
    seq = []
    while not body[n].startswith('.LBB'):
        l = body[n].strip()
        if l and not l.startswith(';'):
            op = l.split()[0]
            if op.startswith('v_mfma'): seq.append('M')
            elif op.startswith('ds_read'): seq.append('r')
            elif op.startswith('ds_write'): seq.append('w')
            elif op.startswith('global_load') or op.startswith('buffer_load'): seq.append('G')
            elif op.startswith('global_store'): seq.append('S')
            elif op == 's_waitcnt': seq.append('[' + l.split(None, 1)[1].replace('cnt', '') + ']')
            elif op == 's_barrier': seq.append('|BAR|')
            elif op == 's_nop': seq.append('n' + l.split()[1])
            else: seq.append('.')
        n += 1
    print('loop@%d:' % loop, ''.join(seq))
