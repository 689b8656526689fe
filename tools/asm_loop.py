"""Instruction mix of a kernel's hottest loop in a hipcc -S dump: python tools/asm_loop.py file.s mangled_substring"""
import sys, collections
s = open(sys.argv[1]).read()
key = sys.argv[2]
i = s.index(key)
i = s.index(':', i)
j = s.index('.Lfunc_end', i)
body = s[i:j].split('\n')
loops = [n for n, l in enumerate(body) if 'Loop Header' in l]
for n, l in enumerate(body):
    if 'scratch_' in l:
        print('scratch', n, l.strip())
print('lines', len(body), 'loops at', loops)
for loop in loops:
    c = collections.Counter()
    for l in body[loop + 1:]:
        if l.startswith('.LBB'):
            break
        op = l.strip().split()[0] if l.strip() else ''
        if op and not op.startswith(';'):
            c[op] += 1
    print('loop', loop, dict(c.most_common(14)))
