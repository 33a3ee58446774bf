#!/usr/bin/env python3
"""Count instructions per kernel in a hipcc -save-temps .s file (static count, not executed count)."""
import sys, re, collections

def main(path):
    cur = None
    ops = collections.OrderedDict()
    for line in open(path):
        m = re.match(r'^(_Z\w+|[A-Za-z_]\w*):\s*(;.*)?$', line)
        if m and not line.startswith('.'):
            cur = m.group(1)
            ops[cur] = collections.Counter()
            continue
        s = line.strip()
        if cur is None or not s or s[0] in ';.' or s.endswith(':'):
            continue
        op = s.split()[0]
        if re.match(r'^(v_|s_|ds_|global_|buffer_|flat_|scratch_)', op):
            ops[cur][op] += 1
        if op == 's_endpgm':
            cur = None
    for name, c in ops.items():
        tot = sum(c.values())
        if tot < 10:
            continue
        v = sum(n for o, n in c.items() if o.startswith('v_'))
        s = sum(n for o, n in c.items() if o.startswith('s_'))
        mem = tot - v - s
        print(f"{name[:60]:60s} total={tot} valu={v} salu={s} mem={mem}")
        print("    " + ", ".join(f"{o}:{n}" for o, n in c.most_common(16)))

if __name__ == '__main__':
    main(sys.argv[1])
