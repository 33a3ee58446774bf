#!/usr/bin/env python3
"""Count instructions per kernel in a hipcc -save-temps .s file (static count, not executed count)."""
import sys, re, collections

# Issue cost in full-rate slots (one wave64 VALU instruction per 2 cycles per SIMD), from the
# measured rates in profiles/r01_valu_issue_rates.txt: these opcodes issue at full rate,
# every other VALU opcode measured (v_alignbit_b32, v_add3_u32, v_perm_b32, v_lshlrev_b32,
# v_cmp_*, v_min/max, v_and_or, v_xad, v_bfe, 64-bit shifts/adds, DPP/SDWA, ...) at half rate.
FULL_RATE = {"v_add_u32_e32", "v_sub_u32_e32", "v_subrev_u32_e32", "v_xor_b32_e32", "v_and_b32_e32", "v_or_b32_e32",
             "v_lshrrev_b32_e32", "v_ashrrev_i32_e32", "v_mov_b32_e32", "v_bitop3_b32", "v_not_b32_e32", "v_cndmask_b32_e32"}


def slots(counter):
    return sum(n * (1 if op in FULL_RATE else 2) for op, n in counter.items() if op.startswith("v_"))


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
        print(f"{name[:60]:60s} total={tot} valu={v} valu_slots={slots(c)} salu={s} mem={mem}")
        print("    " + ", ".join(f"{o}:{n}" for o, n in c.most_common(16)))

if __name__ == '__main__':
    main(sys.argv[1])
