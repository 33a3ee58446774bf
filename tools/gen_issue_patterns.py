#!/usr/bin/env python3
"""Generates tools/issue_patterns.inc: fixed instruction sequences (explicit VGPRs, one asm block per loop body, so hipcc
can neither reorder them nor put s_nop between them) for tools/issue_patterns.hip.  Question they answer: at the
clock the chip holds, which property of the SHA-256 instruction stream costs VALU issue slots -- producer/consumer
distance, operand banks of the 3-source instructions, or the order of half-rate and full-rate instructions?

Slots: full-rate instruction = 1, half-rate (v_alignbit_b32, v_add3_u32) = 2 (profiles/r01_valu_issue_rates.txt).
"""
import itertools
import os
import sys

HALF = {"v_alignbit_b32", "v_add3_u32"}


class I:
    """One VALU instruction over virtual registers (strings); `lits` are emitted verbatim."""

    def __init__(self, op, dst, srcs, tail=""):
        self.op, self.dst, self.srcs, self.tail = op, dst, list(srcs), tail

    def slots(self):
        return 2 if self.op in HALF else 1


def rot(d, s, n):
    return I("v_alignbit_b32", d, [s, s], f", {n}")


def shr(d, s, n):
    return I("v_lshrrev_b32", d, [s], f"@{n}")     # special form: v_lshrrev_b32 d, n, s


def bit3(d, a, b, c, tt):
    return I("v_bitop3_b32", d, [a, b, c], f" bitop3:{tt:#x}")


def add(d, a, b):
    return I("v_add_u32", d, [a, b])


def add3(d, a, b, c):
    return I("v_add3_u32", d, [a, b, c])


def xor(d, a, b):
    return I("v_xor_b32", d, [a, b])


def sha_rounds(p, nrounds, with_schedule):
    """`nrounds` SHA-256 rounds of one hash stream with register-name prefix p, in the order hipcc emits for the shipped
    code (every consumer right behind its producers).  K+w is a register (`p`kw) unless with_schedule, in which case the
    message-schedule step of a round >= 16 is generated too (ring of 16 w registers)."""
    s = [f"{p}s{i}" for i in range(8)]      # a..h live in s[(i - t) & 7]
    w = [f"{p}w{i}" for i in range(16)]
    t0, t1, t2, t3, u0, u1, u2 = (f"{p}t{i}" for i in range(7))
    out = []
    for t in range(nrounds):
        a, b, c, d, e, f, g, h = (s[(i - t) & 7] for i in range(8))
        kw = f"{p}kw"
        if with_schedule:
            i = t & 15
            out += [rot(u0, w[(i + 1) & 15], 7), rot(u1, w[(i + 1) & 15], 18), shr(u2, w[(i + 1) & 15], 3), bit3(u0, u0, u1, u2, 0x96),
                    rot(u1, w[(i + 14) & 15], 17), rot(u2, w[(i + 14) & 15], 19), shr(t3, w[(i + 14) & 15], 10), bit3(u1, u1, u2, t3, 0x96),
                    add3(w[i], w[i], u0, w[(i + 9) & 15]), add(w[i], w[i], u1), add(t3, f"{p}k", w[i])]
            kw = t3
            # (t3 is consumed by the add below before ch overwrites it)
            out += [rot(t0, e, 25), rot(t1, e, 11), rot(t2, e, 6), bit3(t0, t2, t1, t0, 0x96), add(h, kw, h), bit3(t3, e, f, g, 0xca),
                    add3(h, h, t0, t3)]
        else:
            out += [rot(t0, e, 25), rot(t1, e, 11), rot(t2, e, 6), bit3(t0, t2, t1, t0, 0x96), bit3(t3, e, f, g, 0xca), add3(h, h, t0, t3),
                    add(h, kw, h)]
        out += [rot(t0, a, 22), rot(t1, a, 13), rot(t2, a, 2), bit3(t3, a, b, c, 0xe8), bit3(t0, t2, t1, t0, 0x96), add(d, h, d),
                add3(h, t0, h, t3)]
    return out


def ssa_temps(prog, keep):
    """Gives every definition of a register outside `keep` a fresh name, so that re-used temporaries do not chain
    instructions together through WAR/WAW dependences."""
    version, cur, out = {}, {}, []
    for ins in prog:
        srcs = [cur.get(r, r) for r in ins.srcs]
        d = ins.dst
        if d not in keep:
            version[d] = version.get(d, 0) + 1
            cur[d] = f"{d}.{version[d]}"
            d = cur[d]
        out.append(I(ins.op, d, srcs, ins.tail))
    return out


def interleave(*streams):
    out = []
    for group in itertools.zip_longest(*streams):
        out += [x for x in group if x is not None]
    return out


def list_schedule(prog, latency):
    """Greedy list scheduling of `prog` (program order defines the dependences: RAW, WAR, WAW over virtual registers):
    at every step issue the ready instruction with the longest dependent chain behind it whose producers are at least
    `latency` instructions back; if none qualifies, the one whose producers are oldest."""
    n = len(prog)
    preds = [set() for _ in range(n)]
    last_w, readers = {}, {}
    for k, ins in enumerate(prog):
        for r in ins.srcs:
            if r in last_w:
                preds[k].add(("raw", last_w[r]))
        if ins.dst in last_w:
            preds[k].add(("waw", last_w[ins.dst]))
        for rd in readers.get(ins.dst, []):
            if rd != k:
                preds[k].add(("war", rd))
        for r in ins.srcs:
            readers.setdefault(r, []).append(k)
        last_w[ins.dst] = k
        readers[ins.dst] = []
    succs = [[] for _ in range(n)]
    for k in range(n):
        for _, p in preds[k]:
            succs[p].append(k)
    height = [0] * n
    for k in reversed(range(n)):
        height[k] = prog[k].slots() + max([height[s] for s in succs[k]], default=0)
    done_at, order, remaining = {}, [], set(range(n))
    while remaining:
        pos = len(order)
        ready = [k for k in remaining if all(p in done_at for _, p in preds[k])]

        def slack(k):   # how far back the youngest RAW producer is
            raws = [done_at[p] for kind, p in preds[k] if kind == "raw"]
            return pos - max(raws) if raws else 10 ** 6
        ok = [k for k in ready if slack(k) > latency]
        pick = max(ok, key=lambda k: (height[k], -k)) if ok else max(ready, key=lambda k: (slack(k), height[k], -k))
        done_at[pick] = pos
        order.append(pick)
        remaining.discard(pick)
    return [prog[k] for k in order]


def allocate(prog, first_reg, live_out, bank_policy):
    """Virtual -> physical VGPRs.  Registers named in `live_out` (and anything read before it is written) are live over
    the whole loop body and get their own register; temporaries are recycled after their last read.
    bank_policy: 'natural' lowest free register; 'spread' prefers a bank not used by the instruction's other operands;
    'clash' prefers the bank of another operand."""
    writes_first = set()
    seen = set()
    for ins in prog:
        for r in ins.srcs:
            if r not in seen:
                writes_first.add(r)   # read before any write: loop-carried
        seen.add(ins.dst)
        seen.update(ins.srcs)
    carried = sorted(set(writes_first) | set(live_out))
    phys, nxt = {}, first_reg
    for r in carried:
        phys[r] = nxt
        nxt += 1
    last_use = {}
    for k, ins in enumerate(prog):
        for r in ins.srcs + [ins.dst]:
            last_use[r] = k
    free, out = [], []
    top = nxt
    cur = dict(phys)
    for k, ins in enumerate(prog):
        srcs_p = [cur[r] for r in ins.srcs]
        # sources read for the last time here free their register before the destination is chosen (dst may reuse it)
        for r in ins.srcs:
            if r not in carried and last_use[r] == k and r in cur and cur[r] not in free and r != ins.dst:
                free.append(cur[r])
        if ins.dst in carried:
            d = phys[ins.dst]
        elif ins.dst in cur and last_use[ins.dst] > k and ins.dst in ins.srcs:
            d = cur[ins.dst]          # read-modify-write of a temporary keeps its register
        else:
            if not free:
                free.append(top)
                top += 1
            banks = {p & 3 for p in srcs_p}
            cand = sorted(free)
            if bank_policy == "spread":
                pick = next((c for c in cand if (c & 3) not in banks), cand[0])
            elif bank_policy == "clash":
                pick = next((c for c in cand if (c & 3) in banks), cand[0])
            else:
                pick = cand[0]
            free.remove(pick)
            d = pick
        cur[ins.dst] = d
        out.append((ins, d, srcs_p))
        if ins.dst not in carried and last_use[ins.dst] == k:
            free.append(d)
    return out, carried, phys, top


def emit(alloc):
    lines = []
    for ins, d, srcs in alloc:
        if ins.tail.startswith("@"):
            lines.append(f"{ins.op} v{d}, {ins.tail[1:]}, v{srcs[0]}")
        else:
            lines.append(f"{ins.op} v{d}, " + ", ".join(f"v{s}" for s in srcs) + ins.tail)
    return lines


def same_bank_sources(alloc):
    """3-source instructions with two or more DISTINCT source registers in one bank."""
    n2 = n3 = 0
    for ins, d, srcs in alloc:
        u = sorted(set(srcs))
        if len(u) >= 2:
            banks = [s & 3 for s in u]
            m = max(banks.count(b) for b in set(banks))
            n2 += m == 2
            n3 += m >= 3
    return n2, n3


def min_raw_distance(alloc):
    last = {}
    dist = []
    for k, (ins, d, srcs) in enumerate(alloc):
        for s in srcs:
            if s in last:
                dist.append(k - last[s])
        last[d] = k
    return min(dist) if dist else 0, sum(1 for x in dist if x == 1), sum(1 for x in dist if x == 2)


PATTERNS = []   # dicts: name, phases (list of line lists), alt (line list for odd wavefronts or None), threads, barrier, instrs, slots, top, note


def add_raw(name, phases, instrs, slots, top, note, alt=None, threads=256, barrier=False):
    PATTERNS.append(dict(name=name, phases=phases, alt=alt, threads=threads, barrier=barrier, instrs=instrs, slots=slots, top=top, note=note))


def lower(prog, bank_policy="natural", live_out=()):
    alloc, carried, phys, top = allocate(prog, 16, live_out, bank_policy)
    return alloc, emit(alloc), top


def add_pattern(name, prog, note, bank_policy="natural", live_out=(), **kw):
    alloc, lines, top = lower(prog, bank_policy, live_out)
    n2, n3 = same_bank_sources(alloc)
    mind, d1, d2 = min_raw_distance(alloc)
    slots = sum(i.slots() for i in prog)
    add_raw(name, [lines], len(prog), slots, top, f"{note}; {len(prog)} instr, {slots} slots; RAW at distance 1: {d1}, 2: {d2}; "
            f"3-src instr with 2 sources in a bank: {n2}, 3: {n3}; VGPRs v16..v{top - 1}", **kw)


def simple(name, op_fn, n, note, **kw):
    add_pattern(name, [op_fn(k) for k in range(n)], note, **kw)


R = [f"r{i}" for i in range(48)]


def set1():
    """Round 3, first pass: producer/consumer distance, operand banks, the SHA round in several orders, H/F order."""
    # ---- single-opcode streams: producer/consumer distance -------------------------------------------------------
    for d in (1, 2, 4, 8):
        simple(f"xor_d{d}", lambda k, d=d: xor(R[k % d], R[k % d], R[40]), 64, f"v_xor_b32 (VOP2), {d} independent chain(s)")
        simple(f"align_d{d}", lambda k, d=d: rot(R[k % d], R[k % d], 7), 64, f"v_alignbit_b32, {d} independent chain(s)")
        simple(f"bit3_d{d}", lambda k, d=d: bit3(R[k % d], R[k % d], R[40], R[41], 0x96), 64, f"v_bitop3_b32, {d} independent chain(s)")
        simple(f"add_d{d}", lambda k, d=d: add(R[k % d], R[k % d], R[40]), 64, f"v_add_u32, {d} independent chain(s)")
        simple(f"add3_d{d}", lambda k, d=d: add3(R[k % d], R[k % d], R[40], R[41]), 64, f"v_add3_u32, {d} independent chain(s)")
    # ---- operand banks of the 3-source instructions (explicit physical registers) --------------------------------
    def explicit(name, op, dsts, srcs, tail, note):
        lines = [f"{op} v{dsts[k % len(dsts)]}, " + ", ".join(f"v{s}" for s in srcs) + tail for k in range(64)]
        slots = 64 * (2 if op in HALF else 1)
        add_raw(name, [lines], 64, slots, 64, note + f"; 64 instr, {slots} slots")
    for op, tail in (("v_bitop3_b32", " bitop3:0x96"), ("v_add3_u32", "")):
        short = "bit3" if "bitop" in op else "add3"
        explicit(f"{short}_banks_3diff", op, [16, 17, 18, 19, 20, 21, 22, 23], [45, 50, 55], tail, f"{op}: sources in three different banks")
        explicit(f"{short}_banks_2same", op, [16, 17, 18, 19, 20, 21, 22, 23], [44, 48, 53], tail, f"{op}: two sources in one bank")
        explicit(f"{short}_banks_3same", op, [16, 17, 18, 19, 20, 21, 22, 23], [44, 48, 52], tail, f"{op}: three sources in one bank")
    explicit("align_2regs_samebank", "v_alignbit_b32", [16, 17, 18, 19, 20, 21, 22, 23], [44, 48], ", 7", "v_alignbit_b32: two source registers in one bank")
    explicit("align_2regs_diffbank", "v_alignbit_b32", [16, 17, 18, 19, 20, 21, 22, 23], [44, 49], ", 7", "v_alignbit_b32: two source registers in two banks")
    # ---- the SHA-256 round as hipcc orders it, and re-scheduled --------------------------------------------------
    state = lambda p: [f"{p}s{i}" for i in range(8)] + [f"{p}kw"]   # noqa: E731
    base = ssa_temps(sha_rounds("x", 8, False), set(state("x")))
    add_pattern("rounds8_hipcc_order", base, "8 rounds, one hash stream, the order hipcc emits", live_out=state("x"))
    add_pattern("rounds8_hipcc_order_spread", base, "the same, destination banks chosen away from the sources'", "spread", live_out=state("x"))
    add_pattern("rounds8_hipcc_order_clash", base, "the same, destination banks chosen INTO the sources' banks", "clash", live_out=state("x"))
    for lat in (1, 2, 3, 4):
        add_pattern(f"rounds8_sched_lat{lat}", list_schedule(base, lat), f"8 rounds, one stream, list-scheduled for producer distance > {lat}", live_out=state("x"))
    basey = ssa_temps(sha_rounds("y", 8, False), set(state("y")))
    two = interleave(base, basey)
    add_pattern("rounds8x2_interleaved", two, "two independent hash streams, instruction by instruction", live_out=state("x") + state("y"))
    add_pattern("rounds8x2_sched_lat3", list_schedule(base + basey, 3),
                "two independent hash streams, list-scheduled for producer distance > 3", live_out=state("x") + state("y"))
    wst = lambda p: [f"{p}s{i}" for i in range(8)] + [f"{p}w{i}" for i in range(16)] + [f"{p}k"]   # noqa: E731
    full = ssa_temps(sha_rounds("x", 16, True), set(wst("x")))
    fully = ssa_temps(sha_rounds("y", 16, True), set(wst("y")))
    add_pattern("rounds16w_hipcc_order", full, "16 rounds WITH the message-schedule step, one stream, program order", live_out=wst("x"))
    for lat in (2, 4):
        add_pattern(f"rounds16w_sched_lat{lat}", list_schedule(full, lat), f"16 rounds with schedule, list-scheduled for producer distance > {lat}", live_out=wst("x"))
    add_pattern("rounds16w_x2_sched_lat4", list_schedule(full + fully, 4),
                "two streams of 16 rounds with schedule, list-scheduled for producer distance > 4", live_out=wst("x") + wst("y"))
    # ---- order of half-rate and full-rate instructions, all independent -------------------------------------------
    for order in ("HHHHFFFF", "HFHFHFHF", "HHFFHHFF", "HHHFFHFF", "FFFFFFHH", "HHHHHHFF"):
        add_pattern(f"mix_{order}", mix(order * 8), f"independent v_alignbit (H) and v_xor (F) in the order {order}")


def mix(order, f_ops=None):
    """Independent instructions (12 chains) in the given order: H = v_alignbit_b32, F = v_xor_b32 (or f_ops[k], cycling)."""
    f_ops = f_ops or [lambda r: xor(r, r, R[40])]
    prog, nf = [], 0
    for k, c in enumerate(order):
        r = R[k % 12]
        if c == "H":
            prog.append(rot(r, r, 7))
        elif c == "A":
            prog.append(add3(r, r, R[40], R[41]))
        else:
            prog.append(f_ops[nf % len(f_ops)](r))
            nf += 1
    return prog


def set2():
    """Second pass: WHEN do full-rate instructions get their 2-cycle issue?  (First pass: in every stream that mixes
    half-rate and full-rate instructions each instruction cost about 4 cycles, whatever its kind, order or distance.)"""
    fx = lambda r: xor(r, r, R[40])                      # noqa: E731
    fa = lambda r: add(r, r, R[40])                      # noqa: E731
    fl = lambda r: shr(r, r, 1)                          # noqa: E731
    fb = lambda r: bit3(r, r, R[40], R[41], 0x96)        # noqa: E731
    add_pattern("F_xor", mix("F" * 64), "pure v_xor_b32")
    add_pattern("H_align", mix("H" * 64), "pure v_alignbit_b32")
    add_pattern("allF_xor_add", mix("F" * 64, [fx, fa]), "full-rate only: v_xor / v_add_u32 alternating")
    add_pattern("allF_xor_lshr", mix("F" * 64, [fx, fl]), "full-rate only: v_xor / v_lshrrev alternating")
    add_pattern("allF_xor_bit3", mix("F" * 64, [fx, fb]), "full-rate only: v_xor (VOP2) / v_bitop3 (VOP3) alternating")
    add_pattern("allF_4kinds", mix("F" * 64, [fx, fa, fl, fb]), "full-rate only: xor, add, lshr, bitop3 cycling")
    add_pattern("allH_align_add3", mix("HA" * 32), "half-rate only: v_alignbit / v_add3 alternating")
    lines = [f"v_xor_b32_e64 v{16 + k % 12}, v{16 + k % 12}, v56" for k in range(64)]
    add_raw("F_xor_e64", [lines], 64, 64, 57, "pure v_xor_b32 in its 8-byte VOP3 encoding; 64 instr, 64 slots")
    lines = [f"v_add_u32 v{16 + k % 12}, 0x428a2f98, v{16 + k % 12}" for k in range(64)]
    add_raw("F_add_literal", [lines], 64, 64, 57, "pure v_add_u32 with a 32-bit literal (8 bytes); 64 instr, 64 slots")
    for n in (16, 64, 256):
        add_pattern(f"runs_F{n}_H{n}", mix("F" * n + "H" * n), f"{n} full-rate then {n} half-rate instructions, waves not synchronised")
    for n in (7, 15, 31, 63):
        add_pattern(f"rareH_F{n}_H1", mix(("F" * n + "H") * (256 // (n + 1))), f"one v_alignbit after every {n} v_xor")
    add_pattern("rareF_H7_F1", mix("HHHHHHHF" * 16), "one v_xor after every 7 v_alignbit")
    # scalar instructions inside a full-rate stream
    alloc, lines, top = lower(mix("F" * 64))
    with_nop = []
    for k, ln in enumerate(lines):
        with_nop.append(ln)
        if k % 8 == 7:
            with_nop.append("s_nop 0")
    add_raw("F_xor_snop_every8", [with_nop], 64, 64, max(top, 57), "pure v_xor with an s_nop 0 after every 8th; 64 VALU instr, 64 slots")
    with_smov = []
    for k, ln in enumerate(lines):
        with_smov.append(ln)
        if k % 4 == 3:
            with_smov.append(f"s_mov_b32 s{20 + (k // 4) % 8}, {0x428a2f98 + k:#x}")
    add_raw("F_xor_smov_every4", [with_smov], 64, 64, max(top, 57), "pure v_xor with an s_mov_b32 literal after every 4th; 64 VALU instr, 64 slots")
    # different wavefronts, different streams
    alloc, rot_lines, _ = lower(mix("F" * 64))
    add_raw("F_xor_oddwaves_shifted", [rot_lines], 64, 64, 57, "pure v_xor; odd wavefronts run the same stream shifted by 3 instructions (never the same instruction word at the same time)",
            alt=rot_lines[3:] + rot_lines[:3])
    _, add_lines, _ = lower(mix("F" * 64, [fa]))
    add_raw("F_even_xor_odd_add", [rot_lines], 64, 64, 57, "even wavefronts pure v_xor, odd wavefronts pure v_add_u32", alt=add_lines)
    _, h_lines, _ = lower(mix("H" * 32))
    add_raw("even_F64_odd_H32", [rot_lines], 48, 64, 57, "even wavefronts 64 v_xor, odd wavefronts 32 v_alignbit per iteration: every wavefront is pure, the SIMD sees a mix (instr/slots = averages)",
            alt=h_lines)
    # phases separated by workgroup barriers: 1024-lane workgroups = 4 wavefronts per SIMD
    _, f64, _ = lower(mix("F" * 64))
    _, h64, _ = lower(mix("H" * 64))
    _, f256, _ = lower(mix("F" * 256))
    _, h256, _ = lower(mix("H" * 256))
    add_raw("wg1024_F64_H64_nobarrier", [f64 + h64], 128, 192, 57, "1024-lane workgroups: 64 F then 64 H, no barrier", threads=1024)
    add_raw("wg1024_F64_bar_H64_bar", [f64, h64], 128, 192, 57, "1024-lane workgroups: 64 F | s_barrier | 64 H | s_barrier", threads=1024, barrier=True)
    add_raw("wg1024_F256_bar_H256_bar", [f256, h256], 512, 768, 57, "1024-lane workgroups: 256 F | s_barrier | 256 H | s_barrier", threads=1024, barrier=True)
    add_raw("wg1024_F64", [f64], 64, 64, 57, "1024-lane workgroups: pure F", threads=1024)
    add_raw("wg1024_H64", [h64], 64, 128, 57, "1024-lane workgroups: pure H", threads=1024)
    state = lambda p: [f"{p}s{i}" for i in range(8)] + [f"{p}kw"]   # noqa: E731
    base = ssa_temps(sha_rounds("x", 8, False), set(state("x")))
    add_pattern("rounds8_hipcc_order", base, "8 SHA rounds, one hash stream, the order hipcc emits (reference point)", live_out=state("x"))
    add_pattern("rounds8_wg1024", base, "the same in 1024-lane workgroups", live_out=state("x"), threads=1024)


def sha_rounds_2phase(p, nrounds, with_schedule=True):
    """SHA-256 rounds of one hash stream re-ordered into ONE run of half-rate instructions (all rotates of the round and
    of its message-schedule step) followed by ONE run of full-rate instructions (shifts, three-input logic, every
    addition as a two-operand add).  Returns a list of (h_run, f_run) per round."""
    s = [f"{p}s{i}" for i in range(8)]
    w = [f"{p}w{i}" for i in range(16)]
    T = lambda n: f"{p}{n}"   # noqa: E731
    out = []
    for t in range(nrounds):
        a, b, c, d, e, f, g, h = (s[(i - t) & 7] for i in range(8))
        i = t & 15
        hrun = [rot(T("e0"), e, 25), rot(T("e1"), e, 11), rot(T("e2"), e, 6), rot(T("a0"), a, 22), rot(T("a1"), a, 13), rot(T("a2"), a, 2)]
        frun = []
        if with_schedule:
            w1, w14, w9 = w[(i + 1) & 15], w[(i + 14) & 15], w[(i + 9) & 15]
            hrun += [rot(T("p0"), w1, 7), rot(T("p1"), w1, 18), rot(T("q0"), w14, 17), rot(T("q1"), w14, 19)]
            frun += [shr(T("p2"), w1, 3), shr(T("q2"), w14, 10), bit3(T("p0"), T("p0"), T("p1"), T("p2"), 0x96), bit3(T("q0"), T("q0"), T("q1"), T("q2"), 0x96),
                     add(T("p0"), T("p0"), T("q0")), add(w[i], w[i], w9), add(w[i], w[i], T("p0")), add(T("kw"), T("k"), w[i])]
        frun += [bit3(T("e0"), T("e2"), T("e1"), T("e0"), 0x96), bit3(T("c"), e, f, g, 0xca), bit3(T("a0"), T("a2"), T("a1"), T("a0"), 0x96), bit3(T("m"), a, b, c, 0xe8),
                 add(h, h, T("e0")), add(h, h, T("c")), add(h, h, T("kw")), add(d, d, h), add(T("a0"), T("a0"), T("m")), add(h, h, T("a0"))]
        out.append((hrun, frun))
    return out


def set3():
    """Third pass: is it the issue ARBITER?  (Second pass: full-rate instructions get their 2-cycle issue only while every
    wavefront on the SIMD is in a full-rate run -- barrier-separated runs reach 0.94-0.97 of the issue peak, the same runs
    unsynchronised 0.75.)  s_setprio around the half-rate or the full-rate runs; the SHA round in two-run form with and
    without barriers; per-wavefront duration spread (starvation shows as a wide spread)."""
    state = lambda q: [f"{q}s{i}" for i in range(8)] + [f"{q}kw"]   # noqa: E731
    wst = lambda q: [f"{q}s{i}" for i in range(8)] + [f"{q}w{i}" for i in range(16)] + [f"{q}k"]   # noqa: E731

    def with_prio(lines_tagged, hi_kind):
        """lines_tagged: [(line, 'H'|'F')]; s_setprio 1 in front of every run of hi_kind, s_setprio 0 behind it."""
        out, cur = [], None
        for ln, kind in lines_tagged:
            if kind != cur:
                out.append("s_setprio 1" if kind == hi_kind else "s_setprio 0")
                cur = kind
            out.append(ln)
        return out

    def tagged(prog, **kw):
        alloc, lines, top = lower(prog, **kw)
        return [(ln, "H" if ins.op in HALF else "F") for (ins, _, _), ln in zip(alloc, lines)], top

    for name, prog, kw in (("mix_HHHHFFFF", mix("HHHHFFFF" * 8), {}), ("runs_F64_H64", mix("F" * 64 + "H" * 64), {}),
                           ("rounds8", ssa_temps(sha_rounds("x", 8, False), set(state("x"))), dict(live_out=state("x"))),
                           ("rounds16w", ssa_temps(sha_rounds("x", 16, True), set(wst("x"))), dict(live_out=wst("x")))):
        tl, top = tagged(prog, **kw)
        instrs, slots = len(prog), sum(i.slots() for i in prog)
        add_raw(f"{name}_plain", [[ln for ln, _ in tl]], instrs, slots, top, f"{name}: no priority changes; {instrs} instr, {slots} slots")
        add_raw(f"{name}_prioH", [with_prio(tl, "H")], instrs, slots, top, f"{name}: s_setprio 1 around every run of half-rate instructions")
        add_raw(f"{name}_prioF", [with_prio(tl, "F")], instrs, slots, top, f"{name}: s_setprio 1 around every run of full-rate instructions")
    # the SHA round in two-run form (k hash streams per lane), with and without barriers, 512- and 1024-lane workgroups
    for k in (1, 2):
        streams = [sha_rounds_2phase("xyzv"[j], 16, True) for j in range(k)]
        live = sum((wst("xyzv"[j]) for j in range(k)), [])
        # one flat program in run order (register allocation over the whole loop body), cut back into runs afterwards
        flat, cuts = [], []
        for t in range(16):
            for which in (0, 1):
                run = []
                for j in range(k):
                    run += streams[j][t][which]
                flat += run
                cuts.append(len(flat))
        flat = ssa_temps(flat, set(live))
        alloc, lines, top = lower(flat, live_out=live)
        runs, lo = [], 0
        for c in cuts:
            runs.append(lines[lo:c])
            lo = c
        instrs, slots = len(flat), sum(i.slots() for i in flat)
        note = f"16 SHA rounds with schedule, {k} hash stream(s) per lane, per round one run of rotates then one run of full-rate instructions; {instrs} instr, {slots} slots"
        for threads in (512, 1024):
            add_raw(f"sha2run_k{k}_t{threads}_nobar", [sum(runs, [])], instrs, slots, top, note + "; no barriers", threads=threads)
            add_raw(f"sha2run_k{k}_t{threads}_bar", runs, instrs, slots, top, note + "; s_barrier between runs", threads=threads, barrier=True)
        tl = [(ln, "H" if ins.op in HALF else "F") for (ins, _, _), ln in zip(alloc, lines)]
        add_raw(f"sha2run_k{k}_prioH", [with_prio(tl, "H")], instrs, slots, top, note + "; s_setprio 1 around the rotate runs")
    full = ssa_temps(sha_rounds("x", 16, True), set(wst("x")))
    for threads in (512, 1024):
        add_pattern(f"rounds16w_t{threads}", full, f"16 rounds with schedule in hipcc's order, {threads}-lane workgroups (reference point)", live_out=wst("x"), threads=threads)


def set4():
    """Fourth pass: does priority help a LONE wavefront (the latency-bound top of the reduction runs one per SIMD)?  The same
    16 rounds with the priority toggled around the complex runs, raised once and for all, or left alone."""
    wst = lambda q: [f"{q}s{i}" for i in range(8)] + [f"{q}w{i}" for i in range(16)] + [f"{q}k"]   # noqa: E731
    prog = ssa_temps(sha_rounds("x", 16, True), set(wst("x")))
    alloc, lines, top = lower(prog, live_out=wst("x"))
    tl = [(ln, "H" if ins.op in HALF else "F") for (ins, _, _), ln in zip(alloc, lines)]
    instrs, slots = len(prog), sum(i.slots() for i in prog)

    def toggled(hi_kind, level=1):
        out, cur = [], None
        for ln, kind in tl:
            if kind != cur:
                out.append(f"s_setprio {level}" if kind == hi_kind else "s_setprio 0")
                cur = kind
            out.append(ln)
        return out + ["s_setprio 0"]
    plain = [ln for ln, _ in tl]
    add_raw("r16_plain", [plain], instrs, slots, top, "16 rounds with schedule, hipcc order, priority untouched")
    add_raw("r16_prioH", [toggled("H")], instrs, slots, top, "s_setprio 1 around every complex run")
    add_raw("r16_prioH3", [toggled("H", 3)], instrs, slots, top, "s_setprio 3 around every complex run")
    add_raw("r16_const1", [["s_setprio 1"] + plain], instrs, slots, top, "s_setprio 1 once per loop body, never lowered")
    add_raw("r16_const3", [["s_setprio 3"] + plain], instrs, slots, top, "s_setprio 3 once per loop body, never lowered")
    add_raw("r16_nops", [[x if not x.startswith("s_setprio") else "s_nop 0" for x in toggled("H")]], instrs, slots, top, "an s_nop 0 wherever prioH has an s_setprio (same scalar instruction count, no priority change)")
    add_pattern("H_align", mix("H" * 64), "pure v_alignbit_b32")
    add_pattern("F_xor", mix("F" * 64), "pure v_xor_b32")


def set5():
    """Fifth pass: WHICH complex instructions need the raised priority?  The 16 rounds with s_setprio around the rotates only,
    around the three-operand adds only, with two levels, and with the simple run lowered below a base priority instead."""
    wst = lambda q: [f"{q}s{i}" for i in range(8)] + [f"{q}w{i}" for i in range(16)] + [f"{q}k"]   # noqa: E731
    prog = ssa_temps(sha_rounds("x", 16, True), set(wst("x")))
    alloc, lines, top = lower(prog, live_out=wst("x"))
    kinds = [("R" if ins.op == "v_alignbit_b32" else ("A" if ins.op == "v_add3_u32" else "F")) for (ins, _, _) in alloc]
    instrs, slots = len(prog), sum(i.slots() for i in prog)

    def toggled(level_of):
        """level_of: kind -> priority while such instructions run."""
        out, cur = [], 0
        for ln, k in zip(lines, kinds):
            want = level_of[k]
            if want != cur:
                out.append(f"s_setprio {want}")
                cur = want
            out.append(ln)
        return out + ["s_setprio 0"]
    add_raw("p_plain", [lines], instrs, slots, top, "16 rounds, priority untouched")
    add_raw("p_all_complex", [toggled({"R": 1, "A": 1, "F": 0})], instrs, slots, top, "priority 1 for rotates and add3 (what the pass does)")
    add_raw("p_rot_only", [toggled({"R": 1, "A": 0, "F": 0})], instrs, slots, top, "priority 1 for rotates only")
    add_raw("p_add3_only", [toggled({"R": 0, "A": 1, "F": 0})], instrs, slots, top, "priority 1 for add3 only")
    add_raw("p_two_levels", [toggled({"R": 2, "A": 1, "F": 0})], instrs, slots, top, "priority 2 for rotates, 1 for add3")
    add_raw("p_two_levels_b", [toggled({"R": 1, "A": 2, "F": 0})], instrs, slots, top, "priority 1 for rotates, 2 for add3")
    add_raw("p_base1", [toggled({"R": 3, "A": 3, "F": 1})], instrs, slots, top, "priority 3 for complex, 1 for simple")
    # the same with every 3rd add3 split (the balance of the two slots)
    import copy
    sp, n = [], 0
    for ins in prog:
        if ins.op == "v_add3_u32":
            n += 1
            if n % 3 == 0 and ins.dst not in ins.srcs[1:]:
                sp.append(I("v_add_u32", ins.dst, [ins.srcs[0], ins.srcs[1]]))
                sp.append(I("v_add_u32", ins.dst, [ins.dst, ins.srcs[2]]))
                continue
        sp.append(copy.copy(ins))
    alloc2, lines2, top2 = lower(sp, live_out=wst("x"))
    k2 = [("C" if ins.op in HALF else "F") for (ins, _, _) in alloc2]
    out, cur = [], 0
    for ln, k in zip(lines2, k2):
        want = 1 if k == "C" else 0
        if want != cur:
            out.append(f"s_setprio {want}")
            cur = want
        out.append(ln)
    add_raw("p_split3", [out + ["s_setprio 0"]], len(sp), slots, top2, f"every 3rd add3 split ({len(sp)} instr), priority 1 for complex")


def write(out):
    with open(out, "w") as f:
        f.write("// generated by tools/gen_issue_patterns.py -- do not edit\n")
        for p in PATTERNS:
            name, top = p["name"], max(p["top"], 57)
            if top > 120:
                sys.exit(f"{name}: needs v{top}")
            clob = ", ".join(f'"v{r}"' for r in range(16, top)) + ', "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27"'
            init = " ".join(f'"v_add_u32 v{r}, {(r * 2654435761) & 0xffffffff:#x}, %0\\n\\t"' for r in range(16, top))

            def block(lines):
                return "asm volatile(" + " ".join(f'"{ln}\\n\\t"' for ln in lines) + f" : : : {clob});"
            body = ""
            if p["alt"] is not None:
                body = f"if (odd) {{ {block(p['alt'])} }} else {{ {block(p['phases'][0])} }}"
            else:
                for ph in p["phases"]:
                    body += block(ph) + (" __syncthreads(); " if p["barrier"] else " ")
            waves = p["threads"] // 64
            f.write(f"""__global__ __launch_bounds__({p['threads']}) void k_{name}(uint32_t* out, unsigned long long* stamps, int iters)
{{
    uint32_t seed = threadIdx.x * 2654435761u + blockIdx.x * 977u, res;
    const bool odd = (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) & 1) != 0; (void)odd;
    asm volatile({init} : : "v"(seed) : {clob});
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {{ {body} }}
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    asm volatile("v_xor_b32 %0, v16, v17" : "=v"(res) : : {clob});
    out[blockIdx.x * blockDim.x + threadIdx.x] = res;
    if ((threadIdx.x & 63u) == 0u) {{
        unsigned long long* o = stamps + ((size_t)blockIdx.x * {waves} + (threadIdx.x >> 6)) * 4;
        o[0] = t0; o[1] = r0; o[2] = t1; o[3] = r1;
    }}
}}
""")
        f.write("struct Pattern { const char* name; const char* note; void (*kern)(uint32_t*, unsigned long long*, int); int instrs, slots, threads; };\n")
        f.write("static const Pattern PATTERN_TABLE[] = {\n")
        for p in PATTERNS:
            f.write(f'    {{"{p["name"]}", "{p["note"]}", k_{p["name"]}, {p["instrs"]}, {p["slots"]}, {p["threads"]}}},\n')
        f.write("};\n")


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "set2"
    {"set1": set1, "set2": set2, "set3": set3, "set4": set4, "set5": set5}[which]()
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "issue_patterns.inc")
    write(out)
    print(f"{which}: {len(PATTERNS)} patterns -> {out}")
    for p in PATTERNS:
        print(f"  {p['name']}: {p['note']}")


if __name__ == "__main__":
    main()
