"""Issue-priority pass over the gfx950 assembly hipcc emits for the kernels (part of the build: build.py).

Why.  A gfx950 SIMD issues up to two VALU instructions per 4-cycle turn, from two different wavefronts: one may be of
any kind, the other only a "simple" one (v_add_u32, v_xor/and/or_b32, v_lshrrev_b32, v_bitop3_b32 on VGPRs, v_mov ...).
The "complex" kinds (v_alignbit_b32, v_add3_u32, v_perm_b32, compares, 64-bit shifts ...) show up as half rate in
single-opcode streams because only one of them fits a turn.  SHA-256 is 58 % rotates and three-operand adds, so it
should run at about max(N/2, N_complex) turns -- but with every wavefront at the same priority the arbiter pairs
nothing in a stream that mixes the two kinds: each instruction, simple or complex, costs a whole turn (measured 4.0
cycles per instruction on the shipped node hash at 2.38 GHz, 2 to 8 wavefronts per SIMD, whatever the order, the
producer distance or the operand banks: profiles/r03_issue_patterns_*.txt).  Raising the wavefront's priority for the
duration of every run of complex instructions (s_setprio 1 ... s_setprio 0) makes the arbiter take the complex
instruction first and fill the turn's second slot with a simple one from a wavefront at priority 0: the same
instruction stream then issues at 2.3-2.4 cycles per instruction (profiles/r03_issue_patterns_set3.txt).

What.  For every basic block of every kernel: s_setprio 1 in front of each maximal run of complex VALU instructions,
s_setprio 0 behind it.  Runs separated by at most `gap` simple instructions are merged (fewer toggles).  Code that
manages its own priority (a kernel prologue inside s_setprio 3 ... s_setprio 0) is left alone while its priority is
raised.  Nothing else is touched: same instructions, same registers, same order -- results are bit-identical.

    python3 -m vk_merkle_roots_amd.isa_prio_pass in.s out.s [--gap N]
"""
import re
import sys

# VALU opcodes that fit the second ("simple") issue slot: full rate in single-opcode streams
# (profiles/r01_valu_issue_rates.txt, profiles/r03_issue_patterns_set2.txt).
SIMPLE = {
    "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_xor_b32", "v_and_b32", "v_or_b32", "v_not_b32", "v_lshrrev_b32", "v_ashrrev_i32",
    "v_mov_b32", "v_bitop3_b32", "v_cndmask_b32", "v_accvgpr_write_b32", "v_accvgpr_read_b32", "v_nop",
}
# Complex VALU opcodes whose one-per-turn issue was MEASURED (profiles/r03_issue_patterns_set1-5.txt, r01_valu_issue_rates.txt):
# exact names, and prefixes ending in "*".  classify() treats every opcode outside SIMPLE as complex -- the safe price -- but a
# toolchain that starts to emit an opcode nobody measured must not move bench.py's floor silently: audit() below lists those.
COMPLEX_MEASURED = {
    "v_alignbit_b32", "v_add3_u32", "v_perm_b32", "v_lshlrev_b32", "v_cndmask_b32", "v_bitop3_b32", "v_cmp_*", "v_cmpx_*",
    "v_lshlrev_b64", "v_lshrrev_b64", "v_lshl_add_u64", "v_mov_b64", "v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32",
}
# Complex by assumption (address arithmetic and lane traffic at the edges of a hash block; never in the round function): priced
# as complex, tolerated up to this many PER HASH BLOCK -- more means the compiler has begun to use one inside the rounds.
COMPLEX_ASSUMED = {"v_add_lshl_u32", "v_lshl_add_u32", "v_lshl_or_b32", "v_sub_co_u32", "v_subrev_co_u32", "v_subb_co_u32", "v_subbrev_co_u32", "v_add_co_u32",
                   "v_addc_co_u32", "v_writelane_b32", "v_readlane_b32", "v_readfirstlane_b32", "v_min_u32", "v_max_u32", "v_and_or_b32", "v_or3_b32"}
ASSUMED_PER_BLOCK = 16
# Hash blocks every kernel must show (hash_blocks()): the node hash once per code path that hashes pairs, a 64-byte block and
# the digest hash per map_kernel instantiation.  A listing with other counts is not the code the static counts were taken from.
EXPECTED_HASH_BLOCKS = {"ELi64ELi5ELb1": 3,   # map_kernel MODE 5 (two blocks per trip: two block bodies + the digest); first match wins
                        "reduce_pass_kernel": 1, "reduce_level_kernel": 1, "reduce_collapse_kernel": 2, "reduce_tail_kernel": 3, "map_kernel": 2,
                        "reduce_pass_proofs_kernel": 1, "reduce_collapse_proofs_kernel": 2, "reduce_tail_proofs_kernel": 3,
                        "map_persist_kernel": 4, "map_hash_sorted_kernel": 2}   # the last two: experiments build (staged + per-lane loop; block + digest)

_INSTR = re.compile(r"^\s+([a-z_0-9]+)\s*(.*)$")
_LABEL = re.compile(r"^[.\w$]+:")


def classify(line):
    """'C' complex VALU, 'S' simple VALU, 'P' an explicit s_setprio, 'B' ends a basic block, 'O' anything else that is an
    instruction, None for labels / directives / comments (labels end the block too, see callers)."""
    if _LABEL.match(line):
        return "L"
    m = _INSTR.match(line)
    if not m:
        return None
    op, rest = m.group(1), m.group(2)
    if op.startswith(";") or op.startswith("."):
        return None
    if op == "s_setprio":
        return "P"
    if op.startswith("s_cbranch") or op in ("s_branch", "s_endpgm", "s_setpc_b64", "s_swappc_b64", "s_barrier"):
        return "B"
    if not op.startswith("v_"):
        return "O"
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if op.endswith("_dpp") or op.endswith("_sdwa") or " row_" in rest or "quad_perm" in rest:
        return "C"
    if base in SIMPLE:
        if base == "v_cndmask_b32" and op.endswith("_e64"):
            return "C"
        if base == "v_bitop3_b32" and re.search(r"(^|[\s,])(s\d+|s\[|vcc|exec|0x[0-9a-f]+|-?\d+)\b", rest.split("bitop3")[0]):
            return "C"     # an SGPR / constant operand makes it half rate
        return "S"
    return "C"


_ADD3 = re.compile(r"^(\s+)v_add3_u32\s+(v\d+),\s*([^,]+),\s*([^,]+),\s*([^,\s]+)\s*$")


def _kernel_of(line):
    """The (mangled) name when `line` is a function label, else None."""
    if _LABEL.match(line) and not line.startswith(".L"):
        return line.split(":")[0]
    return None


def split_add3(lines, every, skip=()):
    """Balance the two issue slots: v_add3_u32 is a complex instruction, two v_add_u32 are simple ones.  The node hash has
    2 108 complex and 1 500 simple instructions, i.e. the complex slot is the longer queue; every add3 that is split moves
    one instruction from it to the other.  `every` = k splits every k-th v_add3_u32 of each kernel (0 = none).
    d = a + b + c  ->  d = p + q ; d = r + d   with r != d, d allowed among p, q, and a VGPR in the VOP2 src1 position."""
    if not every:
        return lines, 0
    out, n, done = [], 0, 0
    isv = lambda x: re.fullmatch(r"v\d+", x) is not None   # noqa: E731
    skipping = False
    for ln in lines:
        k = _kernel_of(ln)
        if k is not None:
            n = 0
            skipping = any(pat in k for pat in skip)
        if skipping:
            out.append(ln)
            continue
        m = _ADD3.match(ln.split(";")[0].rstrip() if ";" in ln else ln.rstrip("\n"))
        if not m:
            out.append(ln)
            continue
        n += 1
        if n % every:
            out.append(ln)
            continue
        ind, d, a, b, c = m.group(1), m.group(2), m.group(3).strip(), m.group(4).strip(), m.group(5).strip()
        srcs = [a, b, c]
        pick = None
        for ri in (2, 1, 0):
            r = srcs[ri]
            p, q = [srcs[k] for k in range(3) if k != ri]
            if r == d or (srcs.count(d) > 1):
                continue
            if not (isv(p) or isv(q)):
                continue
            nonv = [x for x in (p, q, ) if not isv(x)]
            if len(nonv) > 1:
                continue
            x, y = (p, q) if isv(q) else (q, p)       # y: the VGPR that goes into src1
            pick = (x, y, r)
            break
        if pick is None:
            out.append(ln)
            continue
        x, y, r = pick
        out.append(f"{ind}v_add_u32_e32 {d}, {x}, {y}\n")
        out.append(f"{ind}v_add_u32_e32 {d}, {r}, {d}\n")
        done += 1
    return out, done


def transform(lines, gap=0, level=1, split_every=0, skip=(), rotate_level=None):
    """skip: kernels (substrings of their names) left exactly as hipcc emitted them -- the latency-bound ones, launched
    with one wavefront per SIMD: a lone wavefront has nobody to pair with, and every s_setprio (and the second
    instruction of a split add3) is one more issue turn on its critical path."""
    lines, nsplit = split_add3(lines, split_every, skip)
    out, stats = [], {"runs": 0, "complex": 0, "simple_inside": 0, "kernels": 0, "add3_split": nsplit}
    in_text = False
    skipping = False
    base_prio = 0
    i, n = 0, len(lines)
    while i < n:
        ln = lines[i]
        st = ln.lstrip()
        if st.startswith(".text") or (st.startswith(".section") and ".text" in st):   # template kernels sit in .section .text.<name> (comdat)
            in_text = True
        elif re.match(r"\.(section|rodata|data)\b", st):
            in_text = False
        kind = classify(ln) if in_text else None
        if kind == "L" and not ln.startswith(".L"):   # a function symbol: every kernel starts at priority 0
            base_prio = 0
            stats["kernels"] += 1
            skipping = any(pat in ln for pat in skip)
        if in_text and skipping:
            out.append(ln)
            i += 1
            continue
        if kind == "P":
            m = re.search(r"s_setprio\s+(\d+)", ln)
            base_prio = int(m.group(1)) if m else 0
        if kind != "C" or base_prio != 0:
            out.append(ln)
            i += 1
            continue
        # a run of complex instructions starts here: extend it over complex instructions, comments/directives, scalar and
        # memory instructions and up to `gap` simple VALU instructions at a time; never over labels, branches, barriers
        # or priority changes
        j, last_c, simple_seen = i, i, 0
        k = i + 1
        while k < n:
            c = classify(lines[k])
            if c == "C":
                last_c, simple_seen = k, 0
            elif c == "S":
                simple_seen += 1
                if simple_seen > gap:
                    break
            elif c is None or c == "O":
                pass           # comments, directives, scalar / memory instructions ride along
            else:
                break
            k += 1
        run = lines[j:last_c + 1]
        stats["runs"] += 1
        stats["complex"] += sum(1 for x in run if classify(x) == "C")
        stats["simple_inside"] += sum(1 for x in run if classify(x) == "S")
        if rotate_level is None or rotate_level == level:
            out.append(f"\ts_setprio {level}\n")
            out.extend(run)
        else:   # two levels: rotates above the other complex instructions (profiles/r03_issue_patterns_set5.txt)
            cur = None
            for x in run:
                if classify(x) == "C":
                    want = rotate_level if "v_alignbit_b32" in x else level
                    if want != cur:
                        out.append(f"\ts_setprio {want}\n")
                        cur = want
                out.append(x)
        out.append("\ts_setprio 0\n")
        i = last_c + 1
    return out, stats


def hash_blocks(lines):
    """Static VALU counts of the SHA-256 basic blocks of every kernel in a (transformed) assembly listing: a basic block
    with more than 100 rotates is one unrolled hash -- the node hash of the reduce kernels (three compressions), the
    per-64-byte-block compression and the digest hash of the map kernels.  {kernel: [{valu, complex, simple, rotates}]}
    in program order.  bench.py prices the kernels with these (issue turns = max(valu / 2, complex))."""
    out, cur, bb = {}, None, None

    def close():
        if cur is not None and bb and bb["rotates"] > 100:
            out[cur].append(bb)
    for ln in lines:
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            close()
            cur, bb = m.group(1), {"valu": 0, "complex": 0, "simple": 0, "rotates": 0}
            out[cur] = []
            continue
        if cur is None:
            continue
        if re.match(r"^\.LBB\w+:", ln):
            close()
            bb = {"valu": 0, "complex": 0, "simple": 0, "rotates": 0}
            continue
        k = classify(ln)
        if k in ("C", "S"):
            bb["valu"] += 1
            bb["complex" if k == "C" else "simple"] += 1
            if "v_alignbit_b32" in ln:
                bb["rotates"] += 1
        if ln.strip().startswith("s_endpgm"):      # ends a block, not the kernel: early exits come before the hashes in some layouts
            close()
            bb = {"valu": 0, "complex": 0, "simple": 0, "rotates": 0}
        if ln.startswith(".Lfunc_end"):
            close()
            cur, bb = None, None
    return {k: v for k, v in out.items() if v}


def _base(op):
    return re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)


def _known(base):
    if base in SIMPLE or base in COMPLEX_MEASURED:
        return "measured"
    if any(p.endswith("*") and base.startswith(p[:-1]) for p in COMPLEX_MEASURED):
        return "measured"
    if base in COMPLEX_ASSUMED:
        return "assumed"
    return None


def audit(lines):
    """The guard between the issue model and the compiler.  Over the SHA-256 basic blocks of a (transformed)
    listing -- the blocks bench.py's floor is computed from -- returns
        {"hash_valu": {mnemonic: count}, "unclassified": [...], "blocks": {kernel: n}, "block_count_errors": [...]}
    `unclassified` names every VALU mnemonic in a hash block that is neither in SIMPLE nor in COMPLEX_MEASURED, and every
    COMPLEX_ASSUMED one used more than ASSUMED_PER_BLOCK times in one block; `block_count_errors` every kernel whose number of
    hash blocks is not EXPECTED_HASH_BLOCKS'.  Both must be empty for the static counts to mean what DESIGN.md 3.1 says."""
    hist, unclassified, blocks = {}, [], {}
    unraised = []          # hash blocks in which the pass raised nothing (a kernel it was told to skip is expected here; any other is a
                           # block it took for part of a raised-priority region: its base priority is tracked in textual, not control-flow, order)
    cur, ops, rot, raised = None, {}, 0, 0

    def close():
        nonlocal ops, rot, raised
        if cur is not None and rot > 100:
            blocks[cur] = blocks.get(cur, 0) + 1
            if raised == 0:
                unraised.append(cur)
            for b, n in ops.items():
                hist[b] = hist.get(b, 0) + n
                k = _known(b)
                if k is None:
                    unclassified.append(f"{cur}: {b} x{n} (not measured)")
                elif k == "assumed" and n > ASSUMED_PER_BLOCK:
                    unclassified.append(f"{cur}: {b} x{n} in one hash block (assumed complex, tolerated up to {ASSUMED_PER_BLOCK})")
        ops, rot, raised = {}, 0, 0
    for ln in lines:
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            close()
            cur = m.group(1)
            continue
        if cur is not None and re.match(r"^\s+s_setprio\s+[12]\b", ln):
            raised += 1
        if cur is None:
            continue
        if re.match(r"^\.LBB\w+:", ln):
            close()
            continue
        if classify(ln) in ("C", "S"):
            b = _base(_INSTR.match(ln).group(1))
            ops[b] = ops.get(b, 0) + 1
            if b == "v_alignbit_b32":
                rot += 1
        if ln.strip().startswith("s_endpgm"):      # ends a block, not the kernel (see hash_blocks)
            close()
        if ln.startswith(".Lfunc_end"):
            close()
            cur = None
    close()
    errors = []
    for kernel, n in sorted(blocks.items()):
        want = next((v for k, v in EXPECTED_HASH_BLOCKS.items() if k in kernel), None)
        if want is None:
            errors.append(f"{kernel}: {n} hash block(s) in a kernel nobody listed in EXPECTED_HASH_BLOCKS")
        elif n != want:
            errors.append(f"{kernel}: {n} hash block(s), expected {want}")
    return {"hash_valu": dict(sorted(hist.items())), "unclassified": sorted(set(unclassified)), "blocks": blocks, "block_count_errors": errors,
            "hash_blocks_without_raised_runs": sorted(set(unraised))}


def verify(original, transformed):
    """Build-time self-check: the transformed listing is the original plus inserted `s_setprio` lines, with some
    `v_add3_u32 d, a, b, c` replaced by two `v_add_u32` that add the same three operands into d -- and nothing else.
    Returns a list of differences (empty = verified)."""
    diffs = []
    i = j = 0
    n, m = len(original), len(transformed)
    prio = re.compile(r"^\ts_setprio \d+\n?$")
    add = re.compile(r"^\s+v_add_u32_e32\s+(v\d+),\s*([^,]+),\s*([^,\s]+)\s*$")
    while i < n or j < m:
        if i < n and j < m and original[i] == transformed[j]:
            i += 1; j += 1
            continue
        if j < m and prio.match(transformed[j]):
            j += 1
            continue
        a3 = _ADD3.match((original[i].split(";")[0].rstrip() if ";" in original[i] else original[i].rstrip("\n"))) if i < n else None
        if a3 and j + 1 < m:
            x, y = add.match(transformed[j].rstrip("\n")), add.match(transformed[j + 1].rstrip("\n"))
            if x and y:
                d = a3.group(2)
                srcs = sorted(t.strip() for t in (a3.group(3), a3.group(4), a3.group(5)))
                first = [x.group(2).strip(), x.group(3).strip()]
                second = [y.group(2).strip(), y.group(3).strip()]
                if x.group(1) == d and y.group(1) == d and d in second:
                    second.remove(d)
                    if sorted(first + second) == srcs and second[0] != d:
                        i += 1; j += 2
                        continue
        diffs.append(f"line {i + 1} of the input / {j + 1} of the output: {(original[i] if i < n else '<end>').strip()!r} vs {(transformed[j] if j < m else '<end>').strip()!r}")
        if len(diffs) >= 8:
            break
        i += 1; j += 1
    return diffs


def main(argv):
    gap = 0
    args = [a for a in argv if not a.startswith("--")]
    for k, a in enumerate(argv):
        if a == "--gap":
            gap = int(argv[k + 1])
            args.remove(argv[k + 1])
    src, dst = args[0], args[1]
    with open(src) as f:
        lines = f.readlines()
    out, stats = transform(lines, gap)
    with open(dst, "w") as f:
        f.writelines(out)
    print(f"isa_prio_pass: {stats['runs']} runs, {stats['complex']} complex VALU instructions raised, {stats['simple_inside']} simple ones inside merged runs (gap {gap})")


if __name__ == "__main__":
    main(sys.argv[1:])
