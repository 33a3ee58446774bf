"""The issue-priority pass (vk_merkle_roots_amd/isa_prio_pass.py) changes WHEN instructions issue, never what is computed:
it may only insert s_setprio and replace a v_add3_u32 by two v_add_u32 with the same three addends.  Checked on a small
listing and on the assembly of the library in the tree (build/obj/.../device.s, when this checkout has built it).  No GPU."""
import os
import re

import pytest

from conftest import ROOT

LISTING = """\t.text
\t.amdgcn_target "amdgcn-amd-amdhsa--gfx950"
_Z6kernelPj:                            ; @_Z6kernelPj
; %bb.0:
\ts_load_dwordx2 s[0:1], s[0:1], 0x0
\tv_alignbit_b32 v1, v0, v0, 7
\tv_alignbit_b32 v2, v0, v0, 18
\tv_lshrrev_b32_e32 v3, 3, v0
\tv_bitop3_b32 v1, v1, v2, v3 bitop3:0x96
\tv_add3_u32 v4, v1, v0, s2
\tv_add3_u32 v4, v4, v1, v2
\tv_add_u32_e32 v5, v4, v1
\ts_cbranch_scc0 .LBB0_2
.LBB0_1:
\ts_setprio 3
\tv_alignbit_b32 v1, v0, v0, 7
\ts_setprio 0
\tv_perm_b32 v6, v1, v2, s3
\tv_bitop3_b32 v7, v1, s4, v3 bitop3:0xca
\tv_xor_b32_e32 v8, v7, v6
.LBB0_2:
\ts_endpgm
\t.section\t.rodata,"a",@progbits
\t.amdhsa_kernel _Z6kernelPj
\t.end_amdhsa_kernel
\t.section\t.text._Z5otherPj,"axG",@progbits,_Z5otherPj,comdat
_Z5otherPj:
\tv_add3_u32 v1, v1, v2, v3
\tv_add3_u32 v1, v2, v3, v1
\tv_add3_u32 v1, v1, v1, v3
\tv_add3_u32 v1, s1, s2, v3
\ts_endpgm
"""


def _lines():
    return [ln + "\n" for ln in LISTING.split("\n")]


def test_runs_of_complex_instructions_are_wrapped_and_nothing_else_changes():
    from vk_merkle_roots_amd import isa_prio_pass as P
    out, stats = P.transform(_lines(), gap=0)
    text = "".join(out)
    # block 0: [alignbit alignbit] lshr bitop3 [add3 add3] add -- two runs; the scalar load rides outside
    assert "\ts_setprio 1\n\tv_alignbit_b32 v1, v0, v0, 7\n\tv_alignbit_b32 v2, v0, v0, 18\n\ts_setprio 0\n\tv_lshrrev_b32_e32" in text
    assert "\ts_setprio 1\n\tv_add3_u32 v4, v1, v0, s2\n\tv_add3_u32 v4, v4, v1, v2\n\ts_setprio 0\n\tv_add_u32_e32 v5" in text
    # code that manages its own priority is left alone while it is raised; after its s_setprio 0 the pass resumes, and a
    # v_bitop3_b32 with an SGPR operand counts as complex
    assert "\ts_setprio 3\n\tv_alignbit_b32 v1, v0, v0, 7\n\ts_setprio 0\n\ts_setprio 1\n\tv_perm_b32 v6, v1, v2, s3\n\tv_bitop3_b32 v7, v1, s4, v3 bitop3:0xca\n\ts_setprio 0\n\tv_xor_b32_e32" in text
    # kernels in their own .text.<name> section (templates) are covered too
    assert text.count("s_setprio 1") == 4 and stats["kernels"] == 2
    # removing what was inserted gives the input back, line for line
    inserted = {"\ts_setprio 1\n"}
    back, skip_next_zero = [], False
    src = _lines()
    i = 0
    for ln in out:
        if i < len(src) and ln == src[i]:
            back.append(ln)
            i += 1
        else:
            assert ln in ("\ts_setprio 1\n", "\ts_setprio 0\n"), ln
    assert back == src
    # merged runs (gap 2) wrap the two simple instructions between the complex runs of block 0
    out1, _ = P.transform(_lines(), gap=2)
    assert "".join(out1).count("s_setprio 1") < text.count("s_setprio 1")
    # a skipped kernel is untouched
    out2, _ = P.transform(_lines(), gap=0, skip=("_Z5other",))
    assert "".join(out2).split("_Z5otherPj:\n")[1] == LISTING.split("_Z5otherPj:\n")[1] + "\n"


def test_add3_split_keeps_the_three_addends_and_never_clobbers_one_it_still_needs():
    from vk_merkle_roots_amd import isa_prio_pass as P
    out, n = P.split_add3(_lines(), 1)
    text = "".join(out)
    # d = a + b + c with d distinct: d = a + b; d = c + d (SGPR goes to src0: VOP2 takes a VGPR in src1)
    assert "\tv_add_u32_e32 v4, v1, v0\n\tv_add_u32_e32 v4, s2, v4\n" in text
    # d among the sources: the first add consumes it
    assert "\tv_add_u32_e32 v4, v1, v4\n\tv_add_u32_e32 v4, v2, v4\n" in text or "\tv_add_u32_e32 v4, v4, v1\n\tv_add_u32_e32 v4, v2, v4\n" in text
    assert "\tv_add_u32_e32 v1, v2, v1\n\tv_add_u32_e32 v1, v3, v1\n" in text or "\tv_add_u32_e32 v1, v1, v2\n\tv_add_u32_e32 v1, v3, v1\n" in text
    assert "\tv_add_u32_e32 v1, v3, v1\n\tv_add_u32_e32 v1, v2, v1\n" in text or "\tv_add_u32_e32 v1, v2, v3\n" in text
    # d twice among the sources, or two non-VGPR addends in the first add: left as it is
    assert "\tv_add3_u32 v1, v1, v1, v3\n" in text
    # every split pair adds exactly the original three operands
    src = [l for l in _lines() if "v_add3_u32" in l]
    assert n == len(src) - 1
    # every k-th only
    out2, n2 = P.split_add3(_lines(), 2)      # the counter restarts in every kernel
    assert n2 == 3 and "".join(out2).count("v_add3_u32") == len(src) - 3
    assert "\tv_add_u32_e32 v1, s1, v3\n\tv_add_u32_e32 v1, s2, v1\n" in "".join(out2)


def _semantics(lines):
    """(dst, sorted addends) of every v_add3_u32 / pair of v_add_u32 produced by the split, in order."""
    out = []
    for ln in lines:
        m = re.match(r"\s+v_add3_u32\s+(v\d+),\s*([^,]+),\s*([^,]+),\s*(\S+)", ln)
        if m:
            out.append((m.group(1), sorted(x.strip() for x in m.groups()[1:])))
    return out


def test_the_assembly_of_the_library_in_the_tree_differs_only_by_what_the_pass_may_insert(native):
    work = os.path.join(ROOT, "build", "obj", "libvkmr_hip.so")
    before, after = os.path.join(work, "device.s"), os.path.join(work, "device_prio.s")
    if not (os.path.exists(before) and os.path.exists(after)):
        pytest.skip("no build/obj in this checkout (the library was built elsewhere)")
    from vk_merkle_roots_amd import build, isa_prio_pass as P
    src = open(before).readlines()
    dst = open(after).readlines()
    if os.path.getmtime(after) < os.path.getmtime(build.HIP_LIB) - 600:
        pytest.skip("build/obj is older than the library")
    # undo the pass: drop inserted s_setprio 1 / the s_setprio 0 that closes each run, fuse split pairs back
    expect, _ = P.split_add3(src, build.SPLIT_ADD3_EVERY, build.LATENCY_BOUND_KERNELS)
    got = []
    open_run = False
    for ln in dst:
        if ln == "\ts_setprio 1\n":
            open_run = True
            continue
        if ln == "\ts_setprio 0\n" and open_run:
            open_run = False
            continue
        got.append(ln)
    assert got == expect
    # and the split itself kept every addend: compare against the add3 it replaced
    k = 0
    e_iter = iter(expect)
    for ln in src:
        m = P._ADD3.match(ln.split(";")[0].rstrip() if ";" in ln else ln.rstrip("\n"))
        e = next(e_iter)
        if m and e != ln:
            e2 = next(e_iter)
            a = re.match(r"\s+v_add_u32_e32\s+(v\d+),\s*([^,]+),\s*(\S+)", e)
            b = re.match(r"\s+v_add_u32_e32\s+(v\d+),\s*([^,]+),\s*(\S+)", e2)
            assert a and b and a.group(1) == b.group(1) == m.group(2) and b.group(3) == m.group(2)
            assert sorted([a.group(2).strip(), a.group(3).strip(), b.group(2).strip()]) == sorted(x.strip() for x in m.groups()[2:])
            k += 1
    assert k > 0


def _hash_listing(extra=""):
    """A listing with one kernel whose only loop body looks like a SHA-256 block to hash_blocks()/audit(): more than 100 rotates."""
    body = []
    for r in range(128):
        body += [f"\tv_alignbit_b32 v1, v{r % 8}, v{r % 8}, {1 + r % 30}\n", "\tv_bitop3_b32 v2, v1, v3, v4 bitop3:0x96\n", "\tv_add3_u32 v5, v2, v6, v1\n",
                 "\tv_add_u32_e32 v6, v5, v2\n", "\tv_lshrrev_b32_e32 v7, 3, v6\n"]
    return (["\t.text\n", "_Z18reduce_pass_kernelPKN8vkmr_dev4NodeE9SliceGeomPS0_j:\n", "; %bb.0:\n", "\ts_load_dwordx2 s[0:1], s[0:1], 0x0\n", ".LBB0_1:\n"] + body +
            ([extra] if extra else []) + ["\ts_cbranch_scc1 .LBB0_1\n", ".LBB0_2:\n", "\ts_endpgm\n"])


def test_an_opcode_nobody_measured_is_reported_not_silently_priced():
    """classify() prices every unknown VALU mnemonic as complex and hash_blocks() finds hashes by their rotates; a
    compiler that starts to emit another opcode inside the rounds, or lays the hashes out in other blocks, would move both the
    speed and the floor bench.py divides by without a test noticing.  audit() names such opcodes and such block counts."""
    from vk_merkle_roots_amd import isa_prio_pass as P
    clean = P.audit(_hash_listing())
    assert clean["unclassified"] == [] and clean["block_count_errors"] == [] and clean["hash_valu"]["v_alignbit_b32"] == 128
    assert list(clean["blocks"].values()) == [1]
    # one opcode renamed in the listing: red
    renamed = [ln.replace("v_add_u32_e32 v6", "v_xad_u32 v6", 1) if k == 8 else ln for k, ln in enumerate(_hash_listing())]
    assert any("v_xad_u32" in u for u in P.audit(renamed)["unclassified"])
    # an opcode that is only ASSUMED complex is tolerated at a block's edge, not inside the rounds
    assert P.audit(_hash_listing("\tv_add_lshl_u32 v9, v1, v2, 2\n"))["unclassified"] == []
    many = _hash_listing("".join("\tv_add_lshl_u32 v9, v1, v2, 2\n" for _ in range(P.ASSUMED_PER_BLOCK + 1)))
    many = [x for ln in many for x in (ln.splitlines(keepends=True) if ln.count("\n") > 1 else [ln])]
    assert any("tolerated" in u for u in P.audit(many)["unclassified"])
    # the same hash in two basic blocks of a kernel that should hold one: red
    twice = _hash_listing()
    twice = twice[:-1] + [".LBB0_3:\n"] + twice[5:-3] + ["\ts_endpgm\n"]
    assert any("expected 1" in e for e in P.audit(twice)["block_count_errors"])


def test_verify_accepts_what_the_pass_does_and_nothing_else():
    from vk_merkle_roots_amd import isa_prio_pass as P
    src = _lines()
    out, _ = P.transform(src, gap=0, split_every=2)
    assert P.verify(src, out) == []
    broken = list(out)
    k = next(i for i, ln in enumerate(broken) if "v_lshrrev_b32_e32 v3, 3, v0" in ln)
    broken[k] = broken[k].replace("3, v0", "4, v0")
    assert P.verify(src, broken)
    dropped = [ln for ln in out if "v_perm_b32" not in ln]
    assert P.verify(src, dropped)


def test_the_shipped_library_is_covered_by_the_issue_model(native):
    """The record the build writes beside the product library: every VALU opcode in its hash blocks is one the issue
    measurements covered, every kernel shows the hash blocks the static counts assume, and the pass's output was verified."""
    import json
    path = os.path.splitext(native.HIP_LIB)[0] + ".isa.json"
    if not os.path.exists(path):
        pytest.skip("the library in the tree was built without the issue pass (llvm tools absent)")
    rec = json.load(open(path))
    assert rec["audit"]["unclassified"] == [], rec["audit"]["unclassified"]
    assert rec["audit"]["block_count_errors"] == [], rec["audit"]["block_count_errors"]
    assert "verify" in rec["verified"]
    # every hash block of a kernel the pass works on got its priority toggles: the pass tracks a kernel's own s_setprio (the map
    # prologue's) in textual order, and a hash block laid out inside such a region would silently run unpaired
    from vk_merkle_roots_amd.build import LATENCY_BOUND_KERNELS
    assert [k for k in rec["audit"]["hash_blocks_without_raised_runs"] if not any(s in k for s in LATENCY_BOUND_KERNELS)] == []
    names = " ".join(rec["audit"]["blocks"])
    for kernel in ("reduce_pass_kernel", "reduce_collapse_kernel", "reduce_tail_kernel", "reduce_level_kernel", "map_kernel"):
        assert kernel in names
    assert sum(1 for k in rec["audit"]["blocks"] if "map_kernel" in k) == 5    # the five shipped instantiations: staged, per-lane x 2 launch widths, two-block x 2
