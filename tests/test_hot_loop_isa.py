"""CPU (compile only): the steady-state loop of the headline instantiations of gemv_pair_kernel must stay straight-line.

Round 2 put a runtime-guarded store (the column-block residual, `YSource::res_out`) behind every row's barrier: hipcc
turned it into `s_and_saveexec` + `s_cbranch_execz` + a vmcnt(0)-guarded load/store block in front of the next tile's
`global_load_dwordx4` (cfg2 loop 333 -> 370 instructions, 2 -> 11 branches) and the in-loop headline lost 4.5 %.  The store
is a compile-time template flag now (CB); this test disassembles the three headline instantiations (tests/isa_probe.hip,
~1 s) and fails when the loop grows branches, a full `vmcnt(0)` drain (only the DRAIN form has one, by design), or
instructions again."""
import collections
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _functions(asm):
    fn, cur = {}, None
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m and "gemv_pair_kernel" in m.group(1):
            cur = m.group(1)
            fn[cur] = []
            continue
        if cur is not None:
            if line.startswith(".Lfunc_end"):
                cur = None
                continue
            fn[cur].append(line.rstrip())
    return fn


def _hot_loop(lines):
    """The backward-branch body with the most 16-byte global loads = the software-pipelined steady state."""
    labels = {}
    for i, l in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    best = None
    for i, l in enumerate(lines):
        m = re.match(r"\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            body = lines[labels[m.group(1)]:i + 1]
            nload = sum("global_load_dwordx4" in b for b in body)
            if best is None or nload > best[0]:
                best = (nload, body)
    assert best is not None, "no loop found"
    ops = []
    for l in best[1]:
        l = l.split(";")[0].strip()
        if l and not l.startswith("."):
            ops.append(l)
    return ops


# mangled-name fragment -> (16-byte loads per trip, max instructions, full drains allowed per trip)
# The fp32 forms end a trip with ONE vmcnt(0) + register moves: hipcc defers a step's g += A_i r_i behind the next step's
# loads, the tile registers are therefore live one step longer than written and the rotation closes with copies (round-1
# ISA, measured at 88-90 % of the roofline in that form; DESIGN.md section 3 "What the ISA showed").  More than that = regression.
EXPECT = {
    "IfLi512ELi4ELi1ELb1ELi2ELb1ELi3E": (12, 345, 1),            # cfg2: 3 tiles x 4 chunks per trip, counted vmcnt inside
    "IfLi1024ELi4ELi1ELb1ELi4ELb1ELi2E": (8, 270, 3),            # cfg4: 2 tiles x 4 chunks, DRAIN (one vmcnt(0) per step by design)
    "INS_6bf16_tELi512ELi4ELi1ELb1ELi2ELb1ELi3E": (12, 500, 0),  # cfg5: bf16, 3 tiles, counted vmcnt only
}


@pytest.mark.timeout(300)
def test_headline_loops_are_straight_line(tmp_path):
    from fastoptsolver_amd import build
    src = os.path.join(ROOT, "tests", "isa_probe.hip")
    out = tmp_path / "probe.s"
    subprocess.run([build.hipcc_path(), "-O3", "--offload-arch=gfx950", "-std=c++17", "--cuda-device-only", "-S", "-o", str(out),
                    src], check=True, capture_output=True)
    fns = _functions(out.read_text())
    assert len(fns) == 3, list(fns)
    for frag, (loads, max_len, drains) in EXPECT.items():
        name = [k for k in fns if frag in k]
        assert len(name) == 1, (frag, list(fns))
        ops = _hot_loop(fns[name[0]])
        c = collections.Counter(o.split()[0] for o in ops)
        branches = sum(v for k, v in c.items() if k.startswith("s_cbranch") or k == "s_branch")
        assert c["global_load_dwordx4"] == loads, (frag, c["global_load_dwordx4"])
        # the loop's own back edge (+ its exit test) and nothing else: no exec-masked block inside the pipeline
        assert branches <= 2, (frag, branches, [o for o in ops if o.startswith("s_cbranch") or o.startswith("s_branch")])
        assert c["s_cbranch_execz"] == 0 and c["s_cbranch_execnz"] == 0, (frag, dict(c))
        # no global store / atomic in the steady state (slabs are written once, in the epilogue)
        assert not [o for o in ops if o.startswith("global_store") or o.startswith("global_atomic")], frag
        full_drains = sum(1 for o in ops if re.match(r"s_waitcnt\s+vmcnt\(0\)", o))
        assert full_drains <= drains, (frag, full_drains)
        assert len(ops) <= max_len, (frag, len(ops))
        # one workgroup barrier per row step
        assert c["s_barrier"] == loads // 4, (frag, c["s_barrier"])
