#!/usr/bin/env python3
"""Instruction-class counts of the render kernels' loops, from the disassembly of the shipped build.

    python profiles/make_isa_mix.py > profiles/r06_render_isa_mix.txt          # CPU only: hipcc -S cross-compiles gfx950

render.hip is compiled to assembly with the Makefile's flags (--offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize), the two instantiations
the headline step launches -- k_render_forward_q<true, 0> (normal image, no extra colour set) and k_render_backward_q<false, false, 0, false> --
are cut into basic blocks, natural loops are found from the backward branches, and every loop body (innermost blocks only: a block belongs to
the innermost loop that contains it) is counted by instruction class:

    plain    v_* that are none of the below (one issue slot: 1.32 ns per wave64 instruction and SIMD as measured, r02_issue_rate_microbench.txt)
    pk       v_pk_*                      two-wide fp32 (2.0 ns)
    dpp      any VALU instruction with a DPP / row_* / quad_perm modifier, v_readlane / v_readfirstlane / v_writelane, v_permlane*, ds_swizzle-free
             cross-lane moves (2.0 ns)
    trans    v_exp / v_log / v_rcp / v_rsq / v_sqrt / v_sin / v_cos (3.4 ns)
    lds      ds_*                        vmem  global_* / buffer_* / flat_* / scratch_*          salu  s_* except waits, nops and branches
    branch   s_cbranch_* / s_branch      wait  s_waitcnt / s_nop / s_barrier ...

`bench.py` reads the "wide_fraction" of the loop that dominates each kernel -- (pk + dpp + trans weighted as listed) -- from this file
(bench.load_isa_mix) instead of carrying literals.  Static counts say what a loop ITERATION costs; how often each loop runs comes from the
counting instantiations (profiles/r06_render_loop_trips.txt, written on the GPU by profiles/render_loop_trips.py)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "emd_amd", "csrc")
KERNELS = {
    "k_render_forward_q": "k_render_forward_qILb1ELi0E",            # <NORMAL = true, NX = 0>
    "k_render_backward_q": "k_render_backward_qILb0ELb0ELi0ELb0E",   # <NORMAL = false, ABS = false, NX = 0, STATS = false>
}
NS = {"plain": 1.32, "pk": 2.0, "dpp": 2.0, "trans": 3.4}          # measured issue time per wave64 instruction and SIMD, 8 waves resident
TRANS = re.compile(r"^v_(exp|log|rcp|rsq|sqrt|sin|cos)_")
DPP = re.compile(r"(row_shr|row_shl|row_ror|row_bcast|row_mirror|row_half_mirror|row_share|row_xmask|row_newbcast|quad_perm|wave_shr|wave_shl|wave_ror|wave_rol|dpp8|_dpp\b)")
CROSS = re.compile(r"^v_(readlane|readfirstlane|writelane|permlane|mov_b32_dpp|bpermute)")


def classify(op, text):
    if op.startswith("v_"):
        if TRANS.match(op):
            return "trans"
        if DPP.search(text) or CROSS.match(op):
            return "dpp"
        if op.startswith("v_pk_"):
            return "pk"
        return "plain"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_endpgm")):
        return "branch"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep", "s_wait", "s_code_end")):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return "other"


def assemble(extra_flags=()):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "render.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-fno-slp-vectorize", f"-I{os.path.join(ROOT, 'include')}", "-S",
               "--cuda-device-only", *extra_flags, os.path.join(CSRC, "render.hip"), "-o", out]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL, cwd=CSRC)
        return open(out).read().splitlines()


def function_body(lines, mangled_part):
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and mangled_part in l and ":" in l)
    body = []
    for l in lines[start + 1:]:
        if l.startswith(".Lfunc_end"):
            break
        body.append(l)
    return body


def blocks_of(body):
    """-> [(label, [(op, text)])] in layout order; the entry block is labelled '<entry>'."""
    blocks, cur, name = [], [], "<entry>"
    for l in body:
        s = l.strip()
        if not s or s.startswith((";", ".")) and not re.match(r"^\.LBB\d+_\d+:", s):
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            blocks.append((name, cur))
            name, cur = m.group(1), []
            continue
        s = s.split(";")[0].strip()
        if not s:
            continue
        op = s.split()[0]
        cur.append((op, s))
    blocks.append((name, cur))
    return blocks


def loops_of(blocks):
    """Natural loops by layout: a branch in block j to a label at block i <= j closes the loop [i, j].  -> [(i, j)] sorted, nested by containment."""
    index = {name: k for k, (name, _) in enumerate(blocks)}
    loops = set()
    for j, (_, ins) in enumerate(blocks):
        for op, text in ins:
            if op.startswith(("s_cbranch", "s_branch")):
                tgt = text.split()[-1]
                if tgt in index and index[tgt] <= j:
                    loops.add((index[tgt], j))
    # merge loops with the same header (several back edges)
    by_head = {}
    for i, j in loops:
        by_head[i] = max(by_head.get(i, j), j)
    return sorted(by_head.items())


def count(ins):
    c = {}
    for op, text in ins:
        k = classify(op, text)
        c[k] = c.get(k, 0) + 1
    return c


def report(kernel, mangled, lines):
    body = function_body(lines, mangled)
    blocks = blocks_of(body)
    loops = loops_of(blocks)
    owner = {}
    for k in range(len(blocks)):
        inside = [(i, j) for i, j in loops if i <= k <= j]
        owner[k] = min(inside, key=lambda ij: ij[1] - ij[0]) if inside else None
    total = count([x for _, ins in blocks for x in ins])
    out = [f"kernel {kernel}  ({mangled})  blocks {len(blocks)}  loops {len(loops)}  instructions {sum(total.values())}"]
    cols = ("plain", "pk", "dpp", "trans", "lds", "vmem", "salu", "branch", "wait")
    out.append("  " + f"{'region':34s}" + "".join(f"{c:>7s}" for c in cols) + "   valu  wide_fraction  issue_ns/iter")

    def line(name, c):
        valu = sum(c.get(k, 0) for k in ("plain", "pk", "dpp", "trans"))
        wide = (c.get("pk", 0) + c.get("dpp", 0) + c.get("trans", 0)) / valu if valu else 0.0
        ns = sum(c.get(k, 0) * NS[k] for k in NS)
        return "  " + f"{name:34s}" + "".join(f"{c.get(k, 0):7d}" for k in cols) + f"  {valu:5d}  {wide:13.3f}  {ns:13.1f}"
    out.append(line("whole kernel (static)", total))
    rows = []
    for (i, j) in loops:
        own = [x for k in range(i, j + 1) if owner[k] == (i, j) for x in blocks[k][1]]
        depth = sum(1 for a, b in loops if a <= i and j <= b) - 1
        c = count(own)
        rows.append(((i, j), depth, c))
        out.append(line(f"{'  ' * depth}loop {blocks[i][0]}..{blocks[j][0]} (own blocks)", c))
    straight = count([x for k in range(len(blocks)) if owner[k] is None for x in blocks[k][1]])
    out.append(line("outside every loop", straight))
    return out, rows, blocks


def main():
    lines = assemble()
    print("# profiles/make_isa_mix.py: instruction classes of the render kernels' loops, from `hipcc -S --cuda-device-only` of emd_amd/csrc/render.hip with the Makefile's flags.")
    print("# A block is counted in the innermost loop that contains it.  issue_ns/iter = plain x 1.32 + (pk + dpp) x 2.0 + trans x 3.4 ns (profiles/r02_issue_rate_microbench.txt):")
    print("# the vector-issue time of ONE pass through the loop's own blocks on one SIMD with 8 waves resident.  wide_fraction = (pk + dpp + trans) / valu.")
    summary = {}
    k6_loops = []
    for kernel, mangled in KERNELS.items():
        out, rows, blocks = report(kernel, mangled, lines)
        print()
        print("\n".join(out))
        # the loop that carries the per-(pixel, entry) arithmetic: the DEEPEST loop, among those the one with the most VALU instructions of its own
        best = max(rows, key=lambda r: (r[1], sum(r[2].get(k, 0) for k in ("plain", "pk", "dpp", "trans"))))
        if kernel == "k_render_forward_q" and len(rows) == 4:
            best = rows[3]          # K6's loops are siblings: the compositing drain (78.5 % of the kernel's vector instructions, r06_render_loop_trips.txt) is the last
        c = best[2]
        valu = sum(c.get(k, 0) for k in ("plain", "pk", "dpp", "trans"))
        summary[kernel] = dict(loop=f"{blocks[best[0][0]][0]}..{blocks[best[0][1]][0]}", valu=valu, plain=c.get("plain", 0), pk=c.get("pk", 0), dpp=c.get("dpp", 0),
                               trans=c.get("trans", 0), lds=c.get("lds", 0), vmem=c.get("vmem", 0), salu=c.get("salu", 0),
                               wide_fraction=round((c.get("pk", 0) + c.get("dpp", 0) + c.get("trans", 0)) / max(valu, 1), 4),
                               ns_per_valu=round(sum(c.get(k, 0) * NS[k] for k in NS) / max(valu, 1), 4))
        if kernel == "k_render_forward_q" and len(rows) == 4:
            # K6's four loops in layout (= source) order: the scan -> cull -> drain rounds, the scan of the list words, the exact footprint test of the
            # queued entries (it holds the kernel's only store inside a loop: the survivor ids), the compositing drain (unrolled twice)
            roles = ("round", "scan", "cull", "drain")
            assert rows[2][2].get("vmem", 0) >= 1 and rows[3][2].get("vmem", 0) == 0, "K6's loops are not in the expected order"
            for role, (ij, depth, c) in zip(roles, rows):
                valu = sum(c.get(k, 0) for k in ("plain", "pk", "dpp", "trans"))
                k6_loops.append(f"LOOP k_render_forward_q {role} valu={valu} lds={c.get('lds', 0)} vmem={c.get('vmem', 0)} salu={c.get('salu', 0)} "
                                f"unroll={2 if role == 'drain' else 1} ns={sum(c.get(k, 0) * NS[k] for k in NS):.1f}")
            st = count([x for k_ in range(len(blocks)) if not any(i <= k_ <= j for (i, j), _, _ in rows) for x in blocks[k_][1]])
            k6_loops.append(f"LOOP k_render_forward_q straight valu={sum(st.get(k, 0) for k in ('plain', 'pk', 'dpp', 'trans'))} lds={st.get('lds', 0)} "
                            f"vmem={st.get('vmem', 0)} salu={st.get('salu', 0)} unroll=1 ns={sum(st.get(k, 0) * NS[k] for k in NS):.1f}")
    print()
    print("# ---- K6's loops by role (profiles/render_loop_trips.py multiplies these by the measured trip counts) ----")
    for l in k6_loops:
        print(l)
    print()
    print("# ---- machine-readable summary of each kernel's hottest loop (bench.load_isa_mix reads these lines) ----")
    for kernel, s in summary.items():
        print("MIX " + kernel + " " + " ".join(f"{k}={v}" for k, v in s.items()))


if __name__ == "__main__":
    main()
