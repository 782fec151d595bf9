#!/usr/bin/env python3
"""Static instruction budget of blend_bwd_q_kernel<9> per 16-entry group, by what the instructions are FOR.

  hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize -std=c++17 -gline-tables-only --cuda-device-only -S \
        gs-2m_amd/csrc/blend_bwd_q.hip -o /tmp/bwd_g.s ; python3 tools/isa_budget.py /tmp/bwd_g.s > profiles/r05_bwd_isa_budget.md

Every instruction of the kernel's listing carries the source line the compiler attributes it to (.loc).  Lines are mapped to
categories (the table below names the line ranges of gs-2m_amd/csrc/blend_bwd_q.hip at this commit); the block loop (4 iterations
per group, not unrolled) is weighted x 4, everything else of the group x 1.  Inline-asm scans and lines of common.h are
attributed through the line of their call site when inlined."""
import re, sys, collections

src = open(sys.argv[1]).read().split("\n")
KERNEL = "blend_bwd_q_kernelILi9E"
start = [i for i, l in enumerate(src) if re.match(r"^_Z.*" + KERNEL + r".*:", l)][0]
end = [i for i, l in enumerate(src) if i > start and l.startswith(".Lfunc_end")][0]
files = {}
for l in src:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', l)
    if m:
        files[int(m.group(1))] = m.group(3)
    else:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"', l)
        if m:
            files[int(m.group(1))] = m.group(2)

# (first line, last line, category, weight per group) of blend_bwd_q.hip -- the per-group part; earlier lines: kernel prologue
CATS = [
    (1, 42, "kernel prologue (per quadrant, not per group)", 0),
    (43, 63, "block: T scan (inline asm: Kogge-Stone product over the 16 entry lanes, 4 DPP levels x 2 chains per row step)", 4),
    (64, 82, "block: S scan (inline asm: Kogge-Stone sum, 4 DPP levels x 2 chains per row step)", 4),
    (83, 145, "kernel prologue (per quadrant, not per group)", 0),
    (146, 238, "list walk: entry / record loads, row pointer, flush of the previous group's rows (per group)", 1),
    (239, 259, "group start: first colour.gradient MFMA block, operand splats", 1),
    (260, 282, "block: LDS operand reads (T, S, n_contrib, gradient columns, next A operands)", 4),
    (283, 299, "block: dx, dy, power (written order), exp2", 4),
    (300, 312, "block: alpha tests (position, power <= 0, alpha >= 1/255), clamp, 1 - alpha, rcp", 4),
    (313, 316, "block: T scan call site + transmittance", 4),
    (317, 328, "block: w, colour.gradient pick-up, S scan (DPP), dL/dalpha", 4),
    (329, 333, "block: write-back of the running T / S (lane j == 15)", 4),
    (334, 342, "block: s = opacity dL/dalpha G, mean-gradient terms, |.| channels", 4),
    (343, 348, "block: W x Ggrad MFMAs (2 per row step)", 4),
    (349, 351, "block: moments of s (m0, m1, m2)", 4),
    (352, 362, "block: next block's colour.gradient MFMAs", 4),
    (363, 364, "flush of the previous group's rows (call site)", 1),
    (365, 367, "next group's loads (call site)", 1),
    (368, 400, "epilogue: pixel-column reduction (permlane swaps), row assembly in LDS", 1),
    (401, 414, "zero rows behind the last contributor / loop head (per quadrant)", 0),
    (415, 430, "group install (survivor registers) + loop control", 1),
]

def cat_of(fname, line):
    if not fname.endswith("blend_bwd_q.hip"):
        return None
    for a, b, c, w in CATS:
        if a <= line <= b:
            return c, w
    return "other lines of blend_bwd_q.hip", 1

KIND = lambda op: ("mfma" if op.startswith("v_mfma") else "valu_dpp" if False else "trans" if op in ("v_exp_f32_e32", "v_rcp_f32_e32", "v_exp_f32_e64", "v_rcp_f32_e64")
                   else "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_")) else "other")
cur = (None, 0)
inl = None  # the blend_bwd_q.hip line an inlined helper of another file was called from: approximated by the last .loc of the main file
last_main = ("blend_bwd_q.hip", 0)
tab = collections.defaultdict(lambda: collections.Counter())
weights = {}
for l in src[start + 1:end]:
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        fname, line = files.get(int(m.group(1)), "?"), int(m.group(2))
        cur = (fname, line)
        if fname.endswith("blend_bwd_q.hip"):
            last_main = cur
        continue
    t = l.strip()
    if not t or t.startswith((".", ";", "//")) or t.endswith(":"):
        continue
    op = t.split()[0]
    c = cat_of(*cur) or cat_of(*last_main)
    name, w = c
    weights[name] = w
    k = KIND(op)
    if "dpp" in t and op.startswith("v_"):
        k = "valu_dpp"
    if op == "s_nop" or op == "s_waitcnt":
        k = "wait/nop"
    tab[name][k] += 1

cols = ["valu", "valu_dpp", "trans", "mfma", "lds", "vmem", "salu", "wait/nop"]
print("# blend_bwd_q_kernel<9>: static instruction budget per 16-entry group (tools/isa_budget.py)\n")
print("| what | x per group | " + " | ".join(cols) + " | vector instructions per group |")
print("|---|---|" + "---|" * (len(cols) + 1))
tot = collections.Counter(); grand = 0
order = [c for _, _, c, _ in CATS] + ["other lines of blend_bwd_q.hip"]
for name in order:
    if name not in tab:
        continue
    w = weights[name]
    vec = (tab[name]["valu"] + tab[name]["valu_dpp"] + tab[name]["trans"] + tab[name]["mfma"]) * w
    grand += vec
    for k in cols:
        tot[k] += tab[name][k] * w
    print(f"| {name} | {w} | " + " | ".join(str(tab[name][k]) for k in cols) + f" | {vec} |")
print("| **per group (weighted)** | | " + " | ".join(str(tot[k]) for k in cols) + f" | **{grand}** |")
ARITH = ("block: dx, dy", "block: alpha tests", "block: T scan", "block: S scan", "block: w, colour", "block: s = opacity", "block: W x Ggrad", "block: moments", "block: next block's", "group start", "epilogue")
ar = sum((tab[n]["valu"] + tab[n]["valu_dpp"] + tab[n]["trans"] + tab[n]["mfma"]) * weights[n] for n in tab if n.startswith(ARITH))
print(f"\nThe reference's arithmetic in this formulation (alpha evaluation, the two recurrences as scans, dL/dalpha, the gradient terms and their sums over pixels, the per-group reduction): **{ar} of {grand} vector instructions = {100.0 * ar / grand:.0f} %**; the rest is the list walk (entry / record / row addressing, the deferred row flush), LDS operand reads and their addressing, loop control.")
