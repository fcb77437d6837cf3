#!/usr/bin/env python3
"""BENCH-ONLY (timing only, results wrong): copies of the library that bound the two energy levers cdna_hip_programming.md
rule 28 names for an MFMA-dense loop the chip clocks down, on the bf16 similarity kernel.  All three stage +inf as every
threshold (nothing passes: same epilogue work whatever the similarities are):
    e_base       the kernel as built
    e_halfreads  half of the LDS operand reads of a K-step are skipped (the second A block of each pair and the second
                 K half's B blocks reuse the registers of the first): the bound for ANY re-tiling that reads fewer LDS
                 bytes per MFMA (a 128 x 128 wave tile would save a third)
    e_morereads  +14 discarded ds_read_b128 on top of the 24 of every K-step (same MFMA operands): the slope of the clock
                 against LDS read bytes without e_halfreads' confound (its skipped reads repeat MFMA operands)
    e_l2all      every workgroup stages the SAME two operand tiles (query tile 0, bank tile 0 of the launch): all operand
                 traffic is served by the L2, nothing comes from beyond it -- the bound for ANY tile / launch order
                 that raises the L2 hit rate
usage: python tools/ablation/make_energy_variants.py  -> image-classification_amd/libmi_knn_e_{base,halfreads,l2all}.so
       then on the GPU box: tools/ablation/energy_probe.sh"""
import os
import shutil
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "image-classification_amd")

NOSPEC_OLD = "    if (N > S0_COLS && k <= mi::WSEL_MAX_K && pl.cap > 0) {"
NOSPEC_NEW = "    if (false && N > S0_COLS && k <= mi::WSEL_MAX_K && pl.cap > 0) {"
TAU = ("            else t = p.tau[q0 + tid];", "            else t = INFINITY;")
HALF = [
    ("            a0[1] = *(const bf16x8 *)(la + 2048 + coff0);", "            a0[1] = a0[0];"),
    ("                    an[1] = *(const bf16x8 *)(la + (2 * np + 1) * 2048 + co);", "                    an[1] = an[0];"),
    ("                        for (int j = 0; j < 4; ++j) b1[j] = *(const bf16x8 *)(lb + j * 2048 + coff1);",
     "                        for (int j = 0; j < 4; ++j) b1[j] = b0[j];"),
]
# the opposite probe, free of the confound of e_halfreads (its skipped reads also REPEAT MFMA operands, which lowers the
# datapath's toggling): every A-pair read is issued twice, the second copy (a neighbouring 16-byte chunk of the same rows)
# is kept live and discarded -- +14 of 24 ds_read_b128 per K-step with identical MFMA operands
MORE = [
    ("                    an[1] = *(const bf16x8 *)(la + (2 * np + 1) * 2048 + co);",
     "                    an[1] = *(const bf16x8 *)(la + (2 * np + 1) * 2048 + co);\n"
     "                    { bf16x8 x0 = *(const bf16x8 *)(la + (2 * np) * 2048 + (co ^ 16)), x1 = *(const bf16x8 *)(la + (2 * np + 1) * 2048 + (co ^ 16));\n"
     "                      asm volatile(\"\" :: \"v\"(x0), \"v\"(x1)); }"),
]
L2ALL = [
    ("        int qa = q0 + r; qa = qa < p.B ? qa : p.B - 1;", "        int qa = r; qa = qa < p.B ? qa : p.B - 1;"),
    ("        int xb = n0 + r;", "        int xb = p.col0 + r;"),
    ("            int r = q0 + la * (inner >> p.qsh) + idx; r = r < p.B ? r : p.B - 1;",
     "            int r = la * (inner >> p.qsh) + idx; r = r < p.B ? r : p.B - 1;"),
    ("            int r = n0 + lb * (inner & ((1 << p.qsh) - 1)) + (idx - la);",
     "            int r = p.col0 + lb * (inner & ((1 << p.qsh) - 1)) + (idx - la);"),
]
for name, edits in (("e_base", []), ("e_halfreads", HALF), ("e_l2all", L2ALL), ("e_morereads", MORE)):
    with tempfile.TemporaryDirectory() as d:
        shutil.copytree(os.path.join(PKG, "csrc"), os.path.join(d, "pkg", "csrc"))
        shutil.copytree(os.path.join(ROOT, "include"), os.path.join(d, "include"))
        p = os.path.join(d, "pkg", "csrc", "sim_kernel.h")
        s = open(p).read()
        for old, new in [TAU] + edits:
            assert s.count(old) == 1, old
            s = s.replace(old, new)
        open(p, "w").write(s)
        # no speculation: with nothing passing, every row would look like a failed speculation and be redone
        h = os.path.join(d, "pkg", "csrc", "mi_knn.hip")
        hs = open(h).read()
        assert hs.count(NOSPEC_OLD) == 1
        open(h, "w").write(hs.replace(NOSPEC_OLD, NOSPEC_NEW))
        out = os.path.join(PKG, f"libmi_knn_{name}.so")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
                               "-o", out, "mi_knn.hip"], cwd=os.path.join(d, "pkg", "csrc"))
        print(out)
