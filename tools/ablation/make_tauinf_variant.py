#!/usr/bin/env python3
"""BENCH-ONLY: build libmi_knn_tauinf.so, a copy of the library whose filter kernel stages +inf as every row's threshold,
so that no candidate passes whatever the operands hold (the results are meaningless).  With it the SAME instruction stream
can be timed on random, constant and zero operands: what changes between those runs is the energy the MFMA datapath and
the operand movement draw, i.e. the clock the chip grants (tools/ablation/toggle_probe.py).
    python tools/ablation/make_tauinf_variant.py && MI_KNN_LIB=.../libmi_knn_tauinf.so python tools/ablation/toggle_probe.py
"""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "image-classification_amd")

with tempfile.TemporaryDirectory() as d:
    shutil.copytree(os.path.join(PKG, "csrc"), os.path.join(d, "pkg", "csrc"))
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(d, "include"))
    p = os.path.join(d, "pkg", "csrc", "sim_kernel.h")
    s = open(p).read()
    old = "            else t = p.tau[q0 + tid];"
    assert s.count(old) == 1
    s = s.replace(old, "            else t = INFINITY;")
    # no speculation: with nothing passing, every row would look like a failed speculation and be redone
    h = os.path.join(d, "pkg", "csrc", "mi_knn.hip")
    hs = open(h).read()
    nospec = "    if (N > S0_COLS && k <= mi::WSEL_MAX_K && pl.cap > 0) {"
    assert hs.count(nospec) == 1
    open(h, "w").write(hs.replace(nospec, nospec.replace("if (N >", "if (false && N >")))
    suffix = ""
    if len(sys.argv) > 1 and sys.argv[1] == "nowait":      # timing only, a data race: the K loop starts without waiting for stage 0
        old2 = "    stage_wait(std::false_type{});\n    __syncthreads();\n"
        assert s.count(old2) == 1
        s = s.replace(old2, "    __syncthreads();\n")
        suffix = "_nowait"
    open(p, "w").write(s)
    out = os.path.join(PKG, "libmi_knn_tauinf%s.so" % suffix)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
                           "-o", out, "mi_knn.hip"], cwd=os.path.join(d, "pkg", "csrc"))
    print(out)
