#!/usr/bin/env python3
"""BENCH-ONLY: build libmi_knn_seam.so, a diagnostic copy of the library whose STAMPED filter kernel records, per workgroup,
the shader-clock time at its first and last instruction and the CU it ran on (HW_ID / XCC_ID), so that the idle time of a CU
between two consecutive workgroups (dispatch + the tile prologue) can be read off: tools/ablation/seam_probe.py."""
import os
import shutil
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "image-classification_amd")


def rep(s, old, new):
    assert s.count(old) == 1, (s.count(old), old)
    return s.replace(old, new, 1)


with tempfile.TemporaryDirectory() as d:
    shutil.copytree(os.path.join(PKG, "csrc"), os.path.join(d, "pkg", "csrc"))
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(d, "include"))
    p = os.path.join(d, "pkg", "csrc", "sim_kernel.h")
    s = open(p).read()
    s = rep(s, "    auto kstep = [&](int t, auto dma_tag, auto pf_tag) {",
            "    unsigned long long seam_k0 = 0ull;\n"
            "    if constexpr (STAMP) seam_k0 = __builtin_amdgcn_s_memtime();      // the K loop starts here\n"
            "    auto kstep = [&](int t, auto dma_tag, auto pf_tag) {")
    s = rep(s, "            if (tid == 0) {\n"
               "                const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();\n"
               "                unsigned long long *o = p.stamps + 2ll * (bid % p.stamp_cap);\n"
               "                o[0] = c1 - stamp_c0;\n"
               "                o[1] = r1 - stamp_r0;\n"
               "            }",
            "            if (tid == 0) {\n"
            "                const unsigned long long c1 = __builtin_amdgcn_s_memtime();\n"
            "                const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);\n"
            "                unsigned long long *o = p.stamps + 2ll * (((long long)bid * 2) % p.stamp_cap);\n"
            "                o[0] = stamp_c0; o[1] = c1; o[2] = ((unsigned long long)xcc << 32) | hw; o[3] = seam_k0;\n"
            "            }")
    open(p, "w").write(s)
    out = os.path.join(PKG, "libmi_knn_seam.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
                           "-o", out, "mi_knn.hip"], cwd=os.path.join(d, "pkg", "csrc"))
    print(out)
