#!/usr/bin/env python3
"""BENCH-ONLY: libmi_knn_substep.so -- the STAMPED f32 filter kernel records, per wave, how long each of the 8 sub-steps
(32 MFMAs of v_mfma_f32_16x16x4_f32 each) of a K-step takes, summed over the tile (stamps at sub-step boundaries only,
no LDS wait attached to them).  Read with tools/ablation/substep_probe.py."""
import os
import shutil
import subprocess
import sys
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
            "    unsigned long long sub[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};\n"
            "    auto kstep = [&](int t, auto dma_tag, auto pf_tag) {\n"
            "        unsigned long long ts[10];")
    s = rep(s, "            for (int kk = 0; kk < 8; ++kk) {\n                float *ac = (kk & 1) ? a1 : a0, *an = (kk & 1) ? a0 : a1;\n",
            "            for (int kk = 0; kk < 8; ++kk) {\n"
            "                if constexpr (STAMP) { __builtin_amdgcn_sched_barrier(0); ts[kk] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }\n"
            "                float *ac = (kk & 1) ? a1 : a0, *an = (kk & 1) ? a0 : a1;\n")
    s = rep(s, "                MI_MM(2);\n            }\n            MI_MM(32);\n",
            "                MI_MM(2);\n            }\n            MI_MM(32);\n"
            "            if constexpr (STAMP) { __builtin_amdgcn_sched_barrier(0); ts[8] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }\n")
    s = rep(s, "        stage_wait(pf_tag);\n        __syncthreads();\n    };",
            "        stage_wait(pf_tag);\n        __syncthreads();\n"
            "        if constexpr (STAMP && DT == DT_F32) {\n"
            "            ts[9] = __builtin_amdgcn_s_memtime();\n"
            "            for (int i = 0; i < 9; ++i) sub[i] += ts[i + 1] - ts[i];\n"
            "        }\n    };")
    s = rep(s, "            if (tid == 0) {\n"
               "                const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();\n"
               "                unsigned long long *o = p.stamps + 2ll * (bid % p.stamp_cap);\n"
               "                o[0] = c1 - stamp_c0;\n"
               "                o[1] = r1 - stamp_r0;\n"
               "            }",
            "            if (lane == 0) {\n"
            "                unsigned long long *o = p.stamps + 2ll * (((long long)bid * 64 + w * 8) % p.stamp_cap);\n"
            "                for (int i = 0; i < 9; ++i) o[i] = sub[i];\n"
            "            }")
    open(p, "w").write(s)
    out = os.path.join(PKG, "libmi_knn_substep%s.so" % (sys.argv[1] if len(sys.argv) > 1 else ""))
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC"] + sys.argv[2:]
                          + ["-o", out, "mi_knn.hip"], cwd=os.path.join(d, "pkg", "csrc"))
    print(out)
