#!/usr/bin/env python3
"""BENCH-ONLY (timing only, results wrong): copies of the library with parts of the filter epilogue's candidate
hand-off removed, to apportion what a candidate costs in the candidate-dense early stages (tools/trace_step.sh):
    noatomic  every hit is stored at slot 0 of its segment (no LDS slot request)
    nostore   the slot is requested, nothing is stored
    nohit     the hit test is kept, hits do nothing
    nomask    as built, but the per-row tile bitmask is not updated (no global atomicOr)
    nocnt     as built, but neither the per-tile counts nor the tile bitmask are written
usage: python tools/ablation/make_epilogue_variants.py   -> image-classification_amd/libmi_knn_{noatomic,nostore,nohit}.so

These are the arms behind DESIGN.md 5.3 (1) and apply to the epilogue as it was when they were measured: one slot request
and one store per survivor and column block, i.e. the tree of `git checkout e739797~1`.  From e739797 on the first survivor
of every lane of a row group goes out in one store instruction (what these arms found) and the anchors below no longer
exist: run this from a checkout of that parent commit."""
import os
import shutil
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "image-classification_amd")
OLD = "seg[atomicAdd(lcnt + rl, 1u)] = make_uint2(__float_as_uint(v), (unsigned)(col_base + j * 16));"
NEW = {
    "noatomic": "seg[0] = make_uint2(__float_as_uint(v), (unsigned)(col_base + j * 16));",
    "nostore": "atomicAdd(lcnt + rl, 1u);",
    "nohit": "asm volatile(\"\" :: \"v\"(v));",
}
MASK_OLD = "if (n) atomicOr(p.tmask + (long long)(q0 + tid) * p.mask_words + (st >> 5), 1u << (st & 31));"
CNT_OLD = "p.cnt2[(long long)st * p.bpad + q0 + tid] = (unsigned short)n;           // one contiguous block per tile"
NEW["nomask"] = None
NEW["nocnt"] = None
for name, new in NEW.items():
    with tempfile.TemporaryDirectory() as d:
        shutil.copytree(os.path.join(PKG, "csrc"), os.path.join(d, "pkg", "csrc"))
        shutil.copytree(os.path.join(ROOT, "include"), os.path.join(d, "include"))
        p = os.path.join(d, "pkg", "csrc", "sim_kernel.h")
        s = open(p).read()
        if not (s.count(OLD) == 1 and s.count(MASK_OLD) == 1 and s.count(CNT_OLD) == 1):
            raise SystemExit("make_epilogue_variants.py: the per-survivor epilogue these arms patch is not in this tree "
                             "(it was replaced in e739797): run from `git checkout e739797~1`")
        if name == "nomask":
            s = s.replace(MASK_OLD, "asm volatile(\"\" :: \"v\"(n));")
        elif name == "nocnt":
            s = s.replace(MASK_OLD, "asm volatile(\"\" :: \"v\"(n));").replace(CNT_OLD, "")
        else:
            s = s.replace(OLD, new)
        open(p, "w").write(s)
        # no speculation: a variant that drops candidates would make every row look like a failed speculation
        h = os.path.join(d, "pkg", "csrc", "mi_knn.hip")
        hs = open(h).read()
        nospec = "    if (N > S0_COLS && k <= mi::WSEL_MAX_K && pl.cap > 0) {"
        assert hs.count(nospec) == 1
        open(h, "w").write(hs.replace(nospec, nospec.replace("if (N >", "if (false && N >")))
        out = os.path.join(PKG, f"libmi_knn_{name}.so")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
                               "-o", out, "mi_knn.hip"], cwd=os.path.join(d, "pkg", "csrc"))
        print(out)
