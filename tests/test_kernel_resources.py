"""Register budgets of the hot kernels, read from the code objects inside the built libdbgk.so (no GPU needed).

Why this is a test: the level-2 kernel runs two 512-thread workgroups per CU -- four waves per SIMD, 128 VGPRs each.  A change that
cost it four more registers (130) compiled, passed every parity test and silently halved its occupancy: the level-2 / build pair of
the cfg2 step went from 9.2 to 12.8 ms (round 4).  The kernels below have such a cliff right above their present allocation; the
limits are the hardware's (512 VGPRs per SIMD lane / waves per SIMD), not tuning targets.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "dbg_assembly_amd", "lib", "libdbgk.so")
LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_metadata(tmp_path):
    work = tmp_path / "co"
    work.mkdir()
    so = work / "libdbgk.so"
    shutil.copy(LIB, so)   # (llvm-objdump writes the bundles next to its input)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", str(so)], check=True, capture_output=True, cwd=work)
    meta = {}
    for f in sorted(work.iterdir()):
        if "gfx950" not in f.name:
            continue
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", str(f)], check=True, capture_output=True, text=True).stdout
        name = None
        for line in notes.splitlines():
            m = re.match(r"\s+\.name:\s+(\S+)", line)
            if m:
                name = m.group(1)
                meta.setdefault(name, {})
            m = re.match(r"\s+\.(vgpr_count|agpr_count|private_segment_fixed_size|max_flat_workgroup_size):\s+(\d+)", line)
            if m and name:
                meta[name][m.group(1)] = int(m.group(2))
    return meta


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "llvm-readelf")), reason="the ROCm LLVM tools are not installed")
def test_hot_kernels_stay_inside_their_register_budget(tmp_path):
    assert os.path.exists(LIB), "libdbgk.so not built (python -c 'import __graft_entry__ as g; g.build()')"
    meta = kernel_metadata(tmp_path)
    assert len(meta) > 100, "could not read the kernel descriptors of libdbgk.so"

    def check(pattern, max_vgpr, max_scratch, at_least=1):
        hits = [(n, m) for n, m in meta.items() if re.search(pattern, n)]
        assert len(hits) >= at_least, (pattern, len(hits))
        for n, m in hits:
            regs = m["vgpr_count"] + m.get("agpr_count", 0)
            assert regs <= max_vgpr, "%s: %d VGPRs, its occupancy needs <= %d" % (n, regs, max_vgpr)
            assert m["private_segment_fixed_size"] <= max_scratch, "%s spills %d bytes" % (n, m["private_segment_fixed_size"])

    # level 2, up to 1024 final buckets (graph records and the 32-bit KFREQ records): one workgroup of 16 waves per CU (round 5: 1024 threads,
    # 16 K-record tiles) = four waves per SIMD.  (A few dwords of scratch are tolerated: constants of the bucket-overflow path -- the key of a
    # record that found its final bucket full -- held across the tile loop; that path runs for heavy hitters only)
    check(r"k_scatter_l2ILi0ELi1024ELb[01]E", 128, 32, at_least=2)
    # the region build and the KFREQ block build: 32 waves per CU
    check(r"k_build_regionsILi0E", 64, 0, at_least=8)
    check(r"k_kf_build_blocksI", 64, 0, at_least=4)
    # level 1, 1024 threads per workgroup: 128 VGPRs is all there is; the forms the bench workloads take must not spill
    check(r"k_extract_scatter_uniformILi0ELi[0-3]ELi15ELb0ELb0ELb1ELb[01]E", 128, 0, at_least=6)   # regular tiles (cfg2, cfg3)
    check(r"k_extract_scatter_uniformILi0ELi3ELi1[56]ELb0ELb0ELb0ELb0E", 128, 0, at_least=2)       # KFREQ, direct blocks (cfg4)
    check(r"k_extract_scatter_prefixILi[0-2]ELi1[56]ELb[01]E", 128, 0, at_least=12)                 # mixed lengths (cfg2t): pipelined (Lb1) and plain
    check(r"k_extract_scatter_uniformILi0ELi[0-2]ELi1[56]ELb[01]ELb0ELb0ELb0ELb1ELb1E", 128, 0, at_least=12)  # equal / mostly equal lengths, any lane count, pipelined
    check(r"k_extract_scatter_uniformILi0ELi[0-2]ELi(8|12)ELb[01]ELb1ELb0ELb0ELb1ELb1E", 128, 0, at_least=12)  # the linear form, pipelined (every rank of a job on >= 4 GPUs)
    check(r"k_wide_scatter_l1_uniformILi[0-2]E", 128, 0, at_least=3)                               # k <= 63 (cfg5)
    # every level-1 form at all: never beyond the file, and no large spills
    check(r"k_extract_scatter(_uniform|_prefix|_lin)?I", 128, 64, at_least=40)
