"""The fused Jacobian launches must not touch scratch memory.  Round 5's PMC pass showed 2.8 MB of writes per line launch that nobody
could account for: the kernels' parameter structs had gone to scratch (a select between members of two structs compiles to a select of
their ADDRESSES, which pins both structs in memory — 48 bytes of scratch stores per lane at the top of every workgroup), and in round 6
a few lines of per-launch bookkeeping behind `blockIdx.x == 0` did the same to the point kernel (700 bytes per lane, 24 MB of writes per
launch, +8 us).  The compiler's own metadata says when it happens: private_segment_fixed_size of the two kernels, from the assembly of
csrc/jacobian_kernels.hip (cross-compiled for gfx950, no GPU needed)."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_fused_jacobian_kernels_use_no_scratch(tmp_path):
    src = os.path.join(ROOT, "pl-viwo_amd", "csrc", "jacobian_kernels.hip")
    out = str(tmp_path / "jacobian_kernels.s")
    # (the flags of pl-viwo_amd/Makefile)
    r = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-Wno-unused-function", "-Wno-unused-result",
                        "--cuda-device-only", "-S", "-o", out, src], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    text = open(out).read()
    found = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)", text):
        found[m.group(1)] = int(m.group(2))
    fused = {k: v for k, v in found.items() if "jacobian_nullspace_kernel" in k}
    assert len(fused) == 2, sorted(found)
    assert all(v == 0 for v in fused.values()), fused
    assert any("spec_select_kernel" in k for k in found)
