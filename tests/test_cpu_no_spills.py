"""Every gfx950 kernel in the built library runs without scratch memory: a register spill in a latency-bound kernel is a memory
round trip per spilled value (the LM's linear kernels lost 13 % of a decode step to 28-80 bytes of spills before this was checked).
Reads the kernel descriptors' metadata (.private_segment_fixed_size, .vgpr_spill_count) out of the code objects bundled in
liblds.so with the ROCm LLVM tools; skipped where those tools are not installed."""
import os
import re
import shutil
import subprocess

import pytest

from conftest import ROOT

LLVM = "/opt/rocm/lib/llvm/bin"
LIB = os.path.join(ROOT, "latent-diffusion-speech_amd", "lds", "liblds.so")


@pytest.mark.skipif(not (os.path.exists(os.path.join(LLVM, "llvm-objdump")) and os.path.exists(os.path.join(LLVM, "llvm-readelf"))),
                    reason="ROCm LLVM tools not installed")
def test_no_kernel_uses_scratch(tmp_path):
    assert os.path.exists(LIB), "build the library first (__graft_entry__.build())"
    lib = tmp_path / "liblds.so"
    shutil.copy(LIB, lib)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", str(lib)], check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=tmp_path)
    objs = [f for f in os.listdir(tmp_path) if "gfx950" in f]      # the bundles are extracted next to the input
    assert objs, "no gfx950 code object found in the library"
    kernels, bad = 0, []
    for f in objs:
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", str(tmp_path / f)], check=True, stdout=subprocess.PIPE, text=True).stdout
        for blk in notes.split(".agpr_count")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk)
            priv = re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk)
            spill = re.search(r"\.vgpr_spill_count:\s+(\d+)", blk)
            if not (name and priv):
                continue
            kernels += 1
            if int(priv.group(1)) != 0 or (spill and int(spill.group(1)) != 0):
                bad.append((name.group(1), int(priv.group(1)), int(spill.group(1)) if spill else None))
    assert kernels > 100, kernels
    assert not bad, f"kernels with scratch / spills: {bad}"
