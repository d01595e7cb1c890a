"""CPU (compile only): no kernel of libfos_hip.so may spill registers or use scratch.  A menu geometry that starts
spilling after an unrelated edit silently halves the bandwidth of that configuration (it happened to the bf16
default in round 1: 84 % -> 46 % of the roofline), so this is checked at build level."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_library_kernels_do_not_spill(tmp_path):
    from concurrent.futures import ThreadPoolExecutor
    from fastoptsolver_amd import build
    with ThreadPoolExecutor(max_workers=len(build.UNITS)) as pool:          # the four translation units of the library
        outs = list(pool.map(lambda u: build.compile_unit(u, extra=["-Rpass-analysis=kernel-resource-usage"],
                                                          out=str(tmp_path / (u + ".o")), verbose=False).stderr, build.UNITS))
    out = "\n".join(outs)
    kernels, cur = {}, None
    for line in out.splitlines():
        m = re.search(r"remark:\s+Function Name:\s+(\S+)", line)
        if m:
            cur = m.group(1)
            kernels[cur] = {}
            continue
        m = re.search(r"remark:\s+(VGPRs Spill|SGPRs Spill|ScratchSize \[bytes/lane\]|VGPRs):\s+(\d+)", line)
        if m and cur:
            kernels[cur][m.group(1)] = int(m.group(2))
    assert len(kernels) > 40, "resource-usage remarks not found"
    # The one exception: the opt-in persistent step at n = 8192 (fused_step.hpp, NQ = 4) keeps y as matrix-instruction
    # operands (64 VGPRs), a panel in flight (64) and its staging addresses live at once and spills a handful of registers
    # (measured with them: 80 % of the roofline; the default two-launch step is the product path).  Bounded, not waived.
    allowed = {k for k in kernels if "fista_fused_kernelILi4E" in k and kernels[k].get("VGPRs Spill", 0) <= 8}
    bad = {k: v for k, v in kernels.items()
           if (v.get("VGPRs Spill", 0) or v.get("ScratchSize [bytes/lane]", 0)) and k not in allowed}
    assert not bad, f"kernels that spill / use scratch: {bad}"
    assert any("gemv_pair_kernel" in k for k in kernels) and any("residual_batch_mfma" in k for k in kernels)
