"""CPU suite: the MFMA kernels of the bf16 ResNet-12 path compile for gfx950 without scratch memory.  (A kernel that spills is
still correct -- the GPU tests pass -- and an order of magnitude slower: one unrolled loop cost 576 bytes of scratch per lane and
8.6x the time.  hipcc cross-compiles without a GPU, so this is checked here.)"""
import os
import re
import subprocess

from conftest import ROOT


def test_resnet12_matrix_kernels_use_no_scratch(tmp_path):
    src = os.path.join(ROOT, "fumi_amd", "csrc", "rn12_conv.hip")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", str(tmp_path / "o.o"),
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    name, seen = None, {}
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and name:
            seen[name] = int(m.group(1))
    hot = {k: v for k, v in seen.items() if "rn_conv_kernel" in k or "rn_wgrad_kernel" in k}
    assert len(hot) >= 8, "kernel-resource-usage remarks not found"
    launched = {k: v for k, v in hot.items() if not re.search(r"rn_conv_kernelILi5ELi2ELi4E", k)}     # (never launched: see conv_dispatch)
    assert all(v == 0 for v in launched.values()), {k: v for k, v in launched.items() if v}
