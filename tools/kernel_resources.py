"""Kernel resource table of the library: hipcc -Rpass-analysis=kernel-resource-usage over fumi_amd/csrc/*.hip (gfx950), one row per
kernel -- VGPRs, AGPRs, SGPRs, spills, scratch, occupancy, static LDS.   python tools/kernel_resources.py [out.txt] [file.hip ...]"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
FOR_SIZE = {"episode.hip", "hyper.hip", "linhead.hip", "api.hip", "sampler.hip", "glove.hip", "adam.hip"}


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return p.stdout.splitlines()


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else None
    files = sys.argv[2:] or sorted(glob.glob(os.path.join(ROOT, "fumi_amd", "csrc", "*.hip")))
    rows = []
    for f in files:
        opt = "-Os" if os.path.basename(f) in FOR_SIZE else "-O3"             # as __graft_entry__.build() compiles them
        p = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", opt, "-std=c++17", "-fPIC", "-c", f, "-o", "/dev/null",
                            "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True, cwd="/tmp")
        cur = None
        for line in p.stderr.splitlines():
            m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
            if not m:
                continue
            k, v = m.group(1), m.group(2)
            if k == "Function Name":
                cur = {"name": v, "file": os.path.basename(f)}
                rows.append(cur)
            elif cur is not None:
                cur[k] = v
    names = demangle([r["name"] for r in rows])
    lines = ["kernel resource usage (hipcc -Rpass-analysis=kernel-resource-usage, gfx950) of every kernel in fumi_amd/csrc",
             "file | kernel | VGPRs | AGPRs | SGPRs | SGPR spills | VGPR spills | scratch B/lane | waves/SIMD | static LDS B", "---"]
    for r, n in zip(rows, names):
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"^void ", "", n).split("(")[0]
        lines.append(" | ".join([r["file"], n, r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("TotalSGPRs", "?"), r.get("SGPRs Spill", "?"),
                                 r.get("VGPRs Spill", "?"), r.get("ScratchSize [bytes/lane]", "?"), r.get("Occupancy [waves/SIMD]", "?"),
                                 r.get("LDS Size [bytes/block]", "?")]))
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    else:
        sys.stdout.write(text)


if __name__ == "__main__":
    main()
