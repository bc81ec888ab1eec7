"""VGPRs / spills / occupancy of every kernel of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
Usage: python tools/kernel_resources.py waveome_amd/csrc/wv_gemm.hip [extra hipcc flags]"""
import re
import subprocess
import sys

src, extra = sys.argv[1], sys.argv[2:]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Rpass-analysis=kernel-resource-usage",
       "-c", src, "-o", "/dev/null"] + extra
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in err.splitlines():
    m = re.search(r"remark: .*?Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark: .*?\s(VGPRs|AGPRs|VGPRs Spill|SGPRs Spill|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|ScratchSize \[bytes/lane\]): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1)] = int(m.group(2))
dem = subprocess.run(["c++filt"], input="\n".join(rows), capture_output=True, text=True).stdout.splitlines()
print(f"{'VGPR':>5} {'AGPR':>5} {'spill':>5} {'scratch':>7} {'occ':>3} {'LDS':>6}  kernel")
for name, d in zip(dem, rows.values()):
    print(f"{d.get('VGPRs', 0):5d} {d.get('AGPRs', 0):5d} {d.get('VGPRs Spill', 0):5d} {d.get('ScratchSize [bytes/lane]', 0):7d} "
          f"{d.get('Occupancy [waves/SIMD]', 0):3d} {d.get('LDS Size [bytes/block]', 0):6d}  {name[:150]}")
