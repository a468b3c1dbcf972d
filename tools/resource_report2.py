"""Compiler resource report of every kernel of the library, one line per kernel: python tools/resource_report2.py [substring] [extra hipcc flags...]"""
import re, subprocess, sys
pat = sys.argv[1] if len(sys.argv) > 1 else ""
extra = sys.argv[2:]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Rpass-analysis=kernel-resource-usage",
       "-o", "/tmp/libfk_ru.so", "/root/repo/farkle_ii_amd/csrc/farkle_hip.hip", *extra]
out = subprocess.run(cmd, capture_output=True, text=True, cwd="/tmp").stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: (?:[^ ]+ )?\s*Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark: (?:[^ ]+ )?\s*(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).split(" [")[0]] = int(m.group(2))
for name, r in rows.items():
    short = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name).replace("EvNS_8PlayArgsE", "")
    if pat in short:
        print(f"{short:70s} VGPRs {r.get('VGPRs'):4d} scratch {r.get('ScratchSize'):4d} occupancy {r.get('Occupancy')} sgpr-spill {r.get('SGPRs Spill'):3d} vgpr-spill {r.get('VGPRs Spill')}")
