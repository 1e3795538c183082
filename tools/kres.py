#!/usr/bin/env python3
"""Per-kernel register / spill / LDS table of one csrc file as hipcc sees it:
    python tools/kres.py lstm_rec_h256_bf16 [-DLOB_X=1 ...] [--filter substr]"""
import os, re, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
defs = [a for a in sys.argv[2:] if a.startswith("-D")]
flt = sys.argv[sys.argv.index("--filter") + 1] if "--filter" in sys.argv else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", '-DLOB_BUILD_ID="kres"', "-I", f"{R}/include",
       "-I", f"{R}/lstm_ode_bci_amd/csrc", "-c", f"{R}/lstm_ode_bci_amd/csrc/{src}.hip", "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"] + defs
out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)\s*(\[[^\]]*\])?: (\d+) \[-Rpass", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(3))
names = subprocess.run(["/usr/bin/c++filt"], input="\n".join(rows), stdout=subprocess.PIPE, text=True).stdout.splitlines()
rows = {re.sub(r"\(anonymous namespace\)::", "", re.sub(r"^void ", "", n)).split("(")[0]: v for n, v in zip(names, rows.values())}
for k, v in rows.items():
    if flt in k:
        print(f"{k[:110]:110s} VGPR {v.get('VGPRs', -1):4d} AGPR {v.get('AGPRs', -1):4d} spill {v.get('VGPRs Spill', -1):3d} scratch {v.get('ScratchSize', -1):5d} LDS {v.get('LDS Size', -1):6d} occ {v.get('Occupancy', -1)}")
if "error" in out:
    print(out[-3000:])
