"""Register / scratch / LDS use of every kernel of one source file (cross-compiled for gfx950, no GPU needed).
usage: python tools/resusage.py ndjir_amd/csrc/wgrad.hip [substring filter] [extra hipcc flags ...]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else ""
extra = [a for a in sys.argv[2:] if a.startswith("-")]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-fno-slp-vectorize",
       "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark: (?:\s*)([\w \[\]/]+): (.+?) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
print(f"{'kernel':80s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'spill':>6s} {'scratch':>8s} {'LDS':>7s} {'occ':>4s}")
for r in rows:
    if flt and flt not in r["name"]:
        continue
    name = re.sub(r"\(.*", "", r["name"])[:80]
    print(f"{name:80s} {r.get('VGPRs', '?'):>5s} {r.get('AGPRs', '?'):>5s} {r.get('SGPRs', '?'):>5s} {r.get('VGPRs Spill', '?'):>6s} "
          f"{r.get('ScratchSize [bytes/lane]', '?'):>8s} {r.get('LDS Size [bytes/block]', '?'):>7s} {r.get('Occupancy [waves/SIMD]', '?'):>4s}")
