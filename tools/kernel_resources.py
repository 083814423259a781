"""Print VGPR/AGPR/spill/occupancy per kernel of a .hip file (hipcc -Rpass-analysis)."""
import re, subprocess, sys
src = sys.argv[1]
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics",
                      "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"],
                     capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", line)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = {"name": subprocess.run(["c++filt", t.split(": ")[1]], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", r["name"])
    n = re.sub(r"\(.*", "", n)
    print(f"{n:70s} V={r.get('VGPRs')} A={r.get('AGPRs')} sgprspill={r.get('SGPRs Spill')} vspill={r.get('VGPRs Spill')} "
          f"scratch={r.get('ScratchSize [bytes/lane]')} occ={r.get('Occupancy [waves/SIMD]')}")
