"""VGPR / SGPR / scratch / occupancy of every kernel in a .hip file (hipcc -Rpass-analysis=kernel-resource-usage; no GPU needed).
usage: python tools/kernel_resources.py nonlinearsolve.jl_amd/csrc/nk_sstep.hip [substring filter]"""
import os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "nonlinearsolve.jl_amd", "csrc"), "-ffp-contract=off", "-c", src, "-o", "/tmp/_kr.o",
           "-Rpass-analysis=kernel-resource-usage"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur = None
    rows = {}
    for ln in err.splitlines():
        m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill): (\S+)", ln)
        if not m:
            continue
        if m.group(1) == "Function Name":
            cur = m.group(2)
            rows[cur] = {}
        elif cur:
            rows[cur][m.group(1).replace("s Spill", "spill").split(" ")[0]] = m.group(2)
    dem = subprocess.run(["c++filt"], input="\n".join(rows), capture_output=True, text=True).stdout.splitlines()
    print(f"{'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'sspill':>6} {'vspill':>6} {'scratch':>8} {'occ':>4}  kernel")
    for name, dn in zip(rows, dem):
        r = rows[name]
        short = re.sub(r"\(.*", "", dn)
        if filt and filt not in short:
            continue
        print(f"{r.get('VGPRs','?'):>5} {r.get('AGPRs','?'):>5} {r.get('TotalSGPRs','?'):>5} {r.get('SGPRspill','?'):>6} {r.get('VGPRspill','?'):>6} "
              f"{r.get('ScratchSize','?'):>8} {r.get('Occupancy','?'):>4}  {short}")


if __name__ == "__main__":
    main()
