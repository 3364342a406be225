"""Register / scratch / occupancy table of the library's kernels from hipcc -Rpass-analysis=kernel-resource-usage (device code only).

    python tools/kernel_resources.py [substring ...]      (no argument: the kernels of the C2 / C3 training step)

The "VGPR" column of a rocprofv3 summary is NOT the allocation; this is.  Occupancy cliffs matter here: tc2_bwd_kernel's C2
instance sat at exactly 128 VGPRs (two workgroups per CU) and one unrelated edit moved it to 129 (one per CU, + 5 us per launch, pass r5-02)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT = ["tc2_bwd_kernelILi64ELi3ELb1ELi0E", "tc2_ln_fwd_kernelILi64ELi3ELi4ELi4E", "tc2_ln_fwd_kernelILi64ELi3ELi6ELi4E", "tc1_fwd_kernelILi64ELi64ELi3ELi0E",
           "tc1_bwd_kernelILi64ELi64ELi3ELi0E", "gconv_fwd_kernelILi1ELi16E", "gconv_fwd_b16p_kernelILi2", "gconv_bwd2_kernelILi1E", "head_fwd_kernelILi2ELi4E",
           "head_fwd_kernelILi4ELi4E", "tconv_fwd4_kernelILi2ELi4ELb1E", "wgrad_pair_kernelILi4ELi4ELi4ELi2E", "fc_bwd_kernelILi1ELi2E", "reduce_kernel",
           "pack_kernel", "thin_tc1_fwd_kernel", "thin_tc1_bwd2_kernel"]


def main():
    pats = sys.argv[1:] or DEFAULT
    with tempfile.TemporaryDirectory() as tmp:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-c", '-DSTGCN_BACKEND_NAME="hip-gfx950"',
               "-Rpass-analysis=kernel-resource-usage", os.path.join(ROOT, "stgcn_amd", "csrc", "stgcn_capi.hip"), "-o", os.path.join(tmp, "dev.o")]
        err = subprocess.run(cmd, capture_output=True, text=True).stderr
    cur, rows = None, {}
    for line in err.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            rows[cur] = {}
            continue
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur:
                rows[cur][key] = int(m.group(1))
    print("| kernel (mangled) | VGPRs | AGPRs | scratch B/lane | waves/SIMD |")
    print("|---|---|---|---|---|")
    for name, r in sorted(rows.items()):
        if any(p in name for p in pats):
            print(f"| {name[9:90]} | {r.get('vgpr')} | {r.get('agpr')} | {r.get('scratch')} | {r.get('occ')} |")


if __name__ == "__main__":
    main()
