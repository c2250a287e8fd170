"""Register / LDS / spill report of the product kernels (hipcc -Rpass-analysis=kernel-resource-usage; no GPU needed).
    python tools/kernel_resources.py [source.hip ...] [--filter substr]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gst_tacotron_amd", "csrc")
args = [a for a in sys.argv[1:] if not a.startswith("--")]
flt = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--filter=")]
srcs = args or ["skinny_gemm.hip", "dec_front.hip", "gemm_conv.hip", "attention.hip", "gst.hip", "audio.hip"]
for src in srcs:
    with tempfile.TemporaryDirectory() as td:
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-c",
                            os.path.join(CSRC, src), "-o", os.path.join(td, "o.o"), "-Rpass-analysis=kernel-resource-usage"],
                           capture_output=True, text=True)
    cur = None
    rows = {}
    for ln in r.stderr.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", ln)
        if m:
            cur = m.group(1); rows[cur] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\S+)", ln)
        if m and cur:
            rows[cur][m.group(1).strip()] = m.group(2)
    for name, v in rows.items():
        dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if flt and not any(f in dn for f in flt):
            continue
        print("%-100s VGPR %3s AGPR %3s spill %3s scratch %4s occ %2s LDS %6s" % (dn[:100], v.get("VGPRs"), v.get("AGPRs"), v.get("VGPRs Spill"),
              v.get("ScratchSize [bytes/lane]"), v.get("Occupancy [waves/SIMD]"), v.get("LDS Size [bytes/block]")))
