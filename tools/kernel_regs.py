#!/usr/bin/env python3
"""Register / spill / LDS census of every kernel in the built library (code-object metadata).
usage: tools/kernel_regs.py [name-substring]     (TF_LIBDIR=<directory holding the .o files> for another build)"""
import os, re, subprocess, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
LIBDIR = os.environ.get("TF_LIBDIR", os.path.join(HERE, "..", "tinyfusers_amd", "lib"))
LLVM = "/opt/rocm/lib/llvm/bin"
txt = ""
with tempfile.TemporaryDirectory() as t:
    for o in sorted(f for f in os.listdir(LIBDIR) if f.endswith(".o")):     # one device code object per translation unit
        fat, co = os.path.join(t, "fat.bin"), os.path.join(t, "dev.co")
        if subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", os.path.join(LIBDIR, o), os.path.join(t, "x")],
                          capture_output=True).returncode:
            continue                     # no device code in this object
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
        txt += subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
pat = sys.argv[1] if len(sys.argv) > 1 else ""
rows = []
for blk in txt.split("- .agpr_count")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = g("name")
    if pat and pat not in name:
        continue
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
    rows.append(f"{dem[:84]:84s} vgpr {g('vgpr_count'):>4s} sgpr {g('sgpr_count'):>4s} vspill {g('vgpr_spill_count'):>3s} sspill {g('sgpr_spill_count'):>3s} scratch {g('private_segment_fixed_size'):>5s}")
print("\n".join(sorted(rows)))
