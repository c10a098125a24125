#!/usr/bin/env python3
"""Register / spill / LDS census of every kernel in the built library (code-object metadata).
usage: tools/kernel_regs.py [name-substring]     (TF_LIBDIR=<directory holding the .o files> for another build)"""
import os, re, subprocess, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
LIBDIR = os.environ.get("TF_LIBDIR", os.path.join(HERE, "..", "tinyfusers_amd", "lib"))
LLVM = "/opt/rocm/lib/llvm/bin"


def census(libdir=LIBDIR):
    """One dict per kernel of the objects in `libdir`: demangled name, vgpr, sgpr, vspill, sspill, scratch (ints)."""
    txt = ""
    with tempfile.TemporaryDirectory() as t:
        for o in sorted(f for f in os.listdir(libdir) if f.endswith(".o")):     # one device code object per translation unit
            fat, co = os.path.join(t, "fat.bin"), os.path.join(t, "dev.co")
            if subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", os.path.join(libdir, o), os.path.join(t, "x")],
                              capture_output=True).returncode:
                continue                     # no device code in this object
            subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
            txt += subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
    blocks = txt.split("- .agpr_count")[1:]
    g = lambda blk, k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    names = [g(b, "name") for b in blocks]
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    num = lambda v: int(v) if v.isdigit() else -1
    return [dict(name=dem[i].strip() or names[i], mangled=names[i], vgpr=num(g(b, "vgpr_count")), sgpr=num(g(b, "sgpr_count")), vspill=num(g(b, "vgpr_spill_count")),
                 sspill=num(g(b, "sgpr_spill_count")), scratch=num(g(b, "private_segment_fixed_size"))) for i, b in enumerate(blocks)]


if __name__ == "__main__":
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    rows = [f"{k['name'][:84]:84s} vgpr {k['vgpr']:4d} sgpr {k['sgpr']:4d} vspill {k['vspill']:3d} sspill {k['sspill']:3d} scratch {k['scratch']:5d}"
            for k in census() if pat in k["name"]]
    print("\n".join(sorted(rows)))
