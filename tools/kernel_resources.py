"""Per-kernel register / scratch / LDS use of the built library (from the code object's metadata notes).

    python tools/kernel_resources.py [pattern ...]        # substrings of the demangled kernel names

Needs no GPU: reads the objects of the last build (csrc/_obj/*.o) with the ROCm llvm tools.  A kernel that spills shows
scratch > 0 -- the thing to look at after touching a hot kernel.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def _notes_of(obj, td):
    # the device code object sits in .hip_fatbin of the host object: dump that section, then unbundle
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), f"--dump-section=.hip_fatbin={td}/fat.bin", obj,
                           f"{td}/unused.o"], stderr=subprocess.DEVNULL)
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--unbundle",
                           f"--input={td}/fat.bin", f"--output={td}/dev.co",
                           "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"])
    return subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", f"{td}/dev.co"], text=True)


def resources(objs):
    out = []
    with tempfile.TemporaryDirectory() as td:
        for obj in objs:
            try:
                notes = _notes_of(obj, td)
            except subprocess.CalledProcessError:      # a translation unit without device code
                continue
            for blk in notes.split("- .agpr_count:")[1:]:
                g = lambda k: (re.search(rf"\.{k}:\s+(\S+)", blk) or [None, "?"])[1]
                name = re.search(r"\n\s+\.name:\s+(\S+)", blk.split(".args:")[-1] if False else blk)
                names = re.findall(r"\n    \.name:\s+(\S+)", blk)       # the kernel's own .name (4-space indent)
                name = names[-1] if names else "?"
                try:
                    name = subprocess.check_output(["c++filt", name], text=True).strip()
                except Exception:
                    pass
                out.append({"name": name, "vgpr": g("vgpr_count"), "agpr": blk.split("\n", 1)[0].strip(),
                            "sgpr": g("sgpr_count"), "scratch": g("private_segment_fixed_size"),
                            "lds": g("group_segment_fixed_size"), "tu": os.path.basename(obj)})
    return out


def main():
    variant = os.path.splitext(os.environ.get("PP_HIP_LIB", "libpp_hip.so"))[0]
    objdir = os.path.join(ROOT, "3d-object-detection-for-autonomous-navigation_amd", "csrc",
                          "_obj" if variant == "libpp_hip" else "_obj_" + variant)
    objs = sorted(os.path.join(objdir, f) for f in os.listdir(objdir) if f.endswith(".o"))
    pats = sys.argv[1:]
    rows = [r for r in resources(objs) if not pats or any(p in r["name"] for p in pats)]
    rows.sort(key=lambda r: r["name"])
    print(f"{'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'scratch':>8} {'lds':>7}  kernel")
    for r in rows:
        print(f"{r['vgpr']:>5} {r['agpr']:>5} {r['sgpr']:>5} {r['scratch']:>8} {r['lds']:>7}  {r['name'][:110]}")


if __name__ == "__main__":
    main()
