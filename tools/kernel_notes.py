#!/usr/bin/env python3
"""register / spill / scratch / LDS figures of every gfx950 kernel in libpgv_hip.so (llvm-readelf --notes of the
embedded code objects), optionally filtered by a substring:  python tools/kernel_notes.py [substr]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def notes(lib):
    out = []
    with tempfile.TemporaryDirectory() as d:
        # the fat binary: pull the gfx950 code objects out of .hip_fatbin
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--list", "--type=o", "--input=" + lib],
                       capture_output=True)
        fat = os.path.join(d, "fat.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
        data = open(fat, "rb").read()
        # bundles are concatenated: each starts with the magic
        magic = b"__CLANG_OFFLOAD_BUNDLE__"
        starts = [m.start() for m in re.finditer(re.escape(magic), data)]
        for i, st in enumerate(starts):
            piece = os.path.join(d, "bundle%d" % i)
            open(piece, "wb").write(data[st:starts[i + 1] if i + 1 < len(starts) else len(data)])
            co = os.path.join(d, "co%d" % i)
            r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + piece,
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], capture_output=True)
            if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
                continue
            txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
            for blk in txt.split("- .agpr_count:")[1:]:
                def f(key):
                    m = re.search(r"\.%s:\s*(\S+)" % key, blk)
                    return m.group(1) if m else "?"
                name = subprocess.run(["c++filt", f("name")], capture_output=True, text=True).stdout.strip()
                out.append({"name": name, "agpr": blk.split()[0], "vgpr": f("vgpr_count"), "sgpr": f("sgpr_count"),
                            "vgpr_spill": f("vgpr_spill_count"), "sgpr_spill": f("sgpr_spill_count"),
                            "scratch": f("private_segment_fixed_size"), "lds": f("group_segment_fixed_size")})
    return out


if __name__ == "__main__":
    sub = sys.argv[1] if len(sys.argv) > 1 else ""
    lib = os.path.join(ROOT, "pgvector_amd", "lib", "libpgv_hip.so")
    print("%-100s %5s %5s %5s %7s %7s %8s %7s" % ("kernel", "vgpr", "agpr", "sgpr", "v-spill", "s-spill", "scratch", "lds"))
    for k in notes(lib):
        if sub in k["name"]:
            print("%-100s %5s %5s %5s %7s %7s %8s %7s" % (k["name"][:100], k["vgpr"], k["agpr"], k["sgpr"], k["vgpr_spill"],
                                                        k["sgpr_spill"], k["scratch"], k["lds"]))
