"""Register / spill audit of the gfx950 code objects inside a built shared library.

Reads the `.hip_fatbin` section, splits the clang offload bundles, and parses the AMDGPU metadata note
(`llvm-readelf --notes`) of every amdgcn code object: per kernel its VGPR / AGPR / SGPR counts, spill
counts, private (scratch) segment and LDS size.  No GPU needed.

    python3 tools/codeobj_audit.py [gogp_amd/libgogp_hip.so]      # table, worst first

tests/test_codeobj_audit.py asserts the limits (no VGPR spills, <= 256 VGPRs, no scratch) with a
commented allow-list.
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count",
          "private_segment_fixed_size", "group_segment_fixed_size", "max_flat_workgroup_size")


def code_objects(so_path):
    """The amdgcn ELF images bundled in so_path (bytes objects)."""
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, so_path])
        data = open(fat, "rb").read()
    out = []
    for m in re.finditer(MAGIC, data):
        o = m.start()
        p = o + len(MAGIC)
        (n,) = struct.unpack_from("<Q", data, p)
        p += 8
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, p)
            p += 24
            triple = data[p:p + tl].decode()
            p += tl
            if "amdgcn" in triple and size > 0:
                out.append(data[o + off:o + off + size])
    return out


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    d = r.stdout.splitlines()
    return d if len(d) == len(names) else names


def kernels(so_path):
    """[{name, symbol, vgpr_count, ...}] for every kernel of every gfx950 code object in so_path."""
    ks = []
    for img in code_objects(so_path):
        with tempfile.NamedTemporaryFile(suffix=".elf") as f:
            f.write(img)
            f.flush()
            txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f.name], capture_output=True,
                                 text=True, check=True).stdout
        cur = None
        for line in txt.splitlines():
            s = line.strip()
            if s.startswith("- .agpr_count:") or (s.startswith("- .") and line.startswith("  - ")):
                cur = {}
                ks.append(cur)
                s = s[2:]
            if cur is None:
                continue
            m = re.match(r"\.(\w+):\s+(.*)$", s)
            if not m or not line.startswith("    ." if not line.startswith("  - ") else "  - "):
                continue
            key, val = m.group(1), m.group(2).strip()
            if key in FIELDS:
                cur[key] = int(val)
            elif key == "name" and "symbol" not in cur and "name" not in cur:
                cur["symbol"] = val.strip("'")
    ks = [k for k in ks if "symbol" in k and "vgpr_count" in k]
    for k, d in zip(ks, demangle([k["symbol"] for k in ks])):
        k["name"] = re.sub(r"\(.*$", "", d.replace("void ", ""))
    return ks


def main():
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                            "gogp_amd", "libgogp_hip.so")
    ks = kernels(so)
    print("%d kernels in %s" % (len(ks), so))
    print("%5s %5s %5s %7s %7s %8s %7s  %s" % ("vgpr", "agpr", "sgpr", "vspill", "sspill", "scratch", "lds", "kernel"))
    for k in sorted(ks, key=lambda k: (-k.get("vgpr_spill_count", 0), -k.get("sgpr_spill_count", 0), -k["vgpr_count"])):
        print("%5d %5d %5d %7d %7d %8d %7d  %s" % (k["vgpr_count"], k.get("agpr_count", 0), k["sgpr_count"],
                                                  k.get("vgpr_spill_count", 0), k.get("sgpr_spill_count", 0),
                                                  k.get("private_segment_fixed_size", 0),
                                                  k.get("group_segment_fixed_size", 0), k["name"][:110]))


if __name__ == "__main__":
    main()
