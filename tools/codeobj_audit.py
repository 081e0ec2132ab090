"""Register / spill audit of the gfx950 code objects inside a built shared library.

Reads the `.hip_fatbin` section, splits the clang offload bundles, and parses the AMDGPU metadata note
(`llvm-readelf --notes`) of every amdgcn code object: per kernel its VGPR / AGPR / SGPR counts, spill
counts, private (scratch) segment and LDS size.  No GPU needed.

    python3 tools/codeobj_audit.py [gogp_amd/libgogp_hip.so]      # table, worst first
    python3 tools/codeobj_audit.py --check [lib.so]              # exit 1 if a kernel breaks a limit

The limits (no VGPR spills, no scratch, <= 256 VGPRs, no AGPRs, bounded SGPR spills; commented allow-lists
below) are checked at build time (`make audit`, __graft_entry__.build()) and by tests/test_codeobj_audit.py.
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


SGPR_SPILL_LIMIT = 48
#: kernel-name regex -> allowed SGPR spill count, with the reason
SGPR_SPILL_ALLOW = {
    # the chain's diagonal-block kernel with potrf128_lds inlined: measured 7 % faster than the out-of-line
    # call hipcc chooses by itself (313,996 vs 336,912 cycles per 256-block), which also needs a 20-byte
    # private segment for the callee-saved VGPRs.  Round 4: 456 -> 140 (the 16x16 inverses read their coefficients
    # as LDS broadcasts instead of 240 v_readlane results that hipcc hoisted and spilled); what is left are the
    # sixteen lane == j masks of panel16 (hoisted out of the column-block loop) and kernel arguments.  Round 5: 140 -> 60
    # (the pivot pass broadcasts with DPP, pivot16.h: no v_readlane results to hoist); what is left are kernel arguments and
    # the products' loop state
    r"diag256_kernel<true, false, \d+>": 64,
    # one 128-half of the block on its own (option chain_split = 1): the same potrf128_lds / inv16 code (round 5: 91 -> 7)
    r"diag128_kernel": 48,
    # cold path: only gogp_set_factor (restore of stored results) inverts blocks of an existing factor (round 4: 248 -> 6)
    r"diag256_kernel<false, false, \d+>": 48,
    # multi-term / periodic kernels keep the per-pair loop: kind, scale, period and length tables of up to
    # four terms stay live across it.  Not on any BASELINE configuration (those are single radial terms)
    r"grad_reduce_kernel<\d+, (true|false), (double|float), false>": 80,
    # gradient w.r.t. the inputs (full Observe form, the anynoise / warpedtime case studies): N <= a few
    # hundred in the reference; 32 per-dimension accumulators
    r"xgrad_kernel<32>": 100,
}
#: kernels that may use AGPRs: none.  (Round 3 allowed grad_ard_mfma_kernel 64 of them "as MFMA accumulators"; with
#: __launch_bounds__(256, 2) the compiler keeps every instance inside 226 architectural VGPRs and uses no AGPR at
#: all, so the one class of register the removed gradient instances went wrong through is simply absent.)
AGPR_ALLOW = {}


def sgpr_spill_limit(name):
    limit = SGPR_SPILL_LIMIT
    for pat, allowed in SGPR_SPILL_ALLOW.items():
        if re.search(pat, name):
            limit = allowed
    return limit


def agprs_allowed(name):
    for pat, allowed in AGPR_ALLOW.items():
        if re.search(pat, name):
            return allowed
    return 0


def violations(ks):
    """[(kernel name, what)] for every kernel of `ks` (from kernels()) that breaks a limit of the product library."""
    bad = []
    for k in ks:
        n = k["name"]
        if k.get("vgpr_spill_count", 0):
            bad.append((n, "%d VGPR spills" % k["vgpr_spill_count"]))
        if k.get("private_segment_fixed_size", 0):
            bad.append((n, "%d bytes of scratch" % k["private_segment_fixed_size"]))
        if k["vgpr_count"] > 256:
            bad.append((n, "%d VGPRs" % k["vgpr_count"]))
        if k.get("agpr_count", 0) > agprs_allowed(n):
            bad.append((n, "%d AGPRs" % k["agpr_count"]))
        if k.get("sgpr_spill_count", 0) > sgpr_spill_limit(n):
            bad.append((n, "%d SGPR spills (limit %d)" % (k["sgpr_spill_count"], sgpr_spill_limit(n))))
    return bad


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
    args = [a for a in sys.argv[1:] if a != "--check"]
    so = args[0] if args else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                           "gogp_amd", "libgogp_hip.so")
    ks = kernels(so)
    if "--check" in sys.argv[1:]:
        bad = violations(ks)
        for n, what in bad:
            print("codeobj_audit: %s: %s" % (n[:120], what), file=sys.stderr)
        print("codeobj_audit: %d kernels in %s, %d over a limit" % (len(ks), os.path.basename(so), len(bad)))
        sys.exit(1 if bad or len(ks) < 50 else 0)
    print("%d kernels in %s" % (len(ks), so))
    print("%5s %5s %5s %7s %7s %8s %7s  %s" % ("vgpr", "agpr", "sgpr", "vspill", "sspill", "scratch", "lds", "kernel"))
    for k in sorted(ks, key=lambda k: (-k.get("vgpr_spill_count", 0), -k.get("sgpr_spill_count", 0), -k["vgpr_count"])):
        print("%5d %5d %5d %7d %7d %8d %7d  %s" % (k["vgpr_count"], k.get("agpr_count", 0), k["sgpr_count"],
                                                  k.get("vgpr_spill_count", 0), k.get("sgpr_spill_count", 0),
                                                  k.get("private_segment_fixed_size", 0),
                                                  k.get("group_segment_fixed_size", 0), k["name"][:110]))


if __name__ == "__main__":
    main()
