"""Per-kernel register / scratch / LDS figures of the built objects, read from the code objects' metadata notes.

    python tools/kernel_resources.py [pattern]         # table of every kernel whose name contains `pattern`
    python tools/kernel_resources.py --check           # exit 1 if a kernel that counts on its VMEM queue holds scratch

Why --check exists: the RED epilogue of gemm_glds_pair_kernel and the software-pipelined bf16x3 kernel
wait with COUNTED `s_waitcnt vmcnt(n)` for inline-asm loads the compiler cannot see.  A scratch spill or reload that a
later compiler places between those loads and the wait would shift the count and let values be consumed before they
land.  __graft_entry__.build() runs this check, so such a build fails instead of computing wrong sums silently."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "curvecloudnet_amd", "csrc")


def _llvm_bin():
    """The ROCm LLVM tool directory: $ROCM_PATH, `hipconfig --rocmpath`, then /opt/rocm (not one hard-coded path)."""
    roots = [os.environ.get("ROCM_PATH")]
    try:
        roots.append(subprocess.run(["hipconfig", "--rocmpath"], capture_output=True, text=True).stdout.strip())
    except OSError:
        pass
    roots.append("/opt/rocm")
    for r in roots:
        if r and os.path.exists(os.path.join(r, "lib", "llvm", "bin", "llvm-objdump")):
            return os.path.join(r, "lib", "llvm", "bin")
    raise SystemExit("kernel_resources: no llvm-objdump under $ROCM_PATH / `hipconfig --rocmpath` / /opt/rocm")


LLVM = _llvm_bin()
# kernels whose correctness depends on nothing unseen entering the VMEM queue: no scratch, no spills
# (the weight-gradient kernels gemm_tn_glds_kernel<64, 64> / gemm_h_tn_kernel hold ~250 B of scratch since round 2 / 3; they
# wait vmcnt(0) only, which a spill cannot defeat)
NO_SCRATCH = ("gemm_glds_pair_kernel", "gemm_glds_persistent_kernel", "gemm_glds_kernel", "gemm_x3_lean_kernel",
              "gemm_x3_persistent_kernel", "gemm_h_pair_kernel")     # (gemm_x3_pair_kernel: replaced by the lean kernel in round 4)


def kernels_of(obj):
    """[(name, {field: int})] of the gfx950 code object bundled in `obj`."""
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, os.path.basename(obj))
        os.symlink(obj, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], cwd=tmp, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        cos = sorted(f for f in os.listdir(tmp) if "amdgcn" in f)
        if not cos:
            return None                                    # (no device code object extracted: the caller decides what that means)
        # every extracted code object (one per offload architecture)
        notes = "".join(subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, co)],
                                       check=True, capture_output=True, text=True).stdout for co in cos)
    out = []
    for blk in re.split(r"\n  - \.agpr_count:", notes)[1:]:
        blk = ".agpr_count:" + blk
        name = re.search(r"\.name:\s+(\S+)", blk)
        if not name:
            continue
        f = {}
        for key in ("agpr_count", "vgpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size",
                    "vgpr_spill_count", "sgpr_spill_count"):
            m = re.search(r"\.%s:\s+(\d+)" % key, blk)
            f[key] = int(m.group(1)) if m else -1
        out.append((name.group(1), f))
    return out


def demangle(names):
    for tool in ("c++filt", os.path.join(LLVM, "llvm-cxxfilt")):
        try:
            p = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True)
        except FileNotFoundError:
            continue
        if p.returncode == 0:
            return p.stdout.splitlines()
    return names


def main(argv):
    check = "--check" in argv
    pat = next((a for a in argv if not a.startswith("--")), "")
    bad, blind = [], []
    seen = {k: 0 for k in NO_SCRATCH}
    objs = [fn for fn in sorted(os.listdir(CSRC)) if fn.endswith(".o")]
    if check and not objs:
        blind.append("no object files under %s" % CSRC)
    for fn in objs:
        ks = kernels_of(os.path.join(CSRC, fn))
        if ks is None:
            blind.append("no device code object extracted from %s" % fn)
            continue
        names = demangle([k for k, _ in ks])
        for (_, f), name in zip(ks, names):
            short = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0].replace("void ", "")
            held = f["private_segment_fixed_size"] > 0 or f["vgpr_spill_count"] > 0
            if check:
                for k in NO_SCRATCH:
                    if k in short:
                        seen[k] += 1
                if held and any(k in short for k in NO_SCRATCH):
                    bad.append((fn, short, f))
                continue
            if pat in short:
                print("%-18s %-70s vgpr %3d agpr %3d sgpr %3d lds %6d scratch %4d spills %d" % (
                    fn, short[:70], f["vgpr_count"], f["agpr_count"], f["sgpr_count"], f["group_segment_fixed_size"],
                    f["private_segment_fixed_size"], f["vgpr_spill_count"]))
    if check:
        for fn, short, f in bad:
            print("SCRATCH in a counted-wait kernel: %s %s: %d B scratch, %d spilled VGPRs" % (
                fn, short, f["private_segment_fixed_size"], f["vgpr_spill_count"]))
        # a check that looked at nothing must not pass: a renamed kernel or a changed toolchain layout fails here
        blind += ["no built kernel matches the guarded name %r" % k for k, n in seen.items() if n == 0]
        for msg in blind:
            print("kernel_resources --check cannot vouch for the build: " + msg)
        return 1 if bad or blind else 0
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
