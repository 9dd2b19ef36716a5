"""Scan the gfx950 assembly of the inline-asm kernels for a hazard the compiler's recogniser cannot see inside inline asm:
a VALU write of an SGPR (v_readlane_b32 / v_readfirstlane_b32) needs FIVE wait states before a VMEM instruction may use that
SGPR as (part of) its address.  hipcc pads its own VMEM instructions with s_nop; an `asm volatile("global_load_dword ...")`
whose scalar base the compiler fetches from a VGPR lane right in front of the statement gets no padding and then reads the
previous value of the register pair (round 5: the tail-split fix-up took the previous register's row, now and then).

    python tools/asm_hazards.py            # exit 1 and a listing if any such pair is closer than five wait states
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "curvecloudnet_amd", "csrc")
FILES = ["ccn_gemm.hip", "ccn_gemm_tn.hip", "ccn_gemm_h.hip", "ccn_gemm_x3.hip"]
VMEM = re.compile(r"^\s*(global_|buffer_|flat_|scratch_)\w+\s+(.*)$")
SWRITE = re.compile(r"^\s*v_read(?:first)?lane_b32\s+s(\d+)\b")
SPAIR = re.compile(r"s\[(\d+):(\d+)\]")
NOP = re.compile(r"^\s*s_nop\s+(\d+)")


def scan(path):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-S", "--cuda-device-only", "-o", out,
                        path], check=True, stderr=subprocess.DEVNULL)
        lines = open(out).read().splitlines()
    bad = []
    kernel = "?"
    recent = []        # (sgpr index, wait states since the write)
    for ln in lines:
        if ln.startswith("_Z") and ln.rstrip().endswith(":") or (ln.startswith("_Z") and ": " in ln):
            kernel = ln.split(":")[0]
            recent = []
            continue
        text = ln.split(";")[0].rstrip()
        if not text.strip() or text.strip().startswith(".") or text.strip().endswith(":"):
            continue
        m = VMEM.match(text)
        if m:
            for a, b in SPAIR.findall(m.group(2)):
                for reg, age in recent:
                    if int(a) <= reg <= int(b) and age < 5:
                        bad.append((os.path.basename(path), kernel, text.strip(), reg, age))
        nop = NOP.match(text)
        step = 1 + int(nop.group(1)) if nop else 1
        recent = [(r, age + step) for r, age in recent if age + step < 8]
        w = SWRITE.match(text)
        if w:
            recent.append((int(w.group(1)), 0))
    return bad


def main():
    bad = []
    for fn in FILES:
        bad += scan(os.path.join(CSRC, fn))
    for fn, kernel, text, reg, age in bad:
        print("%s %s: `%s` uses s%d %d wait state(s) after a VALU wrote it" % (fn, kernel[:60], text, reg, age))
    print("%d hazard(s)" % len(bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
