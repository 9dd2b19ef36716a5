"""ctypes binding of ``libccn_hip.so`` (the C-ABI in ``include/ccn_hip.h``).

The prototypes are read from the header itself, so the header is the single source of truth for
the boundary.  There is NO fallback: if the shared library is missing, or an op is called without
a GPU tensor, this raises.
"""
import ctypes
import os
import re

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
# (CCN_LIB_PATH: another build of the SAME library for one-box A/B runs of a kernel change -- tools/ only)
LIB_PATH = os.environ.get("CCN_LIB_PATH") or os.path.join(_PKG, "libccn_hip.so")
HEADER_PATH = os.path.join(_ROOT, "include", "ccn_hip.h")
DEBUG_HEADER_PATH = os.path.join(_ROOT, "include", "ccn_hip_debug.h")   # diagnostics / A-B / test hooks: not the boundary

_CTYPES = {
    "int": ctypes.c_int, "int64_t": ctypes.c_int64, "float": ctypes.c_float, "size_t": ctypes.c_size_t,
    "double": ctypes.c_double,
}


def parse_header(path=None):
    """-> {name: (restype, [argtypes])} for every function the header declares (default: the boundary header and the
    diagnostics header together)."""
    if path is None:
        protos = parse_header(HEADER_PATH)
        protos.update(parse_header(DEBUG_HEADER_PATH))
        return protos
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"^\s*#.*$", " ", text, flags=re.M)
    protos = {}
    for m in re.finditer(r"([\w\s\*]+?)\b(ccn_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if "*" in ret:
            restype = ctypes.c_char_p if "char" in ret else ctypes.c_void_p
        else:
            restype = _CTYPES[ret.replace("const", "").strip()]
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    argtypes.append(_CTYPES[a.replace("const", "").split()[0]])
        protos[name] = (restype, argtypes)
    return protos


def _header_abi_version():
    m = re.search(r"^#define\s+CCN_ABI_VERSION\s+(\d+)", open(HEADER_PATH).read(), flags=re.M)
    return int(m.group(1))


ABI_VERSION = _header_abi_version()

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "curvecloudnet_amd: %s is missing -- build it with `make -C curvecloudnet_amd/csrc` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback." % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in parse_header().items():
            if os.environ.get("CCN_LIB_PATH") and not hasattr(handle, name):
                continue                     # (an OLDER build of the library in a tools/ A/B run: entries added since are absent)
            fn = getattr(handle, name)       # AttributeError if the header declares a missing symbol
            fn.restype, fn.argtypes = restype, argtypes
        # (also under CCN_LIB_PATH: an A/B library of another ABI would be called with this header's argument lists -- ADVICE r5)
        if handle.ccn_abi_version() != ABI_VERSION:
            raise RuntimeError("libccn_hip.so ABI version mismatch (header: %d, library %s: %d)"
                               % (ABI_VERSION, LIB_PATH, handle.ccn_abi_version()))
        if os.environ.get("CCN_GEMM_DMA"):       # A/B hook of the GEMM dispatch (include/ccn_hip.h: ccn_gemm_use_dma)
            handle.ccn_gemm_use_dma(int(os.environ["CCN_GEMM_DMA"]))
        if os.environ.get("CCN_GEMM_PAIR_OPT"):  # A/B hook (ccn_gemm_pair_opt)
            handle.ccn_gemm_pair_opt(int(os.environ["CCN_GEMM_PAIR_OPT"]))
        if os.environ.get("CCN_WGRAD_BG"):       # experiment (ccn_gemm_tn_background): bytes of LDS per weight-gradient workgroup
            handle.ccn_gemm_tn_background(int(os.environ["CCN_WGRAD_BG"]))
        if os.environ.get("CCN_FPS_CLAIM"):      # A/B hook (ccn_fps_set_lds_claim)
            handle.ccn_fps_set_lds_claim(int(os.environ["CCN_FPS_CLAIM"]))
        if os.environ.get("CCN_FPS_CLUSTER") and hasattr(handle, "ccn_fps_use_cluster"):   # A/B hook (ccn_fps_use_cluster)
            handle.ccn_fps_use_cluster(int(os.environ["CCN_FPS_CLUSTER"]))
        _lib = handle
    return _lib


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def check(rc, what):
    if rc != 0:
        raise RuntimeError("libccn_hip %s failed (%d): %s" % (what, rc, lib().ccn_last_error().decode()))


PROFILE = None          # bench.py sets this to a list to time launches with HIP events on the launch stream
PROFILE_ONLY = None     # optional name prefix: only these entry points are timed (events cost ~1.5 us each on the GPU)
PROFILE_FILTER = None   # optional predicate(name, args): only the launches it accepts are timed


# entries that are the same product with caller-owned scratch (the last two arguments): recorded as the product, so that
# bench.py's labels and curvecloudnet_amd.costs key on ONE name and ONE argument order per product
PROFILE_ALIAS = {"gemm_nt_ws": "gemm_nt", "gemm_nt_acc_ws": "gemm_nt_acc", "gemm_nt_xf_ws": "gemm_nt_xf",
                 "gemm_nt_red_ws": "gemm_nt_red"}

DEBUG_SYNC = os.environ.get("CCN_DEBUG_SYNC") or None

EXTRA_LAUNCH = None     # experiment (tools/launch_cost.py): a callable launching one trivial kernel after every call


def call(name, *args, work_rows=None):
    """Invoke ``ccn_<name>`` on the current torch stream (appended as the last argument).  ``work_rows``: the row count of
    the edge-sized operand where the prototype does not carry it (CSR aggregations) -- recorded for bench.py's step
    roofline (curvecloudnet_amd.costs), never passed to the library."""
    fn = getattr(lib(), "ccn_" + name)
    if EXTRA_LAUNCH is not None:
        EXTRA_LAUNCH()
    if DEBUG_SYNC is not None:
        # diagnostics (CCN_DEBUG_SYNC=<file>): name + integer arguments of every launch appended BEFORE it, device synchronised
        # AFTER it -- the last line of the file is the launch that faulted
        with open(DEBUG_SYNC, "a") as f:
            f.write("%s %s\n" % (name, [a for a in args if isinstance(a, int)]))
        check(fn(*args, stream()), name)
        torch.cuda.synchronize()
        return
    if PROFILE is None:
        check(fn(*args, stream()), name)
        return
    # (a product with scratch is recorded as the product: its name, its arguments without the trailing scratch pair)
    pname, pargs = (PROFILE_ALIAS[name], args[:-2]) if name in PROFILE_ALIAS else (name, args)
    if ((PROFILE_ONLY is not None and not pname.startswith(PROFILE_ONLY))
            or (PROFILE_FILTER is not None and not PROFILE_FILTER(pname, pargs))):
        check(fn(*args, stream()), name)
        return
    # torch.cuda.Event records on torch's current stream, which is exactly the stream passed to the kernel
    beg, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    beg.record()
    check(fn(*args, stream()), name)
    end.record()
    # integer arguments (sizes / leading dimensions) and the positions of NULL pointers identify the kernel variant
    PROFILE.append((pname, tuple(a for a in pargs if isinstance(a, int)), beg, end,
                    tuple(i for i, a in enumerate(pargs) if a is None), work_rows))


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("curvecloudnet_amd ops run on the GPU only (got a %s tensor); there is no CPU path"
                               % t.device)


def workspace(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
