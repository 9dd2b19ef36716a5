"""Sanitizer builds on the CPU (VERDICT r5 Next 9; GPU AddressSanitizer is not available on the pool): the C part of the oracle under
AddressSanitizer + UndefinedBehaviorSanitizer with a driver that allocates every buffer at exactly the callers' sizes, and the HOST
halves of the HIP library -- argument checks and workspace arithmetic -- called with hostile arguments: they must answer with an error
code (or a size) and never touch the device or fault."""
import ctypes
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_c_under_address_and_ub_sanitizers(tmp_path):
    exe = str(tmp_path / "oracle_sanitize")
    cmd = ["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
           "-ffp-contract=off", "-fopenmp", os.path.join(ROOT, "tests", "c", "oracle_sanitize.c"),
           os.path.join(ROOT, "oracle", "frnn_bruteforce.c"), "-o", exe, "-lm"]
    build = subprocess.run(cmd, capture_output=True, text=True)
    if build.returncode != 0 and ("asan" in build.stderr.lower() or "sanitize" in build.stderr.lower()):
        pytest.skip("this gcc has no sanitizer runtime: %s" % build.stderr.strip().splitlines()[-1])
    assert build.returncode == 0, build.stderr
    env = dict(os.environ, OMP_NUM_THREADS="4", ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    run = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "oracle_sanitize: clean" in run.stdout


def test_library_host_side_rejects_hostile_arguments_without_a_gpu():
    """Every entry point validates on the host before it launches anything: NULL pointers, negative / zero sizes, leading dimensions
    smaller than the row, workspaces that are too small -- a negative return code and a message, never a fault.  The workspace
    queries are pure host arithmetic: monotone in their arguments and finite for the largest sizes BASELINE.json names."""
    from curvecloudnet_amd import _lib
    lib = _lib.lib()
    null = ctypes.c_void_p(None)
    fake = ctypes.c_void_p(0x1000)              # never dereferenced: the checks below fail before any launch
    err = lambda: lib.ccn_last_error().decode()   # noqa: E731
    assert lib.ccn_gemm_nt(null, 8, fake, 8, null, fake, 8, 16, 8, 8, null, null) < 0 and "null" in err()
    assert lib.ccn_gemm_nt(fake, 4, fake, 8, null, fake, 8, 16, 8, 8, null, null) < 0          # lda < K
    assert lib.ccn_gemm_nt(fake, 8, fake, 8, null, fake, 8, -1, 8, 8, null, null) < 0          # negative M
    assert lib.ccn_gemm_tn_ws(fake, 8, fake, 8, fake, 4, 16, 8, 8, null, 0, null) < 0          # lddw < K
    assert lib.ccn_fps(fake, fake, fake, fake, 2, 100, 200, fake, 16, null, fake, null) < 0 and "workspace" in err()
    assert lib.ccn_fps(null, fake, fake, fake, 2, 100, 200, fake, 1 << 20, null, fake, null) < 0
    assert lib.ccn_knn_points(fake, fake, fake, fake, 1, 10, 99, fake, fake, null) < 0 and "K must be" in err()
    assert lib.ccn_ball_query(fake, fake, fake, fake, 0, 1, 1, 1, 0.5, fake, null) < 0          # B = 0
    assert lib.ccn_voxel_keys(fake, fake, fake, 10, 0.0, fake, fake, fake, null) < 0            # voxel size 0
    # workspace arithmetic: grows with the problem, stays finite at BASELINE configs[3]'s 4 x 120 k points and beyond
    sizes = [lib.ccn_fps_workspace_bytes(n, 4) for n in (0, 1, 1000, 480000, 1 << 31)]
    assert sizes == sorted(sizes) and sizes[0] == 4 * 512 and sizes[-1] < 1 << 40
    assert lib.ccn_fps_workspace_bytes(-5, 4) == 0 and lib.ccn_fps_workspace_bytes(5, -1) == 0
    grid = [lib.ccn_frnn_grid_bytes(b, n) for b, n in ((1, 1), (8, 50000), (4, 120000), (16, 1 << 24))]
    assert all(g > 0 for g in grid) and grid[1] < grid[3]
    scan = [lib.ccn_exclusive_scan_workspace_bytes(n) for n in (0, 1, 1 << 20, 1 << 33)]
    assert scan == sorted(scan)
    tn = [lib.ccn_gemm_tn_workspace_bytes(m, 256, 256) for m in (0, 1023, 1024, 1 << 20, 1 << 31)]
    assert tn[0] == 0 and tn[1] == 0 and max(tn) <= 64 << 20
    assert lib.ccn_rank_keys_workspace_bytes(0) <= lib.ccn_rank_keys_workspace_bytes(1 << 20) < 1 << 32
