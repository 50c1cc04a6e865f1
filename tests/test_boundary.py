"""CPU suite, part 2: the drop-in boundary without a GPU -- the C-ABI library loads and exports every
symbol include/pt_api.h declares, compute entry points fail loudly (no CPU fallback), the contract headers
reproduce the reference's layout/known answers, the CLI keeps the reference's argv/exit behaviour, and the
query kernels contain no fused multiply-add in the metric."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "3d-reconstruction-from-point-cloud_amd")


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "pt_api.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pt_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(pkg):
    declared = _declared_symbols()
    assert len(declared) >= 25
    assert declared == sorted(pkg.capi.SYMBOLS), "capi.SYMBOLS and include/pt_api.h disagree"
    out = subprocess.check_output(["nm", "-D", "--defined-only", pkg.capi.LIB_PATH], text=True)
    exported = set(re.findall(r"\bT (pt_[a-z0-9_]+)\b", out))
    missing = [s for s in declared if s not in exported]
    assert not missing, "libpt_hip.so lacks %s" % missing
    L = pkg.capi.lib()
    for s in declared:
        assert getattr(L, s) is not None


def test_point_layout_matches_reference(pkg, oracle):
    lay = oracle.ref_point_layout()                      # from the reference's own Point.h
    dt = pkg.POINT_DTYPE
    assert dt.itemsize == lay["sizeof"] == 80
    assert [dt.fields[n][1] for n in ("ver", "normal", "color", "U", "V")] == \
        [lay["off_ver"], lay["off_normal"], lay["off_color"], lay["off_U"], lay["off_V"]]
    assert pkg.K_REFERENCE == 20                          # reference src/pointsTransfer.cpp:128


def test_contract_headers_selftest():
    exe = os.path.join(PKG, "contract_selftest")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "contract selftest ok" in r.stdout


def test_ply_reader_under_sanitizers(tmp_path):
    """host/ply_io.h (serial reader) on the reference's grammar and on malformed files, and host/ply_fast.h (the parallel
    reader behind the CLI) against it at several thread counts -- built with AddressSanitizer + UBSan (CPU only:
    sanitizers are not available on the GPU pool)."""
    exe = str(tmp_path / "ply_selftest")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-pthread",
                           "-I" + os.path.join(ROOT, "include"), "-o", exe, os.path.join(PKG, "host", "ply_selftest.cpp")])
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0 and "ply selftest ok" in r.stdout, r.stdout + r.stderr


def test_launcher_and_png_writer_under_sanitizers(tmp_path):
    """host/sharded.h -- process launcher (fork / wait / kill) and the ranks' file rendezvous (write_all, wait_for_file, the mmap'ed
    pieces, slab bounds) -- and host/png_write.h (threaded deflate bands) built with AddressSanitizer + UBSan and run on the CPU (VERDICT
    r3, weak 12).  The launcher self-test links libpt_hip.so like the CLI but makes no GPU call; leak checking is off because the HIP
    runtime it pulls in keeps its own allocations until exit."""
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", TMPDIR=str(tmp_path))
    san = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-pthread",
           "-I" + os.path.join(ROOT, "include")]
    exe = str(tmp_path / "launcher_san")
    subprocess.check_call(san + ["-o", exe, os.path.join(PKG, "host", "launcher_selftest.cpp"), "-L" + PKG, "-lpt_hip", "-lz",
                                 "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=180, env=env)
    assert r.returncode == 0 and "launcher selftest ok" in r.stdout, r.stdout + r.stderr
    exe = str(tmp_path / "png_san")
    subprocess.check_call(san + ["-o", exe, os.path.join(PKG, "host", "png_selftest.cpp"), "-lz"])
    for threads in (1, 5):
        r = subprocess.run([exe, str(tmp_path / "s.png"), "257", "333", str(threads)], capture_output=True, text=True, timeout=120, env=env)
        assert r.returncode == 0, r.stdout + r.stderr


def test_cli_keeps_reference_argv_behaviour(tmp_path):
    exe = os.path.join(PKG, "pointsTransfer")
    assert os.path.exists(exe)
    r = subprocess.run([exe], capture_output=True, text=True)            # reference :112-120
    assert r.returncode == 0 and r.stdout.strip() == "Usage: ./pointTransfer <input-point-cloud> <input-mesh>"
    r = subprocess.run([exe, "only-one-arg"], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("Usage:")
    missing = str(tmp_path / "nope.ply")
    r = subprocess.run([exe, missing, missing], capture_output=True, text=True)   # reference :137-141
    assert r.returncode == 0 and r.stderr.strip() == "Cannot read or find point cloud file: " + missing


def test_no_cpu_fallback_without_gpu(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.PtError) as e:
        pkg.PointsTransfer(device=0)
    assert e.value.code == pkg.capi.ERR_HIP
    # raw ABI: a null context is an argument error, never a silent success
    L = pkg.capi.lib()
    assert L.pt_rebuild(None) == pkg.capi.ERR_ARG and L.pt_query_resident(None, 8, None, None) == pkg.capi.ERR_ARG
    h = C.c_void_p()
    assert L.pt_ctx_create(C.byref(h), (C.c_int * 2)(0, 1), 2) == pkg.capi.ERR_ARG     # one context per GPU


def test_product_never_imports_the_oracle():
    bad = []
    for base, _, files in os.walk(PKG):
        if "_build" in base:
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                txt = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r"(^|\n)\s*(from|import)\s+oracle\b|pt_oracle|libpt_oracle", txt):
                    bad.append(f)
    assert not bad, "product files reference the oracle: %s" % bad


def test_metric_has_no_fma_in_query_kernels():
    """reference src/Distance.h:6-11 compiles to 3 mul + 2 add under the reference's flags;
    the HIP query kernels must not contract it (hipcc defaults to -ffp-contract=fast)."""
    subprocess.check_call(["make", "-C", os.path.join(PKG, "csrc"), "-s", "asm"])
    s = open(os.path.join(PKG, "csrc", "_build", "asm", "pt_query-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    assert "knn_kernel" in s and "v_mul_f64" in s and "v_add_f64" in s
    # per kernel body; the tile kernel's BLEND = true instantiations (last but one template argument) are the
    # same metric code plus an epilogue whose fp64 divisions and sqrt expand to FMA sequences -- those are skipped
    checked = 0
    for b in s.split("; -- Begin function ")[1:]:
        name = b.split("\n", 1)[0].strip()
        b = b.split("; -- End function")[0]
        if "s_endpgm" not in b:
            continue
        if "knn_tile_kernel" in name and re.search(r"ELb1ELb[01]ELi\d+ELb[01]EEEv", name):       # <..., WIDE, BLEND = true, DBL, KC, BND>
            continue
        checked += 1
        assert "v_fma_f64" not in b and "v_fmac_f64" not in b, name
    assert checked >= 10
    # no scratch spills in the hot kernels.  Two exceptions, both measured: the HIER = true instantiations of the group and wave kernels
    # (clouds with refined cells only): their register budget is capped for occupancy and the three-level descent, cold code, spills a
    # little; and the wave kernel's plain variant, held to 64 VGPRs for eight waves per SIMD (a handful of cold registers in scratch:
    # 280 ms against 297 at seven waves without spills).
    for b in s.split("; -- Begin function ")[1:]:
        name = b.split("\n", 1)[0].strip()
        m = re.search(r"ScratchSize: (\d+)", b)
        if m is None or "s_endpgm" not in b:
            continue
        hier = ("knn_kernel" in name or "knn_wave_kernel" in name) and re.search(r"ELb1EEEv", name) is not None
        wave = "knn_wave_kernel" in name
        touches = re.search(r"\tscratch_(load|store)|\tbuffer_(load|store)_dword[^\n]*offen|\tbuffer_(load|store)_dword[^\n]*s\[0:3\]", b) is not None
        # The rule is ScratchSize == 0.  One more exception, shown by the ISA itself: a kernel whose SGPRs spill into the LANES of a VGPR
        # (`v_writelane_b32` / `v_readlane_b32`, marked "SGPR spill to VGPR lane" by the compiler) gets a small frame reserved for that
        # VGPR and never touches it -- no scratch_* / buffer_* instruction in the body: register moves, not memory traffic (the fused
        # k-NN + blend tile kernel: 36 bytes, 16 lane writes at entry).
        lane_spill_only = (not touches) and int(m.group(1)) <= 64 and "SGPR spill to VGPR lane" in b
        assert int(m.group(1)) == 0 or lane_spill_only or (hier and int(m.group(1)) <= 512) or (wave and int(m.group(1)) <= 64), (name, m.group(1))


def test_png_writer_roundtrip(tmp_path):
    """host/png_write.h (what writes texture.png in place of cv::imwrite, reference src/pointsTransfer.cpp:613-615): files written
    with 1, 3 and 16 deflate threads decode -- CRCs, Adler-32 of the concatenated bands and all -- to the same pixels."""
    import struct, subprocess, zlib
    import numpy as np
    subprocess.check_call(["make", "-s", "-C", os.path.join(PKG, "host"), "../png_selftest"])
    w, h = 301, 1000
    want = np.empty(w * h * 4, np.uint8)
    s = 12345
    i = np.arange(w * h * 4, dtype=np.uint64)
    lcg = np.empty(w * h * 4, np.uint32)
    for j in range(w * h * 4):                      # (the same LCG as the C++ side)
        s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
        lcg[j] = s
    noisy = ((i // 4 // w) % 7) == 0
    want = np.where(noisy, (lcg >> 24).astype(np.uint8), ((i * 31) >> 3).astype(np.uint8)).reshape(h, w, 4)
    for threads in (1, 3, 16):
        out = tmp_path / ("t%d.png" % threads)
        subprocess.check_call([os.path.join(PKG, "png_selftest"), str(out), str(w), str(h), str(threads)])
        data = open(out, "rb").read()
        assert data[:8] == b"\x89PNG\r\n\x1a\n"
        off, idat = 8, []
        while off < len(data):
            ln, typ = struct.unpack(">I4s", data[off:off + 8])
            body = data[off + 8:off + 8 + ln]
            assert struct.unpack(">I", data[off + 8 + ln:off + 12 + ln])[0] == (zlib.crc32(typ + body) & 0xFFFFFFFF)
            if typ == b"IHDR":
                assert struct.unpack(">IIBBBBB", body) == (w, h, 8, 6, 0, 0, 0)
            if typ == b"IDAT":
                idat.append(body)
            off += 12 + ln
        raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8).reshape(h, w * 4 + 1)
        assert (raw[:, 0] == 0).all()
        assert np.array_equal(raw[:, 1:].reshape(h, w, 4)[:, :, [2, 1, 0, 3]], want)


# ---- multi-process launchers without a GPU ("start, fail loudly, be testable") -------------------------------
def test_cli_launcher_stops_the_job_when_a_rank_fails():
    """host/sharded.h::spawn_and_wait -- the launcher of `pointsTransfer --gpus N`: children reaped as they end, the first failure
    SIGTERMs (then SIGKILLs) the others and is the exit code; a peer that would wait ten minutes is gone within seconds."""
    exe = os.path.join(PKG, "launcher_selftest")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "launcher selftest ok" in r.stdout, r.stdout + r.stderr


def test_cli_sharded_launch_fails_fast_without_devices(tmp_path):
    """`pointsTransfer cloud mesh --gpus 2` where no rank can open its device (this container has no GPU; on a GPU box device 99
    does not exist): non-zero exit within seconds, no rank left behind, the rendezvous directory removed."""
    exe = os.path.join(PKG, "pointsTransfer")
    (tmp_path / "c.ply").write_text("ply\nformat ascii 1.0\nelement vertex 1\nend_header\n0 0 0 0 0 1 1 2 3\n")
    (tmp_path / "m.ply").write_text("ply\nformat ascii 1.0\nelement vertex 0\nelement face 0\nend_header\n")
    r = subprocess.run([exe, str(tmp_path / "c.ply"), str(tmp_path / "m.ply"), "--gpus", "2", "--device", "99"], capture_output=True, text=True,
                       cwd=tmp_path, timeout=60)
    assert r.returncode != 0 and "no usable HIP device" in r.stderr
    dirs = [l.split()[-1] for l in r.stderr.splitlines() if l.startswith("[pt_hip launcher] rendezvous")]
    assert dirs and not os.path.exists(dirs[0])


def _bench(*argv, timeout=180):
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE: the process becomes the launcher (before torch is imported), starts two fresh
    ranks, and stdout carries exactly one JSON line.  --dry-run: rendezvous, fence and max-over-ranks over gloo, no GPU work."""
    import json
    r = _bench("--gpus", "2", "--dry-run", "--backend", "gloo", "--steps", "2")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["dry_run"] is True and out["n_gpus"] == 2 and out["value"] is None


def test_bench_under_torch_distributed_run():
    """The driver's own launch form for N > 1 -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...` -- in dry-run mode on CPU: the ranks are torchrun's, bench.py only reads the environment, and
    rank 0 prints the one JSON line."""
    import json
    import socket
    import sys
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--backend", "gloo", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.lstrip().startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2


def test_bench_launcher_fails_when_a_rank_fails():
    """A rank that dies before the rendezvous: the others sit in init_process_group waiting for it -- the launcher stops them and
    exits with the failed rank's code, promptly."""
    import time
    t0 = time.time()
    r = _bench("--gpus", "3", "--dry-run", "--backend", "gloo", "--fail-rank", "2")
    assert r.returncode == 9 and time.time() - t0 < 120, r.stderr[-2000:]
    assert "stopping the other" in r.stderr and not r.stdout.strip()


def test_bench_refuses_more_ranks_than_devices():
    """--gpus N beyond the visible devices: every rank refuses before entering a collective; non-zero exit with that sentence.
    (No GPU in the CPU container: N = 2 is already too many.  On a GPU box this test asks for one more than there are.)"""
    import torch
    n = torch.cuda.device_count() + 1
    if n < 2:
        n = 2
    r = _bench("--gpus", str(n), "--steps", "1", "--warmup", "0", "--no-cpu-baseline")
    assert r.returncode != 0 and "visible GPUs" in r.stderr and not r.stdout.strip()


def test_python_sources_are_import_clean():
    """Every Python file compiles and uses no undefined names (a NameError in a test that only ever SKIPS is found here, not on the
    first multi-GPU box): py_compile + a symbol-table pass over module- and function-level names."""
    import ast
    import builtins
    import py_compile
    import symtable
    files = []
    for base in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools"), PKG, os.path.join(ROOT, "oracle")):
        files += [os.path.join(base, f) for f in sorted(os.listdir(base)) if f.endswith(".py")]
    assert len(files) > 20
    bad = []
    for path in files:
        py_compile.compile(path, doraise=True)
        src = open(path).read()
        top = symtable.symtable(src, path, "exec")
        module_names = {s.get_name() for s in top.get_symbols() if s.is_assigned() or s.is_imported() or s.is_namespace()} | set(dir(builtins)) | {"__file__", "__name__", "__doc__"}
        star = any(isinstance(n, ast.ImportFrom) and any(a.name == "*" for a in n.names) for n in ast.walk(ast.parse(src)))

        def walk(tab, enclosing):
            local = {s.get_name() for s in tab.get_symbols() if s.is_assigned() or s.is_imported() or s.is_parameter() or s.is_namespace()}
            if tab.get_type() != "class":
                enclosing = enclosing | local
            for sym in tab.get_symbols():
                n = sym.get_name()
                if sym.is_referenced() and not (n in local or n in enclosing or n in module_names or sym.is_free()):
                    if sym.is_global() and not star:
                        bad.append("%s: %s (in %s)" % (os.path.relpath(path, ROOT), n, tab.get_name()))
            for ch in tab.get_children():
                walk(ch, enclosing)
        walk(top, set())
    assert not bad, "undefined names: " + "; ".join(bad)
