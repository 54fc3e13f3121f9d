"""CPU tests of the drop-in boundary: libfpca.so loads, exports every symbol include/fpca.h declares, and refuses to
run without a GPU (no CPU fallback).  No compute calls here."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


HEADERS = ("fpca.h", "fpca_debug.h")  # the drop-in boundary; measurement hooks and hardware diagnostics


def declared_functions(headers=HEADERS):
    names = []
    for h in headers:
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names += re.findall(r"\b(fpca_[a-z0-9_]+)\s*\(", src)
    return sorted(set(n for n in names if n not in ("fpca_allreduce_fn",)))


@pytest.mark.parametrize("header", HEADERS)
def test_header_is_plain_c(header):
    """The headers must compile as C (extern "C", plain pointers and sizes, no C++/torch types)."""
    out = subprocess.run(["gcc", "-std=c99", "-Wall", "-fsyntax-only", "-x", "c", os.path.join(ROOT, "include", header)],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr


def test_drop_in_header_carries_no_lab_bench():
    """include/fpca.h is what a maintainer of the reference binds (svdwide.h:77-81, randompca.h:77-80): the measurement
    hooks and hardware probes live in include/fpca_debug.h, which includes it -- not the other way round."""
    main = declared_functions(("fpca.h",))
    assert not [n for n in main if re.match(r"fpca_(debug|bench|profile)_", n)], main
    assert "fpca_debug.h" not in open(os.path.join(ROOT, "include", "fpca.h")).read()
    dbg = set(declared_functions(("fpca_debug.h",))) - set(main)
    assert dbg and all(re.match(r"fpca_(debug|bench|profile)_", n) for n in dbg), dbg


def test_struct_layouts_match_the_header(built_lib):
    """The ctypes mirrors of fpca_pca_opts / fpca_pca_info have the sizes the compiled library reports (struct_size /
    info_size), and the ABI revision is the header's.  (fpca_pca_default_opts, which wrote the library's own sizes into whatever
    struct the caller had, is gone since ABI 4: fpca_pca_init_opts with the CALLER's sizes is the only initialiser.)"""
    import flashpca_amd
    from flashpca_amd import _lib

    o = _lib.PcaOpts()
    assert not hasattr(flashpca_amd.lib(), "fpca_pca_default_opts")
    flashpca_amd.lib().fpca_pca_init_opts(C.byref(o), C.sizeof(_lib.PcaOpts), C.sizeof(_lib.PcaInfo))
    assert (o.struct_size, o.info_size) == (C.sizeof(_lib.PcaOpts), C.sizeof(_lib.PcaInfo))
    hdr = open(os.path.join(ROOT, "include", "fpca.h")).read()
    assert int(re.search(r"#define FPCA_ABI_VERSION (\d+)", hdr).group(1)) == _lib.ABI_VERSION == flashpca_amd.lib().fpca_abi_version()
    # a caller built against another header is refused before anything is read through the wrong layout
    o.struct_size -= 8
    rc = flashpca_amd.lib().fpca_pca(None, C.byref(o), None, None, None, None, None, None, None)
    assert rc == -1
    # ... and what such a caller does first -- FPCA_PCA_OPTS_INIT(&o), i.e. fpca_pca_init_opts with ITS sizeof -- writes its sizes
    # into the struct and never a byte past them (default_opts used to memset the library's sizeof into the caller's
    # smaller struct)
    small = C.sizeof(_lib.PcaOpts) - 8
    buf = (C.c_ubyte * (C.sizeof(_lib.PcaOpts) + 16))(*([0xAA] * (C.sizeof(_lib.PcaOpts) + 16)))
    flashpca_amd.lib().fpca_pca_init_opts(C.cast(buf, C.POINTER(_lib.PcaOpts)), small, 96)
    o2 = _lib.PcaOpts.from_buffer_copy(bytes(buf)[:C.sizeof(_lib.PcaOpts)])
    assert (o2.struct_size, o2.info_size, o2.ndim, o2.maxiter) == (small, 96, 10, 500)
    assert all(b == 0xAA for b in bytes(buf)[small:])
    rc = flashpca_amd.lib().fpca_pca(None, C.cast(buf, C.POINTER(_lib.PcaOpts)), None, None, None, None, None, None, None)
    assert rc == -1 and b"another include/fpca.h" in flashpca_amd.lib().fpca_last_error()


def test_library_exports_every_declared_symbol(built_lib):
    import flashpca_amd
    from flashpca_amd import _lib

    names = declared_functions()
    assert len(names) >= 25
    L = C.CDLL(built_lib)
    for n in names:
        assert hasattr(L, n), "libfpca.so does not export %s" % n
        assert n in _lib.SIGNATURES, "flashpca_amd/_lib.py has no signature for %s" % n
    assert flashpca_amd.lib().fpca_version().decode() == "0.3.0"


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under flashpca_amd/ (nor the CLI/lib link lines) may reference it."""
    bad = []
    for dp, dn, fn in os.walk(os.path.join(ROOT, "flashpca_amd")):
        if "_build" in dp or "__pycache__" in dp:
            continue
        for f in fn:
            if not f.endswith((".py", ".hpp", ".cpp", ".hip", ".h")) and f != "Makefile":
                continue
            txt = open(os.path.join(dp, f), errors="replace").read()
            code = "\n".join(l for l in txt.splitlines() if not l.lstrip().startswith(("//", "#", "*", "/*")))
            if re.search(r"fpca_oracle|orc_[a-z_]+\(|from oracle|import oracle|libfpca_hostsim|hostsim_", code):
                bad.append(os.path.join(dp, f))
    assert not bad, bad
    out = subprocess.run(["ldd", os.path.join(ROOT, "flashpca_amd", "_build", "libfpca.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out and "hostsim" not in out


def test_shipped_binaries_carry_no_test_switches(built_lib):
    """The environment switches the tests drive (forced missing-indicator modes and split plans, allocation-failure
    injection, the CLI launcher's host-memory transport and failure injection) are compiled into the -DFPCA_TEST_HOOKS build
    only (csrc/common.hpp: FPCA_TEST_ENV): their names must not even occur in the shipped library / CLI."""
    import flashpca_amd as fp

    names = [b"FPCA_CLI_TEST", b"FPCA_DEBUG_", b"FPCA_I8_MODE", b"FPCA_I8_SPLITS", b"FPCA_I8_ABL", b"FPCA_GATHER", b"FPCA_AR_CHUNKS",
             b"FPCA_SPARSE_SIDE_BYTES", b"FPCA_XT_SPLITS", b"FPCA_X_SPLITS", b"FPCA_I8_VERBOSE"]
    for path in (fp.LIB_PATH, fp.CLI_PATH):
        if os.environ.get("FPCA_LIB"):
            pytest.skip("FPCA_LIB overrides the shipped library")
        blob = open(path, "rb").read()
        for n in names:
            assert n not in blob, (path, n)
    hooks = open(fp.HOOKS_LIB_PATH, "rb").read() + open(fp.HOOKS_CLI_PATH, "rb").read()
    for n in (b"FPCA_CLI_TEST_TRANSPORT", b"FPCA_CLI_TEST_KILL_RANK", b"FPCA_I8_MODE", b"FPCA_AR_CHUNKS", b"FPCA_DEBUG_I8_NOMEM"):
        assert n in hooks, n
    # both builds export the same ABI
    L = C.CDLL(fp.HOOKS_LIB_PATH)
    for n in declared_functions():
        assert hasattr(L, n), n


def test_no_cpu_fallback(built_lib):
    """Without a usable gfx950 device every constructor fails loudly with FPCA_ENODEVICE."""
    import flashpca_amd as fp

    if fp.lib().fpca_device_count() > 0:
        pytest.skip("a GPU is present")
    packed = np.zeros((4, 3), dtype=np.uint8)
    with pytest.raises(fp.FpcaError) as e:
        fp.Context.from_packed(packed, 10, 4)
    assert e.value.code == -2 and "no CPU fallback" in str(e.value)
    with pytest.raises(fp.FpcaError):
        fp.Context.synthetic(100, 10)
    with pytest.raises(fp.FpcaError):
        fp.Context.from_bed(os.path.join(ROOT, "tests", "golden", "data_chr1.bed"), 957)


def test_default_opts_match_reference_cli(built_lib):
    import flashpca_amd as fp
    from flashpca_amd._lib import PcaInfo, PcaOpts

    o = PcaOpts()
    fp.lib().fpca_pca_init_opts(C.byref(o), C.sizeof(PcaOpts), C.sizeof(PcaInfo))
    # flashpca.cpp:325 (ndim 10), :426 (maxiter 500), :440 (tol 1e-6), :484 (div p), :276 (seed 1)
    assert (o.ndim, o.maxiter, o.tol, o.divisor, o.seed) == (10, 500, 1e-6, 2, 1)
    assert o.max_applies == 0  # no cap beyond the budget --maxiter implies


def test_gemm_kernels_do_not_spill():
    """Register discipline of the hot kernels: every GEMM kernel must compile without VGPR spills -- a
    spill inside the main loop once cost 15 % while every parity test stayed green."""
    import re
    import subprocess
    import tempfile

    csrc = os.path.join(ROOT, "flashpca_amd", "csrc")
    with tempfile.TemporaryDirectory() as tmp:
        for src, pat in (("kernels_i8.hip", "k_gemm_i8"), ("kernels.hip", "k_xt_b|k_x_t")):
            out = os.path.join(tmp, src + ".s")
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S",
                                   os.path.join(csrc, src), "-o", out], stderr=subprocess.DEVNULL)
            txt = open(out).read()
            names = re.findall(r"\.name:\s+(\S+)", txt)
            spills = re.findall(r"\.vgpr_spill_count:\s+(\d+)", txt)
            assert len(names) == len(spills) and names
            hot = [(n, int(s)) for n, s in zip(names, spills) if re.search(pat, n)]
            assert hot, src
            assert all(s == 0 for _, s in hot), [(n, s) for n, s in hot if s]


def _inflight_hazards(body):
    """Follow one kernel's assembly along EVERY control-flow path (both ways at a conditional branch; a state = program
    counter + the loads in flight, each visited once) and report every instruction that reads or writes a VGPR while a
    global load into it is still outstanding according to the s_waitcnt vmcnt(N) sequence (loads return in issue order)."""
    import re

    def regs(tok):
        out = set()
        for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
            if m.group(1):
                out.update(range(int(m.group(1)), int(m.group(2)) + 1))
            else:
                out.add(int(m.group(3)))
        return frozenset(out)

    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"\s*(\.LBB\w+):", l)] if m}
    ins = []  # (kind, payload) per line
    for l in body:
        t = l.strip()
        if not t or t[0] in ";." or re.match(r"\.?\w+:", t):
            ins.append(None)
            continue
        op = t.split()[0]
        args = t[len(op):]
        if op.startswith("global_load"):
            ins.append(("load", regs(args.split(",")[0])))
        elif op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", t)
            ins.append(("wait", int(m.group(1))) if m else None)
        elif op == "s_branch":
            ins.append(("jump", labels[args.split()[0]]))
        elif op.startswith("s_cbranch"):
            ins.append(("cjump", labels[args.split()[0]]))
        elif op == "s_endpgm":
            ins.append(("end", None))
        else:
            ins.append(("use", regs(args), t))
    bad, seen, stack = {}, set(), [(0, ())]
    while stack:
        pc, fl = stack.pop()
        while pc < len(ins):
            key = (pc, tuple(ln for _, ln in fl))
            if ins[pc] is not None and ins[pc][0] in ("cjump", "jump", "load"):
                if key in seen:
                    break
                seen.add(key)
            x = ins[pc]
            if x is None:
                pc += 1
                continue
            if x[0] == "load":
                fl = fl + ((x[1], pc),)
            elif x[0] == "wait":
                fl = fl[max(0, len(fl) - x[1]):]
            elif x[0] == "jump":
                pc = x[1]
                continue
            elif x[0] == "cjump":
                stack.append((x[1], fl))
            elif x[0] == "end":
                break
            else:
                for dst, ln in fl:
                    if x[1] & dst:
                        bad[(pc, ln)] = (pc, x[2], ln)
            pc += 1
    return sorted(bad.values())


def test_int8_gemm_never_touches_a_register_with_a_load_in_flight():
    """The int8 GEMM issues its global loads by inline asm and waits for them much later with hand-counted s_waitcnt: the
    compiler does not know a destination register is 'in flight' and is free to copy or reuse it (it did, once, when an
    in-flight value was carried around the loop: every parity test on the shapes of the day stayed green while another
    template instance produced noise).  This checks the generated code of EVERY instance: main loop in one straight-line
    block, nothing in flight across its back-edge, and no instruction touching a VGPR before its load has been waited for."""
    import re
    import subprocess
    import tempfile

    csrc = os.path.join(ROOT, "flashpca_amd", "csrc")
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S",
                               os.path.join(csrc, "kernels_i8.hip"), "-o", out], stderr=subprocess.DEVNULL)
        src = open(out).read().split("\n")
    starts = [i for i, l in enumerate(src) if re.match(r"_ZN4fpca4kern9k_gemm_i8\S*:", l)]
    assert len(starts) >= 15
    for st in starts:
        end = next(i for i in range(st, len(src)) if src[i].startswith(".Lfunc_end"))
        bad = _inflight_hazards(src[st:end])
        assert not bad, (src[st][:90], bad[:5])
