"""The C-ABI library loads on a machine without a GPU and exports every symbol
include/hvc_jpeg.h declares; creating a context without a GPU fails loudly with
HVC_E_NO_DEVICE (no CPU fallback).  No compute calls here."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "hvc_jpeg.h")).read()
    return sorted(set(re.findall(r"HVC_API\s+[\w\s\*]+?\b(hvc_\w+)\s*\(", hdr)))


def test_header_and_binding_list_agree():
    import video_coding_amd as hvc
    assert declared_symbols() == sorted(hvc.hvc.SYMBOLS)


def test_library_exports_every_declared_symbol():
    import video_coding_amd as hvc
    hvc.build()
    L = hvc.lib()
    for s in declared_symbols():
        assert hasattr(L, s), s
    assert b"gfx950" in L.hvc_version()


def test_strerror_and_no_device_without_gpu():
    import torch
    import video_coding_amd as hvc
    L = hvc.lib()
    assert L.hvc_strerror(0) == b"ok"
    assert L.hvc_strerror(-2) == b"no usable gfx950 device"
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the no-device path cannot be exercised")
    with pytest.raises(hvc.HvcError) as e:
        hvc.Context(0)
    assert e.value.code == -2


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under video-coding_amd/ may mention it."""
    pkg = os.path.join(ROOT, "video-coding_amd")
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, fn), errors="ignore").read()
                assert "oracle" not in txt.lower().replace("checked against the cpu oracle", ""), os.path.join(dp, fn)


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """include/hvc_jpeg.h compiles as C99 (-pedantic) and a C program taking the address of every
    declared entry point links against libhvc_jpeg.so -- the boundary a cgo / ctypes / OCaml
    `foreign` binding sees.  Host-only calls are executed (no GPU needed)."""
    import subprocess
    import video_coding_amd as hvc
    so = hvc.build()
    syms = declared_symbols()
    src = tmp_path / "abi.c"
    src.write_text(
        '#include <stdio.h>\n#include <string.h>\n#include "hvc_jpeg.h"\n'
        "int main(void) {\n"
        "    typedef void (*anyfn)(void);\n"
        "    const anyfn fn[] = {%s};\n" % ", ".join("(anyfn)%s" % s for s in syms) +
        "    uint16_t q[64];\n"
        "    hvc_jpeg_info info;\n"
        "    int mx = -1;\n"
        "    uint64_t tot = 0, se = 0;\n"
        "    const uint8_t a[4] = {1, 2, 3, 250}, b[4] = {3, 2, 0, 255};\n"
        "    if (hvc_quant_table(0, 50, q) != HVC_OK || q[0] != 16) return 1;   /* Quant_tables.luma at q50 */\n"
        "    if (hvc_jpeg_encoder_layout(1920, 1080, 420, 75, &info) != HVC_OK || info.n_comp != 3) return 2;\n"
        "    if (info.layout[0].blocks_w != 240 || info.layout[0].blocks_h != 136) return 3;\n"
        "    if (hvc_compare_planes(a, b, 4, &mx, &tot, &se) != HVC_OK || mx != 5 || tot != 10 || se != 38) return 4;\n"
        "    if (strcmp(hvc_strerror(HVC_E_NO_DEVICE), \"no usable gfx950 device\")) return 5;\n"
        '    printf("%d symbols\\n", (int)(sizeof fn / sizeof fn[0]));\n'
        "    return 0;\n}\n")
    exe = tmp_path / "abi"
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           str(src), "-o", str(exe), so, "-Wl,-rpath," + os.path.dirname(so)])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, (out.returncode, out.stderr)
    assert out.stdout.strip() == "%d symbols" % len(syms)


def test_every_entry_point_refuses_null_arguments():
    """Every int-returning function of include/hvc_jpeg.h called with nothing but zeros and null pointers (no context, no
    buffers): an hvc_status comes back -- HVC_E_INVALID_ARG but for the two that have nothing to refuse -- and the process
    is still there.  One child process for the whole sweep (a crash must not take the test runner along)."""
    import subprocess
    import sys
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "hvc_jpeg.h")).read(), flags=re.S)
    protos = re.findall(r"HVC_API\s+int\s+(hvc_\w+)\s*\(([^;]*?)\)\s*;", hdr, flags=re.S)
    assert len(protos) >= 45
    calls = [(name, 0 if args.strip() in ("", "void") else len(re.split(r",(?![^()]*\))", args))) for name, args in protos]
    code = "\n".join([
        "import ctypes as C, sys",
        "sys.path.insert(0, %r)" % ROOT,
        "import video_coding_amd as hvc",
        "L = hvc.lib()",
        "for name, n in %r:" % (calls,),
        "    f = getattr(L, name); f.restype = C.c_int; f.argtypes = [C.c_void_p] * n",
        "    print(name, f(*([None] * n)), flush=True)",
        "print('SWEPT')"])
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    seen = dict(line.split() for line in r.stdout.splitlines() if line.startswith("hvc_"))
    assert r.returncode == 0 and "SWEPT" in r.stdout, "crashed behind %s\n%s" % (list(seen)[-1:] or "the start", r.stderr[-2000:])
    assert set(seen) == {name for name, _ in calls}
    nothing_to_refuse = {"hvc_last_hip_error": "0",   # (no context: no error recorded)
                         "hvc_compare_planes": "0"}   # (zero samples: the planes are equal)
    for name, status in seen.items():
        assert status == nothing_to_refuse.get(name, "-1"), (name, status)
