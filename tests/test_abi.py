"""CPU tests: the C-ABI library loads and exports every symbol include/plviwo.h declares; without a
GPU the compute entry points refuse to run (no CPU fallback)."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "plviwo.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(plv_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported(pkg):
    lib = pkg.load_library()
    names = _declared_symbols()
    assert len(names) >= 15
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, f"declared in plviwo.h but not exported: {missing}"
    assert lib.plv_abi_version() == 1


def test_python_binding_covers_header(pkg):
    lib = pkg.load_library()
    bound = set(lib._plv_signatures)
    assert set(_declared_symbols()) <= bound, sorted(set(_declared_symbols()) - bound)


def test_no_device_means_no_compute(pkg):
    """On a machine without a gfx950 the ctx cannot be created: the product has no CPU path."""
    lib = pkg.load_library()
    if lib.plv_device_count() > 0:
        return
    cfg = pkg.default_config()
    h = C.c_void_p()
    rc = lib.plv_ctx_create(C.byref(cfg), C.byref(h))
    assert rc == pkg.PLV_E_NO_DEVICE
    assert b"no" in lib.plv_last_error().lower()


def test_product_does_not_reference_oracle():
    """Nothing under pl-viwo_amd/ may include, link or import the oracle."""
    bad = []
    for d, _, files in os.walk(os.path.join(ROOT, "pl-viwo_amd")):
        if os.sep + "build" in d or os.sep + "lib" in d:
            continue
        for f in files:
            if f.endswith((".hip", ".cpp", ".hpp", ".h", ".py", "Makefile")):
                t = open(os.path.join(d, f), errors="ignore").read()
                if re.search(r'#include\s+"[^"]*oracle|liboracle|import\s+oracle|oracle_lib', t):
                    bad.append(os.path.join(d, f))
    assert not bad, bad


def test_header_is_plain_c_and_links(pkg, tmp_path):
    """include/plviwo.h compiles as C11 (no C++ in the signatures) and a C program links against the shared library."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "cabi.c"
    src.write_text('#include "plviwo.h"\n#include <stdio.h>\n'
                   'int main(void) { plv_config c; plv_config_default(&c, 752, 480);\n'
                   '  printf("%d %d %d\\n", plv_abi_version(), c.width, c.win_size); return 0; }\n')
    exe = tmp_path / "cabi"
    lib_dir = os.path.join(root, "pl-viwo_amd", "lib")
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"), str(src), "-o", str(exe),
                    "-L", lib_dir, "-lplviwo_hip", "-Wl,-rpath," + lib_dir], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert int(out[0]) >= 1 and out[1:] == ["752", "15"]
