"""The C-ABI library loads without a GPU and exports exactly what include/myraytracer_amd.h (the drop-in boundary) and
include/myraytracer_amd_debug.h (diagnostics) declare."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "myraytracer_amd.h")
DEBUG_HEADER = os.path.join(ROOT, "include", "myraytracer_amd_debug.h")


def header_functions(path=None):
    names = set()
    for h in ([path] if path else [HEADER, DEBUG_HEADER]):
        text = open(h).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(mrt_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_the_boundary_header_holds_no_diagnostics():
    """include/myraytracer_amd.h is what INTEGRATION.md cites as the drop-in boundary: every mrt_debug_* entry point lives in
    the debug header, and only there."""
    product, debug = header_functions(HEADER), header_functions(DEBUG_HEADER)
    assert not [n for n in product if n.startswith("mrt_debug_")]
    assert debug and all(n.startswith("mrt_debug_") for n in debug)
    assert "myraytracer_amd_debug.h" not in open(os.path.join(ROOT, "INTEGRATION.md")).read().split("## 1.")[1].split("## 2.")[0]


def test_every_declared_symbol_is_exported(mrt):
    from myraytracer_amd import _lib
    L = C.CDLL(_lib.LIB_PATH)
    declared = header_functions()
    assert len(declared) >= 35
    for name in declared:
        assert hasattr(L, name), f"{name} declared in the header but not exported by the .so"
    assert sorted(_lib.EXPORTS) == declared, "python binding list and header disagree"


def test_abi_version_and_status_strings(mrt):
    from myraytracer_amd import _lib
    L = _lib.load()
    assert L.mrt_abi_version() == 4
    assert L.mrt_status_string(0) == b"ok"
    assert L.mrt_status_string(2) == b"no usable HIP device"
    assert L.mrt_status_string(9) == b"a wait for the GPU passed its deadline"
    assert L.mrt_last_error(None) is not None


def test_build_id_names_the_sources_on_disk(mrt):
    """mrt_build_id() (baked in by the Makefile) equals scripts/source_hash.py over the sources here: the library under test
    was built from this tree (bench.py refuses a headline otherwise)."""
    import sys
    from myraytracer_amd import _lib
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from source_hash import source_sha16
    if os.environ.get("MRT_LIB_OVERRIDE"):
        pytest.skip("a substituted library")
    assert _lib.load().mrt_build_id().decode() == source_sha16()


def test_struct_sizes_match_reference_layouts(mrt):
    from myraytracer_amd import _lib
    assert C.sizeof(_lib.MrtLocals) == 48          # lib.rs:368-377, 16-aligned
    assert C.sizeof(_lib.MrtWorld) == 80           # raw::World 64 B + DielectricRange 16 B
    assert C.sizeof(_lib.MrtSphereRange) == 32 and C.sizeof(_lib.MrtLambertianRange) == 16
    assert C.sizeof(_lib.MrtMetalRange) == 16 and C.sizeof(_lib.MrtDielectricRange) == 16
    assert C.sizeof(_lib.MrtArgs) == 20 and C.sizeof(_lib.MrtSphere) == 36
    assert _lib.MrtLocals.rng_shuffle.offset == 16 and _lib.MrtLocals.framebuffer_weight.offset == 32


def test_no_device_is_a_loud_error_not_a_fallback(mrt):
    """Without a GPU (this container) mrt_create must fail; on the GPU box this test is a no-op."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(mrt.MrtError) as e:
        mrt.State(mrt.Args(16, 16))
    assert e.value.status == 2


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under myraytracer_amd/ or include/ may reference it."""
    for base in ("myraytracer_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hip", ".cpp")):
                    text = open(os.path.join(dirpath, f), errors="ignore").read()
                    assert "rt_oracle" not in text and "pyoracle" not in text, os.path.join(dirpath, f)
                    assert not re.search(r"#include\s+[\"<].*oracle", text), os.path.join(dirpath, f)


def test_hand_issued_scalar_loads_are_safe_in_the_built_isa():
    """scripts/check_isa.py: no compiler-generated instruction touches an SGPR while the sweep's
    inline-asm s_load into it may still be in flight."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_isa.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def build_c_caller(tmp_path):
    """gcc tests/abi_c_caller.c against the in-tree .so, as a C (or Rust FFI) consumer of the header would."""
    import subprocess
    exe = str(tmp_path / "abi_c_caller")
    libdir = os.path.join(ROOT, "myraytracer_amd", "lib")
    subprocess.check_call(["gcc", "-O1", "-Wall", "-Wextra", "-Werror", "-o", exe, os.path.join(ROOT, "tests", "abi_c_caller.c"),
                           "-L" + libdir, "-lmyraytracer_amd", "-Wl,-rpath," + libdir])
    return exe


def test_plain_c_caller_links_and_packs_like_the_reference(mrt, tmp_path):
    """The header compiles as C, the .so links from C, and the reference's 64-byte raw::World of the shipped scene
    (built by hand in C from lib.rs:687-799) is bit-identical to the first 64 bytes mrt_pack_world produces."""
    import subprocess
    r = subprocess.run([build_c_caller(tmp_path), "host"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "host ok" in r.stdout


def test_pytest_collects_only_tests_dir():
    """pytest.ini pins testpaths: scripts/ (GPU experiments) must never be collected."""
    ini = open(os.path.join(ROOT, "pytest.ini")).read()
    assert "testpaths = tests" in ini
    assert not [f for f in os.listdir(os.path.join(ROOT, "scripts")) if f.endswith("_test.py") or f.startswith("test_")]
