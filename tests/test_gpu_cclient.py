"""GPU: the C ABI used from native code -- a hipcc-built client with its own hipMalloc'd buffers and stream, no Python and no torch
in the process (tests/cclient/linear_client.cpp)."""
import os
import shutil
import subprocess
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_native_client_runs_a_linear_layer_through_the_c_abi(tmp_path):
    from mixermdm_amd._lib import lib_path
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    exe = tmp_path / "linear_client"
    libdir = os.path.dirname(lib_path())
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cclient", "linear_client.cpp"), "-o", str(exe), "-L", libdir, "-lmmdm_hip", "-Wl,-rpath," + libdir])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.startswith("OK"), (out.stdout, out.stderr)
