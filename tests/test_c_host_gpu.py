"""The drop-in boundary from a host in plain C: tests/c_host/c_abi_host.c is compiled with gcc
against include/lasso_hip.h + the HIP runtime and linked with liblasso_hip.so and the C oracle;
it runs the fixed-step solve, the line search (trial trace), the objective and the ridge M-step on
hipMalloc'ed buffers and checks them against the fp64 oracle itself."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_c_host_against_the_c_oracle(tmp_path):
    lib_dir = os.path.join(ROOT, "pytorch-lasso_amd", "lasso_amd")
    ora_dir = os.path.join(ROOT, "oracle")
    assert os.path.exists(os.path.join(lib_dir, "liblasso_hip.so")) and os.path.exists(os.path.join(ora_dir, "liblasso_oracle_c.so"))
    exe = str(tmp_path / "c_abi_host")
    cmd = ["gcc", "-O1", "-std=c99", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c_host", "c_abi_host.c"), "-o", exe,
           "-L" + lib_dir, "-llasso_hip", "-L" + ora_dir, "-llasso_oracle_c", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
           "-Wl,-rpath," + lib_dir, "-Wl,-rpath," + ora_dir, "-Wl,-rpath,/opt/rocm/lib"]
    build = subprocess.run(cmd, capture_output=True, text=True)
    assert build.returncode == 0, build.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(run.stdout)
    assert run.returncode == 0, run.stdout + run.stderr
    assert run.stdout.strip().endswith("ok")
