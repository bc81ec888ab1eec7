"""The drop-in boundary from plain C: tests/c_abi/kat1.c includes include/waveome_amd.h and links libwaveome_amd.so -
no Python, no ctypes, no torch in the process.  CPU: the header compiles as strict C99 and the host-only entry points
work; GPU: KAT-1 fitted through wv_fit from C reproduces the notebook's full-precision parameters."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "waveome_amd")


@pytest.fixture(scope="module")
def kat1_binary(tmp_path_factory):
    out = tmp_path_factory.mktemp("c_abi") / "kat1"
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-O1",
           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c_abi", "kat1.c"),
           "-L", LIBDIR, "-lwaveome_amd", "-lm", f"-Wl,-rpath,{LIBDIR}", "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return str(out)


def test_header_is_strict_c_and_the_host_entry_points_work_from_c(kat1_binary):
    r = subprocess.run([kat1_binary, "--host"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "host L-BFGS-B:" in r.stdout


@pytest.mark.gpu
def test_kat1_fitted_from_plain_c(kat1_binary, golden_dir, tmp_path):
    with open(os.path.join(golden_dir, "kat.json")) as fh:
        c = json.load(fh)["kat1"]
    xy = tmp_path / "kat1_xy.txt"
    xy.write_text("\n".join(repr(float(v)) for v in list(c["X"]) + list(c["Y"])) + "\n")
    r = subprocess.run([kat1_binary, str(xy)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "KAT-1 from C" in r.stdout
    print(r.stdout.strip())
