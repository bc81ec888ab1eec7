"""Imported first by the probe scripts that need the diagnostic build (rate probes, GEMM timing experiments): points
WAVEOME_AMD_LIB at tools/libwv_probes.so (`make -C waveome_amd/csrc probes`) unless the caller chose a library."""
import os

_here = os.path.dirname(os.path.abspath(__file__))
_lib = os.path.join(_here, "libwv_probes.so")
if "WAVEOME_AMD_LIB" not in os.environ:
    if not os.path.exists(_lib):
        raise SystemExit("tools/libwv_probes.so is missing: run `make -C waveome_amd/csrc probes` first")
    os.environ["WAVEOME_AMD_LIB"] = _lib
