"""Imported FIRST by the tools that steer the launcher through environment variables (CLSIMHIP_K_POP, CLSIMHIP_GRID,
CLSIMHIP_KERNEL ...): points the Python mirror at the developer build of the library (tools/build_variant.sh dev ->
build_variants/dev.so, -DCLSIMHIP_DEVELOPER).  The default build reads no tuning from the environment (include/clsimhip.h:
clsimhip_set_tuning)."""
import os

_dev = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "build_variants", "dev.so")
if "CLSIMHIP_LIB" not in os.environ:
    if not os.path.exists(_dev):
        raise SystemExit("this tool needs the developer build: run tools/build_variant.sh dev")
    os.environ["CLSIMHIP_LIB"] = _dev
