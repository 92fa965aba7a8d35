"""Import FIRST in a tools script that needs the timing ablations / A/B switches of the tools build:
points ldm_tf2_amd._lib at libldm_hip_tools.so (`make tools`, -DLDM_TOOLS_BUILD).  The product
library libldm_hip.so has none of them compiled in."""
import os

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_PATH = os.path.join(_ROOT, "ldm_tf2_amd", "lib", "libldm_hip_tools.so")
if not os.path.exists(_PATH):
  raise SystemExit(f"{_PATH} is missing: run `make tools` first")
os.environ["LDM_HIP_LIB"] = _PATH
