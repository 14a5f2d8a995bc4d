"""dev helper: build an A/B variant of librfx.so with extra -D flags into build/variants/ (git-ignored, travels with
gpurun).  usage: python tools/build_variant.py NAME -DFOO=1 -DBAR=2 ...   then   RFX_LIB_PATH=build/variants/librfx_NAME.so"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from remixfusion_amd.build import build_library
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = os.path.join(root, "build", "variants")
os.makedirs(d, exist_ok=True)
name, flags = sys.argv[1], sys.argv[2:]
print(build_library(extra=flags, out=os.path.join(d, f"librfx_{name}.so")))
