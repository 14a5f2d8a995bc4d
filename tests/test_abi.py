"""CPU tests of the drop-in boundary: librfx.so builds for gfx950, loads, and exports exactly the
symbols include/rfx.h declares; argument validation paths that need no GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "rfx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rfx_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_every_declared_symbol():
    from remixfusion_amd import _lib
    from remixfusion_amd.build import build_library
    path = build_library()
    assert os.path.exists(path)
    lib = _lib.load()
    declared = header_functions()
    assert len(declared) >= 25
    for fn in declared:
        assert hasattr(lib, fn), f"{fn} declared in include/rfx.h but not exported"
        assert fn in _lib.PROTOTYPES, f"{fn} has no ctypes prototype"
    assert sorted(_lib.PROTOTYPES) == declared
    assert lib.rfx_abi_version() == 10


def test_struct_layouts_match_header():
    from remixfusion_amd import _lib
    assert C.sizeof(_lib.GridDesc) == 8 + 5 * 16 * 4
    assert C.sizeof(_lib.SamplerDesc) == 24
    assert C.sizeof(_lib.FieldDesc) == C.sizeof(_lib.GridDesc) + 8 + 8 + 8 + 4 * 8 + 4 * 4 + 2 * 4 + 8


def test_argument_validation_without_gpu():
    from remixfusion_amd import _lib
    lib = _lib.load()
    assert lib.rfx_tsdf_integrate_workspace_bytes(800, 800, 600, 480, 640) >= 256 + 480 * 640 * 8 + 800 * 800 * 10 * 8
    assert lib.rfx_tsdf_integrate_workspace_bytes(8, 8, 8, 0, 5) == 0 and lib.rfx_tsdf_integrate_workspace_bytes(0, 8, 8, 4, 5) == 0
    assert lib.rfx_field_backward_workspace_bytes(0) == 0 and lib.rfx_field_backward_workspace_bytes(1000) > 1000 * 368 * 4
    z3, z6, z9, z16 = _lib.farr(_lib._F3, [0] * 3), _lib.farr(_lib._F6, [0] * 6), _lib.farr(_lib._F9, [0] * 9), _lib.farr(_lib._F16, [0] * 16)
    assert lib.rfx_tsdf_fill(None, None, None, 10, None) == -1                      # RFX_ERR_ARG
    assert lib.rfx_tsdf_integrate(None, None, None, 4, 4, 4, z3, 0.1, z9, z16, None, None, 4, 4, 0.1, 1.0, 1, 0, z6, 0,
                                  None, 0, None) == -1
    assert lib.rfx_gbv_clear(None, 8, None) == -1
    assert lib.rfx_sample_z(None, None, None, 4, None, None) == -1
    assert lib.rfx_field_forward(None, None, 0, None, None) == 0                     # empty input is a no-op
    s = _lib.SamplerDesc(near=0.1, far=5.0, range_d=0.1, n_range_d=100, n_samples_d=100, perturb=0.0)
    assert lib.rfx_sample_z(C.byref(s), None, None, 0, None, None) == 0
    fake = C.c_void_p(16)
    assert lib.rfx_sample_z(C.byref(s), fake, None, 4, fake, None) == -3             # S > 128: RFX_ERR_UNSUPPORTED
    # a grid level of 4 GiB or more cannot be addressed by the lookups' 32-bit byte offsets: refused, not wrapped (round 5)
    g = _lib.GridDesc()
    g.n_levels, g.n_feat = 1, 4
    g.scale[0], g.res[0], g.size[0], g.offset[0], g.hashed[0] = 699.0, 700, 700 ** 3, 0, 0      # 700^3 float4 entries = 5.5 GB
    assert lib.rfx_grid_encode_forward(C.byref(g), fake, fake, 4, fake, None) == -3
    g.n_feat, g.size[0] = 2, 1 << 29                                                        # 2^29 float2 entries = 4 GiB
    assert lib.rfx_grid_encode_backward(C.byref(g), fake, fake, 4, fake, fake, None, None, 0, None) == -3


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from remixfusion_amd import _lib
    from remixfusion_amd.config import synthetic_config
    from remixfusion_amd.model.Volume import moving_volume

    class T:
        kfx = kfy = kfz = 0.0
        first = 0
    with pytest.raises(_lib.RfxError):
        moving_volume(synthetic_config("office0"), T(), np.eye(4))
    with pytest.raises(_lib.RfxError):
        _lib.ptr(torch.zeros(4))                                                     # CPU tensors are rejected
    # and nothing in the product imports the oracle
    import subprocess
    import sys
    out = subprocess.run([sys.executable, "-c", "import sys; import remixfusion_amd.pipeline, remixfusion_amd.mp_slam.mapper;"
                          "print(any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules))"],
                         capture_output=True, text=True, cwd=ROOT)
    assert out.stdout.strip() == "False", out.stderr
