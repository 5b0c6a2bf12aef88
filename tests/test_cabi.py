"""CPU: the C-ABI library builds/loads and exports every symbol include/lgm_hip.h declares
(no compute calls — there is no GPU in the build container)."""
import ctypes
import os

import pytest

from lgm_hip import _lib


def test_header_parses_and_library_exports_all_symbols():
    protos = _lib.parse_header()
    assert len(protos) >= 30
    for must in ("lgm_conv_xy", "lgm_conv_yx", "lgm_conv_wgrad", "lgm_gn_fwd", "lgm_gn_bwd", "lgm_linattn_fwd",
                 "lgm_attn_bwd", "lgm_adam_step", "lgm_ema_lerp", "lgm_sample_step", "lgm_qsample_target",
                 "lgm_gn_fwd_planes", "lgm_gn_bwd_planes", "lgm_conv3x3_wino_partial", "lgm_conv3x3_wino_fits",
                 "lgm_linattn_bwd_fused", "lgm_linattn_fwd_fused", "lgm_rms_qkv_fused", "lgm_linattn_bwd_deferred",
                 "lgm_attn_bwd_deferred", "lgm_conv_bwd_pair_post", "lgm_colsum_deferred", "lgm_tanh_mse_fwd",
                 "lgm_tanh_mse_bwd", "lgm_vqvae_loss_samples", "lgm_resstack_fwd"):
        assert must in protos
    assert os.path.exists(_lib.LIB_PATH), "run `python __graft_entry__.py build` first"
    dll = ctypes.CDLL(_lib.LIB_PATH)
    for name in protos:
        assert hasattr(dll, name), f"{name} declared in include/lgm_hip.h but not exported"


def test_abi_version_and_error_string():
    L = _lib.lib()
    assert L.lgm_abi_version() == _lib.ABI_VERSION == 7        # the loader refuses a library built for another ABI
    # invalid-argument path works without a GPU: null geometry is rejected before any launch
    with pytest.raises(_lib.LgmError) as e:
        L.lgm_conv_xy(None, None, 0, None, None, None, 0, None, 0, None, 0, None)
    assert "conv" in str(e.value)
    # the size queries of the fused kernels answer without a GPU
    assert L.lgm_linattn_bwd_fused_supported(4, 32, 64) == 1 and L.lgm_linattn_bwd_fused_supported(4, 32, 128) == 0
    assert L.lgm_linattn_fwd_fused_supported(4, 32, 64) == 2 and L.lgm_linattn_fwd_fused_supported(4, 32, 256) == 1
    assert L.lgm_rms_qkv_fused_supported(64, 384) == 2 and L.lgm_rms_qkv_fused_supported(512, 384) == 0
    assert L.lgm_resstack_fwd_supported(4, 4, 128, 128, 32, 2) == 1 and L.lgm_resstack_fwd_supported(8, 8, 128, 128, 32, 2) == 0
    assert L.lgm_linattn_bwd_fused_slabs(128, 1024, 64) == 256 * 384 * 64 * 4


def test_cu_margin_changes_the_launch_plans_without_a_gpu():
    """lgm_set_cu_margin / lgm_cu_margin (host-only): the margin is read back, a plan query answers for 256 - margin workgroup
    slots, more than half of the chip is refused and -1 returns to the environment's default."""
    L = _lib.lib()
    base = L.lgm_cu_margin()
    g = _lib.ConvGeom(64, 32, 32, 64, 32, 32, 64, 3, 3, 1, 1)    # 64 -> 64 at 32 x 32, B = 64: the slab count fills the slots
    try:
        L.lgm_set_cu_margin(0)
        w0 = L.lgm_conv_wgrad_workspace(ctypes.byref(g))
        L.lgm_set_cu_margin(16)
        w16 = L.lgm_conv_wgrad_workspace(ctypes.byref(g))
        L.lgm_set_cu_margin(64)
        assert L.lgm_cu_margin() == 64
        w64 = L.lgm_conv_wgrad_workspace(ctypes.byref(g))
        assert w0 > w16 > w64 > 0                                  # fewer weight-gradient slabs fit 240 / 192 slots than 256
        with pytest.raises(_lib.LgmArgumentError):
            L.lgm_set_cu_margin(129)
        assert L.lgm_cu_margin() == 64
    finally:
        L.lgm_set_cu_margin(-1)
    assert L.lgm_cu_margin() == base


def test_geometry_validation_rejects_inconsistent_shapes():
    L = _lib.lib()
    g = _lib.ConvGeom(1, 8, 8, 4, 9, 8, 4, 3, 3, 1, 1)   # Ho should be 8
    with pytest.raises(_lib.LgmError):
        L.lgm_conv_xy(ctypes.byref(g), 16, 4, 16, None, None, 0, 16, 4, None, 0, None)
    assert L.lgm_conv_wgrad_workspace(ctypes.byref(g)) == -1


def _kernel_symbols():
    """Demangled names of the kernels of liblgm_hip.so, normalised as tools/pmc_kernels.py normalises rocprofv3's rows
    (no return type, no anonymous-namespace qualifier, no argument list).  The host side of a HIP library holds one
    handle object per kernel under the kernel's own mangled name (and a `__device_stub__` function)."""
    import re
    import subprocess
    out = subprocess.run(["nm", "-C", "--defined-only", _lib.LIB_PATH], check=True, capture_output=True, text=True).stdout
    names = set()
    for ln in out.splitlines():
        m = re.match(r"^[0-9a-f]+ \w (.*)$", ln)
        if not m or "__device_stub__" not in m.group(1):
            continue
        k = m.group(1).replace("__device_stub__", "").replace("(anonymous namespace)::", "")
        k = re.sub(r"^void ", "", k)
        names.add(re.sub(r"\(.*$", "", k).strip())
    return names


def test_every_noted_kernel_name_is_a_kernel_of_the_library():
    """`lgm_last_kernel()` is how bench.py attributes HIP-event spans to rocprofv3 rows and looks the
    dominant kernel up in profiles/*_pmc_traffic.json.  A kernel that gained a template argument left a stale name behind
    for a whole round.  Every name the library can note (the LGM_KNAME registry) must be a prefix of a real kernel name,
    and a name that carries template arguments must match one exactly."""
    L = _lib.lib()
    n = L._dll.lgm_kernel_name_count()
    assert n >= 40
    L._dll.lgm_kernel_name.restype = ctypes.c_char_p
    noted = sorted({L._dll.lgm_kernel_name(i).decode() for i in range(n)})
    kernels = _kernel_symbols()
    assert len(kernels) > 100
    for name in noted:
        if name.endswith(">"):
            assert name in kernels, f"noted kernel name {name!r} is not a kernel of liblgm_hip.so"
        else:
            assert any(k == name or k.startswith(name + "<") for k in kernels), \
                f"noted kernel name {name!r} is not a prefix of any kernel of liblgm_hip.so"
    assert "lgmwino4::wino4_conv_kernel<0, false, 0, false>" in noted
    assert L._dll.lgm_kernel_name(n) is None and L._dll.lgm_kernel_name(-1) is None
