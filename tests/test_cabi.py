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
    assert L.lgm_abi_version() == _lib.ABI_VERSION == 2        # the loader refuses a library built for another ABI
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


def test_geometry_validation_rejects_inconsistent_shapes():
    L = _lib.lib()
    g = _lib.ConvGeom(1, 8, 8, 4, 9, 8, 4, 3, 3, 1, 1)   # Ho should be 8
    with pytest.raises(_lib.LgmError):
        L.lgm_conv_xy(ctypes.byref(g), 16, 4, 16, None, None, 0, 16, 4, None, 0, None)
    assert L.lgm_conv_wgrad_workspace(ctypes.byref(g)) == -1
