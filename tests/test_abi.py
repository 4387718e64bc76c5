"""The C-ABI library loads and exports every symbol include/spot_hip.h declares (no GPU needed)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "spot_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(spot_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from spot_amd import lib
    assert os.path.exists(lib.LIB_PATH), "build the library first: make -C spot_amd/csrc"
    so = ctypes.CDLL(lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(so, n), "missing export: " + n


def test_binding_covers_header_and_version():
    from spot_amd import lib
    l = lib.load()
    assert l.spot_version() == 100
    bound = set(lib.SIGNATURES) | {"spot_version", "spot_last_error", "spot_layernorm_bwd_ws_floats", "spot_batchnorm_ws_floats",
                                        "spot_conv3x3_winograd_ws_floats", "spot_conv3x3_winograd_wgrad_ws_floats",
                                        "spot_deform_col2im_ws_floats", "spot_clip_grad_norm_ws_floats", "spot_transt_loss_ws_floats", "spot_mha_fwd_splitk_ws_floats", "spot_mha_drop_bits_words", "spot_conv3x3_winograd_general_ws_floats",
                                        "spot_conv3x3_winograd_general_v_floats"}
    assert bound == set(declared_symbols())


def test_argument_errors_are_reported_without_a_gpu():
    from spot_amd import lib
    l = lib.load()
    rc = l.spot_decode_teacher_f32(None, None, None, None, 0.4, 256, 256, None, None, None, None, 4, 1024, None)
    assert rc < 0
    assert b"null pointer" in l.spot_last_error()
    rc = l.spot_mha_fwd_f32(1, 0, 1, 0, 1, 0, 1, 0, None, None, 1, 16, 16, 2, 64, 0.1, 0.0, 0, None, None)
    assert rc < 0 and b"head_dim" in l.spot_last_error()
