"""ctypes binding of libfounddiff_hip.so (the C ABI declared in include/founddiff_hip.h).

The product path has NO fallback: if the shared library is missing or a call fails, we raise.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FOUNDDIFF_LIB") or os.path.join(HERE, "lib", "libfounddiff_hip.so")   # override: A/B of two builds
LIB_F16_PATH = os.environ.get("FOUNDDIFF_LIB_F16") or os.path.join(HERE, "lib", "libfounddiff_hip_f16.so")

FD_F32, FD_BF16 = 0, 1
FD_OPT_LOW_LATENCY = 0x100
FD_OPT_F32_SPLIT = 0x200
EPI_NONE, EPI_SILU_SPLIT, EPI_RELU, EPI_GATE_RES, EPI_RES_RELU, EPI_GNSILU_ADD, EPI_GNSILU_ADD_FINAL = range(7)
ACT_NONE, ACT_SILU, ACT_GELU, ACT_RELU = range(4)
PRO_NONE, PRO_LN_MOD, PRO_LN_GATE, PRO_LN_GATE_ZRE = range(4)

vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float


class ConvParams(C.Structure):
    _fields_ = [
        ("dtype", i32), ("out_f32", i32),
        ("in0", vp), ("in1", vp),
        ("c0", i32), ("ld0", i32), ("off0", i32),
        ("c1", i32), ("ld1", i32), ("off1", i32),
        ("B", i32), ("H", i32), ("W", i32),
        ("upsample", i32),
        ("KH", i32), ("KW", i32), ("stride", i32), ("pad_h", i32), ("pad_w", i32),
        ("OH", i32), ("OW", i32),
        ("ndir", i32),
        ("weight", vp), ("w_batch_stride", i64), ("w_dir_stride", i64),
        ("bias", vp),
        ("Cout", i32),
        ("out", vp), ("ldo", i32), ("offo", i32), ("out_dir_stride", i64),
        ("epilogue", i32), ("epi_split", i32),
        ("res", vp), ("ld_res", i32), ("off_res", i32),
        ("gate", vp), ("gate_ld", i32),
        ("h", vp), ("gn_mean_rstd", vp), ("gn_gamma", vp), ("gn_beta", vp), ("gn_groups", i32),
        ("stats_partial", vp),
        ("prologue", i32), ("ln_eps", f32),
        ("ln_gamma", vp), ("ln_beta", vp), ("ln_shift", vp), ("ln_scale", vp), ("ln_ld", i32),
        ("ln_z", vp), ("ln_ldz", i32), ("ln_offz", i32),
        ("weight_f8", vp), ("w_scale", vp), ("act_scale", f32), ("f32_split", i32),
        ("fin_w", vp), ("fin_b", f32), ("fin_out", vp), ("fin_mode", i32), ("fin_last", i32), ("fin_img", vp),
        ("fin_xin", vp), ("fin_alpha", f32),
        ("zre_w", vp), ("zre_gamma", vp), ("zre_beta", vp), ("zre_shift", vp), ("zre_scale", vp), ("zre_ld", i32),
        ("zre_eps", f32),
        ("weight_up2x", vp),
        ("weight_split_hi", vp), ("weight_split_lo", vp),
        ("weight_up2x_split_hi", vp), ("weight_up2x_split_lo", vp),
    ]


# name -> (restype, argtypes); every symbol include/founddiff_hip.h declares
SIGNATURES = {
    "fd_version": (i32, []),
    "fd_last_error": (C.c_char_p, []),
    "fd_dev_options": (C.c_char_p, []),
    "fd_conv_mtiles": (i32, [i32, i32]),
    "fd_conv2d": (i32, [C.POINTER(ConvParams), vp]),
    "fd_conv_prologue_ok": (i32, [C.POINTER(ConvParams)]),
    "fd_conv_kernel_id": (i32, [C.POINTER(ConvParams)]),
    "fd_conv_fp8_ok": (i32, [C.POINTER(ConvParams)]),
    "fd_gn_finalize": (i32, [vp, i32, i32, i32, i32, i64, f32, vp, vp]),
    "fd_gn_silu_apply": (i32, [i32, vp, vp, vp, vp, vp, vp, i32, i64, i32, i32, vp]),
    "fd_gn_apply_down4x4_ok": (i32, [i32, i32, i32, i32, i32]),
    "fd_gn_apply_down4x4": (i32, [i32, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "fd_ln_modulate": (i32, [i32, vp, vp, vp, f32, vp, vp, i32, vp, i32, i64, i32, vp]),
    "fd_ln_gate": (i32, [i32, vp, vp, vp, f32, vp, i32, i32, vp, i32, vp, i32, i64, i32, vp]),
    "fd_dwconv3x3": (i32, [i32, vp, i32, i32, vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, vp]),
    "fd_pw_dw3x3_ok": (i32, [i32, i32, i32, i32, i32, i32]),
    "fd_pw_dw3x3": (i32, [i32, vp, i32, i32, i32, vp, vp, f32, vp, vp, i32, vp, i32, vp, vp, i32, vp, i32, i32,
                          i32, vp, i32, i32, i32, i32, i32, vp]),
    "fd_pw_dw3x3_gram_ok": (i32, [i32, i32, i32, i32]),
    "fd_pw_dw3x3_gram_nblk": (i32, [i32, i32]),
    "fd_pw_dw3x3_gram_nblk_opts": (i32, [i32, i32, i32]),
    "fd_pw_dw3x3_gram": (i32, [i32, vp, i32, i32, i32, vp, vp, f32, vp, vp, i32, vp, vp, vp, i32, i32, vp, i32, i32, i32, vp]),
    "fd_pw_dw3x3_proj_ok": (i32, [i32, i32, i32, i32]),
    "fd_pw_dw3x3_proj": (i32, [i32, vp, i32, i32, i32, vp, vp, f32, vp, vp, i32, vp, vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, vp]),
    "fd_pw_dw3x3_f32_ok": (i32, [i32, i32, i32, i32, i32]),
    "fd_pw_dw3x3_f32": (i32, [vp, i32, i32, i32, vp, vp, f32, vp, vp, i32, vp, vp, i32, vp, vp, i32, vp, i32, i32, i32, i32, i32, vp]),
    "fd_pw_dw3x3_gram_f32_ok": (i32, [i32, i32, i32, i32]),
    "fd_pw_dw3x3_gram_f32_nblk": (i32, [i32, i32]),
    "fd_pw_dw3x3_gram_f32": (i32, [vp, i32, i32, i32, vp, vp, f32, vp, vp, i32, vp, vp, vp, i32, vp, i32, i32, i32, vp]),
    "fd_pw_dw3x3_proj_f32_ok": (i32, [i32, i32, i32, i32]),
    "fd_pw_dw3x3_proj_f32": (i32, [vp, i32, i32, i32, vp, vp, f32, vp, vp, i32, vp, vp, vp, i32, vp, vp, i32, vp, i32, i32, i32, i32, i32, vp]),
    "fd_dwconv_gram_ok": (i32, [i32, i32, i32, i32]),
    "fd_dwconv_gram_nblk": (i32, [i32, i32]),
    "fd_dwconv_gram": (i32, [i32, vp, i32, i32, vp, vp, i32, i32, i32, vp]),
    "fd_scan_ws_floats": (i64, [i32, i32, i32, i32, i32]),
    "fd_selective_scan": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "fd_selective_scan_fuses_xproj": (i32, [i32, i32, i32, i32]),
    "fd_selective_scan_plan": (i32, [i32, i32, i32, i32, i32, i32]),
    "fd_selective_scan_xproj": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "fd_selective_scan_fwd_f32": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i64, vp, vp, vp]),
    "fd_chan_attn_nblk": (i32, [i64]),
    "fd_chan_attn_gram": (i32, [i32, vp, i32, i64, i32, vp, vp]),
    "fd_chan_attn_weff": (i32, [i32, vp, i32, vp, vp, vp, i32, i32, vp]),
    "fd_linear": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "fd_sinusoidal": (i32, [vp, vp, i32, i32, vp]),
    "fd_softmax_mul": (i32, [vp, vp, vp, i32, i32, vp]),
    "fd_l2norm_rows": (i32, [vp, vp, i32, i32, f32, vp]),
    "fd_add_f32": (i32, [vp, vp, vp, i64, vp]),
    "fd_pack_planes": (i32, [i32, vp, vp, vp, i32, i64, i32, vp]),
    "fd_init_conv7_ok": (i32, [i32, i32, i32, i32]),
    "fd_init_conv7": (i32, [i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "fd_init_conv7_f32s": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "fd_pack_planes3": (i32, [i32, vp, vp, vp, vp, i32, i64, i32, vp]),
    "fd_cast": (i32, [i32, vp, i32, vp, i64, vp]),
    "fd_final_conv1": (i32, [i32, vp, vp, vp, vp, i64, i32, vp]),
    "fd_avgpool": (i32, [i32, vp, vp, i32, i32, i32, i32, i32, vp]),
    "fd_attnpool_tokens": (i32, [i32, vp, vp, i32, i64, i32, vp]),
    "fd_attnpool_core": (i32, [vp, i32, vp, i32, i32, i32, vp, i32, i32, i32, i32, vp]),
    "fd_res_predictions": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i64, vp]),
    "fd_res_ddim_step": (i32, [vp, vp, vp, vp, f32, f32, i32, vp, i64, vp]),
    "fd_res_step_obj": (i32, [i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i64, vp]),
    "fd_res_posterior_step": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i64, vp]),
    "fd_keyed_normal": (i32, [vp, i32, vp, i32, i64, vp]),
    "fd_ancestral_begin": (i32, [vp, vp, vp, i32, vp]),
    "fd_res_posterior_step_keyed": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i64, vp]),
    "fd_gn_film_silu_apply": (i32, [i32, vp, vp, vp, vp, vp, vp, i32, vp, i32, i64, i32, i32, vp]),
    "fd_chan_ln": (i32, [i32, vp, vp, vp, vp, i64, i32, vp]),
    "fd_linear_attention": (i32, [i32, vp, i32, i64, i32, vp, vp, vp, vp, i32, vp]),
    "fd_attention": (i32, [i32, vp, vp, i32, i64, i32, vp]),
    "fd_lincomb3": (i32, [vp, vp, vp, f32, f32, f32, i32, vp, i64, vp]),
    "fd_metrics_nblk": (i32, [i32, i32]),
    "fd_metrics": (i32, [vp, vp, i32, i32, i32, vp, vp, vp]),
    "fd_affine_f32": (i32, [vp, f32, f32, vp, i64, vp]),
    "fd_axpy_f32": (i32, [vp, vp, f32, vp, i64, vp]),
    "fd_half_format": (i32, []),
}

class FoundDiffHipError(RuntimeError):
    pass


TRACE = None   # set to a list to record (name, args) of every launch (bench.py roofline leg)


class Library:
    """One build of the C ABI, loaded on first use.  Two exist (founddiff_amd/build.py): the default one, whose 16-bit type is
    bfloat16, and the FD_HALF_F16 one (IEEE binary16 in its place; DAEngine mode 'fp16').  Same symbols, same signatures."""

    def __init__(self, path, half_format):
        self.path, self.half_format, self._lib = path, half_format, None

    def lib(self):
        """Load the HIP library (once).  Raises if it has not been built: there is no CPU path."""
        if self._lib is None:
            if not os.path.exists(self.path):
                raise FoundDiffHipError(
                    f"{self.path} not found: build it with `python -m founddiff_amd.build` "
                    "(hipcc --offload-arch=gfx950).  founddiff_amd has no CPU fallback.")
            L = C.CDLL(self.path)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(L, name)       # AttributeError if the .so does not export it
                fn.restype = res
                fn.argtypes = args
            got = L.fd_half_format()
            if got != self.half_format:
                raise FoundDiffHipError(f"{self.path}: fd_half_format() = {got}, expected {self.half_format} "
                                        "(0 = the bfloat16 build, 1 = the binary16 build)")
            self._lib = L
        return self._lib

    def check(self, rc, what=""):
        if rc != 0:
            msg = self.lib().fd_last_error().decode(errors="replace")
            raise FoundDiffHipError(f"{what} failed (rc={rc}): {msg}")

    def call(self, name, *args):
        if TRACE is not None:
            TRACE.append((name, args))
        self.check(getattr(self.lib(), name)(*args), name)


BF16 = Library(LIB_PATH, 0)
F16 = Library(LIB_F16_PATH, 1)
# the default library (every caller that is not an 'fp16' engine)
lib, check, call = BF16.lib, BF16.check, BF16.call
