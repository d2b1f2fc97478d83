"""ctypes binding of libeonerf_hip.so (include/eonerf_hip.h).

PyTorch is plumbing here: it owns device memory (tensor.data_ptr()) and the current HIP stream; every arithmetic
step of the hot path runs inside the HIP library.  There is NO fallback: if the library is missing, cannot be
loaded, or no GPU is present, the product path raises.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("EONERF_LIB") or os.path.join(CSRC, "libeonerf_hip.so")   # EONERF_LIB: A/B builds of the same ABI

EONERF_FP32, EONERF_BF16, EONERF_F16X3 = 0, 1, 2
PRECISIONS = {"fp32": EONERF_FP32, "bf16": EONERF_BF16, "fp16x3": EONERF_F16X3}
F_SHADOWS, F_EVAL, F_TRAIN, F_ONLY_DEPTH, F_RGB_LOSS = 1, 2, 4, 8, 16

SYMBOLS = ["eonerf_version", "eonerf_strerror", "eonerf_create", "eonerf_destroy", "eonerf_param_tensors",
           "eonerf_param_info", "eonerf_param_floats", "eonerf_set_weights", "eonerf_field_workspace_bytes",
           "eonerf_render_workspace_bytes", "eonerf_field_forward", "eonerf_query_density", "eonerf_render_forward",
           "eonerf_render_backward", "eonerf_adam_step", "eonerf_profile_enable", "eonerf_profile_read",
           "eonerf_train_loss", "eonerf_sample_rays", "eonerf_rendering", "eonerf_generate_rays",
           "eonerf_field_train_workspace_bytes", "eonerf_field_forward_train",
           "eonerf_field_backward", "eonerf_set_noise_seed", "eonerf_render_status", "eonerf_device_status", "eonerf_grad_floats",
           "eonerf_grad_seal", "eonerf_profile_name", "eonerf_adam_step_zero_grad", "eonerf_rendering_train", "eonerf_rendering_backward", "eonerf_clock_probe", "eonerf_range_status", "eonerf_set_n_samples", "eonerf_render_backward_loss",
           "eonerf_presample", "eonerf_presample_cancel", "eonerf_grad_early_floats", "eonerf_set_exchange_event"]


class EonerfRpc(C.Structure):
    _fields_ = [("col_num", C.c_double * 20), ("col_den", C.c_double * 20), ("row_num", C.c_double * 20), ("row_den", C.c_double * 20),
                ("row_offset", C.c_double), ("col_offset", C.c_double), ("lat_offset", C.c_double), ("lon_offset", C.c_double),
                ("alt_offset", C.c_double), ("row_scale", C.c_double), ("col_scale", C.c_double), ("lat_scale", C.c_double),
                ("lon_scale", C.c_double), ("alt_scale", C.c_double)]


class EonerfConfig(C.Structure):
    _fields_ = [("n_images", C.c_int), ("precision", C.c_int), ("n_samples", C.c_int), ("radiometric", C.c_int)]


def build(verbose=False):
    """Compile libeonerf_hip.so in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    # up to date?  Decided by CONTENT (a digest of every source, kept beside the library), not by time stamps: a snapshot copied to the
    # GPU box may carry arbitrary mtimes, and the object files do not travel -- a current library must never be rebuilt there
    import hashlib
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp", ".h")) or f == "Makefile")
    srcs.append(os.path.join(_HERE, "..", "include", "eonerf_hip.h"))
    h = hashlib.sha1()
    for f in srcs:
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read())
    digest, stamp = h.hexdigest(), os.path.join(CSRC, "libeonerf_hip.digest")

    def so_digest():
        with open(LIB_PATH, "rb") as fh:
            return hashlib.sha1(fh.read()).hexdigest()

    # the stamp names the sources AND the library build() itself produced from them with the Makefile's own flags: a library that was
    # rebuilt by hand afterwards (an ablation build: make CXXFLAGS=... -DEO_ABL=..) does not match it and is rebuilt from scratch
    recorded = open(stamp).read().split() if os.path.exists(stamp) else []
    if not os.environ.get("EONERF_LIB") and os.path.exists(LIB_PATH) and len(recorded) == 2 and recorded[0] == digest:
        if recorded[1] == so_digest():
            return LIB_PATH
        subprocess.run(["make", "-C", CSRC, "clean"], capture_output=True, text=True)
    cmd = ["make", "-C", CSRC, "-j", str(min(8, os.cpu_count() or 1))]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or r.returncode:
        print(r.stdout[-4000:])
        print(r.stderr[-8000:])
    if r.returncode:
        raise RuntimeError("building libeonerf_hip.so failed")
    if not os.environ.get("EONERF_LIB"):
        with open(stamp, "w") as fh:
            fh.write(digest + " " + so_digest() + "\n")
    return LIB_PATH


_lib = None


def lib():
    """Load the library (building it is an explicit step: __graft_entry__.build() / eonerf_code_amd._lib.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(the EO-NeRF hot path has no non-HIP fallback)")
    L = C.CDLL(LIB_PATH)
    vp, i, sz, fp = C.c_void_p, C.c_int, C.c_size_t, C.c_float
    L.eonerf_version.restype = i
    L.eonerf_strerror.restype = C.c_char_p
    L.eonerf_strerror.argtypes = [i]
    L.eonerf_create.argtypes = [C.POINTER(vp), C.POINTER(EonerfConfig)]
    L.eonerf_destroy.argtypes = [vp]
    L.eonerf_param_tensors.argtypes = [vp]
    L.eonerf_param_info.argtypes = [vp, i, C.POINTER(C.c_char_p), C.POINTER(sz), C.POINTER(i), C.POINTER(i)]
    L.eonerf_param_floats.restype = sz
    L.eonerf_param_floats.argtypes = [vp]
    L.eonerf_set_weights.argtypes = [vp, vp, vp]
    L.eonerf_field_workspace_bytes.restype = sz
    L.eonerf_field_workspace_bytes.argtypes = [vp, i]
    L.eonerf_render_workspace_bytes.restype = sz
    L.eonerf_render_workspace_bytes.argtypes = [vp, i, i]
    L.eonerf_field_forward.argtypes = [vp, vp, vp, vp, vp, i, vp, vp, vp, vp, vp, vp, sz, vp]
    L.eonerf_query_density.argtypes = [vp, vp, vp, i, vp, vp, sz, vp]
    L.eonerf_render_forward.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i, i, vp, vp, vp, sz, vp]
    L.eonerf_render_backward.argtypes = [vp, vp, vp, vp, i, i, vp, vp, vp, sz, vp]
    L.eonerf_presample.argtypes = [vp, vp, vp, vp, i, i, vp, vp, sz, vp]
    L.eonerf_presample_cancel.argtypes = [vp]
    L.eonerf_grad_early_floats.restype = sz
    L.eonerf_grad_early_floats.argtypes = [vp]
    L.eonerf_set_exchange_event.argtypes = [vp, vp, i]
    L.eonerf_render_backward_loss.argtypes = [vp, vp, vp, vp, i, i, vp, vp, i, vp, vp, vp, vp, sz, vp]
    L.eonerf_adam_step.argtypes = [vp, vp, vp, vp, vp, i, fp, fp, fp, fp, fp, vp, vp]
    L.eonerf_adam_step_zero_grad.argtypes = [vp, vp, vp, vp, vp, i, fp, fp, fp, fp, fp, vp, vp]
    L.eonerf_device_status.argtypes = [vp, vp]
    L.eonerf_range_status.argtypes = [vp, vp]
    L.eonerf_grad_floats.restype = sz
    L.eonerf_grad_floats.argtypes = [vp]
    L.eonerf_grad_seal.argtypes = [vp, vp, vp]
    L.eonerf_field_train_workspace_bytes.restype = sz
    L.eonerf_field_train_workspace_bytes.argtypes = [vp, i, i]
    L.eonerf_field_forward_train.argtypes = [vp, vp, vp, vp, vp, i, i, vp, vp, vp, vp, vp, vp, sz, vp]
    L.eonerf_field_backward.argtypes = [vp, vp, vp, i, i, vp, vp, vp, vp, vp, vp, vp, vp, sz, vp]
    L.eonerf_render_status.argtypes = [vp, i, i, vp, sz, vp]
    L.eonerf_train_loss.argtypes = [vp, vp, vp, i, i, vp, vp, vp]
    L.eonerf_generate_rays.argtypes = [C.POINTER(EonerfRpc), vp, vp, C.c_long, i, C.c_double, C.c_double, i, i, C.c_double, C.c_double,
                                       C.POINTER(fp), C.POINTER(fp), vp, vp, vp, vp]
    L.eonerf_sample_rays.argtypes = [vp, vp, vp, vp, i, i, vp, vp, vp, vp, vp, vp, sz, vp]
    L.eonerf_set_noise_seed.argtypes = [vp, C.c_uint64]
    L.eonerf_set_n_samples.argtypes = [vp, i]
    L.eonerf_rendering.argtypes = [vp, vp, vp, vp, vp, vp, vp, i, i, i, vp, vp, vp, vp, vp, vp, vp, sz, vp]
    L.eonerf_rendering_train.argtypes = [vp, vp, vp, vp, vp, vp, vp, i, i, i, vp, vp, vp, vp, vp, vp, vp, sz, vp]
    L.eonerf_rendering_backward.argtypes = [vp, vp, vp, vp, i, i, vp, vp, vp, vp, vp, vp, vp, sz, vp]
    L.eonerf_profile_enable.argtypes = [vp, i]
    L.eonerf_profile_read.argtypes = [vp, i, C.POINTER(fp), C.POINTER(i)]
    L.eonerf_clock_probe.argtypes = [vp, vp, vp]
    L.eonerf_profile_name.restype = C.c_char_p
    L.eonerf_profile_name.argtypes = [i]
    for name in SYMBOLS:
        getattr(L, name)
    _lib = L
    return L


E_RANGE = -6


def check(rc):
    if rc != 0:
        raise RuntimeError(f"libeonerf_hip: {lib().eonerf_strerror(rc).decode()} (code {rc})")


def param_layout(ctx):
    """[(name, offset, rows, cols)] of the flat fp32 parameter buffer."""
    L = lib()
    out = []
    for k in range(L.eonerf_param_tensors(ctx)):
        name, off, r, c = C.c_char_p(), C.c_size_t(), C.c_int(), C.c_int()
        check(L.eonerf_param_info(ctx, k, C.byref(name), C.byref(off), C.byref(r), C.byref(c)))
        out.append((name.value.decode(), off.value, r.value, c.value))
    return out
