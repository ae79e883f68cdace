"""Build + load libradzero_hip.so (the C-ABI of include/radzero_hip.h) through ctypes.

There is NO fallback: if the library is missing or fails to load, importing the product path raises.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
EXP_SRC = os.path.join(PKG_DIR, "..", "tools", "experiments")     # retired experiments' sources: tools build only, never product
# RZ_EXPERIMENTS=1: the TOOLS build (-DRZ_EXPERIMENTS: in-kernel stamp builds of gemm7 / gemm8, the retired attention shapes of rounds 1-2),
# its own library and object directory; the product and every test use the plain build.
EXPERIMENTS = os.environ.get("RZ_EXPERIMENTS") == "1"
LIB_PATH = os.environ.get("RZ_LIB_PATH") or os.path.join(PKG_DIR, "libradzero_hip_experiments.so" if EXPERIMENTS else "libradzero_hip.so")   # RZ_LIB_PATH: A/B of two builds in one gpurun call
SOURCES = ["gemm.hip", "gemm7.hip", "gemm8.hip", "attention.hip", "rowops.hip", "vlcabs.hip", "preprocess.hip", "api.hip"]
if EXPERIMENTS:
    # retired GEMM experiments, never faster than gemm8 inside the step: gemm10 / gemm11 (round 3: other K loops), gemm12 (round 4: two
    # 256x128 workgroups per CU so that epilogues overlap K loops; profiles/r04/gemm12_*.log).  They live in tools/experiments/ (with the
    # generated loops attention.hip's and gemm10.hip's experiment branches include) and are found through -I
    SOURCES += ["gemm10.hip", "gemm11.hip", "gemm12.hip"]
# attention.hip: fmaxf chains on MFMA outputs fuse into v_max3_f32 without a canonicalising v_max each
EXTRA_FLAGS = {"attention.hip": ["-fno-honor-nans"]}
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

RZ_F32, RZ_BF16, RZ_F16 = 0, 1, 2
PROF_FAMILIES = ("attn", "gemm", "rowops", "vlcabs", "post")


def _src_path(src: str) -> str:
    p = os.path.join(CSRC, src)
    return p if os.path.exists(p) or not EXPERIMENTS else os.path.join(EXP_SRC, src)


def _all_deps() -> list:
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(PKG_DIR, "..", "include", "radzero_hip.h")]
    if EXPERIMENTS and os.path.isdir(EXP_SRC):
        deps += [os.path.join(EXP_SRC, f) for f in os.listdir(EXP_SRC)]
    return deps


STAMP_PATH = LIB_PATH + ".stamp"


def _source_digest() -> str:
    """sha256 over the CONTENTS of every source / header the library is built from (+ the build flavour and extra flags): the rebuild decision must not
    depend on file times — a snapshot copied to another machine (gpurun, the driver's round-end run) keeps contents, not necessarily mtime order."""
    import hashlib
    h = hashlib.sha256()
    h.update(("experiments" if EXPERIMENTS else "product").encode() + os.environ.get("RZ_CXXFLAGS", "").encode())
    for d in sorted(os.path.abspath(p) for p in _all_deps() if os.path.isfile(p)):
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _needs_rebuild() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    try:
        return open(STAMP_PATH).read().strip() != _source_digest()
    except OSError:
        pass
    # a library without a stamp (built by an older tree): fall back to file times
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(d) > t for d in _all_deps() if os.path.exists(d))


def build(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 every source, link in-tree (the .so travels with the repo snapshot).

    Safe under `torch.distributed.run` (every rank imports this module at the same time): the whole build runs under an
    exclusive `flock` on radzero_amd/build/.lock, objects and the library are written to per-process temporary names and
    moved into place with `os.replace`, and whoever gets the lock second finds an up-to-date library and does nothing.
    A reader therefore never maps a half-written file."""
    if not force and not _needs_rebuild():
        return LIB_PATH
    if not os.path.exists(HIPCC):
        raise RuntimeError(f"hipcc not found at {HIPCC}; cannot build libradzero_hip.so")
    import fcntl
    obj_dir = os.path.join(PKG_DIR, "build_experiments" if EXPERIMENTS else "build")
    if os.environ.get("RZ_CXXFLAGS", "").strip():        # an A/B build: its objects must never be mistaken for the product's (they are reused by file time alone)
        import hashlib
        obj_dir += "_" + hashlib.sha256(os.environ["RZ_CXXFLAGS"].encode()).hexdigest()[:10]
    os.makedirs(obj_dir, exist_ok=True)
    with open(os.path.join(obj_dir, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _needs_rebuild():       # another process built it while this one waited
                return LIB_PATH
            return _build_locked(obj_dir, verbose, force)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(obj_dir: str, verbose: bool, force: bool = False) -> str:
    flags = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Werror=uninitialized",
             "-Werror=return-type", f"-I{CSRC}"] + (["-DRZ_EXPERIMENTS", f"-I{EXP_SRC}"] if EXPERIMENTS else []) + os.environ.get("RZ_CXXFLAGS", "").split()   # RZ_CXXFLAGS: A/B builds of a tunable (with RZ_LIB_PATH)
    tag = f".{os.getpid()}.tmp"

    def cc(src):
        obj = os.path.join(obj_dir, src.replace(".hip", ".o"))
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > _src_mtime(src):
            return obj                                   # unchanged translation unit: keep its object
        cmd = [HIPCC, *flags, *EXTRA_FLAGS.get(src, []), "-c", _src_path(src), "-o", obj + tag]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
        if r.stderr and (verbose or "-Wuninitialized" in r.stderr or "-Wreturn-type" in r.stderr or "-Wsometimes-uninitialized" in r.stderr):
            print(r.stderr, file=sys.stderr)         # these warnings have been real bugs: never hide them
        os.replace(obj + tag, obj)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(cc, SOURCES))
    r = subprocess.run([HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB_PATH + tag, *objs],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    os.replace(LIB_PATH + tag, LIB_PATH)
    with open(STAMP_PATH + tag, "w") as f:
        f.write(_source_digest() + "\n")
    os.replace(STAMP_PATH + tag, STAMP_PATH)
    return LIB_PATH


def _src_mtime(src: str) -> float:
    """Newest mtime of a translation unit and every header it may include."""
    deps = [_src_path(src)] + [d for d in _all_deps() if d.endswith((".h", ".inc"))]
    return max(os.path.getmtime(d) for d in deps if os.path.exists(d))


class RzImageDesc(ctypes.Structure):
    """rz_image_desc (include/radzero_hip.h)"""
    _fields_ = [("image_dev", ctypes.c_void_p), ("src_dtype", ctypes.c_int32), ("height", ctypes.c_int32), ("width", ctypes.c_int32),
                ("channels", ctypes.c_int32), ("pad_left", ctypes.c_int32), ("pad_top", ctypes.c_int32), ("padded_height", ctypes.c_int32),
                ("padded_width", ctypes.c_int32), ("bounds_h_dev", ctypes.c_void_p), ("coeffs_h_dev", ctypes.c_void_p), ("ksize_h", ctypes.c_int32),
                ("bounds_v_dev", ctypes.c_void_p), ("coeffs_v_dev", ctypes.c_void_p), ("ksize_v", ctypes.c_int32)]


class RzConfig(ctypes.Structure):
    _fields_ = [
        ("compute_dtype", ctypes.c_int32), ("hidden_size", ctypes.c_int32), ("num_attention_heads", ctypes.c_int32),
        ("mlp_ratio", ctypes.c_int32), ("patch_size", ctypes.c_int32), ("num_channels", ctypes.c_int32),
        ("vit_layers", ctypes.c_int32), ("align_layers", ctypes.c_int32), ("vit_layer_norm_eps", ctypes.c_float),
        ("vocab_size", ctypes.c_int32), ("max_position_embeddings", ctypes.c_int32), ("text_layers", ctypes.c_int32),
        ("text_intermediate_size", ctypes.c_int32), ("text_layer_norm_eps", ctypes.c_float),
        ("pad_token_id", ctypes.c_int32), ("shared_layer_norm_eps", ctypes.c_float),
    ]


# every symbol include/radzero_hip.h declares: name -> (restype, argtypes)
_P, _I, _L, _F = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
SYMBOLS = {
    "rz_last_error": (ctypes.c_char_p, []),
    "rz_version": (ctypes.c_char_p, []),
    "rz_create": (_I, [ctypes.POINTER(RzConfig), ctypes.POINTER(_P)]),
    "rz_destroy": (_I, [_P]),
    "rz_load_weight": (_I, [_P, ctypes.c_char_p, _P, _L]),
    "rz_weights_ready": (_I, [_P]),
    "rz_set_position_table": (_I, [_P, _I, _I, _P]),
    "rz_reserve": (_I, [_P, _I, _I, _I, _I]),
    "rz_padded_tokens": (_I, [_P, _I, _I, ctypes.POINTER(_I)]),
    "rz_vision_forward": (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    "rz_text_forward": (_I, [_P, _P, _P, _I, _I, _P, _P, _P]),
    "rz_vlcabs": (_I, [_P, _P, _I, _I, _P, _P, _P, _P]),
    "rz_upsample_maps": (_I, [_P, _P, _L, _I, _I, _I, _I, _I, _P, _P]),
    "rz_grounding_points": (_I, [_P, _P, _L, _I, _I, _I, _I, _P, _P, _P]),
    "rz_grounding_points_ex": (_I, [_P, _P, _L, _I, _I, _I, _I, _I, _P, _P, _P]),
    "rz_upsample_maps_ex": (_I, [_P, _P, _L, _I, _I, _I, _I, _I, _I, _P, _P]),
    "rz_preprocess_image": (_I, [_P, _I, _I, _I, _I, _I, _P, _P, _I, _P, _P, _I, _P, _P, _F, _I, _P, _P, _P]),
    "rz_preprocess_batch_workspace": (ctypes.c_size_t, [ctypes.POINTER(RzImageDesc), _I, _I]),
    "rz_preprocess_batch": (_I, [ctypes.POINTER(RzImageDesc), _I, _I, _P, _P, _F, _I, _P, ctypes.c_size_t, _P, _P]),
    "rz_gemm": (_I, [_I, _I, _P, _P, _P, _P, _I, _I, _I, _P]),
    "rz_gemm_ex": (_I, [_I, _I, _P, _L, _P, _L, _P, _P, _L, _P, _P, _L, _I, _I, _I, _I, _I, _P]),
    "rz_gemm_qkv": (_I, [_I, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P]),
    "rz_gemm_f32_split": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "rz_layernorm": (_I, [_I, _P, _P, _P, _F, _P, _P, _L, _I, _P]),
    "rz_flash_attention": (_I, [_I, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "rz_flash_attention_split_workspace": (ctypes.c_size_t, [_I, _I, _I]),
    "rz_flash_attention_f32_split": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "rz_flash_attention_f32_mx": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "rz_text_embed_ln": (_I, [_I, _P, _P, _P, _P, _P, _F, _P, _P, _I, _I, _I, _I, _I, _P]),
    "rz_text_attention": (_I, [_I, _P, _P, _P, _P, _I, _I, _I, _P]),
    "rz_masked_meanpool": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "rz_rows_dot": (_I, [_P, _L, _P, _L, _P, _P, _I, _I, _I, _I, _L, _L, _L, _P]),
    "rz_image_features": (_I, [_P, _L, _I, _I, _I, _P, _P]),
    "rz_patch_embed": (_I, [_I, _P, _I, _I, _I, _I, _I, _P, _I, _P, _I, _P, _P, _P]),
    "rz_set_option": (_I, [ctypes.c_char_p, _I]),
    "rz_set_model_option": (_I, [_P, ctypes.c_char_p, _I]),
    "rz_get_model_option": (_I, [_P, ctypes.c_char_p, ctypes.POINTER(_I)]),
    "rz_debug_buffer": (_I, [ctypes.c_char_p, _P]),
    "rz_profile_enable": (_I, [_P, _I]),
    "rz_profile_read": (_I, [_P, _P, _P]),
}

_lib = None


def load(auto_build: bool = True) -> ctypes.CDLL:
    """Load the shared library (building it first if the sources are newer and hipcc is present)."""
    global _lib
    if _lib is not None:
        return _lib
    # torch FIRST: its wheel bundles its own libamdhip64.so.7 / libhsa-runtime64.so.1 (same SONAMEs as /opt/rocm's).  Whichever copy is
    # mapped first serves the whole process; if this library pulled in /opt/rocm's before torch initialised the GPU through its own, the
    # process would hold two HIP runtimes and rz_create found no device (seen with `python __graft_entry__.py smoke`: build() then smoke()).
    import torch  # noqa: F401
    if auto_build and os.path.exists(HIPCC):
        build()
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`. "
            "radzero_amd has no CPU / PyTorch fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)           # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class RzError(RuntimeError):
    pass


def check(rc: int, what: str = ""):
    """Map C-ABI status codes to the exceptions the reference raises."""
    if rc == 0:
        return
    msg = load().rz_last_error().decode(errors="replace")
    if rc == 10001:
        raise ValueError(msg)
    if rc == 10003:
        raise NotImplementedError(msg)
    raise RzError(f"{what} failed with status {rc}: {msg}")
