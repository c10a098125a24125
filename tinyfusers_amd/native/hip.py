import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TF_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "lib", "libtinyfusers_hip.so")   # TF_LIB_PATH: an experimental build
HEADER_PATH = os.path.join(os.path.dirname(os.path.dirname(_HERE)), "include", "tinyfusers_hip.h")

_CT = {
    "int": ctypes.c_int, "float": ctypes.c_float, "double": ctypes.c_double, "size_t": ctypes.c_size_t,
    "long long": ctypes.c_longlong, "void*": ctypes.c_void_p, "const void*": ctypes.c_void_p,
    "void**": ctypes.POINTER(ctypes.c_void_p), "const void*const*": ctypes.c_void_p, "void*const*": ctypes.c_void_p, "int*": ctypes.POINTER(ctypes.c_int), "const int*": ctypes.POINTER(ctypes.c_int),
    "float*": ctypes.POINTER(ctypes.c_float), "double*": ctypes.POINTER(ctypes.c_double),
    "long long*": ctypes.POINTER(ctypes.c_longlong), "char*": ctypes.c_char_p, "const char*": ctypes.c_char_p,
    "tfStream_t": ctypes.c_void_p, "tfEvent_t": ctypes.c_void_p, "tfGraph_t": ctypes.c_void_p, "tfComm_t": ctypes.c_void_p,
    "tfComm_t*": ctypes.POINTER(ctypes.c_void_p),
    "tfStream_t*": ctypes.POINTER(ctypes.c_void_p), "tfEvent_t*": ctypes.POINTER(ctypes.c_void_p),
    "tfGraph_t*": ctypes.POINTER(ctypes.c_void_p), "void": None,
    "unsigned": ctypes.c_uint, "tfFunction_t": ctypes.c_void_p, "tfFunction_t*": ctypes.POINTER(ctypes.c_void_p),
}


def _parse_header(path=HEADER_PATH):
    """[(return type, name, [arg types])] for every prototype in include/tinyfusers_hip.h."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"^\s*#.*$", "", src, flags=re.M)
    protos = []
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(tf_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        argt = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                mm = re.match(r"(.*?)(\w+)$", a)          # strip the parameter name
                t = mm.group(1).strip() if mm and mm.group(1).strip() else a
                t = re.sub(r"\s*\*\s*", "*", t)
                argt.append(t)
        protos.append((re.sub(r"\s*\*\s*", "*", ret), name, argt))
    return protos


def declared_symbols():
    return [n for _, n, _ in _parse_header()]


def _load():
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m tinyfusers_amd.build` (hipcc --offload-arch=gfx950). "
            "tinyfusers_amd has no CPU fallback.")
    dll = ctypes.CDLL(LIB_PATH)
    old_build = bool(os.environ.get("TF_LIB_PATH")) and os.environ.get("TF_LIB_ALLOW_MISSING") == "1"   # A/B against an earlier round's library
    for ret, name, argt in _parse_header():
        if old_build and not hasattr(dll, name):
            continue
        fn = getattr(dll, name)          # AttributeError here = header/library mismatch: fail loudly
        fn.restype = _CT[ret]
        fn.argtypes = [_CT[t] for t in argt]
    return dll


lib = _load()


def check(status, name):
    if status != 0:
        msg = lib.tf_last_error()
        raise RuntimeError(f"{name} failed with status {status}" + (f": {msg.decode()}" if msg else ""))


class _Hip:
    """``hip.tf_xxx(args)`` calls the C entry and raises RuntimeError on a non-zero status
    (the reference checks the status at every call site; here it is done once)."""

    def __getattr__(self, name):
        if name.endswith("_16") and not hasattr(lib, name) and os.environ.get("TF_LIB_PATH") and os.environ.get("TF_LIB_ALLOW_MISSING") == "1":
            # same-box A/B against an earlier round's library (tools/ab_lib.sh): it has no dtype-tagged entries -- their float16 namesakes take the call
            f16 = getattr(self, {"tf_conv2d_fused_16": "tf_conv2d_fused_f16", "tf_conv2d_fused_norm_16": "tf_conv2d_fused_norm_f16", "tf_conv2d_gn_16": "tf_conv2d_gn_f16"}.get(name, name[:-3] + "_f16"))

            def call16(dtype, *args):
                assert dtype == 0, f"{name}: the library at TF_LIB_PATH has float16 entries only"
                return f16(*args)
            setattr(self, name, call16)
            return call16
        fn = getattr(lib, name)
        if fn.restype is not ctypes.c_int:
            return fn

        def call(*args):
            check(fn(*args), name)
        call.__name__ = name
        setattr(self, name, call)
        return call


hip = _Hip()

# per-shape GEMM configurations measured on MI355X for the SD-1.x step (tools/tune_best.sh); shapes that are not in the
# table are autotuned on their first eager call
# TF_GEMM_TUNE_TABLE=<path> uses another table, TF_GEMM_TUNE_TABLE= (empty) none: every shape is tuned afresh (tools/tune_best.sh)
_TUNE = os.environ.get("TF_GEMM_TUNE_TABLE", os.path.join(os.path.dirname(_HERE), "gemm_tune_gfx950.txt"))
if _TUNE and os.path.exists(_TUNE):
    lib.tf_gemm_tune_load(_TUNE.encode())
# One process per GPU: with several ranks every rank must pick the same kernels (bit-identical results across ranks, the same kernels in
# every rank's timed region), so nothing is tuned at run time -- a shape without a row in the table is an error naming the shape
# (TF_GEMM_AUTOTUNE=0/1/2 overrides: see tf_gemm_autotune in include/tinyfusers_hip.h)
if os.environ.get("TF_SPLITK_PARTIALS"):          # A/B: 32 = the fp32 split-K partial slabs of rounds 1-3 (default 16: fp16 slabs, fp32 accumulation)
    check(lib.tf_gemm_splitk_partials(int(os.environ["TF_SPLITK_PARTIALS"])), "tf_gemm_splitk_partials")
_mode = os.environ.get("TF_GEMM_AUTOTUNE", "2" if int(os.environ.get("WORLD_SIZE", "1") or 1) > 1 else "")
if _mode:
    check(lib.tf_gemm_autotune(int(_mode)), "tf_gemm_autotune")
