"""Drives the HOST code of libtinyfusers_hip_asan.so (python -m tinyfusers_amd.build --asan-host: ASan + UBSan on the host side of every
translation unit) without a GPU: the tuning-table loader on good and malformed files, the tf_gemm_tune_* table views, the shape predicates
(pure host logic: patch / ping-pong / persistent-kernel eligibility), the argument checks and status paths of the op entries, tf_rtc_* and
tf_comm_* argument handling, the profiling read-backs.  Run by tests/test_abi.py::test_host_code_is_clean_under_asan_ubsan in a child process
whose environment tinyfusers_amd.build.sanitizer_env() prepared; a sanitizer finding aborts the process (abort_on_error / halt_on_error).
Reference convention exercised: integer status codes, nothing aborts (native/cuda/utils.h:32-49 is the reference's only error handling)."""
import ctypes
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
assert os.environ.get("TF_LIB_PATH", "").endswith("_asan.so"), "run me through tinyfusers_amd.build.sanitizer_env()"
import tinyfusers_amd.native as native      # loads the library, parses the header, loads the shipped tuning table (tf_gemm_tune_load)

lib = native.lib
c_int10, c_int5 = ctypes.c_int * 10, ctypes.c_int * 5
checks = 0


def ok(cond, what):
    global checks
    checks += 1
    assert cond, what


# ---- the shipped table through the host-side views
n = ctypes.c_int()
ok(lib.tf_gemm_tune_count(ctypes.byref(n)) == 0 and n.value >= 150, "shipped table rows")
rows = []
for i in range(n.value):
    k, c = c_int10(), c_int5()
    ok(lib.tf_gemm_tune_entry(i, k, c) == 0, "entry")
    c2 = c_int5()
    ok(lib.tf_gemm_tune_query(k, c2) == 0 and list(c) == list(c2), "query == entry")
    rows.append((list(k), list(c)))
ok(lib.tf_gemm_tune_entry(-1, c_int10(), c_int5()) == 10001 and lib.tf_gemm_tune_entry(n.value, c_int10(), c_int5()) == 10001, "entry index range")
ok(lib.tf_gemm_tune_query(None, None) == 10001, "query null")
ok(lib.tf_gemm_tune_query(c_int10(7, 7, 64, 64, 0, 1, 1, 0, 0, 0), c_int5()) == 10004, "query miss")
# ---- save / reload round trip, then malformed tables: truncated rows, non-numbers, absurd values, empty, directory, missing
with tempfile.TemporaryDirectory() as d:
    p = os.path.join(d, "t.txt").encode()
    ok(lib.tf_gemm_tune_save(p) == 0, "save")
    ok(lib.tf_gemm_autotune(0) == 0, "mode 0 clears the table")
    ok(lib.tf_gemm_tune_count(ctypes.byref(n)) == 0 and n.value == 0, "cleared")
    ok(lib.tf_gemm_autotune(1) == 0 and lib.tf_gemm_tune_load(p) == 0, "reload")
    ok(lib.tf_gemm_tune_count(ctypes.byref(n)) == 0 and n.value == len(rows), "round trip keeps every row")
    for j, text in enumerate(["", "1 2 3", "a b c d e f g h i j k l m n o\n", "8192 320 320 320 0 1 1 0 0 1 64 160 1 3\n",      # a row one field short
                              "8192 320 320 320 0 1 1 0 0 1 64 160 1 99 7\n",                                               # variant out of range
                              "8192 320 320 320 0 1 1 0 0 1 -64 160 1 0 0\n", "8192 320 320 320 0 1 1 0 0 1 64 160 4096 0 0\n",  # tile / split out of range
                              "99999999999999999999 1 1 1 1 1 1 1 1 1 64 64 1 0 0\n",                                       # overflowing integer
                              "8192 320 2880 320 0 3 1 0 0 0 192 160 1 6 1\n8192 320 2880 320 0 3 1 0 1 0 192 160 1 6 1\n",  # a pp3 row, and one with act = 1 (refused)
                              "1 1 1 1 1 1 1 1 1 1 64 64 1 0 0" * 2000]):
        q = os.path.join(d, f"bad{j}.txt")
        open(q, "w").write(text)
        ok(lib.tf_gemm_tune_load(q.encode()) == 0, f"malformed table {j} must not fail the load")
    ok(lib.tf_gemm_tune_load(d.encode()) == 0, "a directory")
    ok(lib.tf_gemm_tune_load(os.path.join(d, "missing").encode()) == 0, "a missing file = no cache yet")
    ok(lib.tf_gemm_tune_load(None) == 10001 and lib.tf_gemm_tune_save(None) == 10001, "null paths")
    ok(lib.tf_gemm_tune_save(os.path.join(d, "no", "such", "dir", "x").encode()) == 10001, "unwritable path")
    ok(lib.tf_gemm_tune_trace(1) == 0 and lib.tf_gemm_tune_trace_dump(os.path.join(d, "trace.txt").encode()) == 0 and lib.tf_gemm_tune_trace(0) == 0, "trace")
    ok(lib.tf_gemm_tune_trace_dump(None) == 10001, "trace null")
    ok(lib.tf_prof_dump(None) == 10001, "prof dump null")
# every accepted row is one the loader's own rules allow
ok(lib.tf_gemm_tune_count(ctypes.byref(n)) == 0 and n.value >= len(rows), "rows after the malformed loads")
for i in range(n.value):
    k, c = c_int10(), c_int5()
    lib.tf_gemm_tune_entry(i, k, c)
    ok(c[0] in (64, 128, 192, 256) and c[1] in (64, 128, 160, 256) and 1 <= c[2] <= 32 and 0 <= c[3] <= 8 and c[4] in (0, 1), f"row {list(k)} -> {list(c)}")
# ---- switches
ok(lib.tf_gemm_autotune(3) == 10001 and lib.tf_gemm_autotune(-1) == 10001 and lib.tf_gemm_autotune(2) == 0 and lib.tf_gemm_autotune(1) == 0, "autotune modes")
ok(lib.tf_gemm_splitk_partials(8) == 10001 and lib.tf_gemm_splitk_partials(32) == 0 and lib.tf_gemm_splitk_partials(16) == 0, "slab types")
ok(lib.tf_gemm_debug(1) == 10001 and lib.tf_gemm_debug(4096) == 10001, "ablation bits refused")
for f in (8, 16, 32, 64, 128, 256, 512, 1024, 2048, 8192, 16384, 0):
    ok(lib.tf_gemm_debug(f) == 0, f"debug {f}")
ok(lib.tf_gemm_force_config(256, 160, 4) == 0 and lib.tf_gemm_force_config(0, 0, 0) == 0, "force config")
# ---- shape predicates and sizes: pure host arithmetic over many shapes (incl. degenerate ones)
for N_, H, W, C1, C2, Co in ((2, 64, 64, 320, 0, 320), (8, 96, 96, 320, 0, 320), (8, 48, 48, 640, 640, 640), (8, 24, 24, 2560, 0, 1280), (1, 1, 1, 8, 0, 8), (0, 0, 0, 0, 0, 0),
                             (2, 8, 8, 1280, 1280, 1280), (8, 12, 12, 1280, 0, 1280), (3, 17, 5, 72, 24, 40)):
    for R, st, pad, up in ((3, 1, 1, 0), (3, 2, 1, 0), (1, 1, 0, 0), (3, 1, 1, 1), (5, 1, 2, 0)):
        lib.tf_conv2d_workspace(N_, H, W, C1, C2, Co, R, R, st, pad, up)
        lib.tf_conv2d_fused_workspace(N_, H, W, C1, C2, Co, R, R, st, pad, up, 64, 0)
        lib.tf_conv2d_fp8_workspace(N_, H, W, C1, C2, Co, R, R, st, pad, up)
        lib.tf_conv2d_gn_supported(N_, H, W, C1, C2, Co, R, R, st, pad, up, 0, 0, 32)
        lib.tf_mx8_conv_supported(N_, H, W, C1, C2, Co, R, R, st, pad, up)
        checks += 5
    lib.tf_conv2d_gn_partial_bytes(N_, 32); lib.tf_group_norm_workspace(N_, H * W, C1, 32)
for M, N2, K in ((73728, 2560, 320), (8192, 320, 320), (512, 1280, 11520), (1, 1, 8), (0, 0, 0), (128, 10240, 1280), (73728, 320, 1280)):
    for act in (0, 1):
        lib.tf_linear_workspace(M, N2, K, act); lib.tf_mx8_gemm_supported(M, N2, K, act, act)
        checks += 2
ok(lib.tf_mx8_bytes(0, 32) == 0 and lib.tf_mx8_bytes(4, 64) == 4 * 64 + 4 * 2, "mx8 bytes")
# ---- op entries: argument checks come first (10001 with a message), then -- with no GPU -- a HIP status from the launch
P = ctypes.c_void_p(4096)                     # a non-null "device pointer": nothing dereferences it on the host
nogpu = lib.tf_init(0) != 0
ok(lib.tf_linear_f16(None, None, None, None, None, 4, 4, 8, 0, None, 0, None) == 10001 and b"null tensor" in lib.tf_last_error(), "linear null")
ok(lib.tf_linear_f16(P, P, P, None, None, 4, 4, 12, 0, None, 0, None) == 10001, "linear K % 8")
ok(lib.tf_linear_f16(P, P, P, None, None, 4, 32, 64, 1, None, 0, None) == 10001, "GEGLU needs a bias")
ok(lib.tf_linear_ln_f16(P, P, P, P, P, None, 4, 6, 64, 0, 1e-5, None) == 10001, "ln linear N % 4")
ok(lib.tf_linear_bf16(P, P, P, None, None, 4, 4, 4, None) == 10001, "bf16 linear K")
ok(lib.tf_conv2d_f16(P, P, None, P, None, None, 0, None, 1, 8, 8, 12, 0, 8, 3, 3, 1, 1, 0, None, 0, None) == 10001, "conv C % 8")
ok(lib.tf_conv2d_f16(P, P, None, P, None, None, 0, None, 1, 2, 2, 8, 0, 8, 5, 5, 1, 0, 0, None, 0, None) == 10001, "conv empty output")
ok(lib.tf_conv2d_fused_f16(P, P, None, P, None, None, 0, None, 1, 8, 8, 8, 0, 8, 3, 3, 1, 1, 1, None, 0, P, None, 8, 0, None, 0, 0, None, None) == 10001, "extra sources + upsample")
ok(lib.tf_conv2d_fused_f16(P, P, None, P, None, None, 0, None, 1, 8, 8, 64, 0, 64, 3, 3, 1, 1, 0, None, 0, None, None, 0, 0, P, 16, 32, None, None) == 10001, "gn_partial without gn_chunks")
ch, zw = ctypes.c_int(), ctypes.c_int()
ok(lib.tf_conv2d_fused_f16(P, P, None, P, None, None, 0, None, 1, 8, 8, 64, 0, 64, 3, 3, 1, 1, 0, None, 0, None, None, 0, 0, P, 16, 32, ctypes.byref(ch), None) == 10001, "statistics buffer too small")
ok(lib.tf_conv2d_fused_norm_f16(P, P, None, P, None, None, 0, None, 1, 8, 8, 64, 0, 64, 3, 3, 1, 1, 0, None, 0, None, None, 0, 0, None, 0, 32, None, None, None, None, 1e-5, 1, None, None) == 10001, "fused norm null")
ok(lib.tf_conv2d_gn_f16(P, P, None, P, None, None, 0, None, 1, 8, 8, 64, 0, 64, 3, 3, 1, 1, 0, None, 0, None, None, 0, 0, None, 0, 0, None, P, None, None, 1, 32, None, 0, 0, 32, 1e-5, 1, None) == 10001, "gn conv: gamma without beta / no stats")
ok(lib.tf_conv2d_fp8(P, P, None, P, P, None, None, 0, None, 1, 8, 8, 32, 0, 64, 3, 3, 1, 1, 0, None, 0, None, 0, 0, None, None) == 10001, "fp8 conv C % 64")
ok(lib.tf_conv2d_mx8(P, P, None, P, P, None, None, 0, None, 1, 8, 8, 64, 0, 64, 3, 3, 2, 1, None, 0, None, 0, 0, None, None) == 10001, "mx8 conv stride")
ok(lib.tf_linear_mx8(P, P, P, P, P, P, 8, 64, 64, 1, 1, None, 0, None) == 10001, "mx8 output with a residual")
ok(lib.tf_linear_fp8(P, P, P, P, None, None, 8, 64, 32, 0, 0, None, 0, None) == 10001, "fp8 K % 64")
ok(lib.tf_quantize_mx8_f16(P, P, 4, 48, None) == 10001 and lib.tf_quantize_fp8_f16(P, P, 12, 1.0, None) == 10001, "quantisers")
ok(lib.tf_pack_weight_fp8(P, P, P, 4, 12, None) == 10001, "pack weight")
ok(lib.tf_gemv_f16(P, P, P, None, 9, 4, 8, 0, None) == 10001, "gemv M <= 8")
ok(lib.tf_ln_fold_weights_f16(P, P, P, P, None, None, None, 4, 8, None) == 10001, "ln fold null")
ok(lib.tf_sdpa_f16(P, P, P, P, 1, 1, 4, 4, 12, *([8] * 12), 0, None) == 10001 and lib.tf_sdpa_f16(P, P, P, P, 1, 1, 4, 4, 40, 7, *([8] * 11), 0, None) == 10001, "sdpa HS / strides")
ok(lib.tf_group_norm_f16(P, P, None, P, None, 1, 4, 64, 0, 32, 1e-5, 0, None, 0, None) == 10001, "group norm gamma without beta")
ok(lib.tf_group_norm_f16(P, P, None, None, None, 1, 4, 60, 0, 32, 1e-5, 0, None, 0, None) == 10001, "group norm C % G")
ok(lib.tf_group_norm_f16(P, P, None, None, None, 1, 4, 64, 0, 32, 1e-5, 0, None, 0, None) == 10003, "group norm workspace")
ok(lib.tf_group_norm_apply_f16(P, P, None, None, P, 0, 1, 4, 64, 32, 1e-5, 0, None) == 10001, "apply chunks")
ok(lib.tf_group_norm_apply_cat_f16(P, P, P, None, None, P, 1, 32, P, 1, 32, 1, 4, 64, 32, 32, 1e-5, 0, None) == 10001, "cat sub-groups")
ok(lib.tf_group_norm_apply2_f16(P, P, P, None, None, P, 1, P, 1, 1, 4, 64, 31, 1e-5, 0, None) == 10001, "apply2 odd G")
ok(lib.tf_group_norm_apply_mx8(P, P, None, None, None, P, 1, 32, None, 0, 0, 1, 4, 48, 0, 16, 1e-5, 0, None) == 10001, "mx8 apply C % 32")
ok(lib.tf_layer_norm_f16(P, P, P, None, 4, 64, 1e-5, None) == 10001 and lib.tf_layer_norm_mx8(P, P, None, None, 4, 48, 1e-5, None) == 10001, "layer norm")
ok(lib.tf_memcpy(P, P, 4, 9) == 10001 and lib.tf_memcpy_async(P, P, 4, 0, None) == 10001 and lib.tf_memcpy_2d_async(P, 4, P, 4, 8, 1, None) == 10001, "memcpy kinds / pitches")
ok(lib.tf_malloc(None, 16) == 10001 and lib.tf_stream_create(None) == 10001 and lib.tf_event_create(None) == 10001 and lib.tf_device_count(None) == 10001, "null out pointers")
ok(lib.tf_device_arch(None, 0, 0) == 10001, "arch buffer")
fam_ms, fam_w, fam_n = ctypes.c_double(), ctypes.c_double(), ctypes.c_longlong()
ok(lib.tf_prof_read_family(0, None, None, None) == 10001 and lib.tf_prof_read_family(9, None, None, None) == 10001, "family range")
ok(lib.tf_prof_read_family(1, ctypes.byref(fam_ms), ctypes.byref(fam_w), ctypes.byref(fam_n)) == 0 and fam_n.value == 0, "family read-back")
g1, g2, g3, g4 = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_longlong()
ok(lib.tf_prof_read_full(ctypes.byref(g1), ctypes.byref(g2), ctypes.byref(g3), ctypes.byref(g4)) == 0 and g4.value == 0, "GEMM prof read-back")
if nogpu:
    # a well-formed launch without a device: every path down to the launch runs on the host (tile choice, table lookup, geometry), then HIP says no
    for args in ((8192, 320, 320), (512, 1280, 1280), (128, 1280, 11520), (73728, 2560, 320)):
        rc = lib.tf_linear_f16(P, P, P, P, None, *args, 0, P, 1 << 26, None)
        ok(rc != 0, f"linear {args} without a GPU must report a status, got {rc}")
    rc = lib.tf_conv2d_fused_f16(P, P, None, P, P, None, 0, None, 2, 64, 64, 320, 0, 320, 3, 3, 1, 1, 0, P, 1 << 26, None, None, 0, 0, None, 0, 0, None, None)
    ok(rc != 0, "conv without a GPU")
    ok(lib.tf_sdpa_f16(P, P, P, P, 2, 8, 4096, 4096, 40, *([40 * 4096 * 8, 40 * 4096, 40] * 4), 0, None) != 0, "sdpa without a GPU")
    ok(lib.tf_layer_norm_f16(P, P, P, P, 64, 320, 1e-5, None) != 0, "layer norm without a GPU")
# ---- run-time compilation and the communicator: argument handling (the libraries are opened on first use)
fn = ctypes.c_void_p()
ok(lib.tf_rtc_load(None, b"x", b"f") != 0 and lib.tf_rtc_load(ctypes.byref(fn), None, b"f") != 0 and lib.tf_rtc_load(ctypes.byref(fn), b"x", None) != 0, "rtc nulls")
rc = lib.tf_rtc_load(ctypes.byref(fn), b"this is not HIP source", b"f")
ok(rc != 0 and lib.tf_last_error(), "rtc: a compile error (or no device) is a status with a message")
ok(lib.tf_rtc_launch(None, 1, 1, 1, 64, 1, 1, 0, None, None) != 0, "rtc launch null")
ok(lib.tf_comm_unique_id(None) != 0, "unique id null")
buf = ctypes.create_string_buffer(128)
rc = lib.tf_comm_unique_id(buf)           # librccl opened here; without a device it may still draw an id, or report a status: both are fine, neither may crash
comm = ctypes.c_void_p()
ok(lib.tf_comm_init_rank(None, buf, 1, 0) != 0 and lib.tf_comm_init_rank(ctypes.byref(comm), None, 1, 0) != 0 and lib.tf_comm_init_rank(ctypes.byref(comm), buf, 0, 0) != 0
   and lib.tf_comm_init_rank(ctypes.byref(comm), buf, 2, 2) != 0, "comm init arguments")
ok(lib.tf_bcast(None, P, 16, 0, None) != 0 and lib.tf_comm_destroy(None) in (0, 10001), "bcast / destroy null")
print(f"HOST_SANITIZER_OK {checks} checks, no GPU: {nogpu}")
