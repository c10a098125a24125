"""Device -- mirrors tinyfusers/storage/device.py:11-233, the reference's own-runtime kernel launcher
(NVRTC compile -> cuModuleLoadData -> cuLaunchKernel of five tiny .cu kernels, cached per Device).
Here the five kernels are pre-compiled gfx950 code objects inside libtinyfusers_hip.so and the wrappers call those (same
signatures, same status -> RuntimeError behaviour); ``load_func`` / ``launch_func`` are the run-time path for a caller's OWN kernel
source -- hiprtc + hipModuleLoadData + hipModuleLaunchKernel behind tf_rtc_load / tf_rtc_launch."""
import ctypes

import numpy as np

from ..native import hip


def _p(ptr):
    return ptr.value if isinstance(ptr, ctypes.c_void_p) else int(ptr)


class Device:
    def __init__(self, device: str):
        self.device = device
        self.func_lib = {}
        self.device_id = 0

    def __str__(self):
        return str(self.device)

    def load_func(self, code_str, func_name):
        """storage/device.py:31-77: compile ``code_str`` at run time and return the kernel ``func_name`` -- hiprtc for the device's own
        architecture (gfx950) instead of NVRTC, one C-ABI call (tf_rtc_load) for compile + module load + function lookup; a compilation
        error raises RuntimeError with the compiler's log.  Handles are cached per Device as in the reference (``func_lib``).  Nothing on
        the denoising path needs this: its kernels ship pre-built in libtinyfusers_hip.so; the wrappers below use those."""
        key = (func_name, hash(code_str))
        if key not in self.func_lib:
            fn = ctypes.c_void_p()
            src = code_str if isinstance(code_str, bytes) else str(code_str).encode()
            hip.tf_rtc_load(ctypes.byref(fn), src, func_name.encode() if isinstance(func_name, str) else func_name)
            self.func_lib[key] = fn
        return self.func_lib[key]

    def launch_func(self, func, grid, block, args, shared_mem=0, stream=None):
        """cuLaunchKernel of the reference's wrappers (storage/device.py:89-99): ``args`` are ctypes scalars / pointers (or DeviceArray /
        Tensor handles, passed as their device pointer), ``grid`` and ``block`` up-to-3-tuples."""
        g = tuple(grid) + (1,) * (3 - len(grid))
        b = tuple(block) + (1,) * (3 - len(block))
        keep = []
        for a in args:
            if hasattr(a, "ptr"):
                a = ctypes.c_void_p(a.ptr)
            elif hasattr(a, "dt_ptr"):
                a = ctypes.c_void_p(_p(a.dt_ptr))
            elif isinstance(a, (bool, np.bool_)):
                a = ctypes.c_int(int(a))
            elif isinstance(a, (int, np.integer)):
                # a C int, as in the reference's wrappers; a value that does not fit is a 64-bit size or a pointer the caller must type
                # itself (ctypes.c_longlong / c_void_p): truncating it silently would corrupt the launch
                if isinstance(a, (np.int64, np.uint64)) and not -2 ** 31 <= int(a) < 2 ** 31:
                    a = ctypes.c_longlong(int(a))
                elif not -2 ** 31 <= int(a) < 2 ** 32:
                    raise OverflowError(f"launch_func: integer argument {a} does not fit a C int; pass ctypes.c_longlong / ctypes.c_void_p")
                else:
                    a = ctypes.c_int(int(a) if int(a) < 2 ** 31 else int(a) - 2 ** 32)
            elif isinstance(a, (float, np.floating)):
                a = ctypes.c_double(float(a)) if isinstance(a, np.float64) else ctypes.c_float(float(a))
            elif not isinstance(a, ctypes._SimpleCData) and not isinstance(a, (ctypes.Structure, ctypes.Array, ctypes._Pointer)):
                raise TypeError(f"launch_func: cannot pass {type(a).__name__} to a kernel; use ctypes scalars / pointers, numbers or device arrays")
            keep.append(a)
        params = (ctypes.c_void_p * len(keep))(*[ctypes.cast(ctypes.byref(a), ctypes.c_void_p) for a in keep])
        hip.tf_rtc_launch(func, *g, *b, int(shared_mem), stream, params)

    def add_bias(self, result_pntr, bias_pntr, m, n):
        """storage/device.py:79-102 / add_bias_func.cu: result is a column-major (m x n) cuBLAS result (ldc = m); bias[j] is
        added to every element of column j (the kernel runs with BT = m, OC = n)."""
        hip.tf_add_bias_colmajor_f32(_p(result_pntr), _p(bias_pntr), m, n, None)
        return result_pntr

    def scale_tensor(self, x_ptr, scaler, B, T, NH, OC):
        """storage/device.py:104-127 / scale_tensor_func.cu: in-place scalar multiply of B*T*NH*OC floats."""
        hip.tf_scale_f32(_p(x_ptr), float(np.asarray(scaler).reshape(-1)[0]), B * T * NH * OC, None)
        return x_ptr

    def softmax(self, out_tensor, inp_tensor):
        """storage/device.py:129-157 / softmax_func.cu: row softmax over (N, OC) fp32."""
        N, OC = inp_tensor.shape
        hip.tf_softmax_rows_f32(_p(out_tensor.dt_ptr), _p(inp_tensor.dt_ptr), N, OC, None)
        return out_tensor

    def _permute(self, out_tensor, in_tensor, axes):
        shape = tuple(in_tensor.shape)
        nd = len(shape)
        sh = (ctypes.c_int * nd)(*shape)
        ax = (ctypes.c_int * nd)(*axes)
        hip.tf_transpose_f32(_p(out_tensor.dt_ptr), _p(in_tensor.dt_ptr), nd, sh, ax, None)
        out_tensor.shape = tuple(shape[a] for a in axes)
        return out_tensor

    def transpose(self, out_tensor, in_tensor, axes: tuple):
        """storage/device.py:159-195 / transpose.cu: 2-D / 3-D axis permutation."""
        return self._permute(out_tensor, in_tensor, tuple(axes))

    def transpose4d(self, out_tensor, in_tensor, axes: tuple):
        """storage/device.py:197-233 / transpose4d.cu: 4-D axis permutation."""
        return self._permute(out_tensor, in_tensor, tuple(axes))
