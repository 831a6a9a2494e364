"""Device-resident panel path: inputs and outputs stay in HBM (torch tensors are used purely
as device-memory handles; all arithmetic happens in libtsf_amd.so via the `_dev` C-ABI entry
points on torch's current HIP stream).  Used by bench.py and by multi-GPU sharding."""
import ctypes

import numpy as np

from . import _lib


def _torch():
    import torch
    return torch


class DeviceFitOutput(object):
    """Caller-allocated device buffers for tsf_fit_*_dev."""

    def __init__(self, N, stride, n_grids, device):
        torch = _torch()
        self.theta = torch.zeros((N, stride), dtype=torch.float64, device=device)
        self.y_scale = torch.zeros(N, dtype=torch.float64, device=device)
        self.fval = torch.zeros(N, dtype=torch.float64, device=device)
        self.status = torch.zeros(N, dtype=torch.int32, device=device)
        self.n_iter = torch.zeros(N, dtype=torch.int32, device=device)
        self.n_eval = torch.zeros(N, dtype=torch.int32, device=device)
        self.grid = torch.zeros(n_grids * _lib.GRID_DTYPE.itemsize, dtype=torch.uint8, device=device)
        self.n_grids = n_grids
        self._c = _lib.TsfFitOut(self.theta.data_ptr(), self.y_scale.data_ptr(), self.fval.data_ptr(),
                                 self.status.data_ptr(), self.n_iter.data_ptr(),
                                 self.n_eval.data_ptr(), self.grid.data_ptr())

    def grid_numpy(self):
        return np.frombuffer(self.grid.cpu().numpy().tobytes(), dtype=_lib.GRID_DTYPE)


class DeviceForecaster(object):
    """fit + predict on device-resident aligned panels.  One instance per GPU / rank."""

    def __init__(self, spec, device_index=0):
        torch = _torch()
        if not torch.cuda.is_available():
            raise _lib.TsfError('no GPU visible: the device path has no CPU fallback')
        self.spec = spec
        self.cspec = spec.to_c()
        self.device_index = int(device_index)
        self.device = torch.device('cuda', self.device_index)
        self.ctx = _lib.Context(self.device_index)
        self.L = _lib.load()

    def _stream(self):
        return ctypes.c_void_p(_torch().cuda.current_stream(self.device).cuda_stream)

    def set_cost_hints(self, cost):
        """Scheduling hint for the next fit call with len(cost) series (tsf_set_cost_hints): cost[i] = expected
        relative cost of series i, e.g. n_eval of the previous fit of the same panel.  None clears."""
        import numpy as np
        if cost is None:
            self.ctx.check(self.L.tsf_set_cost_hints(self.ctx.handle, None, 0))
            return
        if hasattr(cost, 'detach'):
            cost = cost.detach().cpu().numpy()
        c = np.ascontiguousarray(cost, dtype=np.int32)
        self.ctx.check(self.L.tsf_set_cost_hints(self.ctx.handle, c.ctypes.data, c.shape[0]))

    def set_profiling(self, on=True):
        self.ctx.check(self.L.tsf_set_profiling(self.ctx.handle, int(on)))

    def last_fit_kernel_ms(self):
        ms = ctypes.c_float(0.0)
        self.ctx.check(self.L.tsf_last_fit_kernel_ms(self.ctx.handle, ctypes.byref(ms)))
        return float(ms.value)

    def profile_read(self):
        """Fit-kernel durations (ms) of the profiled calls since set_profiling(True)."""
        buf = (ctypes.c_float * 64)()
        n = ctypes.c_int32(0)
        self.ctx.check(self.L.tsf_profile_read(self.ctx.handle, buf, 64, ctypes.byref(n)))
        return [float(buf[i]) for i in range(n.value)]

    def alloc_fit_output(self, N):
        return DeviceFitOutput(N, self.spec.theta_stride, 1, self.device)

    def fit_aligned(self, ds, y, out, floor=None, cap=None, extra=None):
        """ds: int64 [T] device tensor; y: [N][T] float64/float32/int32 device tensor."""
        torch = _torch()
        N, T = y.shape
        code = {torch.float64: _lib.Y_F64, torch.float32: _lib.Y_F32, torch.int32: _lib.Y_I32}[y.dtype]
        rc = self.L.tsf_fit_aligned_dev(
            self.ctx.handle, ctypes.byref(self.cspec), N, T, ds.data_ptr(), y.data_ptr(), code,
            floor.data_ptr() if floor is not None else None,
            cap.data_ptr() if cap is not None else None,
            extra.data_ptr() if extra is not None else None,
            ctypes.byref(out._c), self._stream())
        self.ctx.check(rc)
        return out

    def predict(self, out, ds_future, yhat, yhat_int=None, floor=None, cap=None, extra_future=None):
        """ds_future: int64 [H] device tensor (shared horizon); yhat: [N][H] float64."""
        N, H = yhat.shape
        rc = self.L.tsf_predict_dev(
            self.ctx.handle, ctypes.byref(self.cspec), N, H, out.theta.data_ptr(),
            out.y_scale.data_ptr(), out.grid.data_ptr(), out.n_grids, ds_future.data_ptr(), 1,
            floor.data_ptr() if floor is not None else None,
            cap.data_ptr() if cap is not None else None,
            extra_future.data_ptr() if extra_future is not None else None,
            yhat.data_ptr(), yhat_int.data_ptr() if yhat_int is not None else None, self._stream())
        self.ctx.check(rc)
        return yhat
