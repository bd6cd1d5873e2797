"""Thin Python driver of the C ABI (``include/mgx.h``): device pointers of torch tensors in, kernels out.

``StepEngine`` is the batched counterpart of the reference's ``Microgrid`` object as far as stepping goes:
``run``/``step`` (microgrid.py:227-325), ``reset`` (microgrid.py:205-219), the discrete action expansion
(priority_list.py:69-167) and the log/observation outputs.  torch is used for memory and streams only.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import check


def _ptr(t):
    return None if t is None else t.data_ptr()


try:                                   # raw hipStream_t of torch's current stream, without building a Stream object
    _get_raw = torch._C._cuda_getCurrentRawStream

    def _raw_stream(idx):
        return _get_raw(idx)
except AttributeError:                 # pragma: no cover
    def _raw_stream(idx):
        return torch.cuda.current_stream(idx).cuda_stream


class StepEngine:
    def __init__(self, batch, obs_dtype=torch.float64, action_dtype=torch.float64):
        if batch.device.type != "cuda":
            raise _lib.MgxError(_lib.MGX_ERR_DEVICE,
                                "StepEngine needs the batch on a GPU (cuda/HIP device); there is no CPU path")
        self.batch = batch
        self.layout = batch.layout
        self.device = batch.device
        self._lib = _lib.lib()
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            L, cols = batch.c_layout(), batch.c_columns()
            check(self._lib.mgx_create(C.byref(L), C.byref(cols), C.byref(self._h)))
        self.window = (self.layout.initial_step, self.layout.final_step)
        self._full_window = self.window
        self._window_start = None
        self._window_t0 = None          # rolling windows: counter value each grid's episode started at
        self._inplace = False           # in-place episodes (reset_episodes)
        self._final_obs = None
        self._obs_compact = False
        self._done_bits = False
        self._dev_counter = False
        self._t = self.layout.initial_step
        self.n_shards = 1
        if batch.forecast_noise is not None:
            self.set_forecast_noise(**batch.forecast_noise)
        self._dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self._only_device = torch.cuda.device_count() == 1
        self.N = self.layout.n_grids
        self.action_dim = self._lib.mgx_action_dim(self._h)
        self.obs_dim = self._lib.mgx_obs_dim(self._h)
        self.log_dim = self._lib.mgx_log_dim(self._h)
        self.log_names = [self._lib.mgx_log_name(self._h, j).decode() for j in range(self.log_dim)]
        assert self.action_dim == self.layout.action_dim and self.obs_dim == self.layout.obs_dim
        assert self.log_names == self.layout.log_names
        self.obs_dtype = torch.float64
        if obs_dtype != torch.float64:
            self.set_obs_dtype(obs_dtype)
        self.action_dtype = torch.float64
        self._action_shape = (self.N, self.action_dim)
        if action_dtype != torch.float64:
            self.set_action_dtype(action_dtype)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.mgx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------------------------------------
    def _call(self, fn, *args):
        """fn(handle, *args, stream) on torch's current stream of the engine's device.  Kept lean: at N = 100k a
        single-step kernel takes ~5 us, so every microsecond of Python here shows up in env-steps/s."""
        idx = self._dev_index
        if self._only_device or torch.cuda.current_device() == idx:      # one visible GPU: nothing to check (~1 us saved)
            rc = fn(self._h, *args, _raw_stream(idx))
        else:
            with torch.cuda.device(idx):
                rc = fn(self._h, *args, _raw_stream(idx))
        if rc:
            check(rc)

    def _empty(self, *shape, dtype=torch.float64):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def set_obs_dtype(self, dtype):
        """Element type of the observation rows: torch.float64 (the reference's arrays) or torch.float32 (the same
        values rounded to nearest on the way out -- half the bytes of the largest output of a step)."""
        if dtype not in (torch.float64, torch.float32):
            raise ValueError("obs_dtype must be torch.float64 or torch.float32")
        check(self._lib.mgx_set_obs_format(self._h, 1 if dtype == torch.float32 else 0))
        self.obs_dtype = dtype

    def set_action_dtype(self, dtype):
        """Element type of the continuous controls of step / step_k: torch.float64, or torch.float32 (what a policy emits;
        widened exactly on device -- the results are those of the float64 path fed ``actions.double()``)."""
        if dtype not in (torch.float64, torch.float32):
            raise ValueError("action_dtype must be torch.float64 or torch.float32")
        check(self._lib.mgx_set_action_format(self._h, 1 if dtype == torch.float32 else 0))
        self.action_dtype = dtype

    def set_obs_state_only(self, flag):
        """True: the ``obs`` output of step / step_discrete / observe / reset receives only the genset / battery state
        columns -- the window columns of that row were written ahead of time by ``observe_windows``."""
        check(self._lib.mgx_set_obs_mode(self._h, 1 if flag else 0))
        self._obs_compact = False

    def set_obs_compact(self, flag):
        """True (``MGX_OBS_ROWS_STATE_COMPACT``): the ``obs`` output of step / step_discrete / observe receives ONLY the
        genset / battery state columns, as a dense [N, S] array (S = ``state_dim``) -- the zero-copy observation contract:
        the window columns are views of ``normalise_series()``'s output."""
        check(self._lib.mgx_set_obs_mode(self._h, 2 if flag else 0))
        self._obs_compact = bool(flag)

    @property
    def state_dim(self):
        return 4 * int(self.layout.has_genset) + 2 * int(self.layout.has_battery)

    def set_done_format(self, bits):
        """``mgx_set_done_format``: the ``done`` output of the fused calls as bytes [K, N] (default) or, ``bits=True``, as
        bit sets [K, ceil(N / 16)] uint16 (``unpack_done_bits`` turns them back into [K, N] bools)."""
        check(self._lib.mgx_set_done_format(self._h, 1 if bits else 0))
        self._done_bits = bool(bits)

    def unpack_done_bits(self, words):
        """[K, ceil(N / 16)] int16 words of a fused call with ``set_done_format(True)`` -> [K, N] bool."""
        K = words.shape[0]
        sh = torch.arange(16, device=words.device, dtype=torch.int32)
        bits = (words.to(torch.int32)[:, :, None] >> sh) & 1
        return bits.reshape(K, -1)[:, :self.N].to(torch.bool)

    def done_steps(self, K):
        """Lock-step ``done`` of the NEXT K steps without a byte of device traffic: done(k) = (t + k >= final_step - 1) for
        every grid (base_timeseries_module.py:124-125) -- a [K, N] bool VIEW (stride 0 along the grids) of a K-vector.
        Not available during per-grid-window episodes (every grid ends on its own: ask the kernel, ``done=True``)."""
        if getattr(self, "_window_start", None) is not None:
            raise RuntimeError("per-grid episodes end per grid: request the kernel's done output")
        t = self.current_step
        d = torch.arange(t, t + int(K), device=self.device) >= (self.window[1] - 1)
        return d[:, None].expand(int(K), self.N)

    def normalise_series(self):
        """``mgx_normalise_series``: the series normalised ONCE, grid-major -- {"load": [N, R], "pv": [N, R],
        "grid": [N, R, 4]} (R = n_steps + horizon + 1, dtype = the engine's obs dtype) -- so that the window columns of the
        observation at step t are the slices ``load[:, t : t + 1 + H]`` etc. (views, no bytes moved per step).  Raises if a bound
        column does not bound its series (the reference's forecast clip would then not be the identity)."""
        L = self.layout
        if getattr(self, "_window_t0", None) is not None:
            raise _lib.MgxError(_lib.MGX_ERR_UNSUPPORTED, "normalise_series: not offered for rolling windows")
        # rows of the series the handle currently steps over: the window buffers during a per-grid-window episode
        T = self._windows["rows"] if getattr(self, "_window_start", None) is not None else L.n_steps
        R = T + L.horizon + 1
        out = {"load": torch.empty(self.N, R, dtype=self.obs_dtype, device=self.device),
               "pv": torch.empty(self.N, R, dtype=self.obs_dtype, device=self.device)}
        if L.has_grid:
            out["grid"] = torch.empty(self.N, R, 4, dtype=self.obs_dtype, device=self.device)
        clipped = torch.zeros(1, dtype=torch.int32, device=self.device)
        self._call(self._lib.mgx_normalise_series, out["load"].data_ptr(), out["pv"].data_ptr(), _ptr(out.get("grid")),
                   clipped.data_ptr())
        n = int(clipped.item())
        if n:
            raise _lib.MgxError(_lib.MGX_ERR_UNSUPPORTED,
                                f"{n} series values lie outside their observation bounds: the forecast clip "
                                f"(forecaster.py:139-149) is not the identity and windows are not slices of one normalised series")
        return out

    def set_ring_pitch(self, rows):
        """Rows between consecutive blocks of the observation rings (``mgx_set_ring_pitch``; default N = dense rings)."""
        check(self._lib.mgx_set_ring_pitch(self._h, int(rows)))
        self._ring_pitch = int(rows)

    def set_ring_layout(self, columns):
        """``mgx_set_ring_layout``: True = column-major ring blocks -- the [N, D] observation of a block is then a view with
        strides (1, pitch) (the pitch must be a multiple of 32: ``set_ring_pitch`` first)."""
        check(self._lib.mgx_set_ring_layout(self._h, 1 if columns else 0))
        self._ring_columns = bool(columns)

    def _check_ring(self, out):
        pitch = getattr(self, "_ring_pitch", self.N)
        want = (1, pitch) if getattr(self, "_ring_columns", False) else (self.obs_dim, 1)      # strides of a block's [N, D] view
        if out.dim() != 3 or tuple(out.shape[1:]) != (self.N, self.obs_dim) or out.dtype != self.obs_dtype \
                or out.device != self.device or (out.stride(1), out.stride(2)) != want \
                or (out.shape[0] > 1 and out.stride(0) != pitch * self.obs_dim):
            raise ValueError(f"ring must be a {self.obs_dtype} tensor [K, {self.N}, {self.obs_dim}] on {self.device} whose "
                             f"blocks are {pitch} rows apart (set_ring_pitch) with strides {want} inside a block (set_ring_layout)")

    def observe_windows(self, K=None, out=None):
        """Observation rows of the next K steps, ``ring[k]`` = the row of step counter t + k (block 0 complete, blocks
        1..K-1 without the state columns): every series value is read and normalised once instead of 1 + horizon times."""
        if out is None:
            pitch = getattr(self, "_ring_pitch", self.N)
            if getattr(self, "_ring_columns", False):
                out = torch.empty((int(K), self.obs_dim, pitch), dtype=self.obs_dtype, device=self.device)[:, :, :self.N].transpose(1, 2)
            else:
                out = torch.empty((int(K), pitch, self.obs_dim), dtype=self.obs_dtype, device=self.device)[:, :self.N]
        self._check_ring(out)
        self._call(self._lib.mgx_observe_windows, int(out.shape[0]), out.data_ptr())
        return out

    def observe_windows_ahead(self, ahead, out):
        """``mgx_observe_windows_ahead``: the window columns of counter values t + ahead .. t + ahead + K - 1 into ``out``
        [K, N, D], written on the engine's prefetch stream behind everything queued on torch's current stream."""
        self._check_ring(out)
        self._call(self._lib.mgx_observe_windows_ahead, int(ahead), int(out.shape[0]), out.data_ptr())
        return out

    def patch_windows(self, mask, ring, first_block, counter_offset=0, restarted=None):
        """``mgx_patch_windows``: recompute the window columns of the grids with ``mask[i] != 0`` in blocks first_block..K-1 of
        ``ring`` (after ``reset_grids*`` replaced their series rows); block first_block = the row of counter value
        current + counter_offset."""
        if mask.dtype == torch.bool:
            mask = mask.view(torch.uint8)
        self._check_ring(ring)
        self._call(self._lib.mgx_patch_windows, mask.data_ptr(), int(ring.shape[0]), ring.data_ptr(), int(first_block),
                   int(counter_offset), _ptr(restarted))

    def prefetch_wait(self):
        """Torch's current stream waits for the last ``observe_windows_ahead``."""
        self._call(self._lib.mgx_prefetch_wait)

    def _obs_buf(self, out):
        D = self.state_dim if getattr(self, "_obs_compact", False) else self.obs_dim
        if out is None:
            return torch.empty((self.N, D), dtype=self.obs_dtype, device=self.device)
        # (a block of a column-major ring is the [N, D] view with strides (1, pitch): the state-only target of a step)
        columns = getattr(self, "_ring_columns", False) and out.dim() == 2 and out.stride() == (1, getattr(self, "_ring_pitch", self.N))
        if tuple(out.shape) != (self.N, D) or out.dtype != self.obs_dtype or not (out.is_contiguous() or columns) \
                or out.device != self.device:
            raise ValueError(f"obs must be a contiguous {self.obs_dtype} tensor of shape ({self.N}, {D}) "
                             f"on {self.device}")
        return out

    def _check_actions(self, actions, lead):
        if actions is None:
            if self.action_dim:
                raise ValueError("actions are required")
            return None
        shape = actions.shape
        if (len(shape) != len(lead) + 2 or shape[-1] != self.action_dim or shape[-2] != self.N
                or (lead and shape[0] != lead[0]) or actions.dtype != self.action_dtype or not actions.is_contiguous()
                or actions.device != self.device):
            want = (*lead, self.N, self.action_dim)
            raise ValueError(f"actions must be a contiguous {self.action_dtype} tensor of shape {want} on {self.device}")
        return actions

    @property
    def current_step(self):
        # host mirror of the handle's step counter (every call that moves it goes through this class or, for fleets,
        # hetero._bump): no C call on the per-step paths.  None = the truth lives on the device (device-counter mode).
        t = self._t
        return t if t is not None else self._lib.mgx_current_step(self._h)

    def use_device_counter(self, enable=True):
        """Keep the step counter on the device so that step calls can be captured in a HIP graph
        (``torch.cuda.graph``) and replayed; see ``mgx_use_device_counter`` in include/mgx.h."""
        self._call(self._lib.mgx_use_device_counter, 1 if enable else 0)
        self._dev_counter = bool(enable)
        self._t = None if enable else self._lib.mgx_current_step(self._h)

    def set_window(self, initial_step, final_step):
        """Episode window of the next reset (what a trajectory_func returns, microgrid.py:221-225)."""
        check(self._lib.mgx_set_window(self._h, int(initial_step), int(final_step)))
        self.window = (int(initial_step), int(final_step))
        self._full_window = self.window

    def set_forecast_noise(self, seed=0, increase_uncertainty=False):
        """GaussianNoiseForecaster switches (needs the *_noise_std columns in the batch)."""
        check(self._lib.mgx_set_forecast_noise(self._h, int(seed) & (2 ** 64 - 1), int(bool(increase_uncertainty))))

    def set_reward_shaper(self, kind):
        check(self._lib.mgx_set_reward_shaper(self._h, int(kind)))

    # ------------------------------------------------------------------------------------------------
    def reset(self, initial_step=None, want_obs=True, out=None):
        obs = self._obs_buf(out) if want_obs else None
        self._call(self._lib.mgx_reset, -1 if initial_step is None else int(initial_step), _ptr(obs))
        if getattr(self, "_window_start", None) is not None:       # a per-grid-window episode ends with a plain reset
            self._window_start = None
            self._window_t0 = None
            self._inplace = False
            self._final_obs = None
            self.window = self._full_window
        if self._t is not None:
            self._t = self.window[0] if initial_step is None else int(initial_step)
        return obs

    def _check_episodes(self, start, length, max_length, mask=None):
        """The reference refuses a trajectory outside the env's window (Microgrid._check_trajectory_func, microgrid.py:181-203:
        ValueError); the kernels clamp instead, which would leave ``current_steps`` at odds with the rows actually walked.
        Validate on the host (one device->host sync per call)."""
        lo, hi = self._full_window
        if int(max_length) > hi - lo:
            return                       # the C ABI refuses this with the reference's own message
        st = start if mask is None else start[mask.view(torch.bool)]
        if st.numel() == 0:
            return
        s_min, s_max = int(st.min().item()), int(st.max().item())
        if s_min < lo:
            raise ValueError(f'trajectory_func returned initial_step value ({s_min}) less than env\'s initial step: ({lo})')
        if s_max >= hi:
            raise ValueError(f'trajectory_func returned values ({s_max}, ...) such that initial_step was greater than or equal to '
                             f'the env\'s final step ({hi}).')
        ln = None if length is None else (length if mask is None else length[mask.view(torch.bool)])
        if ln is not None:
            if int(ln.min().item()) < 1:
                raise ValueError('trajectory_func returned values such that initial_step was greater than or equal to final_step.')
            end = int((st.long() + ln.long()).max().item())
        else:
            end = s_max + int(max_length)
        if end > hi:
            raise ValueError(f'trajectory_func returned final_step value ({end}) greater than env\'s final step: ({hi})')

    def reset_windows(self, start, length=None, max_length=None, want_obs=True, out=None, validate=True):
        """Per-grid episodes (``mgx_reset_windows``): grid i starts at series row ``start[i]`` and reports ``done`` after
        ``length[i]`` steps (``length=None``: ``max_length`` for every grid).  ``start`` / ``length`` are int32 device
        tensors [N].  The window buffers live in the engine and are re-used by the next reset of the same shape."""
        L = self.layout
        if start.dtype != torch.int32 or tuple(start.shape) != (self.N,) or start.device != self.device:
            raise ValueError(f"start must be an int32 tensor of shape ({self.N},) on {self.device}")
        if length is not None:
            if length.dtype != torch.int32 or tuple(length.shape) != (self.N,) or length.device != self.device:
                raise ValueError(f"length must be an int32 tensor of shape ({self.N},) on {self.device}")
            if max_length is None:
                max_length = int(length.max().item())
        if max_length is None:
            raise ValueError("max_length is required when length is None")
        if validate:
            self._check_episodes(start, length, max_length)
        rows = int(max_length) + L.horizon + 1
        w = getattr(self, "_windows", None)
        if w is None or w["rows"] != rows:
            if L.multi:                 # several modules of a kind: series [T, n, N], window buffers [rows, n, N]
                w = dict(rows=rows, load=self._empty(rows, max(1, L.n_load), self.N), pv=self._empty(rows, max(1, L.n_pv), self.N),
                         grid=self._empty(rows, L.n_grid, 4, self.N) if L.has_grid else None,
                         final=self._empty(self.N, dtype=torch.int32))
            else:
                w = dict(rows=rows, load=self._empty(rows, self.N), pv=self._empty(rows, self.N),
                         grid=self._empty(rows, 4, self.N) if L.has_grid else None,
                         final=self._empty(self.N, dtype=torch.int32))
            self._windows = w
        obs = self._obs_buf(out) if want_obs else None
        self._call(self._lib.mgx_reset_windows, start.data_ptr(), _ptr(length), int(max_length), w["load"].data_ptr(),
                   w["pv"].data_ptr(), _ptr(w["grid"]), w["final"].data_ptr() if length is not None else None, _ptr(obs))
        self._window_start = start
        self._window_t0 = None
        self._inplace = False
        self.window = (0, int(max_length))
        self._t = 0
        return obs

    def reset_windows_rolling(self, start, length=None, max_length=None, want_obs=True, out=None, validate=True):
        """Rolling per-grid windows (``mgx_reset_windows_rolling``): as ``reset_windows``, but the window buffers are rings
        (2^p rows >= max_length + horizon + 1), the shared counter never ends, and ``reset_grids`` restarts individual grids
        at any later step.  Single steps only."""
        L = self.layout
        if start.dtype != torch.int32 or tuple(start.shape) != (self.N,) or start.device != self.device:
            raise ValueError(f"start must be an int32 tensor of shape ({self.N},) on {self.device}")
        if length is not None and (length.dtype != torch.int32 or tuple(length.shape) != (self.N,) or length.device != self.device):
            raise ValueError(f"length must be an int32 tensor of shape ({self.N},) on {self.device}")
        if max_length is None:
            raise ValueError("max_length is required (the longest episode any later reset_grids may ask for)")
        if validate:
            self._check_episodes(start, length, max_length)
        need = int(max_length) + L.horizon + 1
        rows = 1 << (need - 1).bit_length()
        w = getattr(self, "_rolling", None)
        if w is None or w["rows"] != rows:
            w = dict(rows=rows, load=self._empty(rows, self.N), pv=self._empty(rows, self.N),
                     grid=self._empty(rows, 4, self.N) if L.has_grid else None, final=self._empty(self.N, dtype=torch.int32))
            self._rolling = w
        obs = self._obs_buf(out) if want_obs else None
        self._call(self._lib.mgx_reset_windows_rolling, start.data_ptr(), _ptr(length), int(max_length), rows,
                   w["load"].data_ptr(), w["pv"].data_ptr(), _ptr(w["grid"]), w["final"].data_ptr(), _ptr(obs))
        self._window_start = start.clone()                 # per-grid: series row of the episode's first step ...
        self._window_t0 = torch.zeros_like(start)          # ... and the counter value it started at
        self.window = (0, int(max_length))
        self._rolling_max = int(max_length)
        self._inplace = False
        self._t = 0
        return obs

    def reset_episodes(self, start, length=None, max_length=None, want_obs=True, out=None, validate=True):
        """Rolling per-grid episodes IN PLACE (``mgx_reset_episodes``): as ``reset_windows_rolling``
        without window buffers -- grid i reads row ``counter + row_off[i]`` of its own series, a (re)start rewrites two words per
        grid.  Single steps only, observation rows per step (no rings).  ``set_auto_reset`` makes the steps restart finished
        grids themselves."""
        if start.dtype != torch.int32 or tuple(start.shape) != (self.N,) or start.device != self.device:
            raise ValueError(f"start must be an int32 tensor of shape ({self.N},) on {self.device}")
        if length is not None and (length.dtype != torch.int32 or tuple(length.shape) != (self.N,) or length.device != self.device):
            raise ValueError(f"length must be an int32 tensor of shape ({self.N},) on {self.device}")
        if max_length is None:
            raise ValueError("max_length is required (the longest episode any later restart may ask for)")
        if validate:
            self._check_episodes(start, length, max_length)
        e = getattr(self, "_episodes", None)
        if e is None:
            e = self._episodes = dict(off=self._empty(self.N, dtype=torch.int32), final=self._empty(self.N, dtype=torch.int32))
        obs = self._obs_buf(out) if want_obs else None
        self._call(self._lib.mgx_reset_episodes, start.data_ptr(), _ptr(length), int(max_length), e["off"].data_ptr(),
                   e["final"].data_ptr(), _ptr(obs))
        self._window_start = start.clone()
        self._window_t0 = torch.zeros_like(start)
        self.window = (0, int(max_length))
        self._rolling_max = int(max_length)
        self._inplace = True
        self._final_obs = None
        self._t = 0
        return obs

    def set_auto_reset(self, enable=True, seed=0, fixed_length=0, lengths_out=None):
        """``mgx_set_auto_reset`` (in-place episodes): every single step restarts the grids whose episode it ends with the draw
        ``reset_grids_random(done, seed, fixed_length)`` would make after the step; the per-grid start rows / restart counters
        (``current_steps``) and ``lengths_out`` are updated in place by the step kernels."""
        if not getattr(self, "_inplace", False):
            raise _lib.MgxError(_lib.MGX_ERR_INVALID, "set_auto_reset: the engine is not stepping in-place episodes (reset_episodes)")
        check(self._lib.mgx_set_auto_reset(self._h, 1 if enable else 0, int(seed) & (2 ** 64 - 1), int(fixed_length),
                                           self._window_start.data_ptr() if enable else None, _ptr(lengths_out) if enable else None,
                                           self._window_t0.data_ptr() if enable else None))
        self._ar_lengths = lengths_out if enable else None          # (kept alive: the kernels write it)

    def set_final_obs(self, buf):
        """``mgx_set_final_obs``: the following single steps also write the observation BEFORE any restart into ``buf`` ([N, D]
        rows in the engine's observation format; None: off)."""
        if buf is not None:
            buf = self._obs_buf(buf)
        check(self._lib.mgx_set_final_obs(self._h, _ptr(buf)))
        self._final_obs = buf

    def reset_grids(self, mask, start, length=None, validate=True):
        """Restart the grids with ``mask[i] != 0`` at the current step (``mgx_reset_grids``): new start rows / lengths for
        them, everything else keeps running.  ``mask`` uint8 / bool [N], ``start`` / ``length`` int32 [N] on the device."""
        if mask.dtype == torch.bool:
            mask = mask.view(torch.uint8)
        for name, t, dt in (("mask", mask, torch.uint8), ("start", start, torch.int32), ("length", length, torch.int32)):
            if t is not None and (t.dtype != dt or tuple(t.shape) != (self.N,) or t.device != self.device or not t.is_contiguous()):
                raise ValueError(f"{name} must be a contiguous {dt} tensor of shape ({self.N},) on {self.device}")
        if self._window_t0 is None:
            raise _lib.MgxError(_lib.MGX_ERR_INVALID, "reset_grids: the engine is not in rolling-window mode")
        if validate:
            self._check_episodes(start, length, self._rolling_max, mask=mask)
        self._call(self._lib.mgx_reset_grids, mask.data_ptr(), start.data_ptr(), _ptr(length))
        m = mask.view(torch.bool)
        # in place: with set_auto_reset the step kernels hold these two tensors' addresses
        self._window_start.copy_(torch.where(m, start, self._window_start))
        self._window_t0.copy_(torch.where(m, torch.full_like(start, self.current_step), self._window_t0))

    def reset_grids_random(self, mask, seed, fixed_length=0, lengths_out=None):
        """``mgx_reset_grids_random``: restart the grids with ``mask[i] != 0`` with episodes drawn ON DEVICE (Philox of
        (seed; grid, counter); ``fixed_length`` > 0: FixedLengthStochasticTrajectory, 0: StochasticTrajectory).  The per-grid
        start rows / restart counters (``current_steps``) are updated in place by the kernel; ``lengths_out`` (int32 [N])
        optionally receives the drawn lengths."""
        if mask.dtype == torch.bool:
            mask = mask.view(torch.uint8)
        if mask.dtype != torch.uint8 or tuple(mask.shape) != (self.N,) or mask.device != self.device or not mask.is_contiguous():
            raise ValueError(f"mask must be a contiguous uint8 / bool tensor of shape ({self.N},) on {self.device}")
        if getattr(self, "_window_t0", None) is None:
            raise _lib.MgxError(_lib.MGX_ERR_INVALID, "reset_grids_random: the engine is not in rolling-window mode")
        self._call(self._lib.mgx_reset_grids_random, mask.data_ptr(), int(seed) & (2 ** 64 - 1), int(fixed_length),
                   self._window_start.data_ptr(), _ptr(lengths_out), self._window_t0.data_ptr())

    def set_shards(self, n_shards):
        """Step in ``n_shards`` independent grid ranges, one internal HIP stream each (``mgx_set_shards``).  While
        n_shards > 1 the stepping calls ignore torch's current stream: bracket them with ``fork()`` / ``join()``."""
        check(self._lib.mgx_set_shards(self._h, int(n_shards)))
        self.n_shards = int(n_shards)

    def set_launch_threads(self, mode=1):
        """Single-step calls in shards: shard j >= 1 issued by a resident host thread of the library while the caller issues shard
        0 (``mgx_set_launch_threads``): 0 never, 1 (default) inside ``step_many``, 2 / True also for every single-step call."""
        mode = 2 if mode is True else int(mode)
        check(self._lib.mgx_set_launch_threads(self._h, mode))

    def fork(self):
        """The shard streams wait for everything queued on torch's current stream (inputs produced there)."""
        self._call(self._lib.mgx_fork)

    def join(self):
        """Torch's current stream waits for everything issued to the shard streams (outputs are then safe to read)."""
        self._call(self._lib.mgx_join)

    def shard_streams(self):
        """``torch.cuda.ExternalStream`` views of the internal shard streams (for timing events)."""
        n = getattr(self, "n_shards", 1)
        if n <= 1:
            return []
        return [torch.cuda.ExternalStream(self._lib.mgx_shard_stream(self._h, j), device=self.device) for j in range(n)]

    def check_step(self, actions, normalized=True, out=None):
        """Dry run of ``step`` (``mgx_check_step``): int32 mask [N] of the requests the reference would refuse with
        ``raise_errors=True``; state and step counter are not touched."""
        actions = self._check_actions(actions, ())
        mask = out if out is not None else self._empty(self.N, dtype=torch.int32)
        self._call(self._lib.mgx_check_step, _ptr(actions), 1 if normalized else 0, mask.data_ptr())
        return mask

    def action_bounds(self, out=None):
        """``mgx_action_bounds``: (lo, hi) [N, A] float64 -- the normalised interval ``sample_action(strict_bound=True)`` draws
        every action column from at the CURRENT state and row (base_module.py:326-356)."""
        A = self.layout.action_dim
        lo, hi = out if out is not None else (self._empty(self.N, A), self._empty(self.N, A))
        self._call(self._lib.mgx_action_bounds, lo.data_ptr(), hi.data_ptr())
        return lo, hi

    def step_many(self, actions, normalized=True, want_obs=False, want_log=False, done=True, out=None):
        """K single-step launches issued by one call (``mgx_step_many``): actions [K, N, A] -> reward [K, N], done [K, N],
        optionally obs [K, N, D] and log [K, L, N]."""
        K = int(actions.shape[0])
        actions = self._check_actions(actions, (K,))
        out = out or {}
        reward = out.get("reward") if out.get("reward") is not None else self._empty(K, self.N)
        d = (out.get("done") if out.get("done") is not None else self._empty(K, self.N, dtype=torch.uint8)) if done else None
        if want_obs and getattr(self, "_obs_compact", False):
            raise ValueError("step_many writes whole rows per step: not offered with compact state rows")
        obs = (out.get("obs") if out.get("obs") is not None
               else torch.empty((K, self.N, self.obs_dim), dtype=self.obs_dtype, device=self.device)) if want_obs else None
        log = (out.get("log") if out.get("log") is not None else self._empty(K, self.log_dim, self.N)) if want_log else None
        self._call(self._lib.mgx_step_many, _ptr(actions), K, 1 if normalized else 0, reward.data_ptr(), _ptr(d), _ptr(obs),
                   _ptr(log))
        if self._t is not None:
            self._t += K
        return obs, reward, d, log

    def observe(self, out=None):
        obs = self._obs_buf(out)
        self._call(self._lib.mgx_observe, _ptr(obs))
        return obs

    def step(self, actions, normalized=True, want_obs=True, want_log=False, out=None, want_done=True):
        """One Microgrid.run for every grid.  Returns (obs|None, reward, done, log|None); ``out`` may hold
        preallocated ``obs`` / ``reward`` / ``done`` / ``log`` tensors.  ``want_done=False``: no per-grid done bytes are
        written (in lock-step `done` is the same for every grid: ``done_steps``)."""
        actions = self._check_actions(actions, ())
        reward = done = obs = log = None
        if out:
            reward, done, obs, log = out.get("reward"), out.get("done"), out.get("obs"), out.get("log")
        if reward is None:
            reward = self._empty(self.N)
        if not want_done:
            done = None
        elif done is None:
            done = self._empty(self.N, dtype=torch.uint8)
        obs = self._obs_buf(obs) if want_obs else None
        if not want_log:
            log = None
        elif log is None:
            log = self._empty(self.log_dim, self.N)
        self._call(self._lib.mgx_step, _ptr(actions), 1 if normalized else 0, reward.data_ptr(), _ptr(done),
                   _ptr(obs), _ptr(log))
        if self._t is not None:
            self._t += 1
        return obs, reward, done, log

    def step_k(self, actions, normalized=True, reward=True, done=False, soc_trace=False, status_trace=False,
               ret_acc=None, log=False, out=None):
        """K fused steps (actions [K, N, A]).  Returns a dict of the requested [K, N] outputs."""
        out = out or {}
        K = int(actions.shape[0]) if actions is not None else int(out["K"])
        actions = self._check_actions(actions, (K,))
        res = {}

        def buf(name, want, *shape, dtype=torch.float64):
            if not want:
                return None
            t = out.get(name)
            if t is None:
                t = self._empty(*shape, dtype=dtype)
            res[name] = t
            return t
        r = buf("reward", reward, K, self.N)
        if getattr(self, "_done_bits", False):
            d = buf("done", done, K, (self.N + 15) // 16, dtype=torch.int16)
        else:
            d = buf("done", done, K, self.N, dtype=torch.uint8)
        s = buf("soc_trace", soc_trace and self.layout.has_battery, K, self.N)
        g = buf("status_trace", status_trace and self.layout.has_genset, K, self.N, dtype=torch.int32)
        lg = buf("log", log, K, self.log_dim, self.N)
        if ret_acc is not None:
            res["ret_acc"] = ret_acc
        self._call(self._lib.mgx_step_k, _ptr(actions), K, 1 if normalized else 0, _ptr(r), _ptr(d), _ptr(s), _ptr(g),
                   _ptr(ret_acc), _ptr(lg))
        if self._t is not None:
            self._t += K
        return res

    def _table_ptr(self, table):
        """ctypes pointer + row count of a priority-list table; the conversion is cached per table object (a Gym loop
        passes the same table every step and this is ~6 us of the ~20 us a discrete step costs on the host)."""
        cached = getattr(self, "_table_cache", None)
        if cached is not None and cached[0] is table:
            return cached[2], cached[3]
        arr = np.ascontiguousarray(table, dtype=np.int32)
        if arr.ndim != 3 or arr.shape[1:] != (3, 2):
            raise ValueError("table must have shape [n_actions, 3, 2]")
        self._table_cache = (table, arr, arr.ctypes.data_as(_lib.c_i32_p), arr.shape[0])
        return self._table_cache[2], self._table_cache[3]

    def step_discrete(self, action_id, table, want_obs=True, want_log=False, want_control=False, out=None, want_done=True):
        """DiscreteMicrogridEnv.step for every grid in one launch: priority-list ids [N] (int32) are expanded and
        stepped in-kernel.  Returns (obs|None, reward, done, log|None, control|None)."""
        if action_id.dtype != torch.int32 or tuple(action_id.shape) != (self.N,) or action_id.device != self.device:
            raise ValueError(f"action_id must be an int32 tensor of shape ({self.N},) on {self.device}")
        tptr, n_lists = self._table_ptr(table)
        out = out or {}
        reward = out.get("reward") if out.get("reward") is not None else self._empty(self.N)
        done = (out.get("done") if out.get("done") is not None else self._empty(self.N, dtype=torch.uint8)) if want_done else None
        obs = self._obs_buf(out.get("obs")) if want_obs else None
        log = (out.get("log") if out.get("log") is not None else self._empty(self.log_dim, self.N)) if want_log else None
        control = (out.get("control") if out.get("control") is not None
                   else self._empty(self.N, self.action_dim)) if want_control else None
        self._call(self._lib.mgx_step_discrete, _ptr(action_id), tptr, n_lists,
                   _ptr(control), reward.data_ptr(), _ptr(done), _ptr(obs), _ptr(log))
        if self._t is not None:
            self._t += 1
        return obs, reward, done, log, control

    def step_lists(self, action_id, lists, want_obs=True, want_log=False, out=None, want_done=True):
        """DiscreteMicrogridEnv.step with priority lists over module instances (``mgx_step_lists``): one launch where the layout holds
        at most two modules of a kind, expand + step through a control buffer otherwise.  ``lists`` as for ``expand_lists``.
        Returns (obs|None, reward, done, log|None)."""
        if action_id.dtype != torch.int32 or tuple(action_id.shape) != (self.N,) or action_id.device != self.device or not action_id.is_contiguous():
            raise ValueError(f"action_id must be a contiguous int32 tensor of shape ({self.N},) on {self.device}")
        if lists.dtype != torch.int32 or lists.dim() != 3 or lists.shape[2] != 3 or lists.device != self.device or not lists.is_contiguous():
            raise ValueError(f"lists must be a contiguous int32 tensor [n_lists, list_len, 3] on {self.device}")
        out = out or {}
        reward = out.get("reward") if out.get("reward") is not None else self._empty(self.N)
        done = (out.get("done") if out.get("done") is not None else self._empty(self.N, dtype=torch.uint8)) if want_done else None
        obs = self._obs_buf(out.get("obs")) if want_obs else None
        log = (out.get("log") if out.get("log") is not None else self._empty(self.log_dim, self.N)) if want_log else None
        L = self.layout
        one_launch = L.multi and max(L.n_genset, L.n_battery, L.n_grid, L.n_load, L.n_pv) <= 2 and min(L.n_load, L.n_pv) >= 1
        control = None
        if not one_launch or out.get("control") is not None:       # (the buffer between the two launches; kept on the engine)
            control = out.get("control")
            if control is None:
                control = getattr(self, "_lists_control", None)
                if control is None:
                    control = self._lists_control = self._empty(self.N, self.action_dim)
        try:
            self._call(self._lib.mgx_step_lists, _ptr(action_id), _ptr(lists), int(lists.shape[0]), int(lists.shape[1]), _ptr(control),
                       reward.data_ptr(), _ptr(done), _ptr(obs), _ptr(log))
        except Exception:
            if control is not None:
                raise
            control = self._lists_control = self._empty(self.N, self.action_dim)      # (in-place episodes, noisy rows: two launches after all)
            self._call(self._lib.mgx_step_lists, _ptr(action_id), _ptr(lists), int(lists.shape[0]), int(lists.shape[1]), _ptr(control),
                       reward.data_ptr(), _ptr(done), _ptr(obs), _ptr(log))
        if self._t is not None:
            self._t += 1
        return obs, reward, done, log

    def rollout_discrete(self, action_id, table, K, reward=True, done=False, soc_trace=False, status_trace=False,
                         ret_acc=None, log=False, out=None):
        """K fused discrete steps with on-device action expansion.  ``action_id`` uint8: [K, N] (an id per step) or
        [N] (one fixed priority list per grid: rule-based control).  Returns the requested [K, N] outputs."""
        out = out or {}
        K = int(K)
        if action_id.dtype != torch.uint8 or action_id.device != self.device or not action_id.is_contiguous() \
                or tuple(action_id.shape) not in ((K, self.N), (self.N,)):
            raise ValueError(f"action_id must be a contiguous uint8 tensor [{K}, {self.N}] or [{self.N}] on {self.device}")
        per_step = int(action_id.dim() == 2)
        tptr, n_lists = self._table_ptr(table)
        res = {}

        def buf(name, want, *shape, dtype=torch.float64):
            if not want:
                return None
            t = out.get(name)
            if t is None:
                t = self._empty(*shape, dtype=dtype)
            res[name] = t
            return t
        r = buf("reward", reward, K, self.N)
        if getattr(self, "_done_bits", False):
            d = buf("done", done, K, (self.N + 15) // 16, dtype=torch.int16)
        else:
            d = buf("done", done, K, self.N, dtype=torch.uint8)
        s = buf("soc_trace", soc_trace and self.layout.has_battery, K, self.N)
        g = buf("status_trace", status_trace and self.layout.has_genset, K, self.N, dtype=torch.int32)
        lg = buf("log", log, K, self.log_dim, self.N)
        if ret_acc is not None:
            res["ret_acc"] = ret_acc
        self._call(self._lib.mgx_rollout_discrete, _ptr(action_id), per_step, tptr,
                   n_lists, K, _ptr(r), _ptr(d), _ptr(s), _ptr(g), _ptr(ret_acc), _ptr(lg))
        if self._t is not None:
            self._t += K
        return res

    def rollout_lists(self, action_id, lists, K, reward=True, done=False, soc_trace=False, status_trace=False, ret_acc=None,
                      log=False, out=None):
        """K fused discrete steps with priority lists over module instances (``mgx_rollout_lists``; every layout).
        ``action_id`` int32 [K, N] (an id per step) or [N] (one fixed list per grid); ``lists`` as for ``expand_lists``."""
        out = out or {}
        K = int(K)
        if action_id.dtype != torch.int32 or action_id.device != self.device or not action_id.is_contiguous() \
                or tuple(action_id.shape) not in ((K, self.N), (self.N,)):
            raise ValueError(f"action_id must be a contiguous int32 tensor [{K}, {self.N}] or [{self.N}] on {self.device}")
        if lists.dtype != torch.int32 or lists.dim() != 3 or lists.shape[2] != 3 or lists.device != self.device \
                or not lists.is_contiguous():
            raise ValueError(f"lists must be a contiguous int32 tensor [n_lists, list_len, 3] on {self.device}")
        res = {}

        def buf(name, want, *shape, dtype=torch.float64):
            if not want:
                return None
            t = out.get(name)
            if t is None:
                t = self._empty(*shape, dtype=dtype)
            res[name] = t
            return t
        r = buf("reward", reward, K, self.N)
        d = buf("done", done, K, self.N, dtype=torch.uint8)
        s = buf("soc_trace", soc_trace and self.layout.has_battery, K, self.N)
        g = buf("status_trace", status_trace and self.layout.has_genset, K, self.N, dtype=torch.int32)
        lg = buf("log", log, K, self.log_dim, self.N)
        if ret_acc is not None:
            res["ret_acc"] = ret_acc
        self._call(self._lib.mgx_rollout_lists, _ptr(action_id), int(action_id.dim() == 2), _ptr(lists), int(lists.shape[0]),
                   int(lists.shape[1]), K, _ptr(r), _ptr(d), _ptr(s), _ptr(g), _ptr(ret_acc), _ptr(lg))
        if self._t is not None:
            self._t += K
        return res

    def expand_discrete(self, action_id, table, out=None, violations=None):
        """priority-list ids [N] (int32) -> unnormalised control [N, A]; ``table`` int32 [n_actions, 3, 2].
        ``violations``: optional int32 [N] tensor that receives the assert the reference's ``_populate_action`` would have failed
        in this state (``_lib.V_EXPAND_*``, 0 = none; priority_list.py:73-154)."""
        if action_id.dtype != torch.int32 or tuple(action_id.shape) != (self.N,) or action_id.device != self.device:
            raise ValueError(f"action_id must be an int32 tensor of shape ({self.N},) on {self.device}")
        tptr, n_lists = self._table_ptr(table)
        control = out if out is not None else self._empty(self.N, self.action_dim)
        self._call(self._lib.mgx_expand_discrete, _ptr(action_id), tptr, n_lists, _ptr(control), _ptr(self._mask_arg(violations)))
        return control

    def _mask_arg(self, mask):
        if mask is not None and (mask.dtype != torch.int32 or tuple(mask.shape) != (self.N,) or mask.device != self.device
                                 or not mask.is_contiguous()):
            raise ValueError(f"violations must be a contiguous int32 tensor of shape ({self.N},) on {self.device}")
        return mask

    def check_discrete(self, action_id, table, out=None):
        """Dry run of ``step_discrete`` (``mgx_check_discrete``): int32 mask [N] -- the assert of ``_populate_action`` the
        reference would fail in this state (``_lib.V_EXPAND_*``), else the mask ``check_step`` gives for the expanded control;
        state and step counter are not touched."""
        if action_id.dtype != torch.int32 or tuple(action_id.shape) != (self.N,) or action_id.device != self.device:
            raise ValueError(f"action_id must be an int32 tensor of shape ({self.N},) on {self.device}")
        tptr, n_lists = self._table_ptr(table)
        mask = self._mask_arg(out) if out is not None else self._empty(self.N, dtype=torch.int32)
        self._call(self._lib.mgx_check_discrete, _ptr(action_id), tptr, n_lists, mask.data_ptr())
        return mask

    def expand_lists(self, action_id, lists, out=None, violations=None):
        """priority-list ids [N] (int32) -> unnormalised control [N, A] for lists over module instances
        (``mgx_expand_lists``): ``lists`` int32 device tensor [n_lists, list_len, 3] of (kind, instance, action).
        ``violations`` as for ``expand_discrete``."""
        if action_id.dtype != torch.int32 or tuple(action_id.shape) != (self.N,) or action_id.device != self.device:
            raise ValueError(f"action_id must be an int32 tensor of shape ({self.N},) on {self.device}")
        if lists.dtype != torch.int32 or lists.dim() != 3 or lists.shape[2] != 3 or lists.device != self.device \
                or not lists.is_contiguous():
            raise ValueError(f"lists must be a contiguous int32 tensor [n_lists, list_len, 3] on {self.device}")
        control = out if out is not None else self._empty(self.N, self.action_dim)
        self._call(self._lib.mgx_expand_lists, _ptr(action_id), _ptr(lists), int(lists.shape[0]), int(lists.shape[1]),
                   _ptr(control), _ptr(self._mask_arg(violations)))
        return control

    def metrics(self, values, out=None):
        """Column sums over the grids: values [M, N] -> [M] (deterministic LDS + wavefront-shuffle reduction)."""
        if values.dim() == 1:
            values = values.unsqueeze(0)
        if values.dtype != torch.float64 or values.shape[1] != self.N or not values.is_contiguous():
            raise ValueError(f"values must be contiguous float64 [M, {self.N}]")
        sums = out if out is not None else self._empty(values.shape[0])
        self._call(self._lib.mgx_metrics, _ptr(values), values.shape[0], _ptr(sums))
        return sums
