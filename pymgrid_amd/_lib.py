"""ctypes binding of ``libmgx.so`` (C ABI in ``include/mgx.h``).

The HIP library is the only compute path: if it is missing or no GPU is visible the package raises -- there is
no CPU fallback (the CPU oracle under ``oracle/`` is test infrastructure and is never imported from here).
"""
import ctypes as C
import json
import os
import re
import subprocess
import sys

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
LIB_PATH = os.environ.get("MGX_LIB") or os.path.join(_PKG, "libmgx.so")   # MGX_LIB: A/B kernel variants
SOURCES = [os.path.join(_PKG, "csrc", f) for f in ("mgx_abi.hip", "mgx_fused.hip", "mgx_kernels.hpp", "mgx_core.hpp")] + \
          [os.path.join(_ROOT, "include", "mgx.h")]
FUSED_PARTS = 6            # MGX_FUSED_PARTS: slices of mgx_fused.hip (the K-step kernels), compiled in parallel

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fPIC"]

MGX_OK, MGX_ERR_INVALID, MGX_ERR_UNSUPPORTED, MGX_ERR_RANGE, MGX_ERR_DEVICE = range(5)
# enum mgx_violation_bit
V_GENSET_RANGE, V_BATTERY_LIMIT, V_GRID_LIMIT, V_GENSET_GOAL, V_GENSET_NEGATIVE, V_NEGATIVE_LIMIT = 1, 2, 4, 8, 16, 32
V_EXPAND_CONSUME, V_EXPAND_PRODUCE, V_EXPAND_SIGN = 64, 128, 256
V_EXPAND = V_EXPAND_CONSUME | V_EXPAND_PRODUCE | V_EXPAND_SIGN      # states in which _populate_action asserts
V_ASSERTS = V_GENSET_GOAL | V_GENSET_NEGATIVE | V_NEGATIVE_LIMIT | V_EXPAND    # the reference raises whatever raise_errors says
ABI_VERSION = 9
ABI_MINOR = 2
# enum mgx_tunable (process-wide launch-shape knobs; set_tunable / get_tunable below)
TUNABLES = ("win_threads", "win_group", "win_pairs", "win_min_lds", "prefetch_pool", "multi_generic", "multi_small_own",
            "grid_major_copy", "fleet_byvalue", "launch_threads", "multi_static")
MAX_INSTANCES = 8          # MGX_MAX_INSTANCES: gensets / batteries / grids per microgrid


class MgxError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"[mgx error {code}] {message}")
        self.code = code


c_double_p = C.POINTER(C.c_double)
c_u32_p = C.POINTER(C.c_uint32)
c_u8_p = C.POINTER(C.c_uint8)
c_i32_p = C.POINTER(C.c_int32)


class Layout(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "struct_size", "n_grids", "n_steps", "horizon", "initial_step", "final_step",
        "has_genset", "has_battery", "has_grid", "n_load", "n_pv", "grid_before_battery", "n_genset", "n_battery", "n_grid",
        "flat_order")]


_F64_COLS = ("bat_min_capacity", "bat_max_capacity", "bat_max_charge", "bat_max_discharge", "bat_efficiency",
             "bat_cost_cycle", "gen_running_min", "gen_running_max", "gen_cost", "gen_co2_per_unit",
             "gen_cost_per_unit_co2")
_F64_COLS2 = ("grid_max_import", "grid_max_export", "grid_cost_per_unit_co2", "loss_load_cost", "overgeneration_cost",
              "load_ts", "pv_ts", "grid_ts", "load_lo", "load_hi", "pv_lo", "pv_hi", "grid_lo", "grid_hi",
              "load_noise_std", "pv_noise_std", "grid_noise_std", "charge", "soc")


# enum mgx_uniform_bit: parameter columns that may hold ONE value for the whole batch (mgx_columns.uniform_mask)
UNIFORM_BITS = ("bat_min_capacity", "bat_max_capacity", "bat_max_charge", "bat_max_discharge", "bat_efficiency", "bat_cost_cycle",
                "gen_running_min", "gen_running_max", "gen_cost", "gen_co2_per_unit", "gen_cost_per_unit_co2", "gen_times",
                "grid_max_import", "grid_max_export", "grid_cost_per_unit_co2", "loss_load_cost", "overgeneration_cost")

# factorised series (mgx_columns, include/mgx.h): base tables [T, PROFILE_PITCH] f64, profile ids / tariff uint8 [N], ratios f64
# [N], outage words [ceil(T / 64), N] (stored as int64: torch has no uint64)
FACTOR_COLUMNS = ("base_load", "base_pv", "base_co2", "load_profile", "pv_profile", "co2_profile", "tariff", "load_ratio",
                  "pv_ratio", "outage_bits")
PROFILE_PITCH = 8          # MGX_PROFILE_PITCH


class Columns(C.Structure):
    _fields_ = ([("struct_size", C.c_int32), ("uniform_mask", C.c_uint32)]
                + [(n, C.c_void_p) for n in _F64_COLS]
                + [("gen_times", C.c_void_p)]
                + [(n, C.c_void_p) for n in _F64_COLS2]
                + [("gen_status", C.c_void_p)]
                + [(n, C.c_void_p) for n in FACTOR_COLUMNS])


COLUMN_NAMES = tuple(n for n, _ in Columns._fields_[2:])


class FleetItem(C.Structure):
    """mgx_fleet_item (include/mgx.h): one batch of a heterogeneous fleet inside ``mgx_fleet_step``."""
    _fields_ = [("struct_size", C.c_int32), ("n_actions", C.c_int32), ("handle", C.c_void_p), ("actions", C.c_void_p),
                ("action_id", C.c_void_p), ("table", c_i32_p), ("reward", C.c_void_p), ("done", C.c_void_p),
                ("obs", C.c_void_p), ("log", C.c_void_p), ("refill_ring", C.c_void_p), ("refill_K", C.c_int32),
                ("refill_ahead", C.c_int32), ("refill_chunk", C.c_int32), ("refill_chunks", C.c_int32),
                ("wait_prefetch", C.c_int32), ("reserved", C.c_int32)]


class Synth(C.Structure):
    """mgx_synth (include/mgx.h): arguments of ``mgx_synthesize_series``."""
    _fields_ = ([("struct_size", C.c_int32), ("n_grids", C.c_int32), ("n_steps", C.c_int32),
                 ("n_load_profiles", C.c_int32), ("n_pv_profiles", C.c_int32), ("n_co2_profiles", C.c_int32)]
                + [(n, C.c_void_p) for n in ("base_load", "base_pv", "base_co2", "load_profile", "pv_profile", "co2_profile",
                                             "load_ratio", "pv_ratio", "tariff", "weak", "outage_per_day",
                                             "outage_duration")]
                + [("seed", C.c_uint64), ("grid_index0", C.c_int64), ("grid_index", C.c_void_p)]
                + [(n, C.c_void_p) for n in ("load_ts", "pv_ts", "grid_ts", "outage_bits")])

class Gen(C.Structure):
    """mgx_gen (include/mgx.h): arguments of ``mgx_generate_columns``."""
    _fields_ = ([(n, C.c_int32) for n in ("struct_size", "n_grids", "n_steps", "n_load_profiles", "n_pv_profiles", "n_co2_profiles",
                                          "mixed_timers", "n_mean_rows")]
                + [("seed", C.c_uint64), ("grid_index0", C.c_int64)]
                + [(n, C.c_void_p) for n in ("grid_index", "base_load", "load_max", "pv_max", "load_bound_max", "pv_bound_max",
                                             "co2_min", "co2_max")]
                + [("tariff_min", C.c_double * 3), ("tariff_max", C.c_double * 3)]
                + [(n, C.c_void_p) for n in ("arch", "load_profile", "pv_profile", "co2_profile", "tariff", "weak", "outage_duration",
                                             "outage_per_day", "load_ratio", "pv_ratio", "load_lo", "load_hi", "pv_lo", "pv_hi",
                                             "grid_lo", "grid_hi", "bat_min_capacity", "bat_max_capacity", "bat_max_charge",
                                             "bat_max_discharge", "charge", "soc", "gen_running_min", "gen_running_max", "gen_times",
                                             "gen_status", "grid_max_import", "grid_max_export", "d_bin_rand", "d_soc0_normal",
                                             "d_outage_normal", "d_size_load", "d_pv_pen", "d_bat_hours", "d_su", "d_wd")])


class EnvSlot(C.Structure):
    """mgx_env_slot (include/mgx.h): one set of rotating output buffers of ``mgx_env_step``."""
    _fields_ = [("reward", C.c_void_p), ("done", C.c_void_p), ("obs", C.c_void_p), ("log", C.c_void_p)]


ENV_MAX_SLOTS = 128        # MGX_ENV_MAX_SLOTS


class EnvPlan(C.Structure):
    """mgx_env_plan (include/mgx.h): what ``mgx_env_bind`` takes."""
    _fields_ = [("struct_size", C.c_int32), ("n_slots", C.c_int32), ("slots", C.POINTER(EnvSlot)), ("ring_K", C.c_int32),
                ("n_actions", C.c_int32), ("rings", C.c_void_p * 3), ("table", c_i32_p)]


# every symbol include/mgx.h declares: (restype, argtypes)
SYMBOLS = {
    "mgx_abi_version": (C.c_int, []),
    "mgx_abi_minor": (C.c_int, []),
    "mgx_set_tunable": (C.c_int, [C.c_int32, C.c_int64]),
    "mgx_get_tunable": (C.c_int, [C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "mgx_set_launch_threads": (C.c_int, [C.c_void_p, C.c_int32]),
    "mgx_action_bounds": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgx_last_error": (C.c_char_p, []),
    "mgx_create": (C.c_int, [C.POINTER(Layout), C.POINTER(Columns), C.POINTER(C.c_void_p)]),
    "mgx_destroy": (None, [C.c_void_p]),
    "mgx_action_dim": (C.c_int32, [C.c_void_p]),
    "mgx_obs_dim": (C.c_int32, [C.c_void_p]),
    "mgx_log_dim": (C.c_int32, [C.c_void_p]),
    "mgx_log_name": (C.c_char_p, [C.c_void_p, C.c_int32]),
    "mgx_current_step": (C.c_int32, [C.c_void_p]),
    "mgx_use_device_counter": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "mgx_set_window": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "mgx_set_reward_shaper": (C.c_int, [C.c_void_p, C.c_int32]),
    "mgx_set_forecast_noise": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int]),
    "mgx_set_obs_format": (C.c_int, [C.c_void_p, C.c_int32]),
    "mgx_set_obs_mode": (C.c_int, [C.c_void_p, C.c_int32]),
    "mgx_set_done_format": (C.c_int, [C.c_void_p, C.c_int32]),
    "mgx_normalise_series": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgx_set_action_format": (C.c_int, [C.c_void_p, C.c_int32]),
    "mgx_observe_windows": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "mgx_observe_windows_ahead": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "mgx_prefetch_wait": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mgx_reset": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "mgx_observe": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgx_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                           C.c_void_p]),
    "mgx_step_k": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                             C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgx_expand_discrete": (C.c_int, [C.c_void_p, C.c_void_p, c_i32_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgx_check_discrete": (C.c_int, [C.c_void_p, C.c_void_p, c_i32_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "mgx_set_ring_pitch": (C.c_int, [C.c_void_p, C.c_int32]),
    "mgx_set_ring_layout": (C.c_int, [C.c_void_p, C.c_int32]),
    "mgx_patch_windows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "mgx_expand_lists": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgx_rollout_lists": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int32, C.c_int32, C.c_int32] + [C.c_void_p] * 7),
    "mgx_step_lists": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32] + [C.c_void_p] * 6),
    "mgx_step_discrete": (C.c_int, [C.c_void_p, C.c_void_p, c_i32_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgx_rollout_discrete": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, c_i32_p, C.c_int32, C.c_int32, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgx_metrics": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "mgx_reset_windows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgx_reset_windows_rolling": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32] + [C.c_void_p] * 6),
    "mgx_reset_grids": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgx_reset_grids_random": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgx_reset_episodes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32] + [C.c_void_p] * 4),
    "mgx_set_auto_reset": (C.c_int, [C.c_void_p, C.c_int32, C.c_uint64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgx_set_final_obs": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mgx_check_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "mgx_step_many": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_void_p]),
    "mgx_env_bind": (C.c_int, [C.c_void_p, C.POINTER(EnvPlan)]),
    "mgx_env_seek": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "mgx_env_position": (C.c_int, [C.c_void_p, c_i32_p, c_i32_p, c_i32_p]),
    "mgx_env_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "mgx_env_step_discrete": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mgx_set_shards": (C.c_int, [C.c_void_p, C.c_int32]),
    "mgx_fork": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mgx_join": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mgx_shard_stream": (C.c_void_p, [C.c_void_p, C.c_int32]),
    "mgx_fleet_step": (C.c_int, [C.POINTER(FleetItem), C.c_int32, C.c_int, C.c_void_p]),
    "mgx_fleet_env_step": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int32, C.c_int, C.c_void_p]),
    "mgx_synthesize_series": (C.c_int, [C.POINTER(Synth), C.c_void_p]),
    "mgx_generate_columns": (C.c_int, [C.POINTER(Gen), C.c_void_p]),
}


def source_hash(extra=()):
    """sha256 (16 hex digits) over the contents of every source the library is built from + the compiler flags: what a built
    libmgx.so is stamped with (``<lib>.srchash``) and what bench.py / the profiles carry to say which kernels they measured."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(SOURCES):
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(HIPCC_FLAGS + list(extra)).encode())
    return h.hexdigest()[:16]


def built_hash(lib_path=None):
    """The source hash the library at ``lib_path`` was built from (None: not built here / no stamp)."""
    try:
        with open((lib_path or LIB_PATH) + ".srchash") as fh:
            return fh.read().strip() or None
    except OSError:
        return None


def up_to_date(lib_path=None):
    """The library exists and its stamp equals the hash of the sources as they are now -- file times play no part (a stale
    .so with a newer mtime than the sources, e.g. after a checkout, does not pass for current)."""
    path = lib_path or LIB_PATH
    return os.path.exists(path) and built_hash(path) == source_hash()


def build(force=False, verbose=False, defs=(), lib_path=None, abi_only=False):
    """Compile csrc/*.hip for gfx950 into pymgrid_amd/libmgx.so (hipcc cross-compiles without a GPU).
    ``defs`` / ``lib_path``: an A/B variant of the kernels (extra -D flags, e.g. ("-DMGX_RING=8",)) built beside the product
    library; load it with MGX_LIB=<lib_path>."""
    if lib_path is not None or defs:
        # abi_only: the variant's flags only concern kernels of mgx_abi.hip (everything but the K-step loops): compile that unit
        # alone and link it with the product build's mgx_fused objects
        return _build(lib_path or LIB_PATH, list(defs), verbose, os.path.basename(lib_path or "variant") + ".o", abi_only=abi_only)
    if os.environ.get("MGX_LIB") and os.path.exists(LIB_PATH) and not force:
        return LIB_PATH                              # an A/B variant named by the caller: loaded as it is, never rebuilt in place
    if not force and up_to_date(LIB_PATH):
        return LIB_PATH
    return _build(LIB_PATH, [], verbose, "", force)


def _build(LIB_PATH, extra_defs, verbose, objtag, force=True, abi_only=False):
    # several ranks may get here at once (torchrun): serialise on a lock file, compile to a temporary name and
    # rename atomically so that nobody ever dlopens a half-written library
    import fcntl
    with open(LIB_PATH + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not extra_defs and up_to_date(LIB_PATH):
                return LIB_PATH                      # another process built it while we waited
            stamp = source_hash(extra_defs)          # (of the sources as they are when the compilers start)
            hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
            tmp = f"{LIB_PATH}.{os.getpid()}.tmp"
            objdir = os.path.join(_PKG, "csrc", "_build" + ("_" + objtag if objtag else ""))
            os.makedirs(objdir, exist_ok=True)
            # translation units: the host side + small kernels, and the slices of the K-step kernels -- compiled in parallel
            units = [(SOURCES[0], [], os.path.join(objdir, "mgx_abi.o"))] + \
                    [(SOURCES[1], [f"-DMGX_FUSED_PART={p}"], os.path.join(objdir, f"mgx_fused_{p}.o")) for p in range(FUSED_PARTS)]
            if abi_only:
                base = os.path.join(_PKG, "csrc", "_build")
                fused = [os.path.join(base, f"mgx_fused_{p}.o") for p in range(FUSED_PARTS)]
                if not all(os.path.exists(f) for f in fused):
                    raise FileNotFoundError("abi_only variants link the product build's mgx_fused objects: build() first")
                units = units[:1]
            procs = []
            for src, defs, obj in units:
                # -Rpass-analysis=kernel-resource-usage: the backend's per-kernel register / scratch figures as remarks (free):
                # kept in <objdir>/resource_usage.json -- a hot kernel that starts using scratch memory costs launch time
                # (24 B of private segment in the single-step kernels once cost 0.4-1 us per launch: NOTES.md)
                cmd = [hipcc] + HIPCC_FLAGS + ["-Rpass-analysis=kernel-resource-usage"] + defs + extra_defs + ["-c", src, "-o", obj]
                if verbose:
                    print(" ".join(cmd))
                procs.append((cmd, subprocess.Popen(cmd, stderr=subprocess.PIPE, text=True)))
            usage = {}
            for cmd, p in procs:
                _, err = p.communicate()
                if p.returncode != 0:
                    sys.stderr.write(err)
                    for _, q in procs:
                        if q.poll() is None:
                            q.kill()
                    raise subprocess.CalledProcessError(p.returncode, cmd)
                usage.update(_parse_resource_usage(err))
                rest = "\n".join(ln for ln in err.splitlines() if "kernel-resource-usage" not in ln and not _REMARK_ECHO.match(ln))
                if rest.strip() and verbose:
                    sys.stderr.write(rest + "\n")
            with open(os.path.join(objdir, "resource_usage.json"), "w") as fh:
                json.dump(usage, fh, indent=0, sort_keys=True)
            link = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared"] + [obj for _, _, obj in units] + (fused if abi_only else []) + ["-o", tmp]
            if verbose:
                print(" ".join(link))
            subprocess.run(link, check=True)
            os.replace(tmp, LIB_PATH)
            with open(LIB_PATH + ".srchash.tmp", "w") as fh:
                fh.write(stamp + "\n")
            os.replace(LIB_PATH + ".srchash.tmp", LIB_PATH + ".srchash")
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


_REMARK_ECHO = re.compile(r"^\s*(\d+ \||\||In file included from)")     # the source echo under each remark


def _parse_resource_usage(stderr_text):
    """{demangled kernel name: {"vgpr", "sgpr", "scratch", "lds"}} out of hipcc's kernel-resource-usage remarks."""
    out, cur = {}, None
    for ln in stderr_text.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", ln)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        if cur is None:
            continue
        for key, pat in (("sgpr", r"TotalSGPRs: (\d+)"), ("vgpr", r"remark:\s+VGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("lds", r"LDS Size \[bytes/block\]: (\d+)"), ("sgpr_spill", r"SGPRs Spill: (\d+)"), ("vgpr_spill", r"VGPRs Spill: (\d+)")):
            m = re.search(pat, ln)
            if m:
                cur[key] = int(m.group(1))
    if out:
        try:
            names = list(out)
            dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
            out = {(d.split("(")[0].replace("void ", "") or n): out[n] for n, d in zip(names, dem)}
        except OSError:
            pass
    return out


def resource_usage():
    """The per-kernel resource figures of the last build on this machine (None when the library was built elsewhere)."""
    f = os.path.join(_PKG, "csrc", "_build", "resource_usage.json")
    try:
        with open(f) as fh:
            return json.load(fh)
    except (OSError, ValueError):
        return None


_lib = None


def lib():
    """Load libmgx.so (must have been built: ``python -c 'import __graft_entry__ as g; g.build()'``)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with __graft_entry__.build() "
                              f"(pymgrid_amd has no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)       # AttributeError if the symbol is not exported
            fn.restype, fn.argtypes = res, args
        if L.mgx_abi_version() != ABI_VERSION:
            raise ImportError(f"libmgx.so ABI {L.mgx_abi_version()} != binding ABI {ABI_VERSION}; rebuild")
        assert C.sizeof(Layout) == 64
        _lib = L
    return _lib


def check(rc):
    if rc != MGX_OK:
        raise MgxError(rc, lib().mgx_last_error().decode())


def set_tunable(name, value):
    """mgx_set_tunable by name (TUNABLES): process-wide launch-shape knobs -- the library reads no environment variables."""
    check(lib().mgx_set_tunable(TUNABLES.index(name), int(value)))


def get_tunable(name):
    """(current value, default) of a tunable."""
    v, d = C.c_int64(), C.c_int64()
    check(lib().mgx_get_tunable(TUNABLES.index(name), C.byref(v), C.byref(d)))
    return v.value, d.value
