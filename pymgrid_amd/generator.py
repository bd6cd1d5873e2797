"""Synthetic microgrid batches drawn with the sizing rules of the reference's ``MicrogridGenerator``
(MicrogridGenerator.py:214-603) -- used by bench.py and the large parity tests.

Per-grid scalars come from a counter-based Philox stream over the GLOBAL grid index, so rank r of W draws exactly
rows [r*N/W, (r+1)*N/W) of the same global batch whatever W is.  Series are base profile x per-grid scale, built
on the device row-block by row-block (the [T, N] arrays never exist on the host).
"""
import numpy as np
import torch

from .batch import BatchLayout, MicrogridBatch, pack_status, pack_times

ARCHS = {"genset+battery": (True, True, False), "battery+grid": (False, True, True),
         "genset+battery+grid": (True, True, True), "loadpv": (False, False, False)}


def _base_profiles(T, seed):
    """5 load shapes, 5 pv shapes, 2 co2 shapes (hourly, peak-normalised) -- stand-ins for data/load, data/pv,
    data/co2 of the reference, which are not available on the GPU box."""
    rs = np.random.Generator(np.random.Philox(key=seed + 0x5eed))
    t = np.arange(T)
    hour, day = t % 24, t // 24
    load, pv, co2 = [], [], []
    for k in range(5):
        daily = 0.55 + 0.3 * np.sin(2 * np.pi * (hour - 7 - k) / 24) + 0.1 * np.sin(4 * np.pi * (hour + k) / 24)
        season = 1.0 + 0.15 * np.cos(2 * np.pi * (day - 30 * k) / 365.0)
        x = np.clip(daily * season * (1 + 0.05 * rs.standard_normal(T)), 0.05, None)
        load.append(x / x.max())
        sun = np.clip(np.sin(np.pi * (hour - 6) / 12.0), 0, None) ** (1.0 + 0.1 * k)
        cloud = np.clip(0.75 + 0.25 * np.cos(2 * np.pi * (day + 20 * k) / 365.0) - 0.3 * rs.random(T), 0, 1)
        y = sun * cloud
        pv.append(y / max(y.max(), 1e-12))
    for k in range(2):
        co2.append(0.25 + 0.1 * k + 0.1 * np.sin(2 * np.pi * (hour - 15) / 24) + 0.02 * rs.random(T))
    return np.stack(load, 1), np.stack(pv, 1), np.stack(co2, 1)


def _tariff(T, pattern):
    """MicrogridGenerator._get_electricity_tariff (:253-285): pattern 1 {0.22, 0.29, 0.59}, pattern 2 {0.08, 0.11}."""
    hour = np.arange(T) % 24
    if pattern == 1:
        return np.where((hour >= 17) & (hour < 21), 0.59, np.where((hour >= 8) & (hour < 23), 0.29, 0.22))
    return np.where((hour >= 8) & (hour < 22), 0.11, 0.08)


def draw_scalars(n_total, seed=42, arch="genset+battery", mixed_timers=False):
    """Per-grid scalar draws for the GLOBAL batch (cheap: a few doubles per grid)."""
    rs = np.random.Generator(np.random.Philox(key=seed))
    d = {}
    d["peak"] = rs.integers(100, 100001, n_total).astype(np.float64)          # load size U{100..100000} (:437-441)
    d["load_pid"] = rs.integers(0, 5, n_total)
    d["pv_pid"] = rs.integers(0, 5, n_total)
    d["pv_pen"] = rs.integers(30, 151, n_total) / 100.0                       # PV penetration 30..150 % (:357)
    d["bat_hours"] = rs.integers(3, 6, n_total).astype(np.float64)            # battery 3..5 h of mean load (:385)
    d["soc0"] = np.clip(rs.standard_normal(n_total), 0.2, 1.0)                # (:239)
    d["su"] = rs.integers(0, 4, n_total) if mixed_timers else np.zeros(n_total, np.int64)
    d["wd"] = rs.integers(0, 4, n_total) if mixed_timers else np.zeros(n_total, np.int64)
    d["weak"] = rs.random(n_total) < 0.5
    d["tariff"] = rs.integers(1, 3, n_total)
    d["co2_pid"] = rs.integers(0, 2, n_total)
    return d


def _hash_uniform(rows, gidx, seed):
    """U[0, 1) per (series row, GLOBAL grid index): a counter-based hash (splitmix64-style mixing in wrapping int64
    arithmetic), so a shard's draw does not depend on how many ranks / shards the batch is split over."""
    def lsr(v, k):                                     # logical shift right of the two's-complement bit pattern
        return (v >> k) & ((1 << (64 - k)) - 1)
    x = rows * -7046029254386353131 + gidx * -4658895280553007687 + (int(seed) * 1000003 + 12345)
    x = (x ^ lsr(x, 30)) * -4658895280553007687
    x = (x ^ lsr(x, 27)) * -7723592293110705685
    x = x ^ lsr(x, 31)
    return lsr(x, 11).to(torch.float64) * (1.0 / 9007199254740992.0)


def generate(n_grids, n_steps=8760, seed=42, arch="genset+battery", horizon=0, device="cuda", rank=0, world=1,
             mixed_timers=False, final_step=0, row_block=256):
    """Build the [rank]-th shard of a global batch of ``n_grids`` microgrids on ``device``."""
    has_genset, has_battery, has_grid = ARCHS[arch]
    if n_grids % world:
        raise ValueError("n_grids must be divisible by the number of ranks")
    per = n_grids // world
    lo, hi = rank * per, (rank + 1) * per
    d = {k: v[lo:hi] for k, v in draw_scalars(n_grids, seed, arch, mixed_timers).items()}
    T, N = n_steps, per
    base_load, base_pv, base_co2 = _base_profiles(T, seed)
    dev = torch.device(device)
    f64 = dict(dtype=torch.float64, device=dev)

    load_scale = d["peak"]                                   # profiles are peak-normalised
    pv_scale = d["peak"] * d["pv_pen"]
    mean_load = base_load.mean(0)[d["load_pid"]] * load_scale
    cols = {}

    def up(a):
        return torch.as_tensor(np.ascontiguousarray(a), **f64)

    bl, bp = up(base_load), up(base_pv)
    lpid, ppid = torch.as_tensor(d["load_pid"], device=dev), torch.as_tensor(d["pv_pid"], device=dev)
    ls, ps = up(load_scale), up(pv_scale)
    load_ts, pv_ts = torch.empty(T, N, **f64), torch.empty(T, N, **f64)
    for r0 in range(0, T, row_block):
        r1 = min(T, r0 + row_block)
        load_ts[r0:r1] = -(bl[r0:r1][:, lpid] * ls)          # stored sign: load <= 0
        pv_ts[r0:r1] = bp[r0:r1][:, ppid] * ps
    cols["load_ts"], cols["pv_ts"] = load_ts, pv_ts
    cols["load_lo"] = -(up(base_load.max(0))[lpid] * ls); cols["load_hi"] = torch.zeros(N, **f64)
    cols["pv_lo"] = torch.zeros(N, **f64);               cols["pv_hi"] = up(base_pv.max(0))[ppid] * ps
    cols["loss_load_cost"] = torch.full((N,), 10.0, **f64)
    cols["overgeneration_cost"] = torch.full((N,), 1.0, **f64)

    if has_battery:                                           # _get_battery / _size_battery (:230-243,:382-386)
        cap = np.ceil(d["bat_hours"] * mean_load)
        cols["bat_max_capacity"] = up(cap)
        cols["bat_min_capacity"] = up(0.2 * cap)
        cols["bat_max_charge"] = up(np.ceil(cap / 4)); cols["bat_max_discharge"] = up(np.ceil(cap / 4))
        cols["bat_efficiency"] = torch.full((N,), 0.9, **f64)
        cols["bat_cost_cycle"] = torch.full((N,), 0.02, **f64)
        cols["soc"] = up(d["soc0"]); cols["charge"] = up(d["soc0"] * cap)     # battery_module.py:96-106
    if has_genset:                                            # _get_genset / _size_genset (:214-228,:372-379)
        rated = np.ceil(d["peak"] / 0.9)
        cols["gen_running_min"] = up(0.05 * rated); cols["gen_running_max"] = up(0.9 * rated)
        cols["gen_cost"] = torch.full((N,), 0.4, **f64)
        cols["gen_co2_per_unit"] = torch.full((N,), 2.0, **f64)
        cols["gen_cost_per_unit_co2"] = torch.full((N,), 0.1, **f64)
        cols["gen_times"] = torch.from_numpy(pack_times(d["su"], d["wd"]).view(np.int32).copy()).to(dev)
        st = pack_status(np.ones(N, np.int64), np.ones(N, np.int64), np.zeros(N, np.int64), d["wd"])   # init on
        cols["gen_status"] = torch.from_numpy(st.view(np.int32).copy()).to(dev)
    if has_grid:                                              # _get_grid (:288-319)
        cols["grid_max_import"] = up(2 * d["peak"]); cols["grid_max_export"] = up(2 * d["peak"])
        cols["grid_cost_per_unit_co2"] = torch.full((N,), 0.1, **f64)
        tariffs = up(np.stack([_tariff(T, 1), _tariff(T, 2)], 1))
        tid = torch.as_tensor(d["tariff"] - 1, device=dev)
        cid = torch.as_tensor(d["co2_pid"], device=dev)
        bc = up(base_co2)
        weak = torch.as_tensor(d["weak"], device=dev)
        gidx = torch.arange(lo, hi, dtype=torch.int64, device=dev).unsqueeze(0)          # GLOBAL grid index
        grid_ts = torch.empty(T, 4, N, **f64)
        for r0 in range(0, T, row_block):
            r1 = min(T, r0 + row_block)
            grid_ts[r0:r1, 0] = tariffs[r0:r1][:, tid]
            grid_ts[r0:r1, 1] = 0.0
            grid_ts[r0:r1, 2] = bc[r0:r1][:, cid]
            rows = torch.arange(r0, r1, dtype=torch.int64, device=dev).unsqueeze(1)
            outage = (_hash_uniform(rows, gidx, seed) < 0.02) & weak              # weak-grid outages (:321-340)
            grid_ts[r0:r1, 3] = (~outage).to(torch.float64)
        cols["grid_ts"] = grid_ts
        cols["grid_lo"] = grid_ts.amin(dim=0).contiguous(); cols["grid_hi"] = grid_ts.amax(dim=0).contiguous()
    layout = BatchLayout(n_grids=N, n_steps=T, horizon=horizon, initial_step=0, final_step=final_step,
                         has_genset=has_genset, has_battery=has_battery, has_grid=has_grid)
    return MicrogridBatch(layout, {k: v.contiguous() for k, v in cols.items()})
