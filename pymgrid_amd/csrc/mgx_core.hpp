// mgx_core.hpp -- per-microgrid device arithmetic of the batched step engine (gfx950 / CDNA4).
//
// One lane owns one microgrid.  Everything here is straight fp64 in the reference's operation order
// (compile with -ffp-contract=off: the reference never fuses a multiply-add), so results are bit-identical
// to the CPU loop of Total-RD/pymgrid v1.2.2 (paths below are relative to src/pymgrid/):
//   Microgrid.run                     microgrid/microgrid.py:227-325
//   MicrogridStep.append/balance      microgrid/utils/step.py:13-36
//   BaseMicrogridModule.step & clips  modules/base/base_module.py:95-274
//   BatteryModule                     modules/battery_module.py:108-147,244-291,332-338
//   GensetModule                      modules/genset_module.py:100-149,188-346,465-517
//   GridModule                        modules/grid_module.py:125-228,314-320
//   LoadModule / RenewableModule      modules/load_module.py:86-111, modules/renewable_module.py:86-110
//   UnbalancedEnergyModule            modules/unbalanced_energy_module.py:28-70
//   ModuleSpace normalise/denormalise utils/space.py:184-231
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mgx.h"

namespace mgx {

// Layout flags: template parameter F of the kernels (one specialisation per module set).
// F_GRID_FIRST (only with battery AND grid): the GridModule precedes the BatteryModule in the microgrid's module list,
// so it is stepped -- and its energies / reward are added to the running sums -- before the battery
// (module_container.py:355-413: controllable = pure sources first, then sources-and-sinks in list order).
enum : int { F_GENSET = 1, F_BATTERY = 2, F_GRID = 4, F_GRID_FIRST = 8 };

// Log columns, in output order.  Names are returned by mgx_log_name().
enum LogCol : int {
    LC_REWARD = 0, LC_FIXED_PROVIDED, LC_FIXED_ABSORBED, LC_CTRL_PROVIDED, LC_CTRL_ABSORBED,
    LC_OVERALL_PROVIDED, LC_OVERALL_ABSORBED,
    LC_LOAD_MET, LC_RENEWABLE_USED, LC_CURTAILMENT, LC_LOSS_LOAD, LC_OVERGENERATION, LC_UNBALANCED_REWARD,
    LC_COMMON_END,
    // genset block (4), battery block (5), grid block (4) follow, present modules only
    LC_GENSET_N = 4, LC_BATTERY_N = 5, LC_GRID_N = 4
};

// Everything a launch needs; passed to kernels BY VALUE (kernarg segment, scalar loads).
struct KArgs {
    mgx_columns c;
    int32_t N, T, H, final_step;
    int32_t obs_dim, log_dim;
    int32_t n_load, n_pv;    // load / renewable modules per grid (1 on the fast path, <= MGX_MAX_MODULES otherwise)
    int32_t n_genset, n_battery, n_grid;   // controllable module instances per grid (0 / 1 on the fast path, <= MGX_MAX_INSTANCES)
    int32_t obs_f32;         // observation rows are written as float (RN of the fp64 value) instead of double
    int32_t act_f32;         // continuous actions arrive as float (widened to double exactly) instead of double
    int32_t obs_state_only;  // obs arguments of step / observe receive ONLY the state columns (windows were prefetched):
                             // 1 = inside full rows [N, D], 2 = as a dense [N, S] array (MGX_OBS_ROWS_STATE_COMPACT)
    int32_t done_bits;       // fused launches write `done` as bit sets ([K, ceil(N / 16)] uint16) instead of bytes
    // mgx_set_ring_layout(MGX_RING_COLUMNS): the blocks of the observation rings are COLUMN-major -- value (grid i, column c) of a
    // block at c * obs_colpitch + i (obs_colpitch = the ring pitch in grids, >= N): a wave's 64 grids are then 512 consecutive
    // bytes of every column, for the refill and for the state columns the step adds (0 = row-major blocks [N, D])
    int32_t obs_colpitch;
    // first column of every module's block inside a flat observation row (mgx_layout.flat_order); fast path only
    int32_t col_load, col_pv, col_gen, col_bat, col_grid;
    int32_t shaper;          // mgx_reward_shaper
    int32_t noise_increase;  // GaussianNoiseForecaster.increase_uncertainty
    uint64_t noise_seed;
    // Device-resident step counter (NULL = the host passes t).  With it a sequence of steps can be captured in a
    // hipGraph and replayed: kernels read the counter, `advance_counter` moves it, the kernel argument t is then an
    // offset relative to it.  A replay that would run past the series is clamped to the last row and flagged.
    int32_t *t_dev;
    // Sub-range [g0, g1) of the grids a stepping launch owns (0, N unless the handle steps in shards: every shard is a
    // launch of its own, on its own HIP stream; column pointers and row strides stay those of the whole batch).
    int32_t g0, g1;
    // Per-grid episode ends (mgx_reset_windows): done_i = t >= grid_final[i] - 1 instead of the batch-wide final_step.
    const int32_t *grid_final;
    // Rolling per-grid windows (mgx_reset_windows_rolling): the window buffers are rings of 2^p rows addressed by
    // (step counter & row_mask); -1 (all ones: the identity) everywhere else.
    int32_t row_mask;
    // Per-grid episodes IN PLACE (mgx_reset_episodes): grid i reads row counter + ep_off[i] of its own
    // series (no window buffers), done_i = counter >= ep_final[i] - 1 (ep_final == grid_final, writable).  NULL everywhere else.
    int32_t *ep_off, *ep_final;
    // mgx_set_auto_reset: a single step restarts the grids whose episode it ends -- mgx_reset_grids_random's draw at the counter
    // value after the step, inside the step kernel -- and its observation is then the first one of the new episode
    int32_t ar_mode;                        // 0 off, 1 on
    int32_t ar_fixed_length, ar_lo, ar_hi, ar_max_length;
    uint64_t ar_seed;
    int32_t *ar_start_io, *ar_length_io, *ar_t0_io;
    void *final_obs;                        // mgx_set_final_obs: the observation BEFORE the restart (rows written inline only)
    // In-place episodes read the series with one row per LANE.  Factorised: c.base_load / base_pv / base_co2 then point at the
    // handle's PROFILE-major copies [MGX_PROFILE_PITCH, pm_pitch] (a lane's consecutive rows share a line; the public [T, 8] layout
    // has a 64-byte line per row, which suits the lock-step kernels where every lane reads the same row).  [T, N] series:
    // c.load_ts / pv_ts / grid_ts point into ONE grid-major copy [N, pm_pitch, 2 or 6] (ts_index).  0 = the public layouts.
    int32_t pm_pitch;
};

// done flag of grid i at step counter t: _done(), base_timeseries_module.py:124-125 (evaluated before the counter moves)
__device__ __forceinline__ uint8_t done_at(const KArgs &a, int64_t i, int32_t t)
{
    const int32_t fin = a.grid_final ? a.grid_final[i] : a.final_step;
    return (uint8_t)(t >= fin - 1);
}

// series row of grid i at step counter t: the counter itself (rolling window buffers: a ring), or the grid's own row during
// in-place episodes.  For the kernels off the hot path; step_kernel / step_discrete_kernel have a compile-time form.
// In-place episodes: a grid that is done and has not been restarted keeps stepping on its own series; once that would leave
// the series (row >= T) it re-reads the LAST row -- the counter of this mode never ends, so the row is clamped here rather
// than refused by the host (include/mgx.h, mgx_reset_episodes).
__device__ __forceinline__ int64_t episode_row(const KArgs &a, int32_t t, int32_t off)
{
    const int64_t row = (int64_t)t + off;
    return row < a.T ? (row < 0 ? 0 : row) : (int64_t)a.T - 1;
}
__device__ __forceinline__ int64_t series_row(const KArgs &a, int64_t i, int32_t t)
{
    return a.ep_off ? episode_row(a, t, a.ep_off[i]) : (int64_t)(t & a.row_mask);
}

__device__ __forceinline__ int32_t resolve_t(const KArgs &a, int32_t t)
{
    if (a.t_dev) {
        const int32_t td = a.t_dev[0] + t;
        if (td >= a.T) { a.t_dev[1] = 1; return a.T - 1; }     // sticky overrun flag (all lanes write the same value)
        return td;
    }
    return t;
}

// Device-counter mode: the LAST workgroup of a stepping kernel to finish moves the counter by `k` steps
// (counter[2] counts finished workgroups).  Every workgroup read the counter at its very start, i.e. before its own
// arrival here, so nobody can observe the new value inside this launch; the next launch sees it (kernel boundary).
// Inside a workgroup the waves are NOT ordered by themselves, hence the barrier: every wave of the arriving workgroup has
// read the counter (resolve_t at its start; waves that left through the range check do not hold the barrier back on
// AMD hardware) before thread 0 signs the workgroup off.  The counter words are written with agent-scope atomics so the
// next launch -- possibly on another XCD's L2 -- reads the new value.
__device__ __forceinline__ void advance_counter_in_kernel(const KArgs &a, int32_t k)
{
    if (a.t_dev == nullptr) return;              // kernel argument: uniform over the whole launch
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned prev = atomicAdd((unsigned *)&a.t_dev[2], 1u);
        if (prev == gridDim.x - 1) {
            const int32_t t = __hip_atomic_load(&a.t_dev[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&a.t_dev[2], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&a.t_dev[0], t + k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// observation kernels: no clamp (t >= T is legal there: the row is end-of-series padding)
__device__ __forceinline__ int32_t resolve_t_obs(const KArgs &a, int32_t t) { return a.t_dev ? a.t_dev[0] + t : t; }

// K-step kernels in device-counter mode: never walk past the series (steps beyond it are dropped and flagged)
__device__ __forceinline__ int32_t resolve_k(const KArgs &a, int32_t t0, int32_t K)
{
    if (a.t_dev && t0 + K > a.T) { a.t_dev[1] = 1; return a.T - t0; }
    return K;
}

struct Params {
    double bat_cmin, bat_cmax, bat_C, bat_D, bat_eta, bat_cost;
    double gen_rmin, gen_rmax, gen_cost, gen_co2, gen_cco2;
    uint32_t gen_times;
    double grid_imp, grid_exp, grid_cco2;
    double ll_cost, og_cost;
};

struct State {
    double charge, soc;
    uint32_t status;
};

// Step-invariant values derived from the parameters (hoisted out of the K-step loop; same operations, same
// operands, hence the same bits as recomputing them every step like the reference does).
struct Derived {
    double bat_lo, bat_sp;     // battery action space: low = -max_discharge/eta, spread (battery_module.py:332-338)
    double gen_sp;             // genset energy action: low 0, spread running_max (genset_module.py:511-517)
    double grid_lo, grid_sp;   // grid action space: low = -max_export, spread (grid_module.py:125-132)
};

struct Inputs {
    double a_goal, a_gen, a_bat, a_grid;     // control (normalised or raw)
    double load, pv;                         // series rows, stored sign
    double g_pimp, g_pexp, g_co2, g_stat;    // grid_ts row
};

struct Outputs {
    double reward;
    double fixed_provided, fixed_absorbed, ctrl_provided, ctrl_absorbed, overall_provided, overall_absorbed;
    double load_met, renewable_used, curtailment, loss_load, overgeneration, unbalanced_reward;
    double genset_production, genset_co2, genset_reward;
    double discharge_amount, charge_amount, battery_reward, soc_pre, charge_pre;
    double grid_import, grid_export, grid_co2, grid_reward;
    // requests the reference would have refused with raise_errors=True (base_module.py:79-93,213-224,265-270) or
    // always refuses (enum mgx_violation_bit, include/mgx.h): bit 0 genset request outside [min, max] production, bit 1
    // battery request above its limit, bit 2 grid request above its limit, bit 3 genset goal outside [0, 1] (AssertionError,
    // genset_module.py:147), bit 4 negative genset energy (a pure source asked to absorb), bit 5 a battery / grid acting at a
    // NEGATIVE limit (`assert absorbed_energy >= 0`, base_module.py:272: a lossy battery one ulp above max_capacity asked to
    // absorb; `assert internal_energy_change <= 0`, battery_module.py:114: one below min_capacity asked to produce);
    // bits 6-8 come from the discrete expansion (populate_core<F, true>)
    uint32_t violations;
};

// ---- ModuleSpace (utils/space.py:204-205,213,224) -------------------------------------------------------
__device__ __forceinline__ double space_spread(double lo, double hi)
{
    double s = hi - lo;
    return s == 0.0 ? 1.0 : s;
}
__device__ __forceinline__ double space_denorm(double lo, double hi, double v) { return lo + space_spread(lo, hi) * v; }
__device__ __forceinline__ double space_norm(double lo, double hi, double v) { return (v - lo) / space_spread(lo, hi); }

// ---- min / max / clip ---------------------------------------------------------------------------------------------
// The reference's `min(a, b)` (= b if b < a else a), `max`, and its clip ladders (base_module.py:213-224,265-270) are
// one v_min_f64 / v_max_f64 each instead of a compare and two 32-bit selects -- a third of the VALU instructions of the
// rule-based rollout were such selects.  For every non-NaN input the VALUE is the one the ladder picks; only the sign of
// a zero result may differ (v_min_f64 orders -0 < +0, `b < a` does not), and no operation downstream can tell +0 from -0
// (no division by a possibly-zero result, no copysign): every comparison with the reference is `==` on values.
// The clip forms assume lo <= hi (min_production <= max_production etc.: what the reference's constructors enforce).
// (Inline asm rather than __builtin_fmin / fmax: for those hipcc canonicalises every operand that is not provably the
// result of an arithmetic instruction -- a `v_max_f64 x, x, x` per loaded parameter and per select result, in every loop
// iteration, which gave back most of the saving.  The asm is not volatile: the compiler may still hoist and CSE it.)
__device__ __forceinline__ double py_min(double a, double b)
{
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double py_max(double a, double b)
{
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double py_clip(double x, double lo, double hi) { return py_min(py_max(x, lo), hi); }

template <int F>
__device__ __forceinline__ void derive(const Params &p, Derived &d)
{
    if constexpr (F & F_BATTERY) {
        d.bat_lo = -p.bat_D / p.bat_eta;
        d.bat_sp = space_spread(d.bat_lo, p.bat_C * p.bat_eta);
    }
    if constexpr (F & F_GENSET) d.gen_sp = space_spread(0.0, p.gen_rmax);
    if constexpr (F & F_GRID) {
        d.grid_lo = -1 * p.grid_exp;
        d.grid_sp = space_spread(d.grid_lo, p.grid_imp);
    }
}

// ---- factorised series (mgx_columns.base_load != NULL) ------------------------------------------------------
// MicrogridGenerator's series are base profile x per-grid ratio (MicrogridGenerator.py:137-147), its import tariff a
// function of the hour of day (:253-285), its co2 series a base profile verbatim (:205-212), its grid status a bit per
// row (:321-340).  The kernels form the values with the single multiply synthesize_series_kernel performs, so a
// factorised batch steps bit-identically to its materialised twin.
constexpr int PP = MGX_PROFILE_PITCH;        // doubles per base-table row (one 64-byte line)

__host__ __device__ __forceinline__ bool factorised(const mgx_columns &c) { return c.base_load != nullptr; }

struct GridFactors {
    double lr, pr;              // load / pv ratio
    uint32_t lp, pp, cp, pat;   // load / pv / co2 profile column, tariff pattern
};

template <int F>
__device__ __forceinline__ void load_factors(const mgx_columns &c, int64_t i, GridFactors &f)
{
    f.lr = c.load_ratio[i]; f.pr = c.pv_ratio[i];
    f.lp = c.load_profile[i]; f.pp = c.pv_profile[i];
    f.cp = 0u; f.pat = 0u;
    if constexpr (F & F_GRID) { f.cp = c.co2_profile[i]; f.pat = c.tariff[i]; }
}

// element (row, profile column p) of a base table: public row-major [T, PP] layout (pm == 0) or the profile-major copy
__device__ __forceinline__ int64_t base_index(int32_t pm, int64_t row, uint32_t p)
{
    return pm ? (int64_t)p * pm + row : row * PP + p;
}

// stored sign: load <= 0, pv >= 0 (base_timeseries_module.py:68-79); one multiply, as _scale_ts (:137-147)
__device__ __forceinline__ double fact_load(double base, double ratio) { return -1.0 * fabs(base * ratio); }
__device__ __forceinline__ double fact_pv(double base, double ratio) { return fabs(base * ratio); }

// MicrogridGenerator._get_electricity_tariff (:253-285): import price by hour of day
__device__ __forceinline__ double tariff_price(int32_t pattern, int32_t row)
{
    const int32_t h = row % 24;
    if (pattern == 1) return (h >= 12 && h < 18) ? 0.59 : ((h < 8 || h >= 21) ? 0.22 : 0.29);
    if (pattern == 2) return ((h >= 0 && h < 5) || (h >= 14 && h < 17)) ? 0.08 : 0.11;
    return 0.0;
}

// grid_status of grid i at row `row` out of the outage words
__device__ __forceinline__ double fact_status(const mgx_columns &c, int64_t N, int64_t i, int64_t row)
{
    if (c.outage_bits == nullptr) return 1.0;
    return ((c.outage_bits[(row >> 6) * N + i] >> (row & 63)) & 1ull) ? 0.0 : 1.0;
}

// the series part of a step's inputs, formed from the factors (global-memory form: base rows out of the caches)
template <int F>
__device__ __forceinline__ void fact_series(const mgx_columns &c, int64_t N, int64_t i, int64_t row, const GridFactors &f, Inputs &in,
                                            int32_t pm = 0)
{
    in.load = fact_load(c.base_load[base_index(pm, row, f.lp)], f.lr);
    in.pv = fact_pv(c.base_pv[base_index(pm, row, f.pp)], f.pr);
    in.g_stat = 1.0;
    if constexpr (F & F_GRID) {
        in.g_pimp = tariff_price((int32_t)f.pat, (int32_t)row); in.g_pexp = 0.0;
        in.g_co2 = c.base_co2[base_index(pm, row, f.cp)];
        in.g_stat = fact_status(c, N, i, row);
    }
}

// Element (row, grid i) of a [T, N] series -- or, during in-place episodes on such series (pm = KArgs.pm_pitch != 0), of the
// handle's GRID-major copy [N, pm, C]: every lane reads its own row there, the C = 2 (load, pv) or 6 (+ the four grid components)
// values of a row are adjacent (one 48-byte read per grid and step) and consecutive rows of a grid share lines.  c.load_ts /
// pv_ts / grid_ts then point at elements 0 / 1 / 2 of that copy.
__device__ __forceinline__ int64_t ts_index(int32_t pm, int64_t N, int64_t row, int64_t i, int C)
{
    return pm ? (i * pm + row) * C : row * N + i;
}
// ... of component cc of the [T, 4, N] grid series
__device__ __forceinline__ int64_t grid_ts_index(int32_t pm, int64_t N, int64_t row, int cc, int64_t i)
{
    return pm ? (i * pm + row) * 6 + cc : (row * 4 + cc) * N + i;
}

// Component `comp` (0 load, 1 pv, 2..5 grid: import price, export price, co2 per kWh, status) of grid i at series row `row`,
// whichever way the batch holds its series.  For the kernels off the hot path (window patches, episode gathers).
__device__ __forceinline__ double series_component(const mgx_columns &c, int64_t N, int comp, int64_t row, int64_t i, int32_t pm = 0)
{
    if (factorised(c)) {
        switch (comp) {
            case 0: return fact_load(c.base_load[base_index(pm, row, c.load_profile[i])], c.load_ratio[i]);
            case 1: return fact_pv(c.base_pv[base_index(pm, row, c.pv_profile[i])], c.pv_ratio[i]);
            case 2: return tariff_price((int32_t)c.tariff[i], (int32_t)row);
            case 3: return 0.0;
            case 4: return c.base_co2[base_index(pm, row, c.co2_profile[i])];
            default: return fact_status(c, N, i, row);
        }
    }
    const int C = c.grid_ts ? 6 : 2;
    if (comp == 0) return c.load_ts[ts_index(pm, N, row, i, C)];
    if (comp == 1) return c.pv_ts[ts_index(pm, N, row, i, C)];
    return c.grid_ts[grid_ts_index(pm, N, row, comp - 2, i)];
}

// ---- loads ----------------------------------------------------------------------------------------------
// parameters of the controllable modules at column index i (= instance * N + grid for layouts with several instances)
// A batch-uniform column (mgx_columns.uniform_mask) holds one value: every lane reads element 0 (a same-address load, one
// cache line for the whole wave) instead of its own element.
__device__ __forceinline__ int64_t col_index(const mgx_columns &c, int bit, int64_t i)
{
    return ((c.uniform_mask >> bit) & 1u) ? (int64_t)0 : i;
}

template <int F>
__device__ __forceinline__ void load_module_params(const mgx_columns &c, int64_t i, Params &p)
{
    if constexpr (F & F_BATTERY) {
        p.bat_cmin = c.bat_min_capacity[col_index(c, MGX_U_BAT_MIN_CAPACITY, i)];
        p.bat_cmax = c.bat_max_capacity[col_index(c, MGX_U_BAT_MAX_CAPACITY, i)];
        p.bat_C = c.bat_max_charge[col_index(c, MGX_U_BAT_MAX_CHARGE, i)];
        p.bat_D = c.bat_max_discharge[col_index(c, MGX_U_BAT_MAX_DISCHARGE, i)];
        p.bat_eta = c.bat_efficiency[col_index(c, MGX_U_BAT_EFFICIENCY, i)];
        p.bat_cost = c.bat_cost_cycle[col_index(c, MGX_U_BAT_COST_CYCLE, i)];
    }
    if constexpr (F & F_GENSET) {
        p.gen_rmin = c.gen_running_min[col_index(c, MGX_U_GEN_RUNNING_MIN, i)];
        p.gen_rmax = c.gen_running_max[col_index(c, MGX_U_GEN_RUNNING_MAX, i)];
        p.gen_cost = c.gen_cost[col_index(c, MGX_U_GEN_COST, i)];
        p.gen_co2 = c.gen_co2_per_unit[col_index(c, MGX_U_GEN_CO2_PER_UNIT, i)];
        p.gen_cco2 = c.gen_cost_per_unit_co2[col_index(c, MGX_U_GEN_COST_PER_UNIT_CO2, i)];
        p.gen_times = c.gen_times[col_index(c, MGX_U_GEN_TIMES, i)];
    }
    if constexpr (F & F_GRID) {
        p.grid_imp = c.grid_max_import[col_index(c, MGX_U_GRID_MAX_IMPORT, i)];
        p.grid_exp = c.grid_max_export[col_index(c, MGX_U_GRID_MAX_EXPORT, i)];
        p.grid_cco2 = c.grid_cost_per_unit_co2[col_index(c, MGX_U_GRID_COST_PER_UNIT_CO2, i)];
    }
}

template <int F>
__device__ __forceinline__ void load_params(const mgx_columns &c, int64_t i, Params &p)
{
    load_module_params<F>(c, i, p);
    p.ll_cost = c.loss_load_cost[col_index(c, MGX_U_LOSS_LOAD_COST, i)];
    p.og_cost = c.overgeneration_cost[col_index(c, MGX_U_OVERGENERATION_COST, i)];
}

template <int F>
__device__ __forceinline__ void load_state(const mgx_columns &c, int64_t i, bool want_soc, State &s)
{
    s.charge = 0.0; s.soc = 0.0; s.status = 0u;
    if constexpr (F & F_BATTERY) {
        s.charge = c.charge[i];
        if (want_soc) s.soc = c.soc[i];      // only the log reads the pre-step SoC
    }
    if constexpr (F & F_GENSET) s.status = c.gen_status[i];
}

template <int F>
__device__ __forceinline__ void store_state(const mgx_columns &c, int64_t i, const State &s)
{
    if constexpr (F & F_BATTERY) { c.charge[i] = s.charge; c.soc[i] = s.soc; }
    if constexpr (F & F_GENSET) c.gen_status[i] = s.status;
}

// actions row [A] of grid i at `act` (row-major [N, A]); series rows at time t
template <int F, typename AT>
__device__ __forceinline__ void load_inputs(const mgx_columns &c, const AT *__restrict__ act,
                                            int64_t N, int64_t i, int64_t t, Inputs &in, int32_t pm = 0)
{
    constexpr int A = 2 * ((F & F_GENSET) != 0) + ((F & F_BATTERY) != 0) + ((F & F_GRID) != 0);
    const AT *a = act + i * A;
    int k = 0;
    if constexpr (F & F_GENSET) { in.a_goal = a[k]; in.a_gen = a[k + 1]; k += 2; }
    if constexpr (F & F_BATTERY) { in.a_bat = a[k]; k += 1; }
    if constexpr (F & F_GRID) { in.a_grid = a[k]; k += 1; }
    if (factorised(c)) {                       // uniform over the launch
        GridFactors f;
        load_factors<F>(c, i, f);
        fact_series<F>(c, N, i, t, f, in, pm);
        return;
    }
    if (pm) {              // in-place episodes on [T, N] series: the grid-major copies (pm is a compile-time 0 in the lock-step kernels)
        constexpr int C = (F & F_GRID) ? 6 : 2;
        in.load = c.load_ts[ts_index(pm, N, t, i, C)];
        in.pv = c.pv_ts[ts_index(pm, N, t, i, C)];
        in.g_stat = 1.0;
        if constexpr (F & F_GRID) {
            const double *g = c.grid_ts + grid_ts_index(pm, N, t, 0, i);
            in.g_pimp = g[0]; in.g_pexp = g[1]; in.g_co2 = g[2]; in.g_stat = g[3];
        }
        return;
    }
    in.load = c.load_ts[t * N + i];
    in.pv = c.pv_ts[t * N + i];
    in.g_stat = 1.0;       // (the same fields are written on both paths: an asymmetric store was sunk behind a selected ADDRESS by
                           //  the optimiser once, which kept `in` in scratch memory -- 24 B of private segment, +0.4 us per launch)
    if constexpr (F & F_GRID) {
        const double *g = c.grid_ts + (t * 4) * N + i;
        in.g_pimp = g[0]; in.g_pexp = g[N]; in.g_co2 = g[2 * N]; in.g_stat = g[3 * N];
    }
}

// ---- GensetModule.update_status (genset_module.py:235-346) on the packed status word ---------------------
// status = current | goal<<8 | steps_until_up<<16 | steps_until_down<<24 ; times = start_up | no_abortion<<8 | wind_down<<16
__device__ __forceinline__ uint32_t genset_update_status(uint32_t st, uint32_t times, double goal_f)
{
    int cur = st & 0xff, gs = (st >> 8) & 0xff, up = (st >> 16) & 0xff, down = st >> 24;
    const int su = times & 0xff, wd = (times >> 16) & 0xff;
    const bool allow = ((times >> 8) & 1u) == 0;      // allow_abortion (genset_module.py:78-88)
    const int goal = goal_f > 0.5 ? 1 : 0;            // Python round(): half-to-even, 0.5 -> 0 (:281)
    if (!(goal == cur && cur == gs)) {                // :284-287
        const bool instant = (su == 0 && goal == 1) || (wd == 0 && goal == 0);
        if (goal != gs && (allow || instant)) gs = goal;   // :289-292
        bool finished = false;                        // _finish_in_progress_change :302-311
        if (up == 0 && gs == 1)        { cur = 1; up = 0;   down = wd; finished = true; }
        else if (down == 0 && gs == 0) { cur = 0; down = 0; up = su;   finished = true; }
        if (!finished) {                              // _non_instantaneous_update :327-346
            if (goal == cur && cur != gs && allow) {  // calling off an in-progress change
                gs = goal;
                if (cur) { up = 0; down = wd; } else { down = 0; up = su; }
            } else if (cur == gs && gs != goal) {
                if (cur) { up = 0; down = wd; } else { down = 0; up = su; }
                gs = goal;
            }
            if (gs != cur) { if (gs == 0) down -= 1; else up -= 1; }
        }
    }
    return (uint32_t)cur | ((uint32_t)gs << 8) | ((uint32_t)(up & 0xff) << 16) | ((uint32_t)(down & 0xff) << 24);
}

// GensetModule.next_status (genset_module.py:360-390)
__device__ __forceinline__ int genset_next_status(uint32_t st, int goal)
{
    const int cur = st & 0xff, up = (st >> 16) & 0xff, down = st >> 24;
    if (goal) return (cur || up == 0) ? 1 : 0;
    return (!cur || down == 0) ? 0 : 1;
}

// Wave-uniform test for the genset fast path: no start-up / wind-down delay anywhere in the wave and every status in
// equilibrium with zero counters.  With zero delays update_status keeps the status inside {off = 0, on = 0x0101}.
template <int F>
__device__ __forceinline__ bool genset_wave_is_instant(const Params &p, const State &s)
{
    if constexpr (F & F_GENSET)
        return __all((p.gen_times == 0u) && (s.status == 0u || s.status == 0x0101u)) != 0;
    return false;
}

// BatteryModule.max_production / max_consumption (battery_module.py:283-291); Python min(a,b) = b if b<a else a
__device__ __forceinline__ double battery_max_production(const Params &p, double charge)
{
    const double b = charge - p.bat_cmin;
    return py_min(p.bat_D, b) * p.bat_eta;
}
__device__ __forceinline__ double battery_max_consumption(const Params &p, double charge)
{
    const double b = p.bat_cmax - charge;
    return py_min(p.bat_C, b) / p.bat_eta;
}

// ---- one Microgrid.run for one grid ---------------------------------------------------------------------
// Sweep order load -> genset -> battery -> grid -> pv -> unbalanced (module_container.py:355-413,
// microgrid.py:255-314).  np.sum over the provided/absorbed lists is a left-to-right running sum for the
// list lengths that occur here (< 8 addends), so the running sums below reproduce MicrogridStep.balance.
//
// The reference's if/else ladders are written as selects (both arms are cheap, lanes of a wave disagree on every
// one of them with random controls): same operations on the taken arm, hence the same values; its min / max / clip
// ladders are v_min_f64 / v_max_f64 (py_min / py_max / py_clip above: same value, the sign of a zero may differ).
// Adding +0.0 where the reference appends nothing to a list leaves the running sum's value unchanged.
//
// want_soc (wave-uniform): compute soc = charge / max_capacity this step (a division); the fused kernel skips it on
// steps whose SoC nobody reads and derives it once at the end (same value: it depends on the final charge only).
// gen_instant (wave-uniform): every genset of the wave has start_up_time == wind_down_time == 0 and an equilibrium
// status, so update_status collapses to "status follows the goal" (instant_up / instant_down, :289-311).
// HAVE_Q: the battery's one quotient by eta was already formed by populate_core on the same pre-step state (bat_q).
template <int F, bool HAVE_Q = false>
__device__ __forceinline__ void step_core(const Params &p, const Derived &d, State &s, const Inputs &in, bool normalized,
                                          bool want_soc, bool gen_instant, Outputs &o, double bat_q = 0.0)
{
    double prov = 0.0, absb = 0.0, reward = 0.0;
    uint32_t viol = 0u;

    // fixed: LoadModule.update (load_module.py:86-111)
    const double L = -1 * in.load;
    o.load_met = L;
    absb += L; reward += 0.0;
    o.fixed_provided = prov; o.fixed_absorbed = absb;

    if constexpr (F & F_GENSET) {
        if (gen_instant) {
            const uint32_t g = in.a_goal > 0.5 ? 1u : 0u;                       // round(): half-to-even (:281)
            s.status = g | (g << 8);
        } else {
            s.status = genset_update_status(s.status, p.gen_times, in.a_goal);  // GensetModule.step :146-149
        }
        const double x = normalized ? 0.0 + d.gen_sp * in.a_gen : in.a_gen;     // act space :511-517, _energy_pos=1
        const double cur = (double)(s.status & 0xff);
        const double mx = cur * p.gen_rmax, mn = cur * p.gen_rmin;              // max/min_production :465-501
        const double e = py_clip(x, mn, mx);                                    // as_source clip base_module.py:213-224
        viol |= ((x > mx) || (x < mn) ? 1u : 0u) | (!(in.a_goal >= 0.0 && in.a_goal <= 1.0) ? 8u : 0u) | (x < 0.0 ? 16u : 0u);
        const double co2 = p.gen_co2 * e;                                       // get_co2
        const double cost = p.gen_cost * e + p.gen_cco2 * co2;                  // get_cost :188-205
        o.genset_production = e; o.genset_co2 = co2; o.genset_reward = -1.0 * cost;
        prov += e; reward += o.genset_reward;
    }

    auto step_battery = [&]() __attribute__((always_inline)) {
        const double x = normalized ? d.bat_lo + d.bat_sp * in.a_bat : in.a_bat;   // space.py:224, bounds :332-338
        o.soc_pre = s.soc; o.charge_pre = s.charge;
        // Each lane needs ONE division by eta: charging -> max_consumption = min(C, cmax - c) / eta (:288-291),
        // discharging -> internal = (-e) / eta (default_transition_model :244-278).  Select the numerator, divide once.
        const bool sink = x < 0;                                                // as_sink(-1.0*x) vs as_source(x)
        const double room = p.bat_cmax - s.charge;
        const double num_sink = py_min(p.bat_C, room);                          // Python min(a, b) = b if b < a else a
        const double mp = battery_max_production(p, s.charge);                  // :283-286
        const double e_src = py_clip(x, 0.0, mp);                               // base_module.py:213-224, min_production 0
        const double num = sink ? num_sink : -1.0 * e_src;
        double q;
        if constexpr (HAVE_Q) q = bat_q; else q = num / p.bat_eta;
        const double ex = -1.0 * x;
        const double e_sink = py_min(ex, q);                                    // base_module.py:265-270
        // (e_sink < 0, i.e. charge above max_capacity, is an AssertionError in the reference, base_module.py:272,
        //  and unspecified here; every valid run has e >= 0 and internal = e * eta.)
        const double e = sink ? e_sink : e_src;
        viol |= (sink ? (ex > q) : (x > mp)) ? 2u : 0u;
        // `assert absorbed_energy >= 0` (base_module.py:272) / `assert internal_energy_change <= 0` (battery_module.py:114): the
        // limit the clip hands on is negative (charge above max_capacity / below min_capacity)
        viol |= !((sink ? e_sink : e_src) >= 0) ? 32u : 0u;
        const double internal = sink ? e * p.bat_eta : ((num < 0) ? q : num * p.bat_eta);
        o.charge_amount = sink ? e : 0.0;
        o.discharge_amount = sink ? 0.0 : e;
        absb += o.charge_amount; prov += o.discharge_amount;
        s.charge = py_max(s.charge + internal, p.bat_cmin);                     // _update_state :125-130
        if (want_soc) s.soc = s.charge / p.bat_cmax;
        o.battery_reward = -1.0 * (fabs(internal) * p.bat_cost);                // get_cost :132-147
        reward += o.battery_reward;
    };
    auto step_grid = [&]() __attribute__((always_inline)) {
        const double x = normalized ? d.grid_lo + d.grid_sp * in.a_grid : in.a_grid;   // _get_bounds :125-132
        const bool sink = x < 0;
        const double ex = -1.0 * x, mc = p.grid_exp * in.g_stat;               // max_consumption :318-320
        const double e_exp = py_min(ex, mc);
        const double mp = p.grid_imp * in.g_stat;                              // max_production :314-316
        const double e_imp = py_clip(x, 0.0, mp);
        viol |= (sink ? (ex > mc) : (x > mp)) ? 4u : 0u;
        viol |= !((sink ? e_exp : e_imp) >= 0) ? 32u : 0u;                     // base_module.py:272 (a negative limit)
        const double co2 = sink ? 0.0 : e_imp * in.g_co2;                      // get_co2_production :199-228
        const double cco2 = -1.0 * p.grid_cco2 * co2;                          // get_co2_cost :176-197
        o.grid_export = sink ? e_exp : 0.0;
        o.grid_import = sink ? 0.0 : e_imp;
        o.grid_co2 = co2;
        o.grid_reward = sink ? in.g_pexp * e_exp + cco2 : -1 * in.g_pimp * e_imp + cco2;   // get_cost :143-174
        absb += o.grid_export; prov += o.grid_import;
        reward += o.grid_reward;
    };
    if constexpr ((F & F_GRID_FIRST) != 0) {
        if constexpr (F & F_GRID) step_grid();
        if constexpr (F & F_BATTERY) step_battery();
    } else {
        if constexpr (F & F_BATTERY) step_battery();
        if constexpr (F & F_GRID) step_grid();
    }

    const double difference = prov - absb;                                     // microgrid.py:277-278
    o.ctrl_provided = prov - o.fixed_provided;                                 // :281
    o.ctrl_absorbed = absb - o.fixed_absorbed;

    // flex modules (:286-314): renewable first, unbalanced energy last
    const bool excess = difference > 0;
    const double need = -difference;
    const double used = excess ? 0.0 : py_min(need, in.pv);                    // renewable_module.py:86-93
    o.renewable_used = used; o.curtailment = in.pv - used;
    const double loss = excess ? 0.0 : need - used;
    const double over = excess ? difference : 0.0;
    o.loss_load = loss; o.overgeneration = over;
    o.unbalanced_reward = -1.0 * (excess ? p.og_cost * over : p.ll_cost * loss);   // unbalanced_energy_module.py:38-70
    prov += used; reward += 0.0;
    prov += loss;
    absb += over;
    reward += o.unbalanced_reward;
    o.overall_provided = prov; o.overall_absorbed = absb;
    o.reward = reward;
    o.violations = viol;
}

// ---- discrete action expansion -------------------------------------------------------------------------
// PriorityListAlgo._populate_action (algos/priority_list/priority_list.py:69-167).  A priority list is packed in
// one word: element k in bits [4k, 4k+3] = module (0 genset, 1 battery, 2 grid) | action << 2 | valid << 3.
struct PLWords {
    uint32_t w[12];
    int32_t n_actions;
};

__device__ __forceinline__ uint32_t pl_select(const PLWords &tab, int32_t id)
{
    // per-lane table lookup out of the kernarg (SGPR) copy: a select chain, no memory access
    uint32_t w = tab.w[0];                 // ids outside [0, n) fall back to list 0 (the reference raises ValueError)
#pragma unroll
    for (int j = 1; j < 12; j++) w = (id == j && j < tab.n_actions) ? tab.w[j] : w;
    return w;
}

// Branch-free form.  Lanes of a wave hold different lists, so the reference's per-element if/else ladder would
// diverge on every slot; instead every slot evaluates the (cheap) genset / grid candidates and selects.  The battery
// is the only module whose answer needs a division (max_consumption = min(C, cmax - c) / eta), and exactly ONE
// division per step is ever needed per lane: when the battery is reached with remaining < 0 it is that one, when it
// is reached with remaining > 0 the battery will act as a source and the step itself needs (-e) / eta
// (default_transition_model).  Pass A finds `remaining` at the battery's slot, the division happens once, pass B
// replays the list with the battery's energy known.  bat_q returns that quotient for step_core<F, true>.
__device__ __forceinline__ bool pl_isclose0(double rem)
{
    return fabs(rem - 0.0) <= 1e-4 + 1e-5 * fabs(0.0);            // np.isclose(remaining_load, 0.0, atol=1e-4) :90
}

// _produce_from_module :138-155 / _consume_in_module :118-136 for a module without a division
__device__ __forceinline__ double pl_energy(double rem, double mn, double mx, double mc, bool is_sink)
{
    const double produce = py_clip(rem, mn, mx);
    const double consume = is_sink ? py_max(rem, -1.0 * mc) : 0.0;
    return pl_isclose0(rem) ? 0.0 : ((rem > 0) ? produce : consume);
}

// gen_instant (wave-uniform, see step_core): next_status(goal) == goal, so the genset's limits under a list element are
// act * running_{min,max} -- loop-invariant for a fixed list, which lets the compiler hoist them out of a K-step loop.
// CHECK: also report, through *xviol, the state in which the reference's _populate_action gives up with an AssertionError
// instead of returning a control (enum mgx_violation_bit): bit 6 `assert module_max_consumption >= 0` (:124: a sink whose
// limit is negative is reached with load left to absorb -- a lossy battery whose charge sits one ulp above max_capacity),
// bit 7 `assert module_production >= 0` (:154: a module whose max_production is negative is asked to produce -- a battery
// below min_capacity), bit 8 `assert total_load >= 0 and renewable >= 0` / `assert remaining_load <= 0.0` (:73, :121: series
// of the wrong sign, NaN).  The reference stops at the FIRST failing assert of its walk down the list: so does the mask (one
// bit).  The control written in such a state is unspecified (the walk goes on with the value the clip ladder gives).
template <int F, bool CHECK = false>
__device__ __forceinline__ void populate_core(const Params &p, const State &s, uint32_t word, Inputs &in, double &bat_q,
                                              double total_load, double renewable, bool gen_instant = false,
                                              uint32_t *xviol = nullptr, bool check_sign = true)
{
    // total_load: sum of the fixed sinks' max_consumption (_get_load :157-164); renewable: np.sum of the flex
    // sources' max_production (_get_renewable :166-167)
    const double rem0 = total_load - renewable;                    // :74
    // per-module limits at the pre-step state
    double g_mx[2] = {0.0, 0.0}, g_mn[2] = {0.0, 0.0};             // next_max/min_production(goal) genset_module.py:392-424
    if constexpr (F & F_GENSET) {
#pragma unroll
        for (int act = 0; act < 2; act++) {
            const double ns = gen_instant ? (double)act : (double)genset_next_status(s.status, act);
            g_mx[act] = ns * p.gen_rmax; g_mn[act] = ns * p.gen_rmin;
        }
    }
    double r_mx = 0.0, r_mc = 0.0;
    if constexpr (F & F_GRID) { r_mx = p.grid_imp * in.g_stat; r_mc = p.grid_exp * in.g_stat; }

    // a list holds every controllable module of the layout at most once: NSLOT slots are enough
    constexpr int NSLOT = ((F & F_GENSET) != 0) + ((F & F_BATTERY) != 0) + ((F & F_GRID) != 0);
    constexpr int NS = NSLOT > 0 ? NSLOT : 1;
    uint32_t el[NS];
#pragma unroll
    for (int k = 0; k < NS; k++) el[k] = (k < NSLOT) ? (word >> (4 * k)) & 0xfu : 0u;
    // candidate energy of the division-free modules at `rem` (genset and / or grid, whichever the layout has)
    auto cheap = [&](int mod, int act, double rem) -> double {
        double e = 0.0;
        if constexpr ((F & F_GENSET) && (F & F_GRID))
            e = (mod == 0) ? pl_energy(rem, g_mn[act], g_mx[act], 0.0, false) : pl_energy(rem, 0.0, r_mx, r_mc, true);
        else if constexpr (F & F_GENSET)
            e = pl_energy(rem, g_mn[act], g_mx[act], 0.0, false);
        else if constexpr (F & F_GRID)
            e = pl_energy(rem, 0.0, r_mx, r_mc, true);
        return e;
    };

    // pass A: remaining load when the battery's element is reached
    double remB = rem0;
    if constexpr (F & F_BATTERY) {
        double rem = rem0;
        bool seen = false;
#pragma unroll
        for (int k = 0; k < NSLOT - 1; k++) {              // the last slot cannot precede the battery
            const bool valid = (el[k] & 8u) != 0;
            const int mod = el[k] & 3u, act = (el[k] >> 2) & 1u;
            const bool isB = valid && mod == 1;
            seen = seen || isB;
            rem -= (valid && !seen) ? cheap(mod, act, rem) : 0.0;
            remB = seen ? remB : rem;                      // still before the battery: remaining after this slot
        }
    }
    // the battery's energy, one division
    double eB = 0.0;
    bat_q = 0.0;
    if constexpr (F & F_BATTERY) {
        const bool close = pl_isclose0(remB);
        const bool produce = !close && remB > 0;
        const double mp = battery_max_production(p, s.charge);                  // battery_module.py:283-286
        const double room = p.bat_cmax - s.charge;
        const double num_sink = py_min(p.bat_C, room);                          // numerator of max_consumption :288-291
        const double e_src = py_clip(remB, 0.0, mp);
        const double num = (close || produce) ? -1.0 * (produce ? e_src : 0.0) : num_sink;
        const double q = num / p.bat_eta;
        const double e_snk = py_max(remB, -1.0 * q);
        eB = close ? 0.0 : (produce ? e_src : e_snk);
        bat_q = q;
    }
    // pass B: replay the list with the battery's energy known
    double rem = rem0, c_goal = 0.0, c_gen = 0.0, c_grid = 0.0;
    uint32_t xv = 0u;
    if constexpr (CHECK) xv = (check_sign && !(total_load >= 0 && renewable >= 0)) ? 256u : 0u;  // :73 (populate_multi: its own test)
#pragma unroll
    for (int k = 0; k < NSLOT; k++) {
        const bool valid = (el[k] & 8u) != 0;
        const int mod = el[k] & 3u, act = (el[k] >> 2) & 1u;
        const double e = (mod == 1) ? eB : cheap(mod, act, rem);
        if constexpr (CHECK) {
            // (at the battery's slot rem == remB, the same running difference: bat_q is then its max_consumption whenever it consumes)
            const bool close = pl_isclose0(rem), produce = !close && rem > 0, consume = !close && !produce;
            const double mc = (mod == 1) ? bat_q : r_mc;
            uint32_t bad = (produce && !(e >= 0)) ? 128u : 0u;                   // :154
            bad = (consume && !(rem <= 0.0)) ? 256u : bad;                       // :121 (NaN)
            bad = (consume && rem <= 0.0 && mod != 0 && !(mc >= 0)) ? 64u : bad; // :124 (sinks only: the genset is skipped)
            xv = (xv == 0u && valid) ? bad : xv;
        }
        const bool isG = valid && mod == 0, isR = valid && mod == 2;
        c_goal = isG ? (double)act : c_goal;                                    // :82-88
        c_gen = isG ? e : c_gen;
        c_grid = isR ? e : c_grid;
        rem -= valid ? e : 0.0;                                                 // :105
    }
    in.a_goal = c_goal; in.a_gen = c_gen; in.a_bat = eB; in.a_grid = c_grid;
    if constexpr (CHECK) { if (xviol) *xviol = xv; }
}

// ---- reward shaping (microgrid/reward_shaping/*.py; MicrogridStep.shaped_reward, utils/step.py:41-46) ----------
// The value RETURNED by step(); the log's "reward" column always holds the unshaped sum.
template <int F>
__device__ __forceinline__ double shaped_reward(int32_t shaper, const Outputs &o)
{
    if (shaper == MGX_SHAPER_PV_CURTAILMENT)            // pv_curtailment_shaper.py:15-17
        return -1.0 * o.curtailment;
    if (shaper == MGX_SHAPER_BATTERY_DISCHARGE) {       // battery_discharge_shaper.py:23-34
        double discharge = 0.0;
        if constexpr (F & F_BATTERY) discharge = o.discharge_amount;
        const double load = o.load_met;
        return load == 0.0 ? 0.0 : (discharge - o.loss_load) / load;   // ZeroDivisionError -> 0.0
    }
    return o.reward;
}

// ---- log row ------------------------------------------------------------------------------------------
// log points at column 0 of grid i; consecutive columns are N apart.
// the log's column stores: write-once streams, non-temporal (full-output fused launch 334 -> 321 us per 64 steps, single step with
// rows + log 11.8 -> 10.8 us: profiles/r06/exp_log_nt_stores.txt; -DMGX_LOG_NT=0 for plain stores)
#ifndef MGX_LOG_NT
#define MGX_LOG_NT 1
#endif
#if MGX_LOG_NT
#define MGX_LOG_ST(ptr, v) __builtin_nontemporal_store((double)(v), (ptr))
#else
#define MGX_LOG_ST(ptr, v) (*(ptr) = (v))
#endif
template <int F>
__device__ __forceinline__ void store_log(double *__restrict__ log, int64_t N, const Outputs &o, uint32_t status)
{
    int k = 0;
    MGX_LOG_ST(log + (k++) * N, o.reward);
    MGX_LOG_ST(log + (k++) * N, o.fixed_provided);   MGX_LOG_ST(log + (k++) * N, o.fixed_absorbed);
    MGX_LOG_ST(log + (k++) * N, o.ctrl_provided);    MGX_LOG_ST(log + (k++) * N, o.ctrl_absorbed);
    MGX_LOG_ST(log + (k++) * N, o.overall_provided); MGX_LOG_ST(log + (k++) * N, o.overall_absorbed);
    MGX_LOG_ST(log + (k++) * N, o.load_met);         MGX_LOG_ST(log + (k++) * N, o.renewable_used);
    MGX_LOG_ST(log + (k++) * N, o.curtailment);      MGX_LOG_ST(log + (k++) * N, o.loss_load);
    MGX_LOG_ST(log + (k++) * N, o.overgeneration);   MGX_LOG_ST(log + (k++) * N, o.unbalanced_reward);
    if constexpr (F & F_GENSET) {
        MGX_LOG_ST(log + (k++) * N, o.genset_production); MGX_LOG_ST(log + (k++) * N, o.genset_co2);
        MGX_LOG_ST(log + (k++) * N, o.genset_reward);     MGX_LOG_ST(log + (k++) * N, (double)status);   // packed word, exact in fp64
    }
    if constexpr (F & F_BATTERY) {
        MGX_LOG_ST(log + (k++) * N, o.discharge_amount);  MGX_LOG_ST(log + (k++) * N, o.charge_amount);
        MGX_LOG_ST(log + (k++) * N, o.battery_reward);    MGX_LOG_ST(log + (k++) * N, o.soc_pre);
        MGX_LOG_ST(log + (k++) * N, o.charge_pre);
    }
    if constexpr (F & F_GRID) {
        MGX_LOG_ST(log + (k++) * N, o.grid_import);       MGX_LOG_ST(log + (k++) * N, o.grid_export);
        MGX_LOG_ST(log + (k++) * N, o.grid_co2);          MGX_LOG_ST(log + (k++) * N, o.grid_reward);
    }
    MGX_LOG_ST(log + (k++) * N, (double)o.violations);
}

// ---- observation (post-step state, series index t = current step) --------------------------------------
// obs row of a grid: [load window (1+H) | pv window (1+H) | genset 4 | battery 2 | grid window 4*(1+H)].
// The windows depend on the series only, the 6 state columns on the post-step state only, so they are produced
// separately: state columns by the lane that owns the grid, windows by wave-sized (64 grids x <=32 columns) work
// items (observe_window_item) that transpose through LDS -- one lane per grid would leave the chip at ~1.5 waves
// per SIMD with D fp64 divisions each in a row.

// normalised value of one series element: rows beyond the series = (lo+hi)/2 (forecaster.py:95,120-137); forecasts
// clipped to the bounds (:139-149); (v - lo) / spread (space.py:213)
__device__ __forceinline__ double obs_series_value(double v, bool in_series, bool is_forecast, double lo, double hi,
                                                   double fill, double sp)
{
    double x = in_series ? v : fill;
    if (in_series && is_forecast) { if (x < lo) x = lo; if (x > hi) x = hi; }
    return (x - lo) / sp;
}

// the 6 state columns (genset_module.py:503-509, battery_module.py:87,323-330), written by the owning lane
template <int F, typename OT>
__device__ __forceinline__ void observe_state_cols(const KArgs &a, const Params &p, const State &s,
                                                   OT *__restrict__ obs_row, int first = -1)
{
    // first < 0: inside a flat row, every block at its column base; else genset (4) then battery (2) from column `first`
    int k = first < 0 ? a.col_gen : first;
    if constexpr (F & F_GENSET) {
        const double su = (double)(p.gen_times & 0xff), wd = (double)((p.gen_times >> 16) & 0xff);
        obs_row[k++] = (OT)space_norm(0.0, 1.0, (double)(s.status & 0xff));
        obs_row[k++] = (OT)space_norm(0.0, 1.0, (double)((s.status >> 8) & 0xff));
        obs_row[k++] = (OT)space_norm(0.0, su, (double)((s.status >> 16) & 0xff));
        obs_row[k++] = (OT)space_norm(0.0, wd, (double)(s.status >> 24));
    }
    if constexpr (F & F_BATTERY) {
        if (first < 0) k = a.col_bat;
        const double min_soc = p.bat_cmin / p.bat_cmax;
        obs_row[k++] = (OT)space_norm(min_soc, 1.0, s.soc);
        obs_row[k++] = (OT)space_norm(p.bat_cmin, p.bat_cmax, s.charge);
    }
}

// H == 0 (no forecaster): the whole row is 2 + 6 (+4) values -- the owning lane stores them directly
template <int F, typename OT>
__device__ __forceinline__ void observe_row_h0(const KArgs &a, int64_t i, int32_t t, const Params &p, const State &s,
                                               OT *__restrict__ obs_row, int32_t pm = 0)
{
    const mgx_columns &c = a.c;
    const int64_t N = a.N;
    const bool in = t < a.T;
    const int64_t tr = t & a.row_mask;                  // row of the series buffers (rolling windows: a ring)
    {
        const double lo = c.load_lo[i], hi = c.load_hi[i];
        const double v = in ? series_component(c, N, 0, tr, i, pm) : 0.0;
        obs_row[a.col_load] = (OT)obs_series_value(v, in, false, lo, hi, (hi + lo) / 2, space_spread(lo, hi));
    }
    {
        const double lo = c.pv_lo[i], hi = c.pv_hi[i];
        const double v = in ? series_component(c, N, 1, tr, i, pm) : 0.0;
        obs_row[a.col_pv] = (OT)obs_series_value(v, in, false, lo, hi, (hi + lo) / 2, space_spread(lo, hi));
    }
    observe_state_cols<F, OT>(a, p, s, obs_row);
    if constexpr (F & F_GRID) {
        const int k = a.col_grid;
#pragma unroll
        for (int cc = 0; cc < 4; cc++) {
            const double lo = c.grid_lo[cc * N + i], hi = c.grid_hi[cc * N + i];
            const double v = in ? series_component(c, N, 2 + cc, tr, i, pm) : 0.0;
            obs_row[k + cc] = (OT)obs_series_value(v, in, false, lo, hi, (hi + lo) / 2, space_spread(lo, hi));
        }
    }
}

// Which series component / horizon step -- or which state value -- observation column `col` shows (W = 1 + H):
// comp 0 load, 1 pv, 2..5 the grid components; 0xffff: a state column, h = its index in (genset 4, battery 2) order.
__device__ __forceinline__ void decode_obs_col(const KArgs &a, bool grid, int32_t col, int32_t W, uint32_t &comp, uint32_t &h)
{
    uint32_t c;
    if ((c = (uint32_t)(col - a.col_load)) < (uint32_t)W) { comp = 0u; h = c; }
    else if ((c = (uint32_t)(col - a.col_pv)) < (uint32_t)W) { comp = 1u; h = c; }
    else if (grid && (c = (uint32_t)(col - a.col_grid)) < (uint32_t)(4 * W)) { comp = 2u + (c & 3u); h = c >> 2; }
    else if (a.n_genset && (c = (uint32_t)(col - a.col_gen)) < 4u) { comp = 0xffffu; h = c; }
    else { comp = 0xffffu; h = 4u * (a.n_genset != 0) + (uint32_t)(col - a.col_bat); }
}

// ---- forecast noise: Philox4x32-10 counter-based generator + Box-Muller ------------------------------------
// GaussianNoiseForecaster._forecast (forecaster.py:262-263): forecast = true future values + N(0, std).  The
// reference draws from numpy's global stream; here every (grid, series component, step, horizon index) owns a
// counter, so the noise is reproducible and independent of batch size / sharding.  Statistical parity only.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t (&out)[4])
{
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// one standard normal for (grid, component id, absolute series row, seed)
__device__ __forceinline__ double forecast_normal(uint64_t seed, int64_t grid, uint32_t comp, int32_t t, int32_t h)
{
    uint32_t r[4];
    philox4x32_10((uint32_t)grid, (uint32_t)((uint64_t)grid >> 32) ^ (comp << 24), (uint32_t)t, (uint32_t)h,
                  (uint32_t)seed, (uint32_t)(seed >> 32), r);
    // 53-bit uniform in (0, 1] and a 32-bit angle
    const double u1 = ((double)(((uint64_t)r[0] << 21) ^ (r[1] >> 11)) + 1.0) * (1.0 / 9007199254740992.0);
    const double u2 = ((double)r[2] + 0.5) * (1.0 / 4294967296.0);
    return sqrt(-2.0 * log(u1)) * cospi(2.0 * u2);     // cospi: no large-argument reduction (no scratch)
}

// U[0, 1) of (seed; grid, row): Philox4x32-10 as a counter-based generator (the generator's series and the episode draws of
// mgx_reset_grids_random use it; pymgrid_amd.generator.synth_uniform_host reproduces it bit for bit)
__device__ __forceinline__ double synth_uniform(uint64_t seed, int64_t grid, int32_t row)
{
    uint32_t r[4];
    philox4x32_10((uint32_t)grid, (uint32_t)((uint64_t)grid >> 32), (uint32_t)row, 0x5eedu, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    return (double)(((uint64_t)r[0] << 21) ^ (r[1] >> 11)) * (1.0 / 9007199254740992.0);      // 53 bits, [0, 1)
}

// One grid's trajectory draw at counter value `counter`, as its own trajectory_func would make it
// (microgrid/trajectory/stochastic.py:9-30): np.random.randint(low, high) = low + min(floor(u * (high - low)), high - low - 1)
__device__ __forceinline__ void episode_draw(uint64_t seed, int64_t i, int32_t counter, int32_t fixed_length, int32_t lo, int32_t hi,
                                             int32_t &s, int32_t &len)
{
    const double u1 = synth_uniform(seed, i, 2 * counter), u2 = synth_uniform(seed, i, 2 * counter + 1);
    auto randint = [](double u, int32_t low, int32_t high) {
        const int32_t span = high - low;
        if (span <= 0) return low;
        const int32_t k = (int32_t)floor(u * (double)span);
        return low + (k < span - 1 ? k : span - 1);
    };
    if (fixed_length > 0) {                                            // FixedLengthStochasticTrajectory (:15-30)
        s = randint(u1, lo, hi - fixed_length);
        len = fixed_length;
    } else {                                                           // StochasticTrajectory (:9-12)
        s = randint(u1, lo, hi - 2);
        const int32_t fin = randint(u2, s, hi);
        len = fin - s;
    }
}

// a start outside the env's window [lo, hi) is clamped into it; the episode lasts 1 .. max_length steps and ends at the env's
// final step at the latest
__device__ __forceinline__ void episode_clamp(int32_t lo, int32_t hi, int32_t max_length, int32_t &s, int32_t &len)
{
    s = s < lo ? lo : (s > hi - 1 ? hi - 1 : s);
    const int32_t room = hi - s;
    len = len < 1 ? 1 : len;
    len = len > max_length ? max_length : len;
    len = len > room ? room : len;
}

// Window work of a wave: it owns G (= 16) grids and the whole window of one time-series module.  Lane = (g, q):
// g = grid within the group, q = horizon phase; lane (g, q) takes horizon steps q, q + Q, q + 2Q, ... (Q = 64 / G),
// JB of them per latency round (7 * 4 = 28 slots cover the usual 24 / 25-step windows in one round), all NC components
// of each.  Loads are 8*G-byte segments along the grids.
constexpr int OBS_JB = 7;

// General form (any horizon, any t): clamped rows, padding beyond the series.
// fetch(row, c): component c of this module at series row `row` for the lane's grid
template <int NC, bool NOISE, typename OT, class Fetch>
__device__ __forceinline__ void observe_window_cols(Fetch fetch, int64_t N,
                                                    const double *__restrict__ lo_col, const double *__restrict__ hi_col,
                                                    int32_t T, int32_t t, int32_t W, int64_t i, int64_t ic, int32_t q, int32_t Q,
                                                    OT *row /* tile + g*LD + first column of this module */,
                                                    const double *__restrict__ noise_std, uint32_t comp_base,
                                                    uint64_t noise_seed, int noise_increase, int32_t row_mask = -1)
{
    double lo[NC], hi[NC], fill[NC], sp[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        lo[c] = lo_col[c * N + ic]; hi[c] = hi_col[c * N + ic];
        fill[c] = (hi[c] + lo[c]) / 2; sp[c] = space_spread(lo[c], hi[c]);
    }
    double std0 = 0.0;
    if constexpr (NOISE) if (noise_std != nullptr) std0 = noise_std[ic];
    for (int32_t hb = 0; hb < W; hb += OBS_JB * Q) {                 // wave-uniform trip count
        double v[OBS_JB][NC];
#pragma unroll
        for (int jj = 0; jj < OBS_JB; jj++) {                         // unconditional, clamped loads (one latency round)
            const int32_t r = t + hb + q + Q * jj;
            const int32_t rc = (r < T ? r : T - 1) & row_mask;
#pragma unroll
            for (int c = 0; c < NC; c++) v[jj][c] = fetch(rc, c);
        }
#pragma unroll
        for (int jj = 0; jj < OBS_JB; jj++) {
            const int32_t h = hb + q + Q * jj;
            const bool in = t + h < T;
            if (h < W) {
                if constexpr (NOISE) if (noise_std != nullptr && h > 0 && in) {     // GaussianNoiseForecaster (:243-263)
                    const double sd = noise_increase ? std0 * (1.0 + log(1.0 + (double)(h - 1))) : std0;
#pragma unroll
                    for (int c = 0; c < NC; c++) v[jj][c] += sd * forecast_normal(noise_seed, i, comp_base + c, t, h);
                }
#pragma unroll
                for (int c = 0; c < NC; c++)
                    row[h * NC + c] = (OT)obs_series_value(v[jj][c], in, h > 0, lo[c], hi[c], fill[c], sp[c]);
            }
        }
    }
}

// ---- fast form: every slot of every round of OBS_JB * Q slots lies inside the series -------------------------
// Split in three so that the caller can put the loads of ALL modules in flight before the first value is consumed.
template <int NC>
struct WinBounds {
    double lo[NC], hi[NC], sp[NC];
};

template <int NC>
__device__ __forceinline__ void window_bounds(const double *__restrict__ lo_col, const double *__restrict__ hi_col, int64_t N,
                                              int64_t ic, WinBounds<NC> &b)
{
#pragma unroll
    for (int c = 0; c < NC; c++) { b.lo[c] = lo_col[c * N + ic]; b.hi[c] = hi_col[c * N + ic]; }
}

template <int NC>
__device__ __forceinline__ void window_bounds_finish(WinBounds<NC> &b)
{
#pragma unroll
    for (int c = 0; c < NC; c++) b.sp[c] = space_spread(b.lo[c], b.hi[c]);
}

// loads of slot jj: row t + Q jj (+ q through lane_off): the row base is wave-uniform (SGPR base + 32-bit lane offset)
template <int NC>
__device__ __forceinline__ void window_issue(const double *__restrict__ ts, int64_t N, int64_t row_stride, int32_t t, int32_t hb,
                                             int32_t Q, uint32_t lane_off, double (&v)[OBS_JB][NC])
{
#pragma unroll
    for (int jj = 0; jj < OBS_JB; jj++) {
        const double *rp = ts + (int64_t)(t + hb + Q * jj) * row_stride;
#pragma unroll
        for (int c = 0; c < NC; c++) v[jj][c] = (rp + c * N)[lane_off];
    }
}

template <int NC, bool NOISE, typename OT>
__device__ __forceinline__ void window_finish(double (&v)[OBS_JB][NC], const WinBounds<NC> &b, int32_t W, int32_t t, int32_t hb,
                                              int64_t i, int64_t ic, int32_t q, int32_t Q, OT *row,
                                              const double *__restrict__ noise_std, uint32_t comp_base, uint64_t noise_seed,
                                              int noise_increase)
{
    double std0 = 0.0;
    if constexpr (NOISE) if (noise_std != nullptr) std0 = noise_std[ic];
#pragma unroll
    for (int jj = 0; jj < OBS_JB; jj++) {
        const int32_t h = hb + q + Q * jj;
        if constexpr (NOISE) if (noise_std != nullptr && h > 0 && h < W) {
            const double sd = noise_increase ? std0 * (1.0 + log(1.0 + (double)(h - 1))) : std0;
#pragma unroll
            for (int c = 0; c < NC; c++) v[jj][c] += sd * forecast_normal(noise_seed, i, comp_base + c, t, h);
        }
#pragma unroll
        for (int c = 0; c < NC; c++) {
            double x = v[jj][c];
            if (h > 0) { if (x < b.lo[c]) x = b.lo[c]; if (x > b.hi[c]) x = b.hi[c]; }   // forecasts are clipped (:139-149)
            const double val = (x - b.lo[c]) / b.sp[c];                                  // space.py:213
            if (h < W) row[h * NC + c] = (OT)val;
        }
    }
}

// =========================================================================================================
// General path: several load / renewable modules per microgrid (the reference's TestMicrogridLoadPV family,
// tests/microgrid/test_microgrid.py:188-427).  Not the hot path: one lane per grid, the provided / absorbed lists of
// MicrogridStep are materialised in private memory and summed exactly as numpy's float64 add.reduce does
// (pairwise_sum: running sum below 8 addends, eight interleaved partial sums above).
// =========================================================================================================
constexpr int MGX_MAX_MODULES = 16;                       // load / renewable modules per grid

// ---- general path: any number of modules per kind ----------------------------------------------------------------------
// MicrogridStep keeps a list of provided and a list of absorbed energies and np.sum()s them three times per step
// (utils/step.py:24-36, microgrid.py:259,277,316).  Which list a battery / grid entry joins depends on the sign of its
// request, so the list lengths differ from grid to grid.  The lists live in LDS, one column per lane (element k of a lane's
// list at list[k * stride]: lanes of a wave hit distinct banks); numpy's summation order depends on the length:
// n < 8 a running sum, n >= 8 eight partial sums combined pairwise, then the tail (DOUBLE_pairwise_sum, n <= 128).
struct StepLists {
    double *prov, *absb;
    int stride, n_prov, n_absb;
    __device__ __forceinline__ void provided(double v) { prov[(n_prov++) * stride] = v; }
    __device__ __forceinline__ void absorbed(double v) { absb[(n_absb++) * stride] = v; }
};

__device__ inline double np_sum_strided(const double *a, int stride, int n)
{
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; i++) res += a[i * stride];
        return res;
    }
    double r[8];
    int i;
#pragma unroll
    for (i = 0; i < 8; i++) r[i] = a[i * stride];
    for (i = 8; i < n - (n % 8); i += 8)
#pragma unroll
        for (int j = 0; j < 8; j++) r[j] += a[(i + j) * stride];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i * stride];
    return res;
}

// addends a lane's lists can hold: provided <= gensets + batteries + grids + renewables + loss load, absorbed <= loads +
// batteries + grids + overgeneration
__host__ __device__ inline int multi_list_capacity(int n_load, int n_pv, int n_genset, int n_battery, int n_grid)
{
    const int a = n_genset + n_battery + n_grid + n_pv + 1, b = n_load + n_battery + n_grid + 1;
    return a > b ? a : b;
}

// One Microgrid.run of grid i with every module list swept in the reference's order (microgrid.py:255-314): fixed modules,
// gensets (pure sources), the source-and-sink names in list order (batteries and grids, every instance of a name in
// order), renewables, unbalanced energy.  Each controllable instance runs the SAME arithmetic as the single-instance
// kernels: step_core<that module only> on zero load / pv, of which only that module's part is kept.  State columns and
// the instance's log block are written as the sweep goes.  `o` receives the microgrid-level outputs (balance columns,
// load / pv / unbalanced columns, reward, violations; discharge_amount / charge_amount = what BatteryDischargeShaper sums).
template <int F, typename AT>
__device__ inline void step_multi_core(const KArgs &a, const AT *__restrict__ act, int64_t i, int32_t t, bool normalized,
                                       StepLists &L, double *__restrict__ log, Outputs &o, uint32_t viol0 = 0u)
{
    const int64_t N = a.N;
    const int NG = a.n_genset, NB = a.n_battery, NR = a.n_grid;
    double reward = 0.0;
    uint32_t viol = viol0;                                // (the expansion's assert mask when the control came from a priority list)
    L.n_prov = 0; L.n_absb = 0;
    o.load_met = 0.0;
    for (int j = 0; j < a.n_load; j++) {                  // fixed modules, module order (microgrid.py:255-257)
        const double Lv = -1 * a.c.load_ts[((int64_t)t * a.n_load + j) * N + i];
        o.load_met += Lv;
        L.absorbed(Lv); reward += 0.0;
    }
    o.fixed_provided = np_sum_strided(L.prov, L.stride, L.n_prov);          // :259-260
    o.fixed_absorbed = np_sum_strided(L.absb, L.stride, L.n_absb);

    const int kg = LC_COMMON_END, kb = kg + LC_GENSET_N * NG, kr = kb + LC_BATTERY_N * NB;    // log blocks
    Inputs in; in.load = 0.0; in.pv = 0.0;
    Outputs oc;
    if constexpr (F & F_GENSET) {
        for (int j = 0; j < NG; j++) {
            const int64_t c = (int64_t)j * N + i;
            Params p; Derived d; State s;
            load_module_params<F_GENSET>(a.c, c, p); derive<F_GENSET>(p, d);
            s.status = a.c.gen_status[c];
            in.a_goal = (double)act[2 * j]; in.a_gen = (double)act[2 * j + 1];
            step_core<F_GENSET>(p, d, s, in, normalized, false, false, oc);
            a.c.gen_status[c] = s.status;
            L.provided(oc.genset_production); reward += oc.genset_reward; viol |= oc.violations;
            if (log) {
                double *q = log + (int64_t)(kg + LC_GENSET_N * j) * N;
                q[0] = oc.genset_production; q[N] = oc.genset_co2; q[2 * N] = oc.genset_reward; q[3 * N] = (double)s.status;
            }
        }
    }
    // BatteryDischargeShaper sums info['battery'][*]['provided_energy'] with a KeyError -> 0.0 fallback for the WHOLE sum
    // (reward_shaping/base.py:10-16): one battery acting as a sink zeroes it
    double discharge_sum = 0.0; bool any_sink = false;
    auto step_batteries = [&]() __attribute__((always_inline)) {
        for (int j = 0; j < NB; j++) {
            const int64_t c = (int64_t)j * N + i;
            Params p; Derived d; State s;
            load_module_params<F_BATTERY>(a.c, c, p); derive<F_BATTERY>(p, d);
            s.charge = a.c.charge[c]; s.soc = a.c.soc[c]; s.status = 0u;
            in.a_bat = (double)act[2 * NG + j];
            step_core<F_BATTERY>(p, d, s, in, normalized, true, false, oc);
            a.c.charge[c] = s.charge; a.c.soc[c] = s.soc;
            // as_source iff the unnormalised request is >= 0 (base_module.py:161-171); a sink logs charge_amount
            const double x = normalized ? d.bat_lo + d.bat_sp * in.a_bat : in.a_bat;
            if (x < 0) { L.absorbed(oc.charge_amount); any_sink = true; }
            else { L.provided(oc.discharge_amount); discharge_sum += oc.discharge_amount; }
            reward += oc.battery_reward; viol |= oc.violations;
            if (log) {
                double *q = log + (int64_t)(kb + LC_BATTERY_N * j) * N;
                q[0] = oc.discharge_amount; q[N] = oc.charge_amount; q[2 * N] = oc.battery_reward;
                q[3 * N] = oc.soc_pre; q[4 * N] = oc.charge_pre;
            }
        }
    };
    auto step_grids = [&]() __attribute__((always_inline)) {
        for (int j = 0; j < NR; j++) {
            const int64_t c = (int64_t)j * N + i;
            Params p; Derived d; State s;
            load_module_params<F_GRID>(a.c, c, p); derive<F_GRID>(p, d);
            s.charge = 0.0; s.soc = 0.0; s.status = 0u;
            const double *g = a.c.grid_ts + (((int64_t)t * NR + j) * 4) * N + i;
            in.g_pimp = g[0]; in.g_pexp = g[N]; in.g_co2 = g[2 * N]; in.g_stat = g[3 * N];
            in.a_grid = (double)act[2 * NG + NB + j];
            step_core<F_GRID>(p, d, s, in, normalized, false, false, oc);
            const double x = normalized ? d.grid_lo + d.grid_sp * in.a_grid : in.a_grid;
            if (x < 0) L.absorbed(oc.grid_export); else L.provided(oc.grid_import);
            reward += oc.grid_reward; viol |= oc.violations;
            if (log) {
                double *q = log + (int64_t)(kr + LC_GRID_N * j) * N;
                q[0] = oc.grid_import; q[N] = oc.grid_export; q[2 * N] = oc.grid_co2; q[3 * N] = oc.grid_reward;
            }
        }
    };
    if constexpr ((F & F_GRID_FIRST) != 0) {
        if constexpr (F & F_GRID) step_grids();
        if constexpr (F & F_BATTERY) step_batteries();
    } else {
        if constexpr (F & F_BATTERY) step_batteries();
        if constexpr (F & F_GRID) step_grids();
    }
    o.discharge_amount = any_sink ? 0.0 : discharge_sum;
    o.charge_amount = 0.0;
    const double provided = np_sum_strided(L.prov, L.stride, L.n_prov);      // :277
    const double consumed = np_sum_strided(L.absb, L.stride, L.n_absb);
    const double difference = provided - consumed;
    o.ctrl_provided = provided - o.fixed_provided; o.ctrl_absorbed = consumed - o.fixed_absorbed;

    const double ll_cost = a.c.loss_load_cost[i], og_cost = a.c.overgeneration_cost[i];
    o.renewable_used = 0.0; o.curtailment = 0.0;
    if (difference > 0) {                                 // :286-299: renewables idle, the excess is overgeneration
        for (int j = 0; j < a.n_pv; j++) {
            o.curtailment += a.c.pv_ts[((int64_t)t * a.n_pv + j) * N + i] - 0.0;
            L.provided(0.0); reward += 0.0;
        }
        const double e = -1.0 * (-1.0 * difference);
        o.overgeneration = e; o.loss_load = 0.0;
        o.unbalanced_reward = -1.0 * (og_cost * e);
        L.absorbed(e);
    } else {                                              // :301-314: renewables in module order, then loss load
        double need = -difference;
        for (int j = 0; j < a.n_pv; j++) {
            const double pv = a.c.pv_ts[((int64_t)t * a.n_pv + j) * N + i];
            const double amt = (pv < need) ? pv : need;
            o.renewable_used += amt; o.curtailment += pv - amt;
            L.provided(amt); reward += 0.0;
            need -= amt;
        }
        o.loss_load = need; o.overgeneration = 0.0;
        o.unbalanced_reward = -1.0 * (ll_cost * need);
        L.provided(need);
    }
    reward += o.unbalanced_reward;
    o.overall_provided = np_sum_strided(L.prov, L.stride, L.n_prov);         // :316-317
    o.overall_absorbed = np_sum_strided(L.absb, L.stride, L.n_absb);
    o.reward = reward;
    o.violations = viol;
    if (log) {
        log[0] = o.reward;
        log[N] = o.fixed_provided;        log[2 * N] = o.fixed_absorbed;
        log[3 * N] = o.ctrl_provided;     log[4 * N] = o.ctrl_absorbed;
        log[5 * N] = o.overall_provided;  log[6 * N] = o.overall_absorbed;
        log[7 * N] = o.load_met;          log[8 * N] = o.renewable_used;
        log[9 * N] = o.curtailment;       log[10 * N] = o.loss_load;
        log[11 * N] = o.overgeneration;   log[12 * N] = o.unbalanced_reward;
        log[(int64_t)(kr + LC_GRID_N * NR) * N] = (double)viol;
    }
}

// ------------------------------------------------------------------------------------------------------
// The SMALL form of the general step: at most MS modules of every kind per grid (what "several modules of a kind" means in
// practice: two gensets, two batteries ...).  step_multi_core walks its module lists with run-time counts, so every instance's
// parameters, state, control and series values are loaded inside the sweep -- one dependent round trip per instance, 5-7 per
// step, in a kernel that is nothing but round trips (10 us per 100 000-grid step for 2 gensets + 2 batteries + 1 grid).  Here the
// counts are bounded at COMPILE time: everything the step reads is requested up front into registers (every load unconditional
// -- an instance the layout does not have re-reads the last one it has: a cache hit -- so that all are in flight together), the
// sweep runs on registers with `j < n` guards around its EFFECTS only, and a K-step loop keeps parameters and state there across
// steps.  Same operations on the same operands in the same order as step_multi_core (the lists and their sums are its), hence
// the same bits: tests/test_multiplicity.py runs both forms on the same batches.
// ------------------------------------------------------------------------------------------------------
constexpr int MS = 2;

// How many modules of each kind the grids of a launch have: read from the launch's KArgs (any layout: the loops keep their `j < n`
// guards and the instances a layout lacks re-read the last one it has) -- or fixed at COMPILE time for the layouts that occur most
// (step_k_multi_small_kernel<F, CountsCT<...>>, mgx_fused.hip part 5): the guards fold away, nothing is loaded twice, the presence
// bits of the provided list become constants.  The same operations on the same operands either way.
struct CountsRT {
    static __device__ __forceinline__ int ng(const KArgs &a) { return a.n_genset; }
    static __device__ __forceinline__ int nb(const KArgs &a) { return a.n_battery; }
    static __device__ __forceinline__ int nr(const KArgs &a) { return a.n_grid; }
    static __device__ __forceinline__ int nl(const KArgs &a) { return a.n_load; }
    static __device__ __forceinline__ int np(const KArgs &a) { return a.n_pv; }
    static constexpr bool is_static = false;
    static constexpr int kNG = 0, kNB = 0, kNR_ = 0, kNP_ = 0;
    static constexpr int max_prov = 0, max_absb = 0, max_mid_prov = 0, max_mid_absb = 0;   // (unused: the slot counts decide)
};
constexpr int MS_CT = 3;     // most instance slots a compile-time-count specialisation may hold
template <int NG_, int NB_, int NR_, int NL_, int NP_>
struct CountsCT {
    static_assert(NG_ <= MS_CT && NB_ <= MS_CT && NR_ <= MS_CT && NL_ >= 1 && NL_ <= MS_CT && NP_ >= 1 && NP_ <= MS_CT, "the register form holds M instances");
    static constexpr int slots = (NG_ > MS || NB_ > MS || NR_ > MS || NL_ > MS || NP_ > MS) ? MS_CT : MS;     // M of the kernel
    static constexpr bool is_static = true;
    static constexpr int kNG = NG_, kNB = NB_, kNR_ = NR_, kNP_ = NP_;
    // most addends MicrogridStep's lists can hold: at the end of the sweep (gensets / discharging batteries / importing grids / renewables /
    // loss load; loads / charging batteries / exporting grids / overgeneration) and after the controllable modules (:277)
    static constexpr int max_prov = NG_ + NB_ + NR_ + NP_ + 1, max_absb = NL_ + NB_ + NR_ + 1, max_mid_prov = NG_ + NB_ + NR_,
                         max_mid_absb = NL_ + NB_ + NR_;
    static __device__ __forceinline__ constexpr int ng(const KArgs &) { return NG_; }
    static __device__ __forceinline__ constexpr int nb(const KArgs &) { return NB_; }
    static __device__ __forceinline__ constexpr int nr(const KArgs &) { return NR_; }
    static __device__ __forceinline__ constexpr int nl(const KArgs &) { return NL_; }
    static __device__ __forceinline__ constexpr int np(const KArgs &) { return NP_; }
};

// M: instance slots per kind the register form holds (MS for the run-time-count form; the compile-time-count specialisations of
// layouts with three modules of a kind use 3 -- slots a layout does not have cost nothing there)
template <int M>
struct MultiRegsT {                      // parameters + dynamic state of one grid (per instance)
    double g_rmin[M], g_rmax[M], g_cost[M], g_co2[M], g_cco2[M];
    uint32_t g_times[M], g_status[M];
    double b_cmin[M], b_cmax[M], b_C[M], b_D[M], b_eta[M], b_cost[M], b_charge[M], b_soc[M];
    double r_imp[M], r_exp[M], r_cco2[M];
    double ll_cost, og_cost;
    Derived d_gen[M], d_bat[M], d_grid[M];      // the step-invariant values of every instance (derive: once per launch, not per step)
};
using MultiRegs = MultiRegsT<MS>;

template <int M>
struct MultiStepInT {                    // what one step reads besides: controls and series rows
    double goal[M], gen[M], bat[M], grd[M];
    double load[M], pv[M], grid[M][4];
};
using MultiStepIn = MultiStepInT<MS>;

// LDS PARKING of step-invariant parameters (round 6, the three-of-a-kind forms): 3 gensets + 3 batteries + 1 grid keep 330 registers
// per lane live across the K-step loop -- one wave per SIMD, so a 100 000-grid launch runs in two rounds of waves.  The parameters a
// step reads ONCE (cost terms, capacity limits, the action-space constants) wait in a per-lane LDS column instead (slot-major,
// PARK_STRIDE lanes per slot: lane-consecutive 8-byte reads, conflict-free) and are read where the sweep needs them; the dynamic state and
// the power limits stay in registers.  The reads are `volatile` so that the compiler does not hoist them back out of the loop into
// registers.  Same operands, same operations: the bits cannot differ.  At most 38 doubles per lane (3 gensets + 3 batteries) = 19 456 B per wave: 8 waves per CU.
constexpr int PARK_STRIDE = 64;
template <class CNT, int M>
struct ParkSlots {                                          // slot rows: per genset 4, per battery 8, loss-load / overgeneration cost
    static constexpr int NG = CNT::is_static ? CNT::kNG : M, NB = CNT::is_static ? CNT::kNB : M;
    static constexpr int G_RMIN = 0, G_COST = NG, G_CO2 = 2 * NG, G_CCO2 = 3 * NG, B0 = 4 * NG, B_CMIN = B0, B_CMAX = B0 + NB, B_ETA = B0 + 2 * NB,
                         B_COST = B0 + 3 * NB, B_C = B0 + 4 * NB, B_D = B0 + 5 * NB, B_LO = B0 + 6 * NB, B_SP = B0 + 7 * NB, LL = B0 + 8 * NB, OG = LL + 1,
                         COUNT = OG + 1;
};
// (an LDS-address-space pointer: a volatile access through a generic pointer stays a flat_load with a 64-bit address and system scope)
typedef __attribute__((address_space(3))) double lds_double;
__device__ __forceinline__ double park_ld(const lds_double *pk, int slot) { return *(const volatile lds_double *)(pk + slot * PARK_STRIDE); }

template <int F, class CNT, int M>
__device__ __forceinline__ void park_multi_regs(const KArgs &a, const MultiRegsT<M> &R, lds_double *pk)
{
    using PS = ParkSlots<CNT, M>;
#pragma unroll
    for (int j = 0; j < M; j++) {
        if constexpr (F & F_GENSET) {
            if (j < CNT::ng(a)) {
                pk[(PS::G_RMIN + j) * PARK_STRIDE] = R.g_rmin[j]; pk[(PS::G_COST + j) * PARK_STRIDE] = R.g_cost[j];
                pk[(PS::G_CO2 + j) * PARK_STRIDE] = R.g_co2[j]; pk[(PS::G_CCO2 + j) * PARK_STRIDE] = R.g_cco2[j];
            }
        }
        if constexpr (F & F_BATTERY) {
            if (j < CNT::nb(a)) {
                pk[(PS::B_CMIN + j) * PARK_STRIDE] = R.b_cmin[j]; pk[(PS::B_CMAX + j) * PARK_STRIDE] = R.b_cmax[j];
                pk[(PS::B_ETA + j) * PARK_STRIDE] = R.b_eta[j]; pk[(PS::B_COST + j) * PARK_STRIDE] = R.b_cost[j];
                pk[(PS::B_C + j) * PARK_STRIDE] = R.b_C[j]; pk[(PS::B_D + j) * PARK_STRIDE] = R.b_D[j];
                pk[(PS::B_LO + j) * PARK_STRIDE] = R.d_bat[j].bat_lo; pk[(PS::B_SP + j) * PARK_STRIDE] = R.d_bat[j].bat_sp;
            }
        }
    }
    pk[PS::LL * PARK_STRIDE] = R.ll_cost; pk[PS::OG * PARK_STRIDE] = R.og_cost;
}

__host__ __device__ inline bool multi_is_small(int n_load, int n_pv, int n_genset, int n_battery, int n_grid)
{
    return n_load >= 1 && n_pv >= 1 && n_load <= MS && n_pv <= MS && n_genset <= MS && n_battery <= MS && n_grid <= MS;
}

template <int F, class CNT = CountsRT, int M = MS>
__device__ __forceinline__ void load_multi_regs(const KArgs &a, int64_t i, MultiRegsT<M> &R)
{
    const int64_t N = a.N;
    const mgx_columns &c = a.c;
    const int NG = CNT::ng(a), NB = CNT::nb(a), NR = CNT::nr(a);
#pragma unroll
    for (int j = 0; j < M; j++) {
        if constexpr (F & F_GENSET) {
            const int64_t q = (int64_t)(j < NG ? j : NG - 1) * N + i;
            R.g_rmin[j] = c.gen_running_min[q]; R.g_rmax[j] = c.gen_running_max[q]; R.g_cost[j] = c.gen_cost[q];
            R.g_co2[j] = c.gen_co2_per_unit[q]; R.g_cco2[j] = c.gen_cost_per_unit_co2[q];
            R.g_times[j] = c.gen_times[q]; R.g_status[j] = c.gen_status[q];
        }
        if constexpr (F & F_BATTERY) {
            const int64_t q = (int64_t)(j < NB ? j : NB - 1) * N + i;
            R.b_cmin[j] = c.bat_min_capacity[q]; R.b_cmax[j] = c.bat_max_capacity[q]; R.b_C[j] = c.bat_max_charge[q];
            R.b_D[j] = c.bat_max_discharge[q]; R.b_eta[j] = c.bat_efficiency[q]; R.b_cost[j] = c.bat_cost_cycle[q];
            R.b_charge[j] = c.charge[q]; R.b_soc[j] = c.soc[q];
        }
        if constexpr (F & F_GRID) {
            const int64_t q = (int64_t)(j < NR ? j : NR - 1) * N + i;
            R.r_imp[j] = c.grid_max_import[q]; R.r_exp[j] = c.grid_max_export[q]; R.r_cco2[j] = c.grid_cost_per_unit_co2[q];
        }
    }
    R.ll_cost = c.loss_load_cost[i]; R.og_cost = c.overgeneration_cost[i];
    // the action-space constants of every instance: the same operations on the same operands as a derive per step, hence the same bits
#pragma unroll
    for (int j = 0; j < M; j++) {
        Params p;
        if constexpr (F & F_GENSET) { p.gen_rmax = R.g_rmax[j]; derive<F_GENSET>(p, R.d_gen[j]); }
        if constexpr (F & F_BATTERY) { p.bat_D = R.b_D[j]; p.bat_eta = R.b_eta[j]; p.bat_C = R.b_C[j]; derive<F_BATTERY>(p, R.d_bat[j]); }
        if constexpr (F & F_GRID) { p.grid_exp = R.r_exp[j]; p.grid_imp = R.r_imp[j]; derive<F_GRID>(p, R.d_grid[j]); }
    }
}

// the dynamic state back into the batch's columns
template <int F, class CNT = CountsRT, int M = MS>
__device__ __forceinline__ void store_multi_state(const KArgs &a, int64_t i, const MultiRegsT<M> &R)
{
    const int64_t N = a.N;
#pragma unroll
    for (int j = 0; j < M; j++) {
        if constexpr (F & F_GENSET) { if (j < CNT::ng(a)) a.c.gen_status[(int64_t)j * N + i] = R.g_status[j]; }
        if constexpr (F & F_BATTERY) { if (j < CNT::nb(a)) { a.c.charge[(int64_t)j * N + i] = R.b_charge[j]; a.c.soc[(int64_t)j * N + i] = R.b_soc[j]; } }
    }
}

template <int F, typename AT, class CNT = CountsRT, int M = MS>
__device__ __forceinline__ void load_multi_step_in(const KArgs &a, const AT *__restrict__ act, int64_t i, int32_t t, MultiStepInT<M> &in)
{
    const int64_t N = a.N;
    const int NG = CNT::ng(a), NB = CNT::nb(a), NR = CNT::nr(a), NL = CNT::nl(a), NP = CNT::np(a);
#pragma unroll
    for (int j = 0; j < M; j++) {
        if constexpr (F & F_GENSET) { const int q = j < NG ? j : NG - 1; in.goal[j] = (double)act[2 * q]; in.gen[j] = (double)act[2 * q + 1]; }
        if constexpr (F & F_BATTERY) { const int q = j < NB ? j : NB - 1; in.bat[j] = (double)act[2 * NG + q]; }
        if constexpr (F & F_GRID) {
            const int q = j < NR ? j : NR - 1;
            in.grd[j] = (double)act[2 * NG + NB + q];
            const double *g = a.c.grid_ts + (((int64_t)t * NR + q) * 4) * N + i;
            in.grid[j][0] = g[0]; in.grid[j][1] = g[N]; in.grid[j][2] = g[2 * N]; in.grid[j][3] = g[3 * N];
        }
        { const int q = j < NL ? j : NL - 1; in.load[j] = a.c.load_ts[((int64_t)t * NL + q) * N + i]; }
        { const int q = j < NP ? j : NP - 1; in.pv[j] = a.c.pv_ts[((int64_t)t * NP + q) * N + i]; }
    }
}

// One Microgrid.run of grid i out of registers: step_multi_core's sweep, line for line, with compile-time instance indices.
// The lists of step_multi_small: slot s of `provided` = gensets [0, MS), then the source-and-sink names in sweep order (batteries and
// grids, MS slots each), renewables [3 MS, 4 MS), loss load 4 MS; of `absorbed` = loads [0, MS), batteries / grids as above,
// overgeneration 3 MS.  A slot is an addend iff its bit is set; the list np.sum sees is the set slots in slot order.  The sums follow
// np_sum_strided: a running sum from 0.0 below 8 addends; with 8 or 9 (only `provided` can: at most ONE of its 9 slots is then unset)
// the first eight in numpy's pairwise order, the ninth added last.
constexpr int SMALL_PROV = 4 * MS + 1, SMALL_ABSB = 3 * MS + 1;
static_assert(MS == 2, "small_pairwise_prov assumes at most 9 addends: one unset slot at 8");

// The same for ANY number of slots (M = 3: 13 provided slots, 10 absorbed ones): the set slots in slot order are the list np.sum sees;
// with n >= 8 addends numpy sums the first eight pairwise and adds the rest one by one (np_sum_strided, n < 16).  The addends are
// compacted into registers with compile-time-unrolled selects on a running count (no indexed register arrays: they would live in scratch).
template <int NSLOT>
__device__ __forceinline__ double slots_pairwise_sum(const double (&e)[NSLOT], uint32_t mask, int n)
{
    static_assert(NSLOT <= 15, "one pairwise block of eight + a tail of at most seven addends");
    constexpr int NT = NSLOT > 8 ? NSLOT - 8 : 1;
    double r[8] = {}, tail[NT] = {};
    int c = 0;
#pragma unroll
    for (int sl = 0; sl < NSLOT; sl++) {
        const bool on = (mask >> sl) & 1u;
#pragma unroll
        for (int q = 0; q < 8; q++) r[q] = (on && c == q) ? e[sl] : r[q];
#pragma unroll
        for (int q = 0; q < NT; q++) tail[q] = (on && c == 8 + q) ? e[sl] : tail[q];
        c += on ? 1 : 0;
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
#pragma unroll
    for (int q = 0; q < NT; q++) if (8 + q < n) res += tail[q];
    return res;
}

// `provided` with 8 or 9 addends (n = popcount(mask) >= 8: at most one of the nine slots unset): numpy's pairwise order over the
// first eight, the ninth added last (np_sum_strided).  Below eight addends -- always for `absorbed`, SMALL_ABSB = 7 -- numpy's sum
// is the running sum from 0.0 in list order, which the sweep keeps as it appends (the slots are in append order).
static_assert(SMALL_ABSB < 8, "the absorbed list of the register form is summed as a running sum");
__device__ __forceinline__ double small_pairwise_prov(const double (&e)[SMALL_PROV], uint32_t mask, int n)
{
    const uint32_t miss = ~mask & ((1u << SMALL_PROV) - 1u);          // no bit (n = 9) or one (n = 8)
    const int m = miss ? __ffs((int)miss) - 1 : SMALL_PROV;
    double r[8];
#pragma unroll
    for (int k = 0; k < 8; k++) r[k] = (k < m) ? e[k] : e[k + 1];     // the first eight addends
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    if (n == 9) res += e[8];
    return res;
}

// The provided list's sum for a COMPILE-TIME layout whose list can hold at most nine addends (3 gensets + 3 batteries + 1 grid + 1
// renewable + loss load): the slots the layout can fill at all are known at compile time, and with eight or nine addends at most ONE
// of them is unset -- small_pairwise_prov's rule on that compile-time slot list instead of slots_pairwise_sum's run-time compaction of
// all 13 slots (13 x 13 selects of doubles per step: half of the three-of-a-kind kernel's 688 VALU instructions per wave-step).
template <int F, class CNT, int M>
struct ProvSlotList {
    int idx[4 * M + 1];
    int n;
    constexpr ProvSlotList() : idx{}, n(0)
    {
        constexpr int SB = (F & F_GRID_FIRST) ? 2 * M : M, SR = (F & F_GRID_FIRST) ? M : 2 * M;
        for (int j = 0; j < CNT::kNG; j++) idx[n++] = j;
        if (SB < SR) { for (int j = 0; j < CNT::kNB; j++) idx[n++] = SB + j; for (int j = 0; j < CNT::kNR_; j++) idx[n++] = SR + j; }
        else { for (int j = 0; j < CNT::kNR_; j++) idx[n++] = SR + j; for (int j = 0; j < CNT::kNB; j++) idx[n++] = SB + j; }
        for (int j = 0; j < CNT::kNP_; j++) idx[n++] = 3 * M + j;
        idx[n++] = 4 * M;
    }
};

template <int F, class CNT, int M>
__device__ __forceinline__ double ct_pairwise_prov(const double (&pe)[4 * M + 1], uint32_t pm, int n)
{
    constexpr ProvSlotList<F, CNT, M> SL{};
    static_assert(SL.n == CNT::max_prov && SL.n >= 8 && SL.n <= 9, "eight or nine possible addends");
    double r[8];
    if constexpr (SL.n == 8) {                       // eight addends: every possible slot is set
#pragma unroll
        for (int k = 0; k < 8; k++) r[k] = pe[SL.idx[k]];
        return ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    } else {
        uint32_t mask9 = 0u;
#pragma unroll
        for (int k = 0; k < 9; k++) mask9 |= ((pm >> SL.idx[k]) & 1u) << k;
        const uint32_t miss = ~mask9 & 0x1ffu;                                // no bit (n = 9) or one (n = 8)
        const int m = miss ? __ffs((int)miss) - 1 : 9;
#pragma unroll
        for (int k = 0; k < 8; k++) {                 // (through an empty asm: a select between two elements of one array becomes an indexed load = scratch)
            double lo = pe[SL.idx[k]], hi = pe[SL.idx[k + 1]];
            asm("" : "+v"(lo)); asm("" : "+v"(hi));
            r[k] = (k < m) ? lo : hi;
        }
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        if (n == 9) res += pe[SL.idx[8]];
        return res;
    }
}

template <int F, class CNT = CountsRT, int M = MS, bool PARK = false>
__device__ __forceinline__ void step_multi_small(const KArgs &a, MultiRegsT<M> &R, const MultiStepInT<M> &sin, int64_t i, bool normalized,
                                                 double *__restrict__ log, Outputs &o, const lds_double *pk = nullptr, uint32_t viol0 = 0u)
{
    const int64_t N = a.N;
    using PS = ParkSlots<CNT, M>;
    // a parked parameter comes from its LDS slot (park_multi_regs), any other from the register copy
    auto cold = [&](int slot, double reg) __attribute__((always_inline)) { if constexpr (PARK) return park_ld(pk, slot); else return reg; };
    const int NG = CNT::ng(a), NB = CNT::nb(a), NR = CNT::nr(a), NL = CNT::nl(a), NP = CNT::np(a);
    double reward = 0.0;
    uint32_t viol = viol0;                                // (the expansion's assert mask when the control came from a priority list)
    // The provided / absorbed lists of MicrogridStep in REGISTERS: running sums kept as the sweep appends (numpy's sum of fewer than
    // eight addends IS the running sum from 0.0 in list order) + one static slot per possible `provided` addend in sweep order with a
    // presence bit each, for the one sum that can see 8 or 9 addends -- the run-time form appends to LDS columns and sums them with a
    // dependent LDS read per addend.
    constexpr int SB = (F & F_GRID_FIRST) ? 2 * M : M, SR = (F & F_GRID_FIRST) ? M : 2 * M;     // first battery / grid slot
    constexpr int NPROV = 4 * M + 1, NABSB = 3 * M + 1;   // slots of the provided / absorbed lists
    // The addends themselves are kept only where a list CAN reach the eight addends from which numpy sums pairwise: by the slot counts
    // with run-time instance counts, by the layout's own maxima with compile-time ones (2 g + 2 b + 1 grid: 7 provided addends at
    // most -- no slots at all, every sum is the running sum)
    constexpr bool TRACK_P = CNT::is_static ? CNT::max_prov >= 8 : NPROV >= 8, TRACK_A = CNT::is_static ? CNT::max_absb >= 8 : NABSB >= 8;
    constexpr bool MID_P = CNT::is_static ? CNT::max_mid_prov >= 8 : 3 * M >= 8, MID_A = CNT::is_static ? CNT::max_mid_absb >= 8 : 3 * M >= 8;
    double pe[TRACK_P ? NPROV : 1] = {};
    double ae[TRACK_A ? NABSB : 1] = {};
    uint32_t am = 0u;
    uint32_t pm = 0u;
    double psum = 0.0, asum = 0.0;                            // running sums of the two lists, from 0.0 in append order
    o.load_met = 0.0;
#pragma unroll
    for (int j = 0; j < M; j++) {                        // fixed modules, module order (microgrid.py:255-257)
        if (j < NL) {
            const double Lv = -1 * sin.load[j];
            o.load_met += Lv;
            asum += Lv; reward += 0.0;
            if constexpr (TRACK_A) { ae[j] = Lv; am |= 1u << j; }
        }
    }
    o.fixed_provided = psum;                                // :259-260 (an empty list: 0.0)
    o.fixed_absorbed = asum;

    const int kg = LC_COMMON_END, kb = kg + LC_GENSET_N * NG, kr = kb + LC_BATTERY_N * NB;    // log blocks
    Inputs in; in.load = 0.0; in.pv = 0.0;
    Outputs oc;
    if constexpr (F & F_GENSET) {
#pragma unroll
        for (int j = 0; j < M; j++) {
            if (j < NG) {
                Params p; Derived d; State s;
                p.gen_rmin = cold(PS::G_RMIN + j, R.g_rmin[j]); p.gen_rmax = R.g_rmax[j]; p.gen_cost = cold(PS::G_COST + j, R.g_cost[j]);
                p.gen_co2 = cold(PS::G_CO2 + j, R.g_co2[j]); p.gen_cco2 = cold(PS::G_CCO2 + j, R.g_cco2[j]); p.gen_times = R.g_times[j];
                d = R.d_gen[j];
                if constexpr (PARK) derive<F_GENSET>(p, d);          // (formed again from running_max, which stays in a register: a compare and a select)
                s.status = R.g_status[j];
                in.a_goal = sin.goal[j]; in.a_gen = sin.gen[j];
                step_core<F_GENSET>(p, d, s, in, normalized, false, false, oc);
                R.g_status[j] = s.status;
                if constexpr (TRACK_P) { pe[j] = oc.genset_production; pm |= 1u << j; } psum += oc.genset_production; reward += oc.genset_reward; viol |= oc.violations;
                if (log) {
                    double *q = log + (int64_t)(kg + LC_GENSET_N * j) * N;
                    MGX_LOG_ST(q + 0, oc.genset_production); MGX_LOG_ST(q + N, oc.genset_co2); MGX_LOG_ST(q + 2 * N, oc.genset_reward); MGX_LOG_ST(q + 3 * N, (double)s.status);
                }
            }
        }
    }
    double discharge_sum = 0.0; bool any_sink = false;   // BatteryDischargeShaper's sum (step_multi_core)
    auto step_batteries = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < M; j++) {
            if (j < NB) {
                Params p; Derived d; State s;
                p.bat_cmin = cold(PS::B_CMIN + j, R.b_cmin[j]); p.bat_cmax = cold(PS::B_CMAX + j, R.b_cmax[j]); p.bat_C = cold(PS::B_C + j, R.b_C[j]);
                p.bat_D = cold(PS::B_D + j, R.b_D[j]); p.bat_eta = cold(PS::B_ETA + j, R.b_eta[j]); p.bat_cost = cold(PS::B_COST + j, R.b_cost[j]);
                d = R.d_bat[j];
                if constexpr (PARK) { d.bat_lo = park_ld(pk, PS::B_LO + j); d.bat_sp = park_ld(pk, PS::B_SP + j); }
                s.charge = R.b_charge[j]; s.soc = R.b_soc[j]; s.status = 0u;
                in.a_bat = sin.bat[j];
                step_core<F_BATTERY>(p, d, s, in, normalized, true, false, oc);
                R.b_charge[j] = s.charge; R.b_soc[j] = s.soc;
                const double x = normalized ? d.bat_lo + d.bat_sp * in.a_bat : in.a_bat;
                if (x < 0) {
                    asum += oc.charge_amount; any_sink = true;
                    if constexpr (TRACK_A) { ae[SB + j] = oc.charge_amount; am |= 1u << (SB + j); }
                }
                else { if constexpr (TRACK_P) { pe[SB + j] = oc.discharge_amount; pm |= 1u << (SB + j); } psum += oc.discharge_amount; discharge_sum += oc.discharge_amount; }
                reward += oc.battery_reward; viol |= oc.violations;
                if (log) {
                    double *q = log + (int64_t)(kb + LC_BATTERY_N * j) * N;
                    MGX_LOG_ST(q + 0, oc.discharge_amount); MGX_LOG_ST(q + N, oc.charge_amount); MGX_LOG_ST(q + 2 * N, oc.battery_reward);
                    MGX_LOG_ST(q + 3 * N, oc.soc_pre); MGX_LOG_ST(q + 4 * N, oc.charge_pre);
                }
            }
        }
    };
    auto step_grids = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < M; j++) {
            if (j < NR) {
                Params p; Derived d; State s;
                p.grid_imp = R.r_imp[j]; p.grid_exp = R.r_exp[j]; p.grid_cco2 = R.r_cco2[j];
                d = R.d_grid[j];
                s.charge = 0.0; s.soc = 0.0; s.status = 0u;
                in.g_pimp = sin.grid[j][0]; in.g_pexp = sin.grid[j][1]; in.g_co2 = sin.grid[j][2]; in.g_stat = sin.grid[j][3];
                in.a_grid = sin.grd[j];
                step_core<F_GRID>(p, d, s, in, normalized, false, false, oc);
                const double x = normalized ? d.grid_lo + d.grid_sp * in.a_grid : in.a_grid;
                if (x < 0) {
                    asum += oc.grid_export;
                    if constexpr (TRACK_A) { ae[SR + j] = oc.grid_export; am |= 1u << (SR + j); }
                }
                else { if constexpr (TRACK_P) { pe[SR + j] = oc.grid_import; pm |= 1u << (SR + j); } psum += oc.grid_import; }
                reward += oc.grid_reward; viol |= oc.violations;
                if (log) {
                    double *q = log + (int64_t)(kr + LC_GRID_N * j) * N;
                    MGX_LOG_ST(q + 0, oc.grid_import); MGX_LOG_ST(q + N, oc.grid_export); MGX_LOG_ST(q + 2 * N, oc.grid_co2); MGX_LOG_ST(q + 3 * N, oc.grid_reward);
                }
            }
        }
    };
    if constexpr ((F & F_GRID_FIRST) != 0) {
        if constexpr (F & F_GRID) step_grids();
        if constexpr (F & F_BATTERY) step_batteries();
    } else {
        if constexpr (F & F_BATTERY) step_batteries();
        if constexpr (F & F_GRID) step_grids();
    }
    o.discharge_amount = any_sink ? 0.0 : discharge_sum;
    o.charge_amount = 0.0;
    // :277: np.sum of the lists so far -- a running sum below eight addends (always with M = 2: at most 6 so far), numpy's pairwise order from eight on
    double provided = psum, consumed = asum;
    if constexpr (MID_P) { const int np_mid = __popc(pm); if (np_mid >= 8) provided = slots_pairwise_sum(pe, pm, np_mid); }
    if constexpr (MID_A) { const int na_mid = __popc(am); if (na_mid >= 8) consumed = slots_pairwise_sum(ae, am, na_mid); }
    const double difference = provided - consumed;
    o.ctrl_provided = provided - o.fixed_provided; o.ctrl_absorbed = consumed - o.fixed_absorbed;

    const double ll_cost = cold(PS::LL, R.ll_cost), og_cost = cold(PS::OG, R.og_cost);
    o.renewable_used = 0.0; o.curtailment = 0.0;
    if (difference > 0) {                                 // :286-299: renewables idle, the excess is overgeneration
#pragma unroll
        for (int j = 0; j < M; j++) {
            if (j < NP) {
                o.curtailment += sin.pv[j] - 0.0;
                if constexpr (TRACK_P) { pe[3 * M + j] = 0.0; pm |= 1u << (3 * M + j); } psum += 0.0; reward += 0.0;
            }
        }
        const double e = -1.0 * (-1.0 * difference);
        o.overgeneration = e; o.loss_load = 0.0;
        o.unbalanced_reward = -1.0 * (og_cost * e);
        asum += e;
        if constexpr (TRACK_A) { ae[3 * M] = e; am |= 1u << (3 * M); }
    } else {                                              // :301-314: renewables in module order, then loss load
        double need = -difference;
#pragma unroll
        for (int j = 0; j < M; j++) {
            if (j < NP) {
                const double pv = sin.pv[j];
                const double amt = (pv < need) ? pv : need;
                o.renewable_used += amt; o.curtailment += pv - amt;
                if constexpr (TRACK_P) { pe[3 * M + j] = amt; pm |= 1u << (3 * M + j); } psum += amt; reward += 0.0;
                need -= amt;
            }
        }
        o.loss_load = need; o.overgeneration = 0.0;
        o.unbalanced_reward = -1.0 * (ll_cost * need);
        if constexpr (TRACK_P) { pe[4 * M] = need; pm |= 1u << (4 * M); } psum += need;
    }
    reward += o.unbalanced_reward;
    o.overall_provided = psum;                            // :316-317
    if constexpr (TRACK_P) {
        const int n_prov = __popc(pm);
        if constexpr (M == MS) { if (n_prov >= 8) o.overall_provided = small_pairwise_prov(pe, pm, n_prov); }
        else if constexpr (CNT::is_static && CNT::max_prov <= 9) { if (n_prov >= 8) o.overall_provided = ct_pairwise_prov<F, CNT, M>(pe, pm, n_prov); }
        else { if (n_prov >= 8) o.overall_provided = slots_pairwise_sum(pe, pm, n_prov); }
    }
    o.overall_absorbed = asum;
    if constexpr (TRACK_A) { const int n_absb = __popc(am); if (n_absb >= 8) o.overall_absorbed = slots_pairwise_sum(ae, am, n_absb); }
    o.reward = reward;
    o.violations = viol;
    if (log) {
        MGX_LOG_ST(log + 0, o.reward);
        MGX_LOG_ST(log + N, o.fixed_provided);        MGX_LOG_ST(log + 2 * N, o.fixed_absorbed);
        MGX_LOG_ST(log + 3 * N, o.ctrl_provided);     MGX_LOG_ST(log + 4 * N, o.ctrl_absorbed);
        MGX_LOG_ST(log + 5 * N, o.overall_provided);  MGX_LOG_ST(log + 6 * N, o.overall_absorbed);
        MGX_LOG_ST(log + 7 * N, o.load_met);          MGX_LOG_ST(log + 8 * N, o.renewable_used);
        MGX_LOG_ST(log + 9 * N, o.curtailment);       MGX_LOG_ST(log + 10 * N, o.loss_load);
        MGX_LOG_ST(log + 11 * N, o.overgeneration);   MGX_LOG_ST(log + 12 * N, o.unbalanced_reward);
        MGX_LOG_ST(log + (int64_t)(kr + LC_GRID_N * NR) * N, (double)viol);
    }
}

// ---- priority lists over module instances, register form (round 6: RuleBasedControl on layouts with several modules of a kind) ----
// The series rows of one step without the controls (load_multi_step_in reads both): what a list rollout reads per step.
template <int F, class CNT = CountsRT, int M = MS>
__device__ __forceinline__ void load_multi_series(const KArgs &a, int64_t i, int32_t t, MultiStepInT<M> &in)
{
    const int64_t N = a.N;
    const int NR = CNT::nr(a), NL = CNT::nl(a), NP = CNT::np(a);
#pragma unroll
    for (int j = 0; j < M; j++) {
        if constexpr (F & F_GRID) {
            const int q = j < NR ? j : NR - 1;
            const double *g = a.c.grid_ts + (((int64_t)t * NR + q) * 4) * N + i;
            in.grid[j][0] = g[0]; in.grid[j][1] = g[N]; in.grid[j][2] = g[2 * N]; in.grid[j][3] = g[3 * N];
        }
        { const int q = j < NL ? j : NL - 1; in.load[j] = a.c.load_ts[((int64_t)t * NL + q) * N + i]; }
        { const int q = j < NP ? j : NP - 1; in.pv[j] = a.c.pv_ts[((int64_t)t * NP + q) * N + i]; }
    }
}

// One priority list [list_len, 3] = (kind, instance, action) packed into a 64-bit word, a byte per element that COUNTS: kind |
// instance << 2 | action << 5 | 0x80.  What populate_multi decides per element at walk time is decided here: padding, a kind outside
// 0..2, an instance the layout does not have and a module met before in this list (priority_list.py:82-88) are dropped, the others
// move up -- a list names every module at most once after that, so the word holds the whole list whenever the layout has at most
// PL_PACK_MAX controllable modules (any list_len).
constexpr int PL_PACK_MAX = 8;
__device__ __forceinline__ uint64_t pack_priority_list(const int32_t *__restrict__ list, int32_t list_len, int NG, int NB, int NR)
{
    uint64_t w = 0;
    uint32_t seen = 0u;
    int n = 0;
    for (int k = 0; k < list_len; k++) {
        const int kind = list[3 * k], j = list[3 * k + 1], act = list[3 * k + 2] != 0;
        bool valid = !(kind < 0 || kind > 2 || j < 0 || j >= (kind == 0 ? NG : kind == 1 ? NB : NR));
        const uint32_t bit = valid ? 1u << (kind * MGX_MAX_INSTANCES + j) : 0u;
        valid = valid && !(seen & bit) && n < PL_PACK_MAX;
        seen |= bit;
        const uint32_t el = (uint32_t)kind | ((uint32_t)j << 2) | ((uint32_t)act << 5) | 0x80u;
        w |= valid ? (uint64_t)el << (8 * n) : 0ull;
        n += valid ? 1 : 0;
    }
    return w;
}

// PriorityListAlgo._populate_action (priority_list.py:69-116) for grid i out of registers: populate_multi's walk -- every element the
// single-module form of populate_core on the load that remains when it is reached -- with the modules' parameters and state taken
// from the register copy R (the state of a K-step loop lives there, not in the batch's columns) by compile-time-unrolled selects on the
// element's instance, and the controls written into `sin` the same way.  Lanes hold different lists: an element is a battery in one
// lane and a genset in the next, so both forms run under their lanes' masks.  Returns the expansion's assert mask like populate_multi.
template <int F, class CNT, int M, bool PARK = false>
__device__ __forceinline__ uint32_t populate_multi_small(const KArgs &a, const MultiRegsT<M> &R, uint64_t plw, MultiStepInT<M> &sin,
                                                         const lds_double *pk = nullptr, bool check = true)
{
    constexpr bool parked = PARK;
    const int NG = CNT::ng(a), NB = CNT::nb(a), NR = CNT::nr(a), NL = CNT::nl(a), NP = CNT::np(a);
    using PS = ParkSlots<CNT, M>;
    // (the candidates pass through an empty asm: hipcc re-forms a chain of selects between elements of one array into ONE load at a
    //  selected address -- a dynamically indexed array, which lives in scratch memory; the same for the select-stores below)
    auto opaque = [](double x) __attribute__((always_inline)) { asm("" : "+v"(x)); return x; };
    auto pick = [&](const double (&arr)[M], int j) __attribute__((always_inline)) {
        double v = opaque(arr[0]);
#pragma unroll
        for (int q = 1; q < M; q++) v = (j == q) ? opaque(arr[q]) : v;
        return v;
    };
    auto put = [&](double (&arr)[M], int j, double e) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < M; q++) { const double old = opaque(arr[q]); arr[q] = (j == q) ? e : old; }
    };
#pragma unroll
    for (int j = 0; j < M; j++) { sin.goal[j] = 0.0; sin.gen[j] = 0.0; sin.bat[j] = 0.0; sin.grd[j] = 0.0; }
    double total_load = 0.0;                                           // _get_load: running sum (:157-164)
#pragma unroll
    for (int j = 0; j < M; j++) if (j < NL) total_load += -1 * sin.load[j];
    double renewable = 0.0;                                            // _get_renewable: np.sum of fewer than eight values (:166-167)
#pragma unroll
    for (int j = 0; j < M; j++) if (j < NP) renewable += sin.pv[j];
    double remaining = total_load - renewable;                         // :74
    uint32_t xv = !(total_load >= 0 && renewable >= 0) ? 256u : 0u;    // :73
    // the gensets' next_max / next_min production for either goal (genset_module.py:392-424) and the grids' limits, once per step
    double g_mx[M][2], g_mn[M][2], r_mx[M], r_mc[M];
#pragma unroll
    for (int j = 0; j < M; j++) {
        if constexpr (F & F_GENSET) {
#pragma unroll
            for (int act = 0; act < 2; act++) {
                const double ns = (double)genset_next_status(R.g_status[j], act);
                double rmin;
                if constexpr (parked) rmin = park_ld(pk, PS::G_RMIN + (j < NG ? j : 0)); else rmin = R.g_rmin[j];
                g_mx[j][act] = ns * R.g_rmax[j]; g_mn[j][act] = ns * rmin;
            }
        }
        if constexpr (F & F_GRID) { r_mx[j] = R.r_imp[j] * sin.grid[j][3]; r_mc[j] = R.r_exp[j] * sin.grid[j][3]; }
    }
    constexpr int LMAX = CNT::is_static ? (CNT::kNG + CNT::kNB + CNT::kNR_) : PL_PACK_MAX;
#pragma unroll
    for (int k = 0; k < (LMAX < PL_PACK_MAX ? LMAX : PL_PACK_MAX); k++) {
        const uint32_t el = (uint32_t)(plw >> (8 * k)) & 0xffu;
        const bool valid = (el & 0x80u) != 0;
        const int kind = el & 3u, j = (el >> 2) & 7u, act = (el >> 5) & 1u;
        const bool close = pl_isclose0(remaining), produce = !close && remaining > 0, consume = !close && !produce;
        double e = 0.0, mc = 0.0;
        if (valid) {
            if (kind == 1) {
                if constexpr (F & F_BATTERY) {                          // populate_core<F_BATTERY>: one division
                    double cmin, cmax, C, D, eta;
                    if constexpr (parked) {
                        const int q = j < NB ? j : 0;
                        cmin = park_ld(pk, PS::B_CMIN + q); cmax = park_ld(pk, PS::B_CMAX + q); C = park_ld(pk, PS::B_C + q);
                        D = park_ld(pk, PS::B_D + q); eta = park_ld(pk, PS::B_ETA + q);
                    } else {
                        cmin = pick(R.b_cmin, j); cmax = pick(R.b_cmax, j); C = pick(R.b_C, j); D = pick(R.b_D, j); eta = pick(R.b_eta, j);
                    }
                    const double charge = pick(R.b_charge, j);
                    Params p; p.bat_cmin = cmin; p.bat_D = D; p.bat_eta = eta;
                    const double mp = battery_max_production(p, charge);                  // battery_module.py:283-286
                    const double room = cmax - charge;
                    const double num_sink = py_min(C, room);
                    const double e_src = py_clip(remaining, 0.0, mp);
                    const double num = (close || produce) ? -1.0 * (produce ? e_src : 0.0) : num_sink;
                    const double q = num / eta;
                    const double e_snk = py_max(remaining, -1.0 * q);
                    e = close ? 0.0 : (produce ? e_src : e_snk);
                    mc = q;
                    put(sin.bat, j, e);
                }
            } else if (kind == 0) {
                if constexpr (F & F_GENSET) {
                    double mn = g_mn[0][0], mx = g_mx[0][0];
#pragma unroll
                    for (int q = 0; q < M; q++)
#pragma unroll
                        for (int c = 0; c < 2; c++) { const bool hit = (j == q) && (act == c); mn = hit ? opaque(g_mn[q][c]) : mn; mx = hit ? opaque(g_mx[q][c]) : mx; }
                    e = pl_energy(remaining, mn, mx, 0.0, false);
                    put(sin.goal, j, (double)act); put(sin.gen, j, e);
                }
            } else {
                if constexpr (F & F_GRID) {
                    const double mx = pick(r_mx, j);
                    mc = pick(r_mc, j);
                    e = pl_energy(remaining, 0.0, mx, mc, true);
                    put(sin.grd, j, e);
                }
            }
            if (check) {                                                         // (wave-uniform: the mask lands in the log's violations column only)
                uint32_t bad = (produce && !(e >= 0)) ? 128u : 0u;               // priority_list.py:154
                bad = (consume && !(remaining <= 0.0)) ? 256u : bad;             // :121 (NaN)
                bad = (consume && remaining <= 0.0 && kind != 0 && !(mc >= 0)) ? 64u : bad;     // :124 (sinks only)
                xv = xv ? xv : bad;                                              // the first assert that fails stops the reference
            }
            remaining -= e;                                                      // :105
        }
    }
    return xv;
}

// PriorityListAlgo._populate_action over module instances (priority_list.py:69-116): `list` holds list_len elements
// (kind, instance, action), kind < 0 = padding.  Every element runs the single-module form of populate_core on the load
// that remains when it is reached; control [A] = (goal, energy) per genset, batteries, grids.
// Returns the assert mask of the expansion (populate_core<F, true>: bits 6-8, the first failing assert of the walk).
template <int F>
__device__ inline uint32_t populate_multi(const KArgs &a, const int32_t *__restrict__ list, int32_t list_len, int64_t i, int32_t t,
                                          double *__restrict__ control)
{
    const int64_t N = a.N;
    const int NG = a.n_genset, NB = a.n_battery, NR = a.n_grid;
    const int A = 2 * NG + NB + NR;
    for (int k = 0; k < A; k++) control[k] = 0.0;
    double total_load = 0.0;                                           // _get_load: running sum (:157-164)
    for (int j = 0; j < a.n_load; j++) total_load += -1 * a.c.load_ts[((int64_t)t * a.n_load + j) * N + i];
    double renewable;                                                  // _get_renewable: np.sum (:166-167)
    {
        const double *pv = a.c.pv_ts + ((int64_t)t * a.n_pv) * N + i;
        renewable = np_sum_strided(pv, (int)N, a.n_pv);
    }
    double remaining = total_load - renewable;                         // :74
    uint32_t seen = 0u;
    uint32_t xv = !(total_load >= 0 && renewable >= 0) ? 256u : 0u;    // :73
    for (int k = 0; k < list_len; k++) {
        const int kind = list[3 * k], j = list[3 * k + 1], act = list[3 * k + 2] != 0;
        // padding, or an element that names a module the layout does not have (the lists live in device memory and cannot
        // be checked by the host: such an element is skipped instead of indexing past the columns)
        if (kind < 0 || kind > 2 || j < 0 || j >= (kind == 0 ? NG : kind == 1 ? NB : NR)) continue;
        const uint32_t bit = 1u << (kind * MGX_MAX_INSTANCES + j);
        if (seen & bit) continue;                                      // :82-88: a module met again is skipped
        seen |= bit;
        const int64_t c = (int64_t)j * N + i;
        const uint32_t word = (uint32_t)kind | ((uint32_t)act << 2) | 8u;
        Params p; State s; Inputs in; double q_unused;
        s.charge = 0.0; s.soc = 0.0; s.status = 0u; in.g_stat = 1.0;
        double e = 0.0;
        uint32_t xe = 0u;
        if (kind == 0) {
            if constexpr (F & F_GENSET) {
                load_module_params<F_GENSET>(a.c, c, p); s.status = a.c.gen_status[c];
                populate_core<F_GENSET, true>(p, s, word, in, q_unused, remaining, 0.0, false, &xe, false);
                control[2 * j] = in.a_goal; control[2 * j + 1] = in.a_gen; e = in.a_gen;
            }
        } else if (kind == 1) {
            if constexpr (F & F_BATTERY) {
                load_module_params<F_BATTERY>(a.c, c, p); s.charge = a.c.charge[c];
                populate_core<F_BATTERY, true>(p, s, word, in, q_unused, remaining, 0.0, false, &xe, false);
                control[2 * NG + j] = in.a_bat; e = in.a_bat;
            }
        } else {
            if constexpr (F & F_GRID) {
                load_module_params<F_GRID>(a.c, c, p);
                in.g_stat = a.c.grid_ts[(((int64_t)t * NR + j) * 4 + 3) * N + i];
                populate_core<F_GRID, true>(p, s, word, in, q_unused, remaining, 0.0, false, &xe, false);
                control[2 * NG + NB + j] = in.a_grid; e = in.a_grid;
            }
        }
        xv = xv ? xv : xe;                                             // the first assert that fails stops the reference
        remaining -= e;                                                // :105
    }
    return xv;
}

}  // namespace mgx
