/*
 * mgx.h -- C ABI of the MI355X-native batched microgrid-step engine (libmgx.so).
 *
 * The reference (Total-RD/pymgrid v1.2.2) is pure Python and has NO FFI: its boundary for this hot path is
 * the Gym surface  DiscreteMicrogridEnv.step / BaseMicrogridEnv.step / reset
 * (src/pymgrid/envs/discrete/discrete.py:109-143, src/pymgrid/envs/base/base.py:165-209) over
 * Microgrid.run / Microgrid.reset (src/pymgrid/microgrid/microgrid.py:227-325, :205-219).
 * This header is what a binding for that path would bind (INTEGRATION.md shows the ctypes stub): every
 * entry point below names the reference interface it replaces.  Plain C types only -- no torch, no HIP
 * types in the signatures (a stream is passed as void*, it is a hipStream_t).
 *
 * Conventions
 *  - N microgrids ("grids") advance in lock-step; all share the step counter t, the series length T and
 *    the episode window [initial_step, final_step) -- unless the episode was started with mgx_reset_windows, which
 *    gives every grid its own start row and episode length (its own trajectory) behind the shared counter.
 *  - Every pointer in mgx_columns and every data argument is a DEVICE pointer owned by the caller
 *    (e.g. torch tensors); the library allocates only a small scratch buffer in mgx_create.
 *  - All arithmetic is IEEE fp64, unfused (no FMA contraction), in the reference's operation order.
 *  - Zero signs: every output equals the reference's VALUE bit for bit, except that a result that is zero may carry the other
 *    sign (+0.0 vs -0.0).  The reference produces -0.0 in places (-1.0 * get_cost(0.0); a flex source asked for `-difference`
 *    with difference == 0.0, microgrid.py:300-314), its min / max return the first argument on ties where v_min_f64 /
 *    v_max_f64 order -0 < +0; no operation of the path divides by such a zero or takes its sign, so the difference cannot
 *    propagate into a non-zero value.  Parity tests compare with `==` (profiles/r03/zero_signs.txt has the census).
 *  - Calls are asynchronous on the given stream and must be issued from one host thread per handle.
 *  - Return value: MGX_OK or an error code; mgx_last_error() gives a thread-local message.
 *
 * Column layouts (struct-of-arrays, grid index fastest):
 *    parameter / state columns   [N]
 *    load_ts, pv_ts              [T, N]     sign as the reference STORES it: load <= 0, pv >= 0
 *                                           (base_timeseries_module.py:68-79); with several load / renewable
 *                                           modules per grid: [T, n_load, N] / [T, n_pv, N], bounds [n, N]
 *    grid_ts                     [T, 4, N]  components import_price, export_price, co2_per_kwh, grid_status
 *                                           (grid_module.py:70)
 *    actions                     [N, A]     A = 2*has_genset + has_battery + has_grid, order
 *                                           genset(goal_status, energy), battery, grid = the reference's
 *                                           controllable sweep order (module_container.py:355-413)
 *    obs                         [N, D]     D = mgx_obs_dim(); order load(1+H), pv(1+H), genset(4), battery(2),
 *                                           grid(4*(1+H), component-minor) -- or, with mgx_layout.flat_order = MGX_FLAT_GYM,
 *                                           battery, genset, grid, load, pv
 *    log                         [L, N]     L = mgx_log_dim(); names from mgx_log_name()
 */
#ifndef MGX_H
#define MGX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MGX_ABI_VERSION 9    /* frozen: struct layouts and the meaning of every v9 entry point do not change any more */
#define MGX_ABI_MINOR 2      /* additions only: 1 = mgx_abi_minor, mgx_set_tunable / mgx_get_tunable, mgx_set_launch_threads,
                              * mgx_action_bounds; 2 = mgx_step_lists */

enum mgx_status {
    MGX_OK = 0,
    MGX_ERR_INVALID = 1,      /* bad argument / inconsistent layout */
    MGX_ERR_UNSUPPORTED = 2,  /* valid in the reference, not offered on device */
    MGX_ERR_RANGE = 3,        /* step counter would leave the time series (IndexError in the reference) */
    MGX_ERR_DEVICE = 4        /* HIP runtime error (message holds hipGetErrorString) */
};

/* element type of the observation rows (mgx_set_obs_format) */
enum mgx_obs_format { MGX_OBS_F64 = 0, MGX_OBS_F32 = 1 };
/* element type of the continuous `actions` of mgx_step / mgx_step_k (mgx_set_action_format) */
enum mgx_action_format { MGX_ACT_F64 = 0, MGX_ACT_F32 = 1 };
/* what the `obs` argument of mgx_step / mgx_step_discrete / mgx_observe / mgx_reset receives (mgx_set_obs_mode) */
enum mgx_obs_rows { MGX_OBS_ROWS_FULL = 0, MGX_OBS_ROWS_STATE_ONLY = 1, MGX_OBS_ROWS_STATE_COMPACT = 2 };
/* what the `done` argument of the fused calls (mgx_step_k, mgx_rollout_discrete) receives (mgx_set_done_format) */
enum mgx_done_format { MGX_DONE_U8 = 0, MGX_DONE_BITS = 1 };

/* Reward shaping functions of the reference (microgrid/reward_shaping/): what step() RETURNS as reward.
 * The log's "reward" column always keeps the unshaped sum (the balance log's `reward` vs `shaped_reward`). */
enum mgx_reward_shaper {
    MGX_SHAPER_NONE = 0,
    MGX_SHAPER_PV_CURTAILMENT = 1,     /* PVCurtailmentShaper:    -curtailment                       */
    MGX_SHAPER_BATTERY_DISCHARGE = 2   /* BatteryDischargeShaper: (discharge - loss_load) / load     */
};

/* Bits of a violations mask (mgx_check_step, mgx_check_discrete, the `violations` outputs of mgx_expand_discrete /
 * mgx_expand_lists, the last log column).  Bits 0-2: requests the reference refuses only with raise_errors=True
 * (base_module.py:79-93,213-224,265-270).  Bits 3-8: states in which the reference ALWAYS gives up with an AssertionError /
 * TypeError -- the device goes on with the clipped value (unspecified there); the mask is how a caller finds out. */
enum mgx_violation_bit {
    MGX_V_GENSET_RANGE = 1,        /* genset request outside [min_production, max_production] */
    MGX_V_BATTERY_LIMIT = 2,       /* battery request above max_production / max_consumption */
    MGX_V_GRID_LIMIT = 4,          /* grid request above max import / export */
    MGX_V_GENSET_GOAL = 8,         /* genset goal_status outside [0, 1] (`assert`, genset_module.py:146-147) */
    MGX_V_GENSET_NEGATIVE = 16,    /* negative genset energy: a pure source asked to absorb */
    MGX_V_NEGATIVE_LIMIT = 32,     /* a battery / grid acting at a NEGATIVE limit: asked to absorb while its max_consumption is
                                    * negative -- `assert absorbed_energy >= 0` (base_module.py:272), a lossy battery whose charge
                                    * was rounded one ulp above max_capacity -- or to produce (even 0.0) while its max_production
                                    * is negative -- `assert internal_energy_change <= 0` (battery_module.py:114), a battery
                                    * below min_capacity */
    MGX_V_EXPAND_CONSUME = 64,     /* discrete expansion: `assert module_max_consumption >= 0` (priority_list.py:124), the same
                                    * battery state met by _consume_in_module with load left to absorb */
    MGX_V_EXPAND_PRODUCE = 128,    /* discrete expansion: `assert module_production >= 0` (priority_list.py:154): a module whose
                                    * max_production is negative (a battery below min_capacity) is asked to produce */
    MGX_V_EXPAND_SIGN = 256        /* discrete expansion: `assert total_load >= 0 and renewable >= 0` / `assert remaining_load
                                    * <= 0.0` (priority_list.py:73,121): series of the wrong sign, NaN */
};
/* The expansion bits are exclusive: the reference stops at the first assert that fails on its way down the priority list,
 * and the mask names that one. */

typedef struct mgx_handle mgx_handle;
typedef void *mgx_stream;     /* hipStream_t */

/* Static shape of the batch.  Replaces the module list handed to Microgrid.__init__ (microgrid.py:100-128). */
typedef struct mgx_layout {
    int32_t struct_size;      /* = sizeof(mgx_layout), ABI guard */
    int32_t n_grids;          /* N */
    int32_t n_steps;          /* T, rows of every time series */
    int32_t horizon;          /* forecast_horizon H of the (oracle) forecaster, 0 = no forecast */
    int32_t initial_step;     /* base_module.py:40 */
    int32_t final_step;       /* resolved as base_timeseries_module.py:317-330 does: <=0 means T */
    int32_t has_genset;       /* 0/1 GensetModule  */
    int32_t has_battery;      /* 0/1 BatteryModule */
    int32_t has_grid;         /* 0/1 GridModule    */
    int32_t n_load;           /* LoadModule count      (1 on the fast path; 0..16 via the general kernels) */
    int32_t n_pv;             /* RenewableModule count (1 on the fast path; 0..16 via the general kernels) */
    int32_t grid_before_battery;   /* 0/1: the GridModule precedes the BatteryModule in the microgrid's module list and is
                                    * therefore stepped / summed first (module_container.py:355-413; the controllable
                                    * sweep is pure sources, then sources-and-sinks in list order).  Changes the last
                                    * bit of the balance sums, not the action / log / observation column order. */
    /* Module multiplicities (the container holds a LIST of modules per name, module_container.py:355-413, and
     * Microgrid.run sweeps every list in order, microgrid.py:262-275): 0 = "as has_*" (0 or 1).  With n > 1 every
     * column of that module kind is [n, N] (instance-major), grid_ts is [T, n_grid, 4, N], grid_lo / grid_hi
     * [n_grid, 4, N]; actions are (goal, energy) per genset, then the batteries, then the grids; the log carries one
     * block per instance; observations one group of state columns / one grid window per instance.  Such layouts run on
     * the general kernels (as do n_load / n_pv != 1): single steps, K-step launches (mgx_step_k, mgx_rollout_lists),
     * observations, priority lists (mgx_expand_lists). */
    int32_t n_genset, n_battery, n_grid;      /* <= MGX_MAX_INSTANCES */
    /* Order of the module blocks inside a flat observation row (envs/base/base.py:128-163,211-223: the reference flattens
     * a gym.spaces.Dict built from a plain dict, and gym sorts the keys of such a Dict):
     *   MGX_FLAT_MODULE (0)  load, pv, genset, battery, grid   (this library's native order)
     *   MGX_FLAT_GYM    (1)  battery, genset, grid, load, pv   (alphabetical by module name: what a policy trained against the
     *                        reference with gym <= 0.26 / gymnasium sees)
     * Only the column BASES of the blocks move (every kernel writes a block at its base): no cost.  One module of every kind
     * per grid only. */
    int32_t flat_order;
} mgx_layout;

enum mgx_flat_order { MGX_FLAT_MODULE = 0, MGX_FLAT_GYM = 1 };

#define MGX_MAX_INSTANCES 8

/* Device columns.  Pointers for absent modules may be NULL. */
/* Batch-UNIFORM parameter columns (mgx_columns.uniform_mask): bit b set = the parameter column b below holds ONE value that
 * every grid shares (the pointer addresses a single element; kernels index it with 0).  MicrogridGenerator gives every microgrid
 * the same battery efficiency / cycle cost, genset cost / co2 figures, unbalanced-energy costs and -- unless drawn -- genset
 * timers (convert/get_module.py:39-97, MicrogridGenerator.py:472): 60 of the 108 parameter bytes a single step reads per grid.
 * Values are the same, the loads become same-address (broadcast) loads out of the caches.  One module of every kind only. */
enum mgx_uniform_bit {
    MGX_U_BAT_MIN_CAPACITY = 0, MGX_U_BAT_MAX_CAPACITY, MGX_U_BAT_MAX_CHARGE, MGX_U_BAT_MAX_DISCHARGE, MGX_U_BAT_EFFICIENCY,
    MGX_U_BAT_COST_CYCLE, MGX_U_GEN_RUNNING_MIN, MGX_U_GEN_RUNNING_MAX, MGX_U_GEN_COST, MGX_U_GEN_CO2_PER_UNIT,
    MGX_U_GEN_COST_PER_UNIT_CO2, MGX_U_GEN_TIMES, MGX_U_GRID_MAX_IMPORT, MGX_U_GRID_MAX_EXPORT, MGX_U_GRID_COST_PER_UNIT_CO2,
    MGX_U_LOSS_LOAD_COST, MGX_U_OVERGENERATION_COST
};

typedef struct mgx_columns {
    int32_t struct_size;      /* = sizeof(mgx_columns) */
    uint32_t uniform_mask;    /* bits of enum mgx_uniform_bit; 0 = every column is [N] */
    /* BatteryModule parameters (battery_module.py:66-93) */
    const double *bat_min_capacity, *bat_max_capacity, *bat_max_charge, *bat_max_discharge;
    const double *bat_efficiency, *bat_cost_cycle;
    /* GensetModule parameters (genset_module.py:61-98) */
    const double *gen_running_min, *gen_running_max, *gen_cost, *gen_co2_per_unit, *gen_cost_per_unit_co2;
    const uint32_t *gen_times;            /* start_up_time | (!allow_abortion) << 8 | wind_down_time << 16 (times <= 255) */
    /* GridModule parameters (grid_module.py:72-101) */
    const double *grid_max_import, *grid_max_export, *grid_cost_per_unit_co2;
    /* UnbalancedEnergyModule parameters (unbalanced_energy_module.py:11-26) */
    const double *loss_load_cost, *overgeneration_cost;
    /* time series */
    const double *load_ts, *pv_ts, *grid_ts;
    /* observation bounds (needed only when observations are requested):
     * base_timeseries_module.py:81-88 (load/pv: min(ts.min(),0), max(ts.max(),0)), grid_module.py:125-132 */
    const double *load_lo, *load_hi, *pv_lo, *pv_hi;      /* [N] */
    const double *grid_lo, *grid_hi;                      /* [4, N] */
    /* GaussianNoiseForecaster (forecast/forecaster.py:220-275): per-grid noise standard deviation of each module's
     * forecast, already scaled by |mean(series)| when relative_noise is set; NULL = OracleForecaster */
    const double *load_noise_std, *pv_noise_std, *grid_noise_std;   /* [N] */
    /* dynamic state, read and written by every step */
    double *charge, *soc;                 /* BatteryModule._current_charge / _soc */
    uint32_t *gen_status;                 /* current | goal<<8 | steps_until_up<<16 | steps_until_down<<24 */
    /* FACTORISED series (optional; base_load != NULL switches it on).  Every series MicrogridGenerator builds is one of a
     * few base profiles times a per-grid scalar (MicrogridGenerator.py:137-147 _scale_ts: ts * (size / ts.max()); :205-212
     * co2 profile verbatim; :253-285 import tariff by hour of day; :321-340 weak-grid outages), so a generated batch is
     * fully described by the base tables + a profile id and a ratio per grid.  In this mode load_ts / pv_ts / grid_ts may be
     * NULL and the kernels FORM the series values as they go, with the same single multiply the generator performs:
     *     load[t, i] = -|base_load[t, load_profile[i]] * load_ratio[i]|     (sign as stored, base_timeseries_module.py:68-79)
     *     pv[t, i]   =  |base_pv[t, pv_profile[i]] * pv_ratio[i]|
     *     grid[t, :, i] = (tariff price of hour t % 24 under pattern tariff[i] (1 or 2), 0, base_co2[t, co2_profile[i]],
     *                      1 - bit t of grid i's outage words)
     * -- bit-identical to the arrays mgx_synthesize_series writes.  The [T, N] series (16 of the 57 B a fused env-step of a
     * Template-4 grid streams; 14 GB per 100 000 grid-years) then never exist; the base tables (<= 560 KB each) stay in
     * the caches.  Base tables are [n_steps, MGX_PROFILE_PITCH] doubles (one 64-byte row per step, unused columns
     * arbitrary), profile ids < MGX_PROFILE_PITCH and tariff in {0, 1, 2} -- a PRECONDITION the library cannot check (device
     * pointers): an id beyond the pitch reads a neighbouring row / table.  outage_bits: [ceil(n_steps / 64), N] words, bit (t & 63) of word
     * t >> 6 set = grid_status 0 at row t; NULL = no outages.  Offered with one module of every kind per grid (the general
     * kernels read materialised series only). */
    const double *base_load, *base_pv, *base_co2;
    const uint8_t *load_profile, *pv_profile, *co2_profile, *tariff;      /* [N] */
    const double *load_ratio, *pv_ratio;                                   /* [N] */
    const uint64_t *outage_bits;
} mgx_columns;

#define MGX_PROFILE_PITCH 8

/* ---- lifetime ------------------------------------------------------------------------------------- */
int mgx_abi_version(void);
int mgx_abi_minor(void);

/* Process-wide launch-shape knobs.  The library reads nothing from the environment: this is the only way to change what the
 * measurements behind DESIGN.md chose, and a consumer can read back what a process runs with.  They change HOW a call is
 * launched (never a value it computes) and take effect for the calls / handles that follow. */
enum mgx_tunable {
    MGX_TUNE_WIN_THREADS = 0,      /* threads of a ring-refill workgroup: 0 = automatic, 256 / 512 / 1024 */
    MGX_TUNE_WIN_GROUP = 1,        /* grids per refill workgroup: 0 = automatic (16 / 32), else a power of two <= 64 */
    MGX_TUNE_WIN_PAIRS = 2,        /* column-major refill stores carry a pair of adjacent grids per lane: -1 = automatic, 0, 1 */
    MGX_TUNE_WIN_MIN_LDS = 3,      /* LDS bytes a refill ahead of the counter asks for at least (occupancy cap): -1 = automatic */
    MGX_TUNE_PREFETCH_POOL = 4,    /* 1: one prefetch stream per device shared by all handles (default 0: one per handle) */
    MGX_TUNE_MULTI_GENERIC = 5,    /* 1: handles CREATED afterwards run every general layout on the run-time-count kernels */
    MGX_TUNE_MULTI_SMALL_OWN = 6,  /* 0: mgx_step_k of small general layouts through the shared kernel (default 1: its own) */
    MGX_TUNE_GRID_MAJOR_COPY = 7,  /* 0: mgx_reset_episodes on [T, N] series gathers rows instead of making its grid-major copy */
    MGX_TUNE_FLEET_BYVALUE = 8,    /* 0: mgx_fleet_step launches the pointer form of the fleet kernel (default 1: by value) */
    MGX_TUNE_LAUNCH_THREADS = 9,   /* mode (0 / 1 / 2) handles created afterwards start with: mgx_set_launch_threads (default 1) */
    MGX_TUNE_MULTI_STATIC = 10,    /* 0: mgx_step_k of small general layouts never takes a compile-time-count specialisation (A/B, tests) */
    MGX_TUNE_COUNT_ = 11
};
int mgx_set_tunable(int32_t id, int64_t value);                                /* MGX_ERR_INVALID: unknown id / value out of range */
int mgx_get_tunable(int32_t id, int64_t *value, int64_t *default_value);       /* either pointer may be NULL */
/* thread-local text of the last error returned on this thread ("" if none) */
const char *mgx_last_error(void);

/* Microgrid.__init__ (microgrid.py:100-128): bind a layout + columns.  The step counter starts at
 * layout->initial_step. */
int mgx_create(const mgx_layout *layout, const mgx_columns *columns, mgx_handle **out);
void mgx_destroy(mgx_handle *h);

/* ---- shape queries -------------------------------------------------------------------------------- */
int32_t mgx_action_dim(const mgx_handle *h);
int32_t mgx_obs_dim(const mgx_handle *h);
int32_t mgx_log_dim(const mgx_handle *h);
const char *mgx_log_name(const mgx_handle *h, int32_t column);   /* NULL if out of range */
int32_t mgx_current_step(const mgx_handle *h);                   /* BaseMicrogridModule.current_step */

/* Keep the step counter in DEVICE memory (enable != 0) so that a sequence of mgx_step / mgx_step_discrete / mgx_step_k /
 * mgx_rollout_discrete / mgx_observe calls can be captured in a hipGraph (stream capture) and replayed: kernels read
 * the counter, the last workgroup of every stepping kernel to finish advances it.  While enabled, launch-time range checks are
 * replaced by an in-kernel clamp + sticky overrun flag (reported as MGX_ERR_RANGE when the mode is switched off), and
 * mgx_current_step() performs a blocking device read (after synchronising the stream of the last call that moved the
 * counter).  Disabling copies the counter back to the host. */
int mgx_use_device_counter(mgx_handle *h, int enable, mgx_stream stream);

/* ---- the hot path --------------------------------------------------------------------------------- */
/* Microgrid.reset (microgrid.py:205-219) -> BaseMicrogridModule.reset (base_module.py:65-77): the step counter
 * returns to initial_step (or to `initial_step` if >= 0: the trajectory_func hook, microgrid.py:221-225);
 * battery charge and genset status are NOT touched (the reference does not restore them).
 * obs [N, D] may be NULL. */
int mgx_reset(mgx_handle *h, int32_t initial_step, void *obs, mgx_stream stream);

/* Episode window [initial_step, final_step) for the next reset -- what a trajectory_func returns
 * (microgrid.py:221-225, microgrid/trajectory/ classes; validated like _check_trajectory_func, microgrid.py:181-203:
 * inside the window given at create, initial < final).  Does not move the step counter; call mgx_reset next. */
int mgx_set_window(mgx_handle *h, int32_t initial_step, int32_t final_step);

/* Per-grid episodes.  In the reference every Microgrid owns its step counter and draws its own trajectory at reset
 * (Microgrid.reset -> _set_trajectory, microgrid.py:205-225; BaseMicrogridModule.reset, base_module.py:65-77,292-296;
 * FixedLengthStochasticTrajectory / StochasticTrajectory, microgrid/trajectory/stochastic.py:9-30).  Batched form:
 * grid i starts at series row start[i] and its episode lasts length[i] steps (NULL: max_length for every grid), i.e.
 * it reports done from counter value length[i] - 1 on (base_timeseries_module.py:124-125 with its own final_step =
 * start[i] + length[i]).  A HIP kernel gathers rows [start[i], start[i] + max_length + horizon] of every grid's series
 * into the caller's window buffers load_w / pv_w [R, N], grid_w [R, 4, N] (R = max_length + horizon + 1; rows beyond
 * the end of a series hold the forecaster's padding value (lo + hi) / 2, forecaster.py:95,120-137), and the handle then
 * steps over those buffers from counter 0: mgx_current_step() = steps since the reset, grid i's own current_step =
 * start[i] + mgx_current_step().  start / length are DEVICE arrays [N]; a start outside the window given at create is
 * clamped into it and a length is cut at the env's final step (the reference raises ValueError in
 * _check_trajectory_func, microgrid.py:181-203: validate on the host if the arrays come from outside).  final_rel [N]
 * (device, int32; required with `length`) receives the episode lengths actually used and must stay alive until the
 * next reset.  Observation bounds stay those of the full series.  Stepping past the longest episode (counter >= max_length)
 * is MGX_ERR_RANGE, like stepping past the end of a series.  mgx_reset() returns to the full series.
 * Layouts with several modules of a kind per grid (series [T, n, N]): load_w [R, n_load, N], pv_w [R, n_pv, N], grid_w
 * [R, n_grid, 4, N]; rings (mgx_observe_windows[_ahead]) work over them as over the full series.
 * MGX_ERR_UNSUPPORTED in device-counter mode, while stepping in shards, with column-major ring blocks. */
int mgx_reset_windows(mgx_handle *h, const int32_t *start, const int32_t *length, int32_t max_length,
                      double *load_w, double *pv_w, double *grid_w, int32_t *final_rel, void *obs, mgx_stream stream);

/* Rolling per-grid windows: N reference microgrids reset one by one, each when ITS episode ends (what a vectorised
 * Gym env does with auto-reset; microgrid.py:205-225 per microgrid, at different times).  As mgx_reset_windows, but the
 * window buffers [ring_rows, N] ([ring_rows, 4, N]) are rings addressed by (step counter & (ring_rows - 1)):
 * ring_rows a power of two >= max_length + horizon + 1.  The shared counter restarts at 0 and never ends; final_abs [N]
 * (device, required) receives the counter value at which each grid's episode has run its length, done_i =
 * counter >= final_abs[i] - 1.  Single steps only (mgx_step, mgx_step_discrete, mgx_step_many, mgx_observe,
 * mgx_observe_windows[_ahead] + mgx_patch_windows, ...); fused launches, shards and the device counter are refused in this
 * mode.  mgx_reset / mgx_reset_windows
 * leave it. */
int mgx_reset_windows_rolling(mgx_handle *h, const int32_t *start, const int32_t *length, int32_t max_length,
                              int32_t ring_rows, double *load_w, double *pv_w, double *grid_w, int32_t *final_abs, void *obs,
                              mgx_stream stream);
/* Restart the grids with mask[i] != 0 at the current counter value: their rows start[i] .. of the full series are
 * gathered into the rings from the current row on, final_abs[i] = counter + length[i] (length NULL: max_length).  Dynamic
 * state is untouched, as in Microgrid.reset.  mask / start / length: device arrays [N]. */
int mgx_reset_grids(mgx_handle *h, const uint8_t *mask, const int32_t *start, const int32_t *length, mgx_stream stream);
/* The same with the new episodes DRAWN ON DEVICE, one draw per restarted grid as its own trajectory_func would make it:
 * fixed_length > 0 = FixedLengthStochasticTrajectory(fixed_length) (start = randint(initial, final - length)),
 * fixed_length == 0 = StochasticTrajectory (start = randint(initial, final - 2), final = randint(start, final));
 * microgrid/trajectory/stochastic.py:9-30.  Uniforms: Philox4x32-10 of (seed; grid, 2 * counter [+ 1]) -- reproducible and
 * independent of how often or in which order grids restart.  start_io / length_io / t0_io (device [N], each may be NULL)
 * receive, for the restarted grids only, the drawn start row, the episode length and the counter value of the restart.
 * Both restart calls also serve in-place episodes (mgx_reset_episodes below): no rows are gathered then. */
int mgx_reset_grids_random(mgx_handle *h, const uint8_t *mask, uint64_t seed, int32_t fixed_length, int32_t *start_io,
                           int32_t *length_io, int32_t *t0_io, mgx_stream stream);

/* Rolling per-grid episodes IN PLACE.  Same model as mgx_reset_windows_rolling (one shared counter that restarts at 0 and
 * never ends, every grid with its own episode: microgrid.py:205-225 per microgrid), but nothing is copied: grid i simply reads
 * row counter + row_off[i] of its own series and reports done_i = counter >= final_abs[i] - 1.  With factorised series
 * (mgx_columns.base_load ...) the rows come out of small cached base tables; for [T, N] arrays the handle makes a grid-major
 * copy per call ([N, T, 2 or 6]: as much device memory again; a grid's own row is then one 16- or 48-byte read) -- if that
 * allocation is refused the lanes gather out of the [T, N] arrays instead (same values, slower).  row_off / final_abs: caller-owned
 * DEVICE arrays [N] that the handle keeps reading AND writing until the next mgx_reset*; start / length as for
 * mgx_reset_windows (length NULL: max_length).  (Re)starts -- mgx_reset_grids, mgx_reset_grids_random -- then rewrite two
 * words per grid instead of gathering rows.  Observation windows reach beyond an episode's end into the grid's series and
 * beyond the series' end into the forecaster's padding, exactly as in lock-step.  Single steps only (as for rolling windows);
 * observation rings work as they do there (mgx_observe_windows[_ahead] + mgx_patch_windows for the grids that restarted).
 * A grid whose episode is over and that has not been restarted (auto-reset off, or a late mgx_reset_grids) keeps stepping on
 * its own series; the shared counter has no end in this mode, so nothing refuses the step that would leave the series:
 * from row n_steps on such a grid re-reads its LAST row (the step stays defined and finite; its observation windows show the
 * forecaster's padding, as they do at the end of a series in lock-step).
 * Layouts with several modules of a kind (ABI minor 1): single steps of the general kernels, every grid reading its own rows of the
 * [T, n, N] series (a per-lane gather; no grid-major copy), restarts and mgx_set_auto_reset as above; observation rows per step
 * only (MGX_OBS_ROWS_FULL: the ring patches are single-instance) and no mgx_set_final_obs. */
int mgx_reset_episodes(mgx_handle *h, const int32_t *start, const int32_t *length, int32_t max_length, int32_t *row_off,
                       int32_t *final_abs, void *obs, mgx_stream stream);
/* Auto-reset inside the step (in-place episodes only): every single step restarts the grids whose episode it ends -- the
 * draw mgx_reset_grids_random(mask = done, seed, fixed_length) would make at the counter value after the step, made by the
 * step kernel itself -- so that the step's `obs` is, for those grids, the first observation of their new episode (what a
 * vectorised Gym env with auto-reset returns) and no further launch is needed.  start_io / length_io / t0_io as for
 * mgx_reset_grids_random (device [N], each may be NULL; kept until the mode is switched off or the next mgx_reset*).
 * enable = 0 switches it off. */
int mgx_set_auto_reset(mgx_handle *h, int32_t enable, uint64_t seed, int32_t fixed_length, int32_t *start_io, int32_t *length_io,
                       int32_t *t0_io);
/* Where the observation BEFORE a restart goes ("final_observation" of a vectorised Gym env): device [N, D] rows in the handle's
 * observation format.  Rows written per step (no ring): the following single steps write every grid's pre-restart row (grids
 * that do not restart get the same row as `obs`).  Prefetched rings (MGX_OBS_ROWS_STATE_ONLY): mgx_patch_windows saves the
 * row of block first_block of every MASKED grid here before it rewrites it; the other rows are left as they are.
 * NULL (default) = off.  In-place episodes only. */
int mgx_set_final_obs(mgx_handle *h, void *final_obs);

/* Microgrid.reward_shaping_func (microgrid.py:105,130): one of enum mgx_reward_shaper. */
int mgx_set_reward_shaper(mgx_handle *h, int32_t shaper);

/* GaussianNoiseForecaster switches: Philox seed of the forecast noise and `increase_uncertainty`
 * (std_j = std * (1 + log(1 + j)) for forecast_j, forecaster.py:243-249).  Noise is drawn per (grid, module
 * component, step, horizon index); statistical -- not bit -- parity with the reference's np.random.normal.
 * With several modules of a kind the *_noise_std columns are [n, N]: one std per module instance, independent streams. */
int mgx_set_forecast_noise(mgx_handle *h, uint64_t seed, int increase_uncertainty);

/* Element type of every `obs` argument below: MGX_OBS_F64 (default; the reference's float64 arrays) or MGX_OBS_F32
 * = the same value rounded to nearest float on the way out (what an RL policy consumes after gym's / torch's cast,
 * envs/base/base.py:211-223: half the bytes of the largest output of the step).  The pointers are `void *` for that
 * reason: [N, D] doubles or [N, D] floats. */
int mgx_set_obs_format(mgx_handle *h, int32_t format);

/* Element type of the continuous controls passed to mgx_step / mgx_step_k: MGX_ACT_F64 (default) or MGX_ACT_F32 -- a
 * policy network emits floats; they are widened to double exactly and every operation after that is the float64 one
 * (= the reference fed `actions.astype(float64)`).  Halves the action stream of the fused kernel (24 -> 12 of 57 B). */
int mgx_set_action_format(mgx_handle *h, int32_t format);

/* Window prefetch.  The time-series windows of an observation (load / pv / grid: current value + forecast, the bulk of
 * the row; base_timeseries_module.py:103-140, forecaster.py:120-149) depend on the series only, never on the actions, and
 * consecutive steps share all but one of their rows.  mgx_observe_windows writes the observation rows of the NEXT K
 * steps -- ring [K, N, D], block k = the row the reference returns at step counter t + k -- reading and normalising
 * every series value once instead of 1 + horizon times.  Block 0 is complete (state columns of the current state); in
 * blocks 1..K-1 the genset / battery state columns are zero: with mgx_set_obs_mode(MGX_OBS_ROWS_STATE_ONLY) the `obs`
 * argument of mgx_step / mgx_step_discrete (pass ring + k*N*D for the step that reaches counter t + k) receives just
 * those columns.  Values are identical to mgx_observe's.  MGX_ERR_UNSUPPORTED with forecast noise (it depends on the
 * (step, horizon index) pair).  Microgrids with several modules of a kind (module_container.py:355-413) are served by the
 * general form of the same kernel -- a window per load / renewable / grid module instance, 4 state columns per genset and
 * 2 per battery -- for lock-step episodes, row- or column-major blocks (tests/test_multi_windows.py: rows == the per-step rows). */
int mgx_observe_windows(mgx_handle *h, int32_t K, void *ring, mgx_stream stream);
/* Rows between consecutive blocks of the rings handed to mgx_observe_windows / mgx_observe_windows_ahead / mgx_fleet_step
 * refills (default: N, i.e. a dense [K, N, D] ring).  With N not a multiple of 16 the blocks of a dense ring are not
 * 128-byte aligned and every 1-KB wave store of the refill begins and ends in a partial line; a ring [K, P, D] with
 * P = N rounded up to 16 rows, block k = the first N rows of ring[k], avoids that. */
int mgx_set_ring_pitch(mgx_handle *h, int32_t rows);
/* Layout of a ring BLOCK.  MGX_RING_ROWS (default): [P, D] row-major -- a grid's observation is D consecutive values, the
 * reference's flat vector (envs/base/base.py:211-223).  MGX_RING_COLUMNS: [D, P] -- value (grid i, column c) at c * P + i, P = the
 * ring pitch (mgx_set_ring_pitch; a multiple of 32: whole 128-byte lines per 32 float / 16 double grids): the SAME [N, D] matrix read with strides (1, P), which is what a policy's
 * first matrix product takes either way.  Why: in a row-major block the step's state columns are 48 bytes at a 8 D-byte stride --
 * 100 000 scattered partial lines per step, 3.5-4 us of a 24-us config-5 fleet step (profiles/r04/exp_fleet_state_patch_cost.txt) --
 * in a column-major block they are six coalesced runs of 8 N bytes.  Applies to mgx_observe_windows[_ahead], the fleets' refills and
 * the state-only `obs` target of the steps (pass the block's base).  A ring written AHEAD of the counter does not touch the state
 * columns of a column-major block at all (they are lines of their own; the step that reaches the block writes them), where a
 * row-major one receives zeros.  Layouts with several modules of a kind per grid included (their 4 n_genset + 2 n_battery state
 * columns).  Lock-step episodes only (mgx_patch_windows and the in-place / rolling modes keep row-major rings). */
enum mgx_ring_layout { MGX_RING_ROWS = 0, MGX_RING_COLUMNS = 1 };
int mgx_set_ring_layout(mgx_handle *h, int32_t layout);
/* Rolling windows with prefetched rings: after mgx_reset_grids* has replaced the series rows of the grids with mask[i] != 0,
 * recompute the window columns of THEIR rows in blocks first_block .. K-1 of `ring` (block first_block = the row of
 * counter value current + ahead).  State columns are left as they are.  A ring written AHEAD of the counter
 * (mgx_observe_windows_ahead) may have read a restarting grid's rows before or while they were replaced: once it is complete
 * (mgx_prefetch_wait), patch into it every grid that restarted since it was launched (first_block = 0 at the moment the
 * counter reaches its block 0).  `restarted` (device [N], may be NULL, must not be `mask`): restarted[i] = 1 for every masked
 * grid -- the accumulator of that later patch. */
int mgx_patch_windows(mgx_handle *h, const uint8_t *mask, int32_t K, void *ring, int32_t first_block, int32_t ahead,
                      uint8_t *restarted, mgx_stream stream);
/* MGX_OBS_ROWS_STATE_COMPACT: `obs` receives ONLY the genset / battery state columns, as a dense [N, S] array,
 * S = 4 * has_genset + 2 * has_battery (the zero-copy observation contract: the window columns are views of the normalised
 * series written once by mgx_normalise_series). */
int mgx_set_obs_mode(mgx_handle *h, int32_t mode);

/* `done` of the fused calls.  MGX_DONE_U8 (default): one byte per grid and step, [K, N].  MGX_DONE_BITS: a bit set per step,
 * [K, W] uint16 words with W = ceil(N / 16), bit (i & 15) of word i >> 4 = done of grid i (little-endian: the row is a
 * ceil(N / 16) * 2-byte bit array) -- 1/8 byte instead of 1 byte per env-step, written by one lane in 16.
 * In lock-step (no per-grid episode ends) `done` is the same for every grid, done(k) = (t0 + k >= final_step - 1)
 * (base_timeseries_module.py:124-125): pass done = NULL and derive it from mgx_current_step / the window. */
int mgx_set_done_format(mgx_handle *h, int32_t format);

/* The zero-copy observation contract for a forecast horizon H > 0.  With the oracle forecaster the window columns of the
 * observation at step t are  norm[t .. t + H]  of a series normalised ONCE ((v - lo) / spread, space.py:207-218; rows past
 * the end of the series = the padding value (lo + hi) / 2, forecaster.py:95,120-137; the forecast clip, forecaster.py:139-149,
 * is the identity because lo / hi bound the series).  mgx_normalise_series writes that normalised copy GRID-major:
 *     load_n, pv_n  [N, R]      R = n_steps + horizon + 1 (the observation after the last step, counter = n_steps, is all padding)
 *     grid_n        [N, R, 4]   (component-minor: the reference's window order falls out of a flat slice)
 * as float64 or float32 (the handle's obs format) so that the window of grid i at step t is the contiguous slice
 * load_n[i, t : t + 1 + H] -- a strided VIEW, no bytes moved per step.  The contract needs every bound column to bound its
 * series (else the reference's forecast clip is not the identity and the views differ from its observations).  That is checked
 * ON DEVICE, asynchronously: the call itself returns MGX_OK; `clipped` (device int32[1], zeroed by the caller) is incremented per
 * offending value and must be read after the stream has synchronised -- a non-zero count means the views are NOT the
 * reference's observations (pymgrid_amd refuses such a batch).  Passing NULL skips the check: only for series known to lie
 * inside their bounds (e.g. bounds computed from the series, base_timeseries_module.py:81-88).  grid_n may be NULL without a
 * GridModule. */
int mgx_normalise_series(mgx_handle *h, void *load_n, void *pv_n, void *grid_n, int32_t *clipped, mgx_stream stream);

/* The same prefetch AHEAD of the counter, overlapped with the steps: block k of `ring` = the window columns of counter
 * value t + ahead + k (ahead >= 1; the state columns of every block are zero and are filled in by the steps that reach
 * them).  The kernel runs on the handle's own prefetch stream, behind everything queued on `stream` so far (the readers
 * of the ring's previous contents); mgx_prefetch_wait(h, s) makes stream s wait for the last prefetch.  Typical use:
 * three rings of K blocks -- while the steps walk ring r, ring r + 1 is being written and ring r - 1 is still readable by
 * the policy (pymgrid_amd/envs.py).  The series rows do not depend on the state, so the overlap needs no other ordering.
 * MGX_ERR_UNSUPPORTED in device-counter mode. */
int mgx_observe_windows_ahead(mgx_handle *h, int32_t ahead, int32_t K, void *ring, mgx_stream stream);
int mgx_prefetch_wait(mgx_handle *h, mgx_stream stream);

/* Normalised observation of the current state (BaseMicrogridModule.to_normalized(state), base_module.py:157;
 * forecast window + end-of-series padding forecaster.py:120-149,215-217). */
int mgx_observe(mgx_handle *h, void *obs, mgx_stream stream);

/* ONE Microgrid.run(control, normalized) for all N grids (microgrid.py:227-325), as BaseMicrogridEnv.step
 * returns it (base.py:169-209): reward [N], done [N] (0/1), optional post-step obs [N, D] and log [L, N].
 * The step counter advances by one.  MGX_ERR_RANGE if the counter is already outside the series. */
int mgx_step(mgx_handle *h, const void *actions, int normalized,
             double *reward, uint8_t *done, void *obs, double *log, mgx_stream stream);

/* K consecutive Microgrid.run calls in ONE launch: parameters and state stay in registers, actions
 * [K, N, A] are streamed.  Outputs (each may be NULL): reward [K, N], done [K, N], soc_trace [K, N],
 * status_trace [K, N] (post-step packed genset status), ret_acc [N] (+= sum of the K rewards per grid),
 * log [K, L, N].  Replaces the user loop `for a in actions: env.step(a)` (README.md:109-111). */
int mgx_step_k(mgx_handle *h, const void *actions, int32_t K, int normalized,
               double *reward, uint8_t *done, double *soc_trace, uint32_t *status_trace,
               double *ret_acc, double *log, mgx_stream stream);

/* DiscreteMicrogridEnv._get_action -> PriorityListAlgo._populate_action (discrete.py:82-88,
 * priority_list.py:69-167): expand one priority-list id per grid into an UNNORMALISED control [N, A]
 * (feed it to mgx_step(..., normalized=0)).  `table` is a HOST array [n_actions, 3, 2] of
 * (module, action) pairs, module 0 genset / 1 battery / 2 grid, -1 = padding; n_actions <= 12.
 * violations [N] uint32 (device, may be NULL) receives, per grid, the assert the reference's _populate_action would have
 * failed in this state (MGX_V_EXPAND_*; 0 = none): there the reference raises AssertionError and returns no control. */
int mgx_expand_discrete(mgx_handle *h, const int32_t *action_id, const int32_t *table, int32_t n_actions,
                        double *control, uint32_t *violations, mgx_stream stream);

/* The same for priority lists over module INSTANCES (microgrids with several gensets / batteries / grids:
 * get_priority_lists enumerates (module name, module number, action) elements, priority_list.py:15-67, and there can
 * be hundreds of lists): `lists` is a DEVICE array int32 [n_lists, list_len, 3] of (kind, instance, action), kind
 * 0 genset / 1 battery / 2 grid, kind -1 = padding.  Works for every layout. */
int mgx_expand_lists(mgx_handle *h, const int32_t *action_id, const int32_t *lists, int32_t n_lists, int32_t list_len,
                     double *control, uint32_t *violations /* as in mgx_expand_discrete; may be NULL */, mgx_stream stream);

/* K fused discrete steps with priority lists over module instances: the general-path counterpart of
 * mgx_rollout_discrete (`for a in ids: env.step(a)`, discrete.py:109-143, or RuleBasedControl.run with one fixed list
 * per grid, rbc.py:64-93).  action_id int32 [K, N] (per_step != 0) or [N]; lists as in mgx_expand_lists; outputs as in
 * mgx_step_k (soc_trace / status_trace report battery 0 / genset 0).  Works for every layout. */
int mgx_rollout_lists(mgx_handle *h, const int32_t *action_id, int per_step, const int32_t *lists, int32_t n_lists,
                      int32_t list_len, int32_t K, double *reward, uint8_t *done, double *soc_trace,
                      uint32_t *status_trace, double *ret_acc, double *log, mgx_stream stream);

/* DiscreteMicrogridEnv.step (discrete.py:109-143) in ONE launch: mgx_expand_discrete + mgx_step(normalized=0) with the
 * control kept in registers.  control [N, A] (optional, may be NULL) receives the expanded control; the other
 * outputs are those of mgx_step. */
int mgx_step_discrete(mgx_handle *h, const int32_t *action_id, const int32_t *table, int32_t n_actions, double *control,
                      double *reward, uint8_t *done, void *obs, double *log, mgx_stream stream);

/* (minor 2) DiscreteMicrogridEnv.step (discrete.py:109-143) for priority lists over module INSTANCES: mgx_expand_lists +
 * mgx_step(normalized=0).  ONE launch -- the list walked and the control kept in registers -- where the layout holds at most two
 * modules of a kind and steps in lock-step or per-grid windows; two launches through `control` otherwise.  control [N, A]: optional
 * where one launch is taken (it receives the expanded control when given), REQUIRED (the buffer between the two launches) for any
 * other layout -- MGX_ERR_INVALID without it.  lists as in mgx_expand_lists; the other outputs are those of mgx_step. */
int mgx_step_lists(mgx_handle *h, const int32_t *action_id, const int32_t *lists, int32_t n_lists, int32_t list_len,
                   double *control, double *reward, uint8_t *done, void *obs, double *log, mgx_stream stream);

/* K fused DiscreteMicrogridEnv steps with the control expanded ON DEVICE: action_id holds priority-list ids as
 * bytes, either [K, N] (per_step != 0: `for a in ids: env.step(a)`, discrete.py:109-143) or [N] (per_step == 0: one
 * fixed list per grid for the whole call = RuleBasedControl.run, algos/rbc/rbc.py:64-93).  `table` as in
 * mgx_expand_discrete; outputs as in mgx_step_k.  No action stream is read at all. */
int mgx_rollout_discrete(mgx_handle *h, const uint8_t *action_id, int per_step, const int32_t *table,
                         int32_t n_actions, int32_t K, double *reward, uint8_t *done, double *soc_trace,
                         uint32_t *status_trace, double *ret_acc, double *log, mgx_stream stream);

/* raise_errors=True (BaseMicrogridModule.__init__, base_module.py:40; as_source / as_sink, :213-224,265-270; _raise_error,
 * :79-93): the reference refuses a request a module cannot meet with a ValueError instead of clipping it.  mgx_step always
 * clips; mgx_check_step is its DRY RUN -- the same arithmetic on a register copy of the state, nothing stored, the counter
 * untouched -- and writes per grid the mask of requests the reference would refuse: bit 0 genset request outside
 * [min, max] production, bit 1 battery request above max_production / max_consumption, bit 2 grid request above its
 * limit, bit 3 genset goal outside [0, 1], bit 4 negative genset energy, bit 5 a battery / grid acting at a negative
 * limit (enum mgx_violation_bit).  A caller that wants raise_errors semantics
 * checks first and steps only when every mask is 0 (the reference has by then already stepped the modules that come
 * before the refusing one in its sweep; here nothing is applied).  violations [N] uint32 (device). */
int mgx_check_step(mgx_handle *h, const void *actions, int normalized, uint32_t *violations, mgx_stream stream);
/* The dry run of mgx_step_discrete: DiscreteMicrogridEnv.step asserts its way through _populate_action
 * (priority_list.py:73,121,124,135,154) and through the step (base_module.py:272) whatever raise_errors says.  violations [N]
 * receives the expansion's assert bit where the reference would have raised inside _populate_action (MGX_V_EXPAND_*: it never
 * steps then), else the mask mgx_check_step would give for the expanded control.  Nothing is stored, the counter does not
 * move.  One module of every kind per grid (layouts with several: mgx_expand_lists with `violations`, then mgx_check_step). */
int mgx_check_discrete(mgx_handle *h, const int32_t *action_id, const int32_t *table, int32_t n_actions, uint32_t *violations,
                       mgx_stream stream);

/* K consecutive mgx_step calls issued by ONE call: the Gym cadence (`for a in actions: env.step(a)`, one kernel launch
 * per env-step, envs/base/base.py:169-209) without a host round trip per step -- from Python a per-step call costs
 * more than the 5 us the kernel takes at N = 100 000.  actions [K, N, A]; reward [K, N]; done [K, N], obs [K, N, D]
 * (block k = the observation after step k) and log [K, L, N] may be NULL. */
int mgx_step_many(mgx_handle *h, const void *actions, int32_t K, int normalized,
                  double *reward, uint8_t *done, void *obs, double *log, mgx_stream stream);

/* The Gym step with NO per-step bookkeeping on the caller's side (`while not done: obs, r, done, info = env.step(a)`,
 * README.md:109-111, envs/discrete/discrete.py:109-143, envs/base/base.py:169-209).  A single-step kernel takes ~5 us at
 * N = 100 000, so whatever the host does between two launches -- choosing the output buffers, walking the observation rings,
 * issuing the next ring's prefetch -- is part of the env-step's time when it is done in an interpreter.  mgx_env_bind hands the
 * handle everything that rotates, once; mgx_env_step / mgx_env_step_discrete then take the controls and a stream and do the rest:
 *   outputs   step j (counted from the bind / the last mgx_env_seek) writes reward / done / log -- and, without rings, the
 *             observation row -- into slots[j % n_slots]: n_slots rotating buffer sets, each valid until the step n_slots later;
 *   rings     ring_K > 0 (the handle in MGX_OBS_ROWS_STATE_ONLY mode): three rings of ring_K row blocks written ahead by
 *             mgx_observe_windows[_ahead].  The env stands on block `ring_pos` of ring `ring_idx`; a step adds its state columns to
 *             the NEXT block (the observation it returns), moves there, and when that enters the following ring it first waits for
 *             that ring's prefetch (mgx_prefetch_wait) and afterwards starts the prefetch of the ring behind it
 *             (mgx_observe_windows_ahead(ahead = ring_K)): exactly the sequence pymgrid_amd/envs.py used to issue call by call.
 * Values are those of mgx_step / mgx_step_discrete, bit for bit (tests/test_env_step.py).  mgx_env_seek tells the handle where
 * the caller stands after anything that moved the rings or the slot outside these calls (mgx_reset + mgx_observe_windows: block 0
 * of ring 0); mgx_env_position reports where the LAST step's outputs are.  Slot and ring pointers are caller-owned device memory. */
typedef struct mgx_env_slot {
    double *reward;               /* [N] */
    uint8_t *done;                /* [N] or NULL (lock-step: derive it from the counter) */
    void *obs;                    /* [N, D] row target of the step (NULL with rings, or when no observation is wanted) */
    double *log;                  /* [L, N] or NULL */
} mgx_env_slot;
#define MGX_ENV_MAX_SLOTS 128
typedef struct mgx_env_plan {
    int32_t struct_size;          /* = sizeof(mgx_env_plan) */
    int32_t n_slots;              /* 1 .. MGX_ENV_MAX_SLOTS */
    const mgx_env_slot *slots;    /* host array [n_slots]; copied */
    int32_t ring_K;               /* 0: no rings */
    int32_t n_actions;            /* discrete: rows of `table` (0: continuous steps only) */
    void *rings[3];               /* ring r = [ring_K] blocks, the handle's ring pitch / layout (mgx_set_ring_pitch / _layout) */
    const int32_t *table;         /* host [n_actions, 3, 2] (mgx_expand_discrete); copied */
} mgx_env_plan;
int mgx_env_bind(mgx_handle *h, const mgx_env_plan *plan);           /* plan == NULL: unbind */
int mgx_env_seek(mgx_handle *h, int32_t next_slot, int32_t ring_idx, int32_t ring_pos);
int mgx_env_position(const mgx_handle *h, int32_t *last_slot, int32_t *ring_idx, int32_t *ring_pos);   /* each may be NULL */
int mgx_env_step(mgx_handle *h, const void *actions, int normalized, mgx_stream stream);
int mgx_env_step_discrete(mgx_handle *h, const int32_t *action_id, mgx_stream stream);

/* BaseMicrogridModule.sample_action(strict_bound=True) (modules/base/base_module.py:326-356, Microgrid.sample_action
 * microgrid.py:337-362): lo / hi [N, A] (device) receive, per action column, the NORMALISED interval a draw must come from to
 * respect the module's instantaneous limits at the current state and row: battery and grid columns
 * [normalize(-max_consumption), normalize(max_production)] (a NaN bound is 0, as there); genset columns [0, 1] (the reference
 * itself raises on a genset with strict_bound: the Python mirror refuses such layouts).  A strict sample is
 * lo + u * (hi - lo), u ~ U[0, 1).  Nothing is stepped. */
int mgx_action_bounds(mgx_handle *h, double *lo, double *hi, mgx_stream stream);

/* Shards.  Grids never interact (no cross-grid term anywhere in Microgrid.run), so the launch sequence of one range of
 * grids owes nothing to another's.  mgx_set_shards(h, S > 1) splits every stepping call (mgx_step, mgx_step_many,
 * mgx_step_k, mgx_step_discrete, mgx_expand_discrete, mgx_rollout_discrete) into S launches over contiguous grid ranges,
 * each on its own internal HIP stream, and the ranges are NOT joined between calls: while one range is in the ramp-up or
 * the tail of a launch another is in full flight (measured: 66-68 instead of 74-76 us per 64-step round of 100 000
 * grids with S = 2).  In this mode the `stream` argument of those calls is ignored; ordering
 * against other streams is explicit:
 *   mgx_fork(h, s)  the shard streams wait for everything queued on s so far  (inputs produced on s are then safe)
 *   mgx_join(h, s)  s waits for everything issued to the shard streams so far (outputs are then safe to read on s)
 * Every other entry point still runs on the stream it is given: bracket it with mgx_join / mgx_fork.  Buffers handed to
 * a stepping call must stay alive until the next mgx_join.  mgx_shard_stream returns the hipStream_t of a shard (for
 * timing events), NULL when shards are off.  Not offered in device-counter mode.  The shard streams are a per-device pool
 * shared by all handles of the process (the runtime multiplexes streams onto a few hardware queues: private pairs per handle
 * ended up on one queue): handles that step in shards at the same time are ordered per stream -- work of two handles on
 * shard j runs in issue order, and mgx_destroy / mgx_set_shards of one handle synchronise pooled streams that other handles
 * may still be using (a wait, never a loss).  The pool is created under a lock: handles may live on different host threads
 * (one thread per handle). */
int mgx_set_shards(mgx_handle *h, int32_t n_shards);
/* Single-step calls in shards (mgx_step, mgx_step_many, mgx_step_discrete, mgx_env_step*): one host thread issues a launch every
 * 3.4-4.3 us, so S launches per env-step from the calling thread make the Gym cadence S times slower than no shards at all.
 * Launch threads: shard j >= 1 is issued by a resident host thread of the library (one per device and shard, shared by all
 * handles, asleep when idle) while the caller issues shard 0.  mode 0: never; 1 (default): inside mgx_step_many, where every
 * thread gets its shard's K launches in one piece; 2: also for every single-step call (one hand-over per call).  A call still
 * returns only when every shard's launches have been issued, and values do not depend on the mode.  Fused calls (mgx_step_k,
 * rollouts) are always issued by the caller.
 * MEASURED (profiles/r06/exp_two_chains.txt, chain_trace_summary.txt; N = 100 000): two chains from two threads run 6.1 us per
 * env-step against 4.8 us for ONE chain from one thread (7.9 us for two chains from one thread): inside one process two threads
 * issue at 6 us per launch each, and a 50 000-grid launch lasts as long as a 100 000-grid one (4.2-4.8 us: the step is a latency
 * chain, not a transfer), so the second chain cannot pay whatever the host does.  Kept because it halves the cost of shards for
 * single steps; not a way below the one-chain cadence. */
int mgx_set_launch_threads(mgx_handle *h, int32_t mode);
int mgx_fork(mgx_handle *h, mgx_stream stream);
int mgx_join(mgx_handle *h, mgx_stream stream);
void *mgx_shard_stream(mgx_handle *h, int32_t shard);

/* Heterogeneous fleets (a population of microgrids with different module sets: one handle per layout).  One call steps
 * every batch of the fleet once -- `for m in microgrids: m.run(control)` over the whole population -- and, for the
 * batches whose observation ring is used up, issues the window prefetch of the next refill_K steps
 * (mgx_observe_windows) behind the steps; no host work between the launches.  An item is a continuous step
 * (actions; `normalized` applies) or, when action_id != NULL, a discrete one (mgx_step_discrete with table / n_actions).
 * All items are validated before the first launch. */
typedef struct mgx_fleet_item {
    int32_t struct_size;          /* = sizeof(mgx_fleet_item) */
    int32_t n_actions;            /* discrete: rows of `table` */
    mgx_handle *handle;
    const void *actions;          /* [N, A] continuous control, or NULL */
    const int32_t *action_id;     /* [N] priority-list ids (device), or NULL */
    const int32_t *table;         /* host [n_actions, 3, 2], discrete only */
    double *reward;               /* [N] */
    uint8_t *done;                /* [N] or NULL */
    void *obs;                    /* [N, D] (or the state-only target inside a ring block) or NULL */
    double *log;                  /* [L, N] or NULL */
    void *refill_ring;            /* [refill_K, N, D] or NULL: window prefetch issued with this step.  Block k = the window columns of
                                   * counter value (counter after this step) + refill_ahead + k.
                                   *   refill_chunks == 0: the whole batch -- mgx_observe_windows (refill_ahead == 0, after the step)
                                   *     or mgx_observe_windows_ahead (refill_ahead >= 1, on the handle's prefetch stream);
                                   *   refill_chunks >= 1 (refill_ahead >= 1): only chunk refill_chunk of refill_chunks of the
                                   *     batch's 16-grid groups, on `stream`, as extra workgroups of the step's own launch: a ring
                                   *     spread over the K - 1 steps before it is needed moves at a constant rate instead of
                                   *     one burst per K steps (state columns zero, as in mgx_observe_windows_ahead). */
    int32_t refill_K;
    int32_t refill_ahead;
    int32_t refill_chunk;
    int32_t refill_chunks;
    int32_t wait_prefetch;        /* != 0: mgx_prefetch_wait(handle, stream) before the step (its obs target lies in a ring that
                                   * mgx_observe_windows_ahead wrote) */
    int32_t reserved;
} mgx_fleet_item;
int mgx_fleet_step(const mgx_fleet_item *items, int32_t n_items, int normalized, mgx_stream stream);
/* The bound form of a fleet's Gym step: every handle carries an env plan (mgx_env_bind: rotating reward / done / row / log slots,
 * three observation rings) and the call is mgx_fleet_step over the items those plans produce -- handle j steps with actions[j]
 * (continuous controls [N, A], or int32 priority-list ids [N] where the plan holds a table) into its next slot and ring block,
 * waits for its prefetched ring when it enters it and has the ring after that one written ahead -- one launch for the steps of
 * all layouts, one C call per fleet step, no per-step bookkeeping on the host (`for env in envs: env.step(a)` of N reference
 * microgrids of different module sets; envs/base/base.py:169-209).  Afterwards every handle's slot and ring position have moved
 * as by mgx_env_step (mgx_env_position).  Nothing moves when the call fails validation.  n_handles <= 64. */
int mgx_fleet_env_step(mgx_handle *const *handles, const void *const *actions, int32_t n_handles, int normalized, mgx_stream stream);

/* MicrogridGenerator's time series on device (MicrogridGenerator.py): load / pv = base profile x size / max(profile)
 * (_scale_ts, :137-147; stored with the modules' sign, base_timeseries_module.py:68-79), import price = tariff pattern
 * 1 / 2 by hour of day (_get_electricity_tariff, :253-285), export price 0, co2 = base co2 profile (:205-212), grid
 * status = 1 except weak-grid outages (_generate_weak_grid_profile, :321-340: a uniform draw per row below
 * outage_per_day / 24 starts an outage whose back-fill covers the duration - 1 rows before it, never row 0; T + 1
 * draws).  Uniforms come from Philox4x32-10(seed; global grid index, row), so rank r of W writes exactly columns
 * [grid_index0, grid_index0 + n_grids) of the same global batch.  Everything is a device pointer; grid_ts (and the
 * arrays only it needs) may be NULL.  No handle: this runs before mgx_create. */
typedef struct mgx_synth {
    int32_t struct_size;          /* = sizeof(mgx_synth) */
    int32_t n_grids, n_steps;
    int32_t n_load_profiles, n_pv_profiles, n_co2_profiles;
    const double *base_load, *base_pv, *base_co2;          /* [T, n_*_profiles], values as in the reference's csv files */
    const int32_t *load_profile, *pv_profile, *co2_profile; /* [N] column of the base arrays */
    const double *load_ratio, *pv_ratio;                    /* [N] size / max(profile) */
    const int32_t *tariff;                                  /* [N] 1 or 2 */
    const int32_t *weak;                                    /* [N] 0 / 1 (rand_weak_grid, :535) */
    const double *outage_per_day;                           /* [N] randn * 3/4 + 0.25 (:291) */
    const int32_t *outage_duration;                         /* [N] randint(1, 8) (:292) */
    uint64_t seed;
    int64_t grid_index0;                                    /* global index of grid 0 ... */
    const int64_t *grid_index;                              /* ... or [N] global indices (a scattered selection); NULL: grid_index0 + i */
    double *load_ts, *pv_ts;                                /* out [T, N]; both NULL: only the outage words are produced */
    double *grid_ts;                                        /* out [T, 4, N] or NULL */
    uint64_t *outage_bits;                                  /* out [ceil(T / 64), N] or NULL: the factorised form of grid_status
                                                             * (mgx_columns.outage_bits), same draws as grid_ts[:, 3] */
} mgx_synth;
int mgx_synthesize_series(const mgx_synth *args, mgx_stream stream);

/* MicrogridGenerator's per-microgrid DRAWS and SIZING on device (MicrogridGenerator.py:346-386 _size_mg / _size_genset /
 * _size_battery, :417-441 _bin_genset_grid / _size_load, :230-243 _get_battery, :288-292 _get_grid, :535-538; module parameters
 * as convert/get_module.py:39-97 derives them): one lane per grid draws the grid's scalars and sizes its modules.  Every
 * output is a device array [n_grids] (NULL = not wanted); nothing of size N ever exists on the host, and rank r of W produces
 * exactly the grids [grid_index0, grid_index0 + n_grids) of the same global batch.
 *   draws   counter-based: U[0, 1) = Philox4x32-10(seed ^ 0x9E3779B97F4A7C15; global grid index, quantity id) -- the reference
 *           consumes numpy's global stream one microgrid at a time, which no batch can reproduce; what is reproduced is every
 *           RULE applied to the draws.  randint(lo, hi) = lo + floor(u (hi - lo)); a standard normal = the sum of 12 uniforms
 *           minus 6 (Irwin-Hall: additions only, so host and device agree bit for bit without sharing a libm).
 *   rules   size_load ~ randint(100, 100001), load / pv file ~ randint(0, n_profiles), pv penetration ~ randint(30, 151),
 *           battery hours ~ randint(3, 6); load_ratio = size_load / max(load profile) (_scale_ts, :137-147); pv size = peak of the
 *           scaled load x penetration / 100, pv_ratio = pv size / max(pv profile); battery capacity = ceil(hours x mean of the
 *           scaled load) -- the mean summed in numpy's pairwise order over the n_steps products, divided by n_steps -- power =
 *           ceil(capacity / 4), min capacity = 0.2 capacity, soc0 = min(max(normal, 0.2), 1), charge = soc0 x capacity; genset
 *           rating = ceil(peak / 0.9), running min / max = 0.05 / 0.9 of it; grid power = floor(2 peak); weak ~ randint(0, 2),
 *           tariff ~ randint(1, 3), outages per day = normal x 3 / 4 + 0.25, outage duration ~ randint(1, 8), co2 file ~
 *           randint(0, n_co2_profiles); genset timers ~ randint(0, 4) when mixed_timers else 0; architecture by
 *           _bin_genset_grid's uniform (< 0.33 genset only, < 0.66 grid only, else both; a weak grid forces a genset).
 * base_load: the load table; *_max / *_min: per-profile extrema (host: a few numbers).  Observation bounds come out as the
 * modules compute them from the series they hold (base_timeseries_module.py:81-88). */
typedef struct mgx_gen {
    int32_t struct_size;          /* = sizeof(mgx_gen) */
    int32_t n_grids, n_steps;
    int32_t n_load_profiles, n_pv_profiles, n_co2_profiles;
    int32_t mixed_timers;
    int32_t n_mean_rows;          /* rows of base_load the sizing rules see: the WHOLE profile (8 760), whatever n_steps is */
    uint64_t seed;
    int64_t grid_index0;
    const int64_t *grid_index;    /* [N] global indices (a scattered selection) or NULL: grid_index0 + i */
    const double *base_load;      /* [n_mean_rows, n_load_profiles] (the mean of the scaled load) */
    const double *load_max, *pv_max;                      /* [n_*_profiles]: max over the whole profile (the sizing rules) */
    const double *load_bound_max, *pv_bound_max;          /* [n_*_profiles]: max over the n_steps rows a module holds (its bounds) */
    const double *co2_min, *co2_max;                      /* [n_co2_profiles], over the n_steps rows */
    double tariff_min[3], tariff_max[3];                  /* by pattern (0 unused) */
    /* outputs, each [N] or NULL */
    uint8_t *arch;                /* 0 genset+battery, 1 battery+grid, 2 genset+battery+grid */
    uint8_t *load_profile, *pv_profile, *co2_profile, *tariff;
    int32_t *weak, *outage_duration;
    double *outage_per_day;
    double *load_ratio, *pv_ratio;
    double *load_lo, *load_hi, *pv_lo, *pv_hi;
    double *grid_lo, *grid_hi;    /* [4, N]; the status rows (component 3) come out as 1 ("never out"): the caller lowers them where
                                   * the outage words it synthesises say otherwise */
    double *bat_min_capacity, *bat_max_capacity, *bat_max_charge, *bat_max_discharge, *charge, *soc;
    double *gen_running_min, *gen_running_max;
    uint32_t *gen_times, *gen_status;
    double *grid_max_import, *grid_max_export;
    /* the raw draws, for whoever wants to re-apply the rules: [N] each or NULL */
    double *d_bin_rand, *d_soc0_normal, *d_outage_normal;
    int32_t *d_size_load, *d_pv_pen, *d_bat_hours, *d_su, *d_wd;
} mgx_gen;
int mgx_generate_columns(const mgx_gen *args, mgx_stream stream);

/* Column sums over the grids, sums[m] = sum_i values[m*N + i] (deterministic two-stage wavefront-shuffle +
 * LDS reduction; the "metrics" vector that is all-reduced across GPUs).  M <= 64. */
int mgx_metrics(mgx_handle *h, const double *values, int32_t M, double *sums, mgx_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* MGX_H */
