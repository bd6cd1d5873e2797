// mgx_kernels.hpp -- HIP kernels of the batched microgrid-step engine, written for gfx950 (MI355X, CDNA4); the host
// side (the C ABI of include/mgx.h: argument checks, launch shapes, streams) is mgx_abi.hip, which includes this file.
//
// Execution shape: one lane per microgrid, 64-lane wavefronts, 256-thread workgroups, SoA columns so every
// global access of a wave is one contiguous 512-byte segment.  The path is element-wise and HBM-bound
// (~40 useful flops vs 189 B per env-step): no MFMA, no LDS tiling of the physics.
// LDS + wavefront shuffles are used where data actually crosses lanes: the [N, D] observation tile
// transpose and the metrics column sums.
//
// Block b runs on XCD b % 8 (MI355X_MICROARCH.md, "Workgroup dispatch"): because block b always owns grids
// [256 b, 256 b + 256), the parameter and state columns of a grid stay in the SAME XCD's 4 MiB L2 across
// the per-step launches -- the blockIdx -> data mapping is deliberately launch-invariant.
#pragma once
#include "mgx_core.hpp"

#include <type_traits>

namespace mgx {

// Workgroup size of the single-step / observation / fleet kernels.  128 since round 6 (profiles/r06/exp_block_size.txt, alternating runs:
// single step 4.94 -> 4.78 us, with row + log + done 10.85 -> 10.25, config-5 float64 fleet step 22.1 -> 21.5 against 256; 64 = 128; 512 = 256).
#ifndef MGX_BLOCK
#define MGX_BLOCK 128
#endif
constexpr int BLOCK = MGX_BLOCK;
// Register-ring depth of the fused kernel (steps of loads in flight).  Re-swept in round 6 (profiles/r06/exp_ring_depth8.txt, _16.txt): the
// headline form 0.698 -> 0.7145 of peak at depth 8 against 4 (0.754 -> 0.773 at 125 000 grids) and 0.558 at 16 (the ring then costs occupancy);
// the full-output form (RICH: it is write-bound, its waves wait on stores) gains up to 16: 314 -> 305 -> 297 us per 64-step launch.
#ifndef MGX_RING
#define MGX_RING 8
#endif
#ifndef MGX_RING_RICH
#define MGX_RING_RICH 16
#endif
#ifndef MGX_BLOCK_K
#define MGX_BLOCK_K 256     // workgroup size of the fused kernel
#endif
constexpr int BLOCK_K = MGX_BLOCK_K;
#ifndef MGX_RING_ROLLOUT
#define MGX_RING_ROLLOUT 8  // ring depth of the discrete rollout kernel for layouts without a GridModule (a slot is two series
#endif                      // values + an id byte; 58.2 vs 61.3 us per 64 steps against depth 4 once the loop was specialised:
                            // profiles/r02/exp_rollout_gpb_ring.txt); with a GridModule (six values per slot) the depth stays 4

// ------------------------------------------------------------------------------------------------------
// Single step: Microgrid.run for N grids (microgrid.py:227-325) + optional obs (base.py:205-209) + log.
// ------------------------------------------------------------------------------------------------------
// the `obs` argument of a single step: state columns only (inside full rows, or as a dense [N, S] array), or -- without a
// forecaster -- the whole 8..12-value row
// Non-temporal stores for the fused kernels' [K, N] output streams.  HOT = the headline's reward + SoC: 0.692 -> 0.709 and 0.695 -> 0.7145
// of peak in alternating runs on one box (profiles/r06/exp_out_nt_stores.txt); OUT = the general form's reward / done / SoC / status:
// inside the noise (283-297 vs 292-298 us per full-output launch), left plain.  ACT = non-temporal loads of the action stream (A/B).
#ifndef MGX_HOT_NT
#define MGX_HOT_NT 1
#endif
#ifndef MGX_ACT_NT
#define MGX_ACT_NT 0
#endif
#ifndef MGX_OUT_NT
#define MGX_OUT_NT 0
#endif
// Non-temporal stores of the H = 0 row tiles: Gym step with rows 7.25 -> 6.79 us, with row + log + done 10.25 -> 9.55 us in alternating
// runs (profiles/r06/exp_rows_nt.txt).  MGX_STATE_NT: the same for the state columns a step adds to a column-major ring block: config-5
// fleet step 20.7 -> 20.15 us (float64 rows), 13.55 -> 12.9 us (float32), profiles/r06/exp_state_nt.txt.
#ifndef MGX_ROWS_NT
#define MGX_ROWS_NT 1
#endif
#ifndef MGX_STATE_NT
#define MGX_STATE_NT 1
#endif
#ifndef MGX_ROWS_TILE
#define MGX_ROWS_TILE 1
#endif
constexpr int ROW_TILE_MAX_D = 12;                      // H = 0 rows: 2 + 4 + 2 + 4 values at most
// does a single-step launch write whole H = 0 rows through the wave-private LDS tiles?  (host: how much dynamic LDS to ask for;
// kernel: whether there is a tile) -- the same predicate on both sides
__host__ __device__ inline bool step_rows_tiled(const KArgs &a, const void *obs)
{
    return MGX_ROWS_TILE && obs != nullptr && a.H == 0 && !a.obs_state_only && a.obs_dim <= ROW_TILE_MAX_D;
}

// Whole H = 0 rows of a FULL wave through a wave-private LDS tile: every lane leaves its D-value row in the tile (row-major, as
// the block is in memory), then the wave streams the tile out with 16-byte stores -- 64 lanes x 16 B = 1 KB of consecutive
// bytes per instruction -- instead of D stores per lane that each touch 64 different lines (a lane's row is 32..96 bytes at a
// stride of its own length: 100 000 grids x 8 partial-line writes per step).  Same values: the row is built by observe_row_h0.
template <int F, typename OT>
__device__ __forceinline__ void observe_row_h0_tiled(const KArgs &a, int64_t i, int32_t t_next, const Params &p, const State &s,
                                                     OT *__restrict__ obs, int32_t pm, OT *tile)
{
    const int lane = threadIdx.x & 63;
    const int32_t D = a.obs_dim;
    observe_row_h0<F>(a, i, t_next, p, s, tile + lane * D, pm);
    __builtin_amdgcn_wave_barrier();                     // (one wave: its LDS traffic is ordered; this keeps the compiler from moving the reads up)
    typedef OT vec2 __attribute__((ext_vector_type(2)));
    typedef OT vec4 __attribute__((ext_vector_type(4)));
    OT *out = obs + (i - lane) * D;                      // the wave's 64 rows are consecutive in memory
    const int32_t total = 64 * D;
    if constexpr (sizeof(OT) == 8) {                     // D is even: a 16-byte pair never straddles the tile's end
        if ((reinterpret_cast<uintptr_t>(out) & 15) == 0) {
            for (int32_t e = 2 * lane; e < total; e += 128) {
                const vec2 v = *reinterpret_cast<const vec2 *>(tile + e);
#if MGX_ROWS_NT
                __builtin_nontemporal_store(v, reinterpret_cast<vec2 *>(out + e));
#else
                *reinterpret_cast<vec2 *>(out + e) = v;
#endif
            }
        } else {                                         // a caller's buffer at an odd 8-byte offset: word stores, same bytes
            for (int32_t e = lane; e < total; e += 64) out[e] = tile[e];
        }
    } else {
        if ((total & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0) {
            for (int32_t e = 4 * lane; e < total; e += 256) {
                const vec4 v = *reinterpret_cast<const vec4 *>(tile + e);
#if MGX_ROWS_NT
                __builtin_nontemporal_store(v, reinterpret_cast<vec4 *>(out + e));
#else
                *reinterpret_cast<vec4 *>(out + e) = v;
#endif
            }
        } else {
            for (int32_t e = 2 * lane; e < total; e += 128) {
                const vec2 v = *reinterpret_cast<const vec2 *>(tile + e);
                *reinterpret_cast<vec2 *>(out + e) = v;
            }
        }
    }
}

template <int F>
__device__ __forceinline__ void store_step_obs(const KArgs &a, void *__restrict__ obs, int64_t i, int32_t t_next, const Params &p,
                                               const State &s, int32_t pm = 0,     // pm: KArgs.pm_pitch during in-place episodes
                                               double *tile = nullptr)             // a wave-private LDS tile for whole H = 0 rows, or NULL
{
    constexpr int NSTATE = 4 * ((F & F_GENSET) != 0) + 2 * ((F & F_BATTERY) != 0);
    if (a.obs_state_only == 2) {            // MGX_OBS_ROWS_STATE_COMPACT
        if (a.obs_f32) observe_state_cols<F>(a, p, s, (float *)obs + i * NSTATE, 0);
        else observe_state_cols<F>(a, p, s, (double *)obs + i * NSTATE, 0);
    } else if (a.obs_state_only && a.obs_colpitch) {        // ... into a COLUMN-major ring block: the wave's grids are adjacent in every column
        const int64_t P = a.obs_colpitch;
        if (a.obs_f32) {
            float st[6] = {0, 0, 0, 0, 0, 0};
            observe_state_cols<F>(a, p, s, st, 0);
            float *b = (float *)obs + i;
#pragma unroll
            for (int j = 0; j < NSTATE; j++) {
                float *q = b + (int64_t)(((F & F_GENSET) != 0 && j < 4) ? a.col_gen + j : a.col_bat + (j - ((F & F_GENSET) ? 4 : 0))) * P;
#if MGX_STATE_NT
                __builtin_nontemporal_store(st[j], q);
#else
                *q = st[j];
#endif
            }
        } else {
            double st[6] = {0, 0, 0, 0, 0, 0};
            observe_state_cols<F>(a, p, s, st, 0);
            double *b = (double *)obs + i;
#pragma unroll
            for (int j = 0; j < NSTATE; j++) {
                double *q = b + (int64_t)(((F & F_GENSET) != 0 && j < 4) ? a.col_gen + j : a.col_bat + (j - ((F & F_GENSET) ? 4 : 0))) * P;
#if MGX_STATE_NT
                __builtin_nontemporal_store(st[j], q);
#else
                *q = st[j];
#endif
            }
        }
    } else if (a.obs_state_only) {          // the window columns of this row were prefetched (obs_windows_k_kernel)
        if (a.obs_f32) observe_state_cols<F>(a, p, s, (float *)obs + i * a.obs_dim);
        else observe_state_cols<F>(a, p, s, (double *)obs + i * a.obs_dim);
    } else if (MGX_ROWS_TILE && tile != nullptr && a.obs_dim <= ROW_TILE_MAX_D && __builtin_amdgcn_read_exec() == ~0ull) {
        // every lane of the wave owns a grid (uniform): the wave's rows leave together
        if (a.obs_f32) observe_row_h0_tiled<F>(a, i, t_next, p, s, (float *)obs, pm, reinterpret_cast<float *>(tile));
        else observe_row_h0_tiled<F>(a, i, t_next, p, s, (double *)obs, pm, tile);
    } else if (a.obs_f32) observe_row_h0<F>(a, i, t_next, p, s, (float *)obs + i * a.obs_dim, pm);
    else observe_row_h0<F>(a, i, t_next, p, s, (double *)obs + i * a.obs_dim, pm);
}

// mgx_set_auto_reset: the grid whose episode this step ended (dn) restarts at once -- the draw of mgx_reset_grids_random at the counter
// value after the step.  Returns the row offset the step's observation is read with (the new episode's when the grid restarted).
__device__ __forceinline__ int32_t episode_auto_restart(const KArgs &a, int64_t i, int32_t t, int32_t off, bool dn)
{
    if (a.ar_mode && dn) {
        int32_t s0, len;
        episode_draw(a.ar_seed, i, t + 1, a.ar_fixed_length, a.ar_lo, a.ar_hi, s0, len);
        episode_clamp(a.ar_lo, a.ar_hi, a.ar_max_length, s0, len);
        off = s0 - (t + 1);
        a.ep_off[i] = off;
        a.ep_final[i] = t + 1 + len;
        if (a.ar_start_io) a.ar_start_io[i] = s0;
        if (a.ar_length_io) a.ar_length_io[i] = len;
        if (a.ar_t0_io) a.ar_t0_io[i] = t + 1;
    }
    return off;
}

// In-place episodes (KArgs.ep_off): what a step adds once the grid's own step is done -- the observation before a restart
// (mgx_set_final_obs), the restart itself when this step ended the grid's episode (mgx_set_auto_reset: the draw of
// mgx_reset_grids_random at the counter value after the step).  Returns the row offset the step's observation is read with.
template <int F>
__device__ __forceinline__ int32_t episode_tail(const KArgs &a, int64_t i, int32_t t, int32_t off, bool dn, const Params &p, const State &s)
{
    if (a.final_obs && a.obs_state_only != 1) store_step_obs<F>(a, a.final_obs, i, t + 1 + off, p, s, a.pm_pitch);   // (rings: mgx_patch_windows saves it)
    return episode_auto_restart(a, i, t, off, dn);
}

// body of one step of grid i (shared by step_kernel and fleet_step_kernel).  EP: in-place per-grid episodes (the grid's series
// row is counter + ep_off[i]; a compile-time form so that the lock-step kernel carries none of it)
template <int F, bool EP = false>
__device__ __forceinline__ void step_body(const KArgs &a, const void *__restrict__ actions, int32_t t, int normalized,
                                          double *__restrict__ reward, uint8_t *__restrict__ done, void *__restrict__ obs,
                                          double *__restrict__ log, int64_t i, double *tile = nullptr)
{
    // all loads first (independent, one latency round), then the arithmetic
    Params p; State s; Inputs in; Outputs o; Derived d;
    int32_t off = 0;
    if constexpr (EP) off = a.ep_off[i];
    const int64_t row = EP ? episode_row(a, t, off) : (int64_t)(t & a.row_mask);
    const int32_t pm = EP ? a.pm_pitch : 0;
    if (a.act_f32) load_inputs<F>(a.c, (const float *)actions, a.N, i, row, in, pm);
    else load_inputs<F>(a.c, (const double *)actions, a.N, i, row, in, pm);
    load_state<F>(a.c, i, log != nullptr, s);
    load_params<F>(a.c, i, p);
    derive<F>(p, d);
    const bool gen_instant = genset_wave_is_instant<F>(p, s);

    step_core<F>(p, d, s, in, normalized != 0, true, gen_instant, o);

    store_state<F>(a.c, i, s);
    reward[i] = shaped_reward<F>(a.shaper, o);
    // _done(): t >= final_step - 1, evaluated before the counter moves (base_timeseries_module.py:124-125)
    const uint8_t dn = done_at(a, i, t);
    if (done) done[i] = dn;
    if (log) store_log<F>(log + i, a.N, o, s.status);
    if constexpr (EP) off = episode_tail<F>(a, i, t, off, dn != 0, p, s);
    // post-step observation (base.py:205-209): without a forecaster the whole 8..12-value row is stored here; with
    // one (H > 0) the host launches obs_rows_wave_kernel behind this kernel and passes obs == nullptr
    if (obs) store_step_obs<F>(a, obs, i, t + 1 + off, p, s, pm, tile);
}

template <int F, bool EP = false>
__global__ __launch_bounds__(BLOCK) void step_kernel(const KArgs a, const void *__restrict__ actions, int32_t t,
                                                     int normalized, double *__restrict__ reward,
                                                     uint8_t *__restrict__ done, void *__restrict__ obs,
                                                     double *__restrict__ log)
{
    t = resolve_t(a, t);
    // one wave-private tile per wave for whole H = 0 rows (store_step_obs): DYNAMIC LDS, asked for by the launches that write such
    // rows only (mgx_abi.hip row_tile_lds) -- every other launch of this kernel keeps its full occupancy
    extern __shared__ __attribute__((aligned(16))) double row_tiles[];
    const int64_t i = (int64_t)a.g0 + (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    // (the pointer is formed whatever the launch asked for: store_step_obs dereferences it exactly when step_rows_tiled() holds)
    if (i < a.g1) step_body<F, EP>(a, actions, t, normalized, reward, done, obs, log, i, row_tiles + (threadIdx.x >> 6) * (64 * ROW_TILE_MAX_D));
    advance_counter_in_kernel(a, 1);
}

// Dry run of one step: which requests would the reference refuse with raise_errors=True (base_module.py:79-93,213-224,
// 265-270)?  The step arithmetic runs on a register copy of the state; only the violations mask leaves the kernel.
template <int F>
__global__ __launch_bounds__(BLOCK) void check_kernel(const KArgs a, const void *__restrict__ actions, int32_t t, int normalized,
                                                      uint32_t *__restrict__ violations)
{
    t = resolve_t(a, t);
    const int64_t i = (int64_t)a.g0 + (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= a.g1) return;
    Params p; State s; Inputs in; Outputs o; Derived d;
    load_state<F>(a.c, i, true, s);
    load_params<F>(a.c, i, p);
    derive<F>(p, d);
    const int64_t row = series_row(a, i, t);
    if (a.act_f32) load_inputs<F>(a.c, (const float *)actions, a.N, i, row, in, a.pm_pitch);
    else load_inputs<F>(a.c, (const double *)actions, a.N, i, row, in, a.pm_pitch);
    step_core<F>(p, d, s, in, normalized != 0, false, false, o);
    violations[i] = o.violations;
}

// ------------------------------------------------------------------------------------------------------
// K fused steps: parameters + state live in registers; actions / series rows stream through a U-slot register
// ring (slot u is refilled with step k+U as soon as step k has been consumed, so U steps of loads are always in
// flight).  Every [K, N] stream is addressed as base + (k*N + i): one shared 64-bit lane offset, SGPR bases.
// ------------------------------------------------------------------------------------------------------
struct FusedOut {
    double *reward;
    uint8_t *done;
    double *soc_trace;
    uint32_t *status_trace;
    double *ret_acc;
    double *log;
};

// One ring slot: the controls stay in their storage type until the step consumes them (widening a float at load time
// makes the prefetch wait for its own data: measured 86 instead of 77 us per launch).
template <typename AT>
struct RawInputs {
    AT a_goal, a_gen, a_bat, a_grid;
    double load, pv, g_pimp, g_pexp, g_co2, g_stat;
};

template <int F, typename AT>
__device__ __forceinline__ Inputs widen(const RawInputs<AT> &r)
{
    Inputs in;
    if constexpr (F & F_GENSET) { in.a_goal = (double)r.a_goal; in.a_gen = (double)r.a_gen; }
    if constexpr (F & F_BATTERY) in.a_bat = (double)r.a_bat;
    if constexpr (F & F_GRID) {
        in.a_grid = (double)r.a_grid;
        in.g_pimp = r.g_pimp; in.g_pexp = r.g_pexp; in.g_co2 = r.g_co2; in.g_stat = r.g_stat;
    }
    in.load = r.load; in.pv = r.pv;
    return in;
}

template <int F, typename AT>
__device__ __forceinline__ void load_inputs_at(const AT *__restrict__ act, const double *__restrict__ lts,
                                               const double *__restrict__ pts, const double *__restrict__ gts,
                                               int64_t N, int64_t i, int64_t off, RawInputs<AT> &in)
{
    constexpr int A = 2 * ((F & F_GENSET) != 0) + ((F & F_BATTERY) != 0) + ((F & F_GRID) != 0);
    const AT *a = act + off * A;
    int k = 0;
    if constexpr (F & F_GENSET) { in.a_goal = a[k]; in.a_gen = a[k + 1]; k += 2; }
    if constexpr (F & F_BATTERY) { in.a_bat = a[k]; k += 1; }
    if constexpr (F & F_GRID) { in.a_grid = a[k]; k += 1; }
    in.load = lts[off];
    in.pv = pts[off];
    if constexpr (F & F_GRID) {
        const double *g = gts + (4 * off - 3 * i);                       // off = k*N + i  ->  (k*4)*N + i
        in.g_pimp = g[0]; in.g_pexp = g[N]; in.g_co2 = g[2 * N]; in.g_stat = g[3 * N];
    }
}

// ---- factorised series inside the fused kernels ------------------------------------------------------------------
// The base-profile rows a launch walks are wave-uniform: a workgroup copies rows [t0 + kb, t0 + kb + FACT_ROWS) of the load /
// pv (/ co2) tables into LDS (64 B per row and table, out of the caches: every workgroup reads the same 8 KB), and a lane
// picks its profile's column with one ds_read per value and step -- read one step ahead, so the LDS latency never meets a
// dependent instruction.  The per-grid factors (profile ids, ratios) are loaded once per launch.
#ifndef MGX_FACT_ROWS
#define MGX_FACT_ROWS 128
#endif
constexpr int FACT_ROWS = MGX_FACT_ROWS;              // rows per LDS chunk (a multiple of every ring depth)

template <int F>
__device__ __forceinline__ void stage_base_rows(const mgx_columns &c, int64_t row0, int32_t n, double *lds, int nthreads)
{
    const int32_t cnt = n * PP;
    const double *__restrict__ bl = c.base_load + row0 * PP;
    const double *__restrict__ bp = c.base_pv + row0 * PP;
    for (int32_t j = threadIdx.x; j < cnt; j += nthreads) {
        lds[j] = bl[j];
        lds[FACT_ROWS * PP + j] = bp[j];
        if constexpr (F & F_GRID) lds[2 * FACT_ROWS * PP + j] = c.base_co2[row0 * PP + j];
    }
}

// base values of one row, as read out of the LDS image (this lane's profile columns)
struct BaseVals { double load, pv, co2; };

template <int F>
__device__ __forceinline__ BaseVals read_base_row(const double *lds, int32_t r, const GridFactors &f)
{
    BaseVals b;
    b.load = lds[r * PP + f.lp];
    b.pv = lds[FACT_ROWS * PP + r * PP + f.pp];
    b.co2 = 0.0;
    if constexpr (F & F_GRID) b.co2 = lds[2 * FACT_ROWS * PP + r * PP + f.cp];
    return b;
}

// grid_status bits of the lane's grid: the word of the current 64-row block + the next one, fetched a block ahead
struct OutageWords {
    uint64_t cur, nxt;
    int64_t wi;
};

__device__ __forceinline__ void outage_init(const mgx_columns &c, int64_t N, int64_t i, int32_t T, int32_t t0, OutageWords &w)
{
    w.wi = t0 >> 6; w.cur = 0; w.nxt = 0;
    if (c.outage_bits) {
        w.cur = c.outage_bits[w.wi * N + i];
        if ((w.wi + 1) * 64 < T) w.nxt = c.outage_bits[(w.wi + 1) * N + i];
    }
}

// grid_status at series row `row` (rows are visited in order; `first`: the launch's first row)
__device__ __forceinline__ double outage_status(const mgx_columns &c, int64_t N, int64_t i, int32_t T, int32_t row, bool first,
                                                OutageWords &w)
{
    if (!first && (row & 63) == 0) {                 // wave-uniform
        w.cur = w.nxt; w.wi += 1; w.nxt = 0;
        if (c.outage_bits && (w.wi + 1) * 64 < T) w.nxt = c.outage_bits[(w.wi + 1) * N + i];
    }
    return ((w.cur >> (row & 63)) & 1ull) ? 0.0 : 1.0;
}

template <typename AT>
struct RawActions {
    AT a_goal, a_gen, a_bat, a_grid;
};

// `row`: the [N, A] block of one step (wave-uniform: an SGPR base), i32: the lane's grid
template <int F, typename AT>
__device__ __forceinline__ void load_actions_at(const AT *__restrict__ row, uint32_t i32, RawActions<AT> &r)
{
    constexpr int A = 2 * ((F & F_GENSET) != 0) + ((F & F_BATTERY) != 0) + ((F & F_GRID) != 0);
    const AT *a = row + i32 * (uint32_t)A;
    int k = 0;
#if MGX_ACT_NT
    if constexpr (F & F_GENSET) { r.a_goal = __builtin_nontemporal_load(a + k); r.a_gen = __builtin_nontemporal_load(a + k + 1); k += 2; }
    if constexpr (F & F_BATTERY) { r.a_bat = __builtin_nontemporal_load(a + k); k += 1; }
    if constexpr (F & F_GRID) { r.a_grid = __builtin_nontemporal_load(a + k); k += 1; }
#else
    if constexpr (F & F_GENSET) { r.a_goal = a[k]; r.a_gen = a[k + 1]; k += 2; }
    if constexpr (F & F_BATTERY) { r.a_bat = a[k]; k += 1; }
    if constexpr (F & F_GRID) { r.a_grid = a[k]; k += 1; }
#endif
}

// `done` of one fused step: a byte per grid, or (KArgs.done_bits) a bit per grid in uint16 words -- lane (i & 15) == 0 of every
// 16-grid group stores the group's word.  Workgroup bases are multiples of 16 grids, so a group never straddles two waves.
__device__ __forceinline__ void store_done(const KArgs &a, uint8_t *__restrict__ done, int64_t off, int64_t i, int32_t k, bool dn)
{
    if (a.done_bits) {
        const unsigned long long b = __ballot(dn);
        const int lane = threadIdx.x & 63;
        if ((lane & 15) == 0) {
            const int64_t W = ((int64_t)a.N + 15) >> 4;
            reinterpret_cast<uint16_t *>(done)[(int64_t)k * W + (i >> 4)] = (uint16_t)(b >> (lane & 48));
        }
    } else {
        done[off] = (uint8_t)dn;
    }
}

// RICH = false: the launch writes neither log rows nor the status trace -- compiled out, together with every value only
// they consume (balance sums, co2, the violations mask): the lean form is the hot one (reward / done / SoC streams).
// FACT: factorised series (mgx_columns.base_load): the ring holds the controls only, the series rows are formed from LDS.
template <int F, int U, typename AT, bool RICH, bool FACT>
__global__ __launch_bounds__(BLOCK_K) void step_k_kernel(const KArgs a, const AT *__restrict__ actions, int32_t t0,
                                                         int32_t K, int normalized, const FusedOut out_rt, int32_t gpb)
{
    FusedOut out = out_rt;
    if constexpr (!RICH) { out.log = nullptr; out.status_trace = nullptr; }
    const int32_t K_launch = K;          // what the host asked for (the counter always moves by this much)
    t0 = resolve_t(a, t0);
    K = resolve_k(a, t0, K);
    // gpb = grids per workgroup (<= BLOCK_K, multiple of 16 = one 128-B line of doubles): chosen by the host so that
    // the busiest CU streams as few grids as possible (fused_grids_per_block)
    const int64_t i = (int64_t)a.g0 + (int64_t)blockIdx.x * gpb + threadIdx.x;
    const bool active = (int32_t)threadIdx.x < gpb && i < a.g1;
    if constexpr (!FACT) { if (!active) return; }
    const int64_t N = a.N;
    constexpr int A_DIM = 2 * ((F & F_GENSET) != 0) + ((F & F_BATTERY) != 0) + ((F & F_GRID) != 0);
    Params p; State s; Derived d;
    const bool norm = normalized != 0;
    const bool want_soc = (out.soc_trace != nullptr) || (out.log != nullptr);
    bool gen_instant = false;
    int32_t k_done = 0;
    if (active) {
        load_state<F>(a.c, i, out.log != nullptr, s);
        load_params<F>(a.c, i, p);
        derive<F>(p, d);
        gen_instant = genset_wave_is_instant<F>(p, s);
        k_done = (a.grid_final ? a.grid_final[i] : a.final_step) - 1 - t0;      // done <=> k >= k_done
    }
    double ret = 0.0;

    // The step loop exists in four forms, picked once per launch by wave-uniform flags and specialised at COMPILE time:
    //   GI   every genset of the wave is instantaneous (no start-up / wind-down delay, equilibrium status): the status FSM
    //        (genset_update_status) is not in the loop at all;
    //   HOT  the launch writes exactly reward + SoC per step (the throughput configuration: lock-step `done` follows from the
    //        counter, no traces, no log): no per-step tests of the output pointers.
    // (Without the specialisation the loop body is ~34 basic blocks per step; the scheduler cannot move loads across them.)
    const bool hot = !RICH && out.reward != nullptr && out.done == nullptr && (!(F & F_BATTERY) || out.soc_trace != nullptr) &&
                     a.shaper == MGX_SHAPER_NONE;
    const uint32_t i32 = (uint32_t)i;             // lane offset of every [K, N] stream: row base in SGPRs + this (no 64-bit VALU
                                                  // address arithmetic per step and stream)
    auto consume = [&](auto gi_tag, auto hot_tag, const Inputs &in, int32_t k, int64_t off) __attribute__((always_inline)) {
        constexpr bool GI = decltype(gi_tag)::value, HOT = decltype(hot_tag)::value;
        Outputs o;
        step_core<F>(p, d, s, in, norm, HOT ? (F & F_BATTERY) != 0 : want_soc, GI, o);
        if constexpr (HOT) {
            const double r = o.reward;
#if MGX_HOT_NT
            __builtin_nontemporal_store(r, (out.reward + (int64_t)k * N) + i32);
            if constexpr (F & F_BATTERY) __builtin_nontemporal_store(s.soc, (out.soc_trace + (int64_t)k * N) + i32);
#else
            (out.reward + (int64_t)k * N)[i32] = r;
            if constexpr (F & F_BATTERY) (out.soc_trace + (int64_t)k * N)[i32] = s.soc;
#endif
            ret += r;
            return;
        }
        const double r = shaped_reward<F>(a.shaper, o);
        {
#if MGX_OUT_NT
            if (out.reward) __builtin_nontemporal_store(r, out.reward + off);
            if (out.done) store_done(a, out.done, off, i, k, k >= k_done);
            if constexpr (F & F_BATTERY) { if (out.soc_trace) __builtin_nontemporal_store(s.soc, out.soc_trace + off); }
            if constexpr (F & F_GENSET) { if (out.status_trace) __builtin_nontemporal_store(s.status, out.status_trace + off); }
#else
            if (out.reward) out.reward[off] = r;
            if (out.done) store_done(a, out.done, off, i, k, k >= k_done);
            if constexpr (F & F_BATTERY) { if (out.soc_trace) out.soc_trace[off] = s.soc; }
            if constexpr (F & F_GENSET) { if (out.status_trace) out.status_trace[off] = s.status; }
#endif
            if (out.log) store_log<F>(out.log + (off - i) * a.log_dim + i, N, o, s.status);
        }
        ret += r;
    };
    // run fn(gi_tag, hot_tag) in the form the wave's flags select
    auto specialised = [&](auto fn) __attribute__((always_inline)) {
        if constexpr ((F & F_GENSET) != 0) {
            if (gen_instant) { if (hot) fn(std::true_type{}, std::true_type{}); else fn(std::true_type{}, std::false_type{}); }
            else { if (hot) fn(std::false_type{}, std::true_type{}); else fn(std::false_type{}, std::false_type{}); }
        } else {
            if (hot) fn(std::false_type{}, std::true_type{}); else fn(std::false_type{}, std::false_type{});
        }
    };

    if constexpr (FACT) {
        __shared__ double base_lds[((F & F_GRID) ? 3 : 2) * FACT_ROWS * PP];
        static_assert(FACT_ROWS % U == 0, "ring slots are addressed by k % U across LDS chunks");
        GridFactors f;
        f.lr = 0.0; f.pr = 0.0; f.lp = 0u; f.pp = 0u; f.cp = 0u; f.pat = 0u;
        OutageWords ow;
        ow.cur = 0; ow.nxt = 0; ow.wi = 0;
        RawActions<AT> ring[U];
        if (active) {
            load_factors<F>(a.c, i, f);
            if constexpr (F & F_GRID) outage_init(a.c, N, i, a.T, t0, ow);
#pragma unroll
            for (int u = 0; u < U; u++)
                if (u < K) load_actions_at<F>(actions + (int64_t)u * N * A_DIM, i32, ring[u]);
        }
        int64_t off = i;                                     // k*N + i
        // the steps of one LDS chunk (the barriers around a chunk stay at kernel scope: every wave of the workgroup meets
        // the same ones, whichever form of the loop it runs)
        auto run_chunk = [&](auto gi_tag, auto hot_tag, int32_t kb, int32_t n) __attribute__((always_inline)) {
            BaseVals nb = read_base_row<F>(base_lds, 0, f);
            for (int32_t k0 = kb; k0 < kb + n; k0 += U) {
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int32_t k = k0 + u;
                    if (k < kb + n) {
                        const RawActions<AT> ra = ring[u];
                        if (k + U < K) load_actions_at<F>(actions + (int64_t)(k + U) * N * A_DIM, i32, ring[u]);
                        Inputs in;
                        if constexpr (F & F_GENSET) { in.a_goal = (double)ra.a_goal; in.a_gen = (double)ra.a_gen; }
                        if constexpr (F & F_BATTERY) in.a_bat = (double)ra.a_bat;
                        in.load = fact_load(nb.load, f.lr);
                        in.pv = fact_pv(nb.pv, f.pr);
                        if constexpr (F & F_GRID) {
                            in.a_grid = (double)ra.a_grid;
                            in.g_pimp = tariff_price((int32_t)f.pat, t0 + k); in.g_pexp = 0.0;
                            in.g_co2 = nb.co2;
                            in.g_stat = outage_status(a.c, N, i, a.T, t0 + k, k == 0, ow);
                        }
                        const int32_t rn = (k + 1 - kb < n) ? k + 1 - kb : n - 1;      // next step's row (LDS, one step ahead)
                        nb = read_base_row<F>(base_lds, rn, f);
                        consume(gi_tag, hot_tag, in, k, off);
                        off += N;
                    }
                }
            }
        };
        for (int32_t kb = 0; kb < K; kb += FACT_ROWS) {      // K is uniform over the launch: the barriers are safe
            const int32_t n = K - kb < FACT_ROWS ? K - kb : FACT_ROWS;
            __syncthreads();                                 // the previous chunk's rows have been consumed
            stage_base_rows<F>(a.c, (int64_t)t0 + kb, n, base_lds, BLOCK_K);
            __syncthreads();
            if (active) specialised([&](auto gi_tag, auto hot_tag) __attribute__((always_inline)) { run_chunk(gi_tag, hot_tag, kb, n); });
        }
        if (!active) return;                                 // (no barrier follows: advance_counter_in_kernel's is skipped
    } else {                                                 //  by returned waves, as in the materialised form)
        // series bases moved to row t0 once (scalar), so row k of this launch is base + k*N
        const double *__restrict__ lts = a.c.load_ts + (int64_t)t0 * N;
        const double *__restrict__ pts = a.c.pv_ts + (int64_t)t0 * N;
        const double *__restrict__ gts = (F & F_GRID) ? a.c.grid_ts + (int64_t)t0 * 4 * N : nullptr;
        specialised([&](auto gi_tag, auto hot_tag) __attribute__((always_inline)) {
            RawInputs<AT> ring[U];
#pragma unroll
            for (int u = 0; u < U; u++)
                if (u < K) load_inputs_at<F>(actions, lts, pts, gts, N, i, (int64_t)u * N + i, ring[u]);

            int64_t off = i;                                         // k*N + i
            for (int32_t k0 = 0; k0 < K; k0 += U) {
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int32_t k = k0 + u;
                    if (k < K) {
                        const Inputs in = widen<F>(ring[u]);
                        if (k + U < K) load_inputs_at<F>(actions, lts, pts, gts, N, i, off + (int64_t)U * N, ring[u]);
                        consume(gi_tag, hot_tag, in, k, off);
                        off += N;
                    }
                }
            }
        });
    }
    if constexpr (F & F_BATTERY) { if (!want_soc) s.soc = s.charge / p.bat_cmax; }
    store_state<F>(a.c, i, s);
    if (out.ret_acc) out.ret_acc[i] += ret;
    advance_counter_in_kernel(a, K_launch);
}

// Launch of a fused kernel as the host side hands it to the slices of mgx_fused.hip (each compiles the kernels of a few
// layouts and takes the launches whose `flags` it owns).
struct FusedLaunch {
    int flags;                       // layout (template parameter F)
    bool act_f32, rich, fact, per_step;
    unsigned blocks;
    hipStream_t stream;
    const KArgs *k;
    const void *actions;             // step_k: controls [K, N, A]
    const PLWords *tab;              // rollout: priority-list table
    const uint8_t *ids;              // rollout: list ids [K, N] or [N]
    int32_t t, K;
    int normalized;
    FusedOut out;
    int32_t gpb;
};
constexpr int MGX_FUSED_PARTS = 6;
// part 5: step_k_multi_small_kernel with COMPILE-TIME instance counts for the layouts listed there; false = no such specialisation
struct MultiStaticLaunch {
    int flags, ng, nb, nr, nl, np;
    unsigned blocks;
    hipStream_t stream;
    const KArgs *k;
    const void *actions;
    int32_t t, K;
    int normalized;
    FusedOut out;
};
bool launch_step_k_multi_static(const MultiStaticLaunch &L);
// ... and mgx_rollout_lists on the same specialisations (rollout_multi_small_kernel)
struct MultiStaticRollout {
    int flags, ng, nb, nr, nl, np;
    unsigned blocks;
    hipStream_t stream;
    const KArgs *k;
    const int32_t *lists;
    int32_t n_lists, list_len;
    const int32_t *ids;
    int per_step;
    int32_t t, K;
    FusedOut out;
};
bool launch_rollout_multi_static(const MultiStaticRollout &L);
bool launch_step_k_p0(const FusedLaunch &L); bool launch_step_k_p1(const FusedLaunch &L); bool launch_step_k_p2(const FusedLaunch &L);
bool launch_step_k_p3(const FusedLaunch &L); bool launch_step_k_p4(const FusedLaunch &L);
bool launch_rollout_p0(const FusedLaunch &L); bool launch_rollout_p1(const FusedLaunch &L); bool launch_rollout_p2(const FusedLaunch &L);
bool launch_rollout_p3(const FusedLaunch &L); bool launch_rollout_p4(const FusedLaunch &L);

// ------------------------------------------------------------------------------------------------------
// Observation of the current state (reset(), or after step_k).
// ------------------------------------------------------------------------------------------------------
template <int F>
__global__ __launch_bounds__(BLOCK) void observe_kernel(const KArgs a, int32_t t, void *__restrict__ obs)
{
    t = resolve_t_obs(a, t);
    const int64_t i = (int64_t)a.g0 + (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= a.g1) return;
    Params p; State s;
    load_state<F>(a.c, i, true, s);
    load_params<F>(a.c, i, p);
    store_step_obs<F>(a, obs, i, t + (a.ep_off ? a.ep_off[i] : 0), p, s, a.pm_pitch);   // whole rows for H == 0 only (the host dispatches)
}

// Observation rows for H > 0 (obs_rows_wave_kernel below).  Every cache line of obs is written whole by one wave
// (partial-line writes from different waves / XCDs cost read-modify-write at the memory side; measured 3x slower).
struct WindowPlan {
    int32_t grid_col_base;   // first obs column of the grid window
    int32_t ld;              // LDS row pitch in doubles (odd: conflict-free column writes)
    int32_t group;           // grids per wave tile: 16, or 8 / 4 / 2 / 1 when a 16-row tile would not fit the LDS
};

// Wave-private row tiles: one 64-lane workgroup per G = plan.group (16) grids.  The wave gathers the windows of its
// grids into an LDS tile [G][LD] (lane = grid x horizon phase), then writes the G rows -- G*D consecutive doubles of
// obs -- with full-wave 16-byte non-temporal stores (the rows are write-once; keeping them out of the caches leaves
// the window rows, which the next 24 steps read again, resident in the 256 MB MALL: measured 65 -> 53 us at D = 156).
// No workgroup is ever waiting for another wave's phase, so load, arithmetic and store phases of different waves
// overlap on a CU (8 tiles of 20 KB per CU at D = 156).
template <int F, bool NOISE, typename OT>
__global__ __launch_bounds__(64) void obs_rows_wave_kernel(const KArgs a, const WindowPlan plan, int32_t t,
                                                           OT *__restrict__ obs)
{
    t = resolve_t_obs(a, t);
    extern __shared__ double tile_raw[];                // [plan.group][plan.ld] of OT
    OT *tile = reinterpret_cast<OT *>(tile_raw);
    const int lane = threadIdx.x;
    const int32_t G = plan.group, Q = 64 / G;
    const int32_t g = lane & (G - 1), q = lane / G;
    const int64_t g0 = (int64_t)blockIdx.x * G;
    const int64_t N = a.N;
    const int32_t W = 1 + a.H, D = a.obs_dim, LD = plan.ld;
    const int64_t i = g0 + g, ic = i < N ? i : g0;
    OT *row = tile + g * LD;
    const int32_t slots = OBS_JB * Q;
    const int32_t W_pad = (W + slots - 1) / slots * slots;
    // wave-uniform: every slot's row exists, lane offsets fit 32-bit byte offsets (in-place episodes: every grid its own rows)
    const int32_t tr = t & a.row_mask;                  // rolling windows: the buffers are rings of row_mask + 1 rows
    const bool fact = factorised(a.c);                  // factorised series: the general form below (rows out of the caches)
    const bool fast = !fact && !a.ep_off && t >= 0 && (int64_t)t + W_pad <= a.T && (int64_t)(4 * Q + 4) * N < (int64_t(1) << 28) &&
                      (a.row_mask == -1 || tr + W_pad <= a.row_mask + 1);               // ... and this window does not wrap
    if (fast) {
        WinBounds<1> bl, bp;
        WinBounds<(F & F_GRID) ? 4 : 1> bg;
        window_bounds<1>(a.c.load_lo, a.c.load_hi, N, ic, bl);
        window_bounds<1>(a.c.pv_lo, a.c.pv_hi, N, ic, bp);
        if constexpr (F & F_GRID) window_bounds<4>(a.c.grid_lo, a.c.grid_hi, N, ic, bg);
        for (int32_t hb = 0; hb < W; hb += slots) {      // one round for the usual 24 / 25-step windows
            double vl[OBS_JB][1], vp[OBS_JB][1], vg[OBS_JB][(F & F_GRID) ? 4 : 1];
            window_issue<1>(a.c.load_ts, N, N, tr, hb, Q, (uint32_t)(q * N + ic), vl);    // all loads of the round in flight
            window_issue<1>(a.c.pv_ts, N, N, tr, hb, Q, (uint32_t)(q * N + ic), vp);
            if constexpr (F & F_GRID) window_issue<4>(a.c.grid_ts, N, 4 * N, tr, hb, Q, (uint32_t)(q * 4 * N + ic), vg);
            if (hb == 0) window_bounds_finish<1>(bl);
            window_finish<1, NOISE, OT>(vl, bl, W, t, hb, i, ic, q, Q, row + a.col_load, a.c.load_noise_std, 0u, a.noise_seed, a.noise_increase);
            if (hb == 0) window_bounds_finish<1>(bp);
            window_finish<1, NOISE, OT>(vp, bp, W, t, hb, i, ic, q, Q, row + a.col_pv, a.c.pv_noise_std, 1u, a.noise_seed, a.noise_increase);
            if constexpr (F & F_GRID) {
                if (hb == 0) window_bounds_finish<4>(bg);
                window_finish<4, NOISE, OT>(vg, bg, W, t, hb, i, ic, q, Q, row + a.col_grid, a.c.grid_noise_std, 2u,
                                        a.noise_seed, a.noise_increase);
            }
        }
    } else if (fact) {
        GridFactors f;
        load_factors<F>(a.c, ic, f);
        const mgx_columns &c = a.c;
        const int32_t ti = t + (a.ep_off ? a.ep_off[ic] : 0);         // in-place episodes: the grid's own series row
        const int32_t pm = a.pm_pitch;
        observe_window_cols<1, NOISE, OT>([&](int32_t r, int) { return fact_load(c.base_load[base_index(pm, r, f.lp)], f.lr); }, N,
                                          a.c.load_lo, a.c.load_hi, a.T, ti, W, i, ic, q, Q, row + a.col_load, a.c.load_noise_std, 0u,
                                          a.noise_seed, a.noise_increase, a.row_mask);
        observe_window_cols<1, NOISE, OT>([&](int32_t r, int) { return fact_pv(c.base_pv[base_index(pm, r, f.pp)], f.pr); }, N,
                                          a.c.pv_lo, a.c.pv_hi, a.T, ti, W, i, ic, q, Q, row + a.col_pv, a.c.pv_noise_std, 1u,
                                          a.noise_seed, a.noise_increase, a.row_mask);
        if constexpr (F & F_GRID)
            observe_window_cols<4, NOISE, OT>([&](int32_t r, int cc) { return series_component(c, N, 2 + cc, r, ic, pm); }, N,
                                              a.c.grid_lo, a.c.grid_hi, a.T, ti, W, i, ic, q, Q, row + a.col_grid,
                                              a.c.grid_noise_std, 2u, a.noise_seed, a.noise_increase, a.row_mask);
    } else {
        const double *lts = a.c.load_ts, *pts = a.c.pv_ts, *gts = a.c.grid_ts;
        const int32_t ti = t + (a.ep_off ? a.ep_off[ic] : 0);         // in-place episodes: the grid's own series row
        const int32_t pm = a.pm_pitch;                                // ... out of the grid-major copies
        constexpr int C = (F & F_GRID) ? 6 : 2;
        observe_window_cols<1, NOISE, OT>([&](int32_t r, int) { return lts[ts_index(pm, N, r, ic, C)]; }, N,
                                          a.c.load_lo, a.c.load_hi, a.T, ti, W, i, ic, q, Q, row + a.col_load, a.c.load_noise_std, 0u,
                                          a.noise_seed, a.noise_increase, a.row_mask);
        observe_window_cols<1, NOISE, OT>([&](int32_t r, int) { return pts[ts_index(pm, N, r, ic, C)]; }, N,
                                          a.c.pv_lo, a.c.pv_hi, a.T, ti, W, i, ic, q, Q, row + a.col_pv, a.c.pv_noise_std, 1u,
                                          a.noise_seed, a.noise_increase, a.row_mask);
        if constexpr (F & F_GRID)
            observe_window_cols<4, NOISE, OT>([&](int32_t r, int cc) { return gts[grid_ts_index(pm, N, r, cc, ic)]; }, N,
                                              a.c.grid_lo, a.c.grid_hi, a.T, ti, W, i, ic, q, Q, row + a.col_grid,
                                              a.c.grid_noise_std, 2u, a.noise_seed, a.noise_increase, a.row_mask);
    }
    if (q == 0) {                                        // the 6 state columns, by the first lane of each grid
        Params p; State s;
        load_state<F>(a.c, ic, true, s);
        load_params<F>(a.c, ic, p);
        observe_state_cols<F, OT>(a, p, s, row);
    }
    __syncthreads();
    const int32_t n_valid = (N - g0 < G) ? (int32_t)(N - g0) : G;
    const int32_t total = n_valid * D;                   // D is even here (one load, one renewable module)
    OT *out = obs + g0 * D;
    typedef OT vec2 __attribute__((ext_vector_type(2)));
    if ((reinterpret_cast<uintptr_t>(out) & (sizeof(vec2) - 1)) == 0) {   // element pair f, f + 1 = 2 lane + 128 j -> (row, column)
        int32_t r = 2 * lane / D, c = 2 * lane - r * D;
        for (int32_t f = 2 * lane; f < total; f += 128) {
            vec2 v2;
            v2.x = tile[r * LD + c];
            v2.y = tile[r * LD + c + 1];                 // D even, c even: the pair never straddles two rows
            __builtin_nontemporal_store(v2, reinterpret_cast<vec2 *>(out + f));
            c += 128;
            while (c >= D) { c -= D; r++; }
        }
    } else {
        int32_t r = lane / D, c = lane - r * D;
        for (int32_t f = lane; f < total; f += 64) {
            __builtin_nontemporal_store(tile[r * LD + c], out + f);
            c += 64;
            while (c >= D) { c -= D; r++; }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// Window prefetch: the observation rows of the NEXT K steps in one launch (mgx_observe_windows).
// The window columns of an observation depend on the series only -- never on the actions -- and the windows of
// consecutive steps overlap in H of their 1 + H rows.  A per-step kernel therefore re-reads (and re-normalises) every
// series value 1 + H times, and those reads miss the 4 MiB L2s (per-XCD window working set 15 MB at N = 100k): they are
// the bound of obs_rows_wave_kernel.  Here a wave reads rows t .. t+K-1+H of its 16 grids ONCE, normalises each value
// ONCE into LDS (clipped / padded form for forecast positions, unclipped form for the "current value" position), and
// writes K row blocks ring[k] (k = 0..K-1) as shifted copies -- with 288 GB of HBM the K*N*D ring is cheap (1 GB at
// K = 8, N = 100k, D = 156).  Block 0 is complete (state columns of the current state); in blocks 1..K-1 the state
// columns are zero and are filled in by the step that reaches them (obs_state_only mode of the step kernels).
// Not offered with forecast noise (noise depends on (t, h), not on t + h: nothing to share).
// ------------------------------------------------------------------------------------------------------
// Workgroup = 4 waves around ONE LDS image of 16 grids: per grid a block of BP doubles
//   [NCOMP][RP]  normalised rows t .. t+R-1 in forecast form (clipped to the bounds, padded beyond the series)
//   [NCOMP][K]   rows t .. t+K-1 in "current value" form (unclipped)
//   [6][K]       state columns: entry 0 = the current state, entries 1..K-1 = 0
// so that output element (block k, grid r, column c) = image[r*BP + map[c] + k] for EVERY kind of column (map[c] =
// offset of the column's k = 0 entry).  Thread (g, q) of the 256 loads rows q, q+16, ... of grid g; wave w then writes
// blocks w, w+4, ... (each 16*D consecutive elements) with 16-byte non-temporal stores.
#ifndef MGX_OBS_KJ
#define MGX_OBS_KJ 2
#endif
#ifndef MGX_WIN_UNROLL
#define MGX_WIN_UNROLL 4        // (1 / 4 / 8 measured: 4 is the best of the three for double rows alone, 263 vs 287 us per refill)
#endif
constexpr int OBS_KJ = MGX_OBS_KJ;                      // rows per thread and latency round (x 16 phases = 32 rows)
constexpr int WIN_U = MGX_WIN_UNROLL;                   // blocks per lane whose image words are read before the first of their stores
constexpr int OBS_K_THREADS = 256;                      // threads that write the ring (phase 2)
constexpr int OBS_P1_THREADS = 1024;                    // most threads a refill launch may have: all of them build the image (phase 1)

// IT: element type of the LDS image -- double, or float where the rows leave as floats (the value is rounded to float once, here,
// instead of at every one of its 1 + H stores: the same bits, half the LDS, twice the grids per workgroup)
template <int NC, typename IT, class Fetch>
__device__ __forceinline__ void windows_k_module(Fetch fetch, int64_t N,
                                                 const double *__restrict__ lo_col, const double *__restrict__ hi_col,
                                                 int32_t T, int32_t t, int32_t R, int32_t K, int64_t ic, int32_t q, int32_t Q,
                                                 IT *nc /* [NC][RP] of this grid */, IT *nu /* [NC][K] */, int32_t RP,
                                                 int32_t row_mask = -1)
{
    double lo[NC], hi[NC], sp[NC], z_lo[NC], z_hi[NC], z_fill[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        lo[c] = lo_col[c * N + ic]; hi[c] = hi_col[c * N + ic];
        sp[c] = space_spread(lo[c], hi[c]);
        z_lo[c] = (lo[c] - lo[c]) / sp[c];                               // a forecast clipped to the lower bound
        z_hi[c] = (hi[c] - lo[c]) / sp[c];                               // ... to the upper bound
        z_fill[c] = ((hi[c] + lo[c]) / 2 - lo[c]) / sp[c];               // a row beyond the series (forecaster.py:95,120-137)
    }
    for (int32_t rb = 0; rb < R; rb += OBS_KJ * Q) {                     // workgroup-uniform trip count
        double v[OBS_KJ][NC];
#pragma unroll
        for (int jj = 0; jj < OBS_KJ; jj++) {                            // unconditional, clamped loads: one latency round
            const int32_t rr = rb + q + Q * jj;                          // rows past the window re-read its last row (a cache
            const int32_t r = t + (rr < R ? rr : R - 1);                 // hit) instead of pulling unused rows out of HBM
            const int32_t rc = (r < T ? (r < 0 ? 0 : r) : T - 1) & row_mask;      // rolling windows: the series buffers are rings
#pragma unroll
            for (int c = 0; c < NC; c++) v[jj][c] = fetch(rc, c);
        }
#pragma unroll
        for (int jj = 0; jj < OBS_KJ; jj++) {
            const int32_t rr = rb + q + Q * jj;                          // row relative to t
            if (rr < R) {
                const bool in = t + rr < T;
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    const double x = v[jj][c];
                    const double n_u = in ? (x - lo[c]) / sp[c] : z_fill[c];                 // unclipped (current value)
                    const double n_c = in ? (x < lo[c] ? z_lo[c] : (x > hi[c] ? z_hi[c] : n_u)) : z_fill[c];
                    nc[c * RP + rr] = (IT)n_c;
                    if (rr < K) nu[c * K + rr] = (IT)n_u;
                }
            }
        }
    }
}

struct WindowsKPlan {
    int32_t grid_col_base;   // first obs column of the grid window
    int32_t group;           // grids per workgroup (16)
    int32_t K;               // steps per launch
    int32_t rp;              // pitch of one component's rows in the image
    int32_t bp;              // pitch of one grid's block in the image (odd)
    int32_t with_state;      // block 0 receives the state columns of the current state (0: a prefetch AHEAD of the counter)
    int32_t group0;          // first group of this launch (a launch may cover a chunk of the batch's groups)
    int32_t pitch;           // rows between consecutive blocks of the ring (>= N; mgx_set_ring_pitch)
    int32_t pairs;           // column-major blocks: a lane stores a PAIR of adjacent grids per instruction (windows_plan)
};

#ifdef MGX_WIN_PLAIN_STORES
#define MGX_WIN_STORE(v, p) (*(p) = (v))
#else
#define MGX_WIN_STORE(v, p) __builtin_nontemporal_store((v), (p))
#endif
// Phase 2 of a refill into COLUMN-major blocks: a lane carries a PAIR of adjacent grids of one column (adjacent words of the block)
// through the K blocks -- 16-byte stores of doubles, 8-byte stores of floats: half the store instructions of a word per lane for the
// same whole lines.  (Four floats per lane, 16-byte stores too, measured no better than two: profiles/r05/exp_refill_col_packs.txt.)
// map[c] >= skip_from: a state column a ring written ahead leaves to the steps.  Returns false (nothing stored) where the ring is
// not aligned for the pair.
template <typename OT, typename IT>
__device__ __forceinline__ bool store_colmajor_packs(const IT *image, int32_t BP, const uint32_t *map, int32_t D, int32_t skip_from,
                                                     int32_t K, int32_t G, int64_t g0, int64_t N, int64_t P, OT *__restrict__ ring, int tid)
{
    constexpr int NPK = 2;
    typedef OT vecp __attribute__((ext_vector_type(NPK)));
    if ((reinterpret_cast<uintptr_t>(ring) & (sizeof(vecp) - 1)) != 0 || G < NPK) return false;
    const int32_t GP = G / NPK, gp = tid & (GP - 1), cq = tid / GP, QP = OBS_K_THREADS / GP;
    const int64_t i0 = g0 + (int64_t)NPK * gp, left = N - i0;
    const int nv = left >= NPK ? NPK : (left > 0 ? (int)left : 0);     // grids of the pack inside the batch (the last group is ragged)
    const IT *img = image + (NPK * gp) * BP;
    OT *outp = ring + i0;
    const int64_t kstride = (int64_t)D * P;
    for (int32_t c = cq; c < D; c += QP) {
        const uint32_t m = map[c];
        if (m >= (uint32_t)skip_from) continue;
        const IT *src = img + m;
        OT *o = outp + (int64_t)c * P;
        int32_t k = 0;
        if constexpr (WIN_U > 1) {                       // WIN_U packs of image words in flight before the first store of the batch
            for (; k + WIN_U <= K; k += WIN_U) {
                IT v[WIN_U][NPK];
#pragma unroll
                for (int u = 0; u < WIN_U; u++)
#pragma unroll
                    for (int e = 0; e < NPK; e++) v[u][e] = src[e * BP + k + u];
#pragma unroll
                for (int u = 0; u < WIN_U; u++) {
                    OT *dst = o + (int64_t)(k + u) * kstride;
                    if (nv == NPK) {
                        vecp vv;
#pragma unroll
                        for (int e = 0; e < NPK; e++) vv[e] = (OT)v[u][e];
                        MGX_WIN_STORE(vv, reinterpret_cast<vecp *>(dst));
                    } else {
#pragma unroll
                        for (int e = 0; e < NPK; e++) if (e < nv) MGX_WIN_STORE((OT)v[u][e], dst + e);
                    }
                }
            }
        }
        for (; k < K; k++) {
            OT *dst = o + (int64_t)k * kstride;
            if (nv == NPK) {
                vecp vv;
#pragma unroll
                for (int e = 0; e < NPK; e++) vv[e] = (OT)src[e * BP + k];
                MGX_WIN_STORE(vv, reinterpret_cast<vecp *>(dst));
            } else {
#pragma unroll
                for (int e = 0; e < NPK; e++) if (e < nv) MGX_WIN_STORE((OT)src[e * BP + k], dst + e);
            }
        }
    }
    return true;
}

// Body shared by obs_windows_k_kernel and the window part of fleet_step_kernel: workgroup `group` (16 grids) of the batch.
// GRID: the layout has a GridModule (6 instead of 2 series components).  `now` (meaningful in the q == 0 lanes, with have_now):
// the state columns of the current state for block 0; without it (a prefetch ahead of the counter) every state column is zero.
template <bool GRID, typename OT>
__device__ __forceinline__ void windows_body(const KArgs &a, const WindowsKPlan &plan, int32_t t, OT *__restrict__ ring,
                                             int64_t group, int32_t nstate, bool have_now, const double (&now)[6], double *image_raw)
{
    typedef typename std::conditional<sizeof(OT) == 4, float, double>::type IT;       // float rows: a float image (windows_k_module)
    IT *image = reinterpret_cast<IT *>(image_raw);
    constexpr int NCOMP = 2 + (GRID ? 4 : 0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Two phases with their own thread counts.  Phase 1 (series rows -> normalised image in LDS) is a chain of dependent loads,
    // divisions and LDS stores per value: with one wave per SIMD nothing hides its latencies -- it took 84-99 of the 200-290 us of
    // a refill (profiles/r05/exp_refill_variants.txt: the "nostore" build) -- so ALL blockDim.x threads of the launch share it
    // (1 024 where the refill runs alone).  Phase 2 (image -> ring stores) stays with the first OBS_K_THREADS threads: how hard a
    // refill leans on the memory system decides what the step launches beside it cost (one workgroup of four storing waves per CU).
    const int32_t NT = (int32_t)blockDim.x;
    const int32_t G = plan.group, K = plan.K, RP = plan.rp, BP = plan.bp;
    const int32_t g = tid & (G - 1), q = tid / G, Q = NT / G;           // phase 1: thread (g, q) of Q per grid
    const int64_t g0 = group * G;
    const int64_t N = a.N;
    const int32_t W = 1 + a.H, D = a.obs_dim, R = K + a.H;
    const int64_t i = g0 + g, ic = i < N ? i : g0;
    const int32_t NU0 = NCOMP * RP, S0 = NU0 + NCOMP * K;
    IT *blk = image + g * BP;
    uint32_t *map = reinterpret_cast<uint32_t *>(image + G * BP);       // [D]

    if (factorised(a.c)) {                               // uniform over the launch: rows formed from the base tables
        GridFactors f;
        load_factors<GRID ? F_GRID : 0>(a.c, ic, f);
        const mgx_columns &c = a.c;
        const int32_t ti = t + (a.ep_off ? a.ep_off[ic] : 0);          // in-place episodes: the grid's own series row
        const int32_t pm = a.pm_pitch;
        windows_k_module<1, IT>([&](int32_t r, int) { return fact_load(c.base_load[base_index(pm, r, f.lp)], f.lr); }, N,
                            a.c.load_lo, a.c.load_hi, a.T, ti, R, K, ic, q, Q, blk, blk + NU0, RP, a.row_mask);
        windows_k_module<1, IT>([&](int32_t r, int) { return fact_pv(c.base_pv[base_index(pm, r, f.pp)], f.pr); }, N,
                            a.c.pv_lo, a.c.pv_hi, a.T, ti, R, K, ic, q, Q, blk + RP, blk + NU0 + K, RP, a.row_mask);
        if constexpr (GRID)
            windows_k_module<4, IT>([&](int32_t r, int cc) { return series_component(c, N, 2 + cc, r, ic, pm); }, N,
                                a.c.grid_lo, a.c.grid_hi, a.T, ti, R, K, ic, q, Q, blk + 2 * RP, blk + NU0 + 2 * K, RP, a.row_mask);
    } else {
        const double *lts = a.c.load_ts, *pts = a.c.pv_ts, *gts = a.c.grid_ts;
        const int32_t ti = t + (a.ep_off ? a.ep_off[ic] : 0);          // in-place episodes: the grid's own series row
        const int32_t pm = a.pm_pitch;                                 // ... out of the grid-major copies
        constexpr int C = GRID ? 6 : 2;
        windows_k_module<1, IT>([&](int32_t r, int) { return lts[ts_index(pm, N, r, ic, C)]; }, N,
                            a.c.load_lo, a.c.load_hi, a.T, ti, R, K, ic, q, Q, blk, blk + NU0, RP, a.row_mask);
        windows_k_module<1, IT>([&](int32_t r, int) { return pts[ts_index(pm, N, r, ic, C)]; }, N,
                            a.c.pv_lo, a.c.pv_hi, a.T, ti, R, K, ic, q, Q, blk + RP, blk + NU0 + K, RP, a.row_mask);
        if constexpr (GRID)
            windows_k_module<4, IT>([&](int32_t r, int cc) { return gts[grid_ts_index(pm, N, r, cc, ic)]; }, N,
                                a.c.grid_lo, a.c.grid_hi, a.T, ti, R, K, ic, q, Q, blk + 2 * RP, blk + NU0 + 2 * K, RP, a.row_mask);
    }
    // State columns: the current state for block 0 and zeros after it -- nstate strips of K words.  A ring written AHEAD of the
    // counter has zeros only: ONE strip of K zeros that every state column maps to (windows_plan sizes the image accordingly --
    // what lets a workgroup take 32 grids of doubles at K = 32: 144 instead of 184 KB).
    if (q == 0) {
        if (have_now) {
#pragma unroll
            for (int j = 0; j < 6; j++) {                // static indices: `now` stays in registers (a dynamic index put it --
                if (j < nstate) {                        // and 64 B of zero-initialisation per thread -- into scratch memory)
                    blk[S0 + j * K] = (IT)now[j];
                    for (int32_t k = 1; k < K; k++) blk[S0 + j * K + k] = (IT)0.0;
                }
            }
        } else {
            for (int32_t k = 0; k < K; k++) blk[S0 + k] = (IT)0.0;
        }
    }
    for (int32_t col = tid; col < D; col += NT) {                       // column -> offset of its k = 0 entry in a block
        uint32_t comp, h;
        decode_obs_col(a, GRID, col, W, comp, h);
        map[col] = comp == 0xffffu ? S0 + (have_now ? h * K : 0) : (h == 0 ? NU0 + comp * K : comp * RP + h);
    }
    __syncthreads();
    if (tid >= OBS_K_THREADS) return;                    // phase 2: the first OBS_K_THREADS threads (no barrier follows)
    const int32_t Q2 = OBS_K_THREADS / G;                // ... thread (g, q) of Q2 per grid (q = tid / G < Q2 here)
    const int32_t n_valid = (N - g0 < G) ? (int32_t)(N - g0) : G;
    const int32_t total = n_valid * D;                   // D is even (one load, one renewable module)
    typedef OT vec2 __attribute__((ext_vector_type(2)));
    const bool wide = (reinterpret_cast<uintptr_t>(ring) & (sizeof(vec2) - 1)) == 0;
    constexpr int KW = OBS_K_THREADS / 64;
    if (a.obs_colpitch) {
        // COLUMN-major blocks: value (block k, column c, grid g0 + g) at (k * D + c) * P + g0 + g.  Thread (g, cq) writes column
        // c = cq, cq + Q, ... of its grid for the wave's blocks k = wave', ...: the G grids of a (k, c) pair -- 16 doubles, or 32
        // floats out of the float image -- are ONE 128-byte line (P and g0 are multiples of G), written whole by G adjacent lanes.  Nothing else ever writes into such a line's bytes
        // except the step's state columns -- which are whole coalesced lines of their own here (a wave's 64 grids x 8 B per column).
        const int64_t P = a.obs_colpitch;
        if (plan.pairs && store_colmajor_packs<OT, IT>(image, BP, map, D, have_now ? 0x7fffffff : S0, K, G, g0, N, P, ring, tid)) return;
        const bool in_batch = i < N;
        OT *outc = ring + g0 + g;
        for (int32_t c = q; c < D; c += Q2) {
            const uint32_t m = map[c];
            // a ring written ahead of the counter leaves the state columns to the steps: here they are lines of their own, so
            // not writing them costs nothing (in row-major blocks the same holes make partial lines: MGX_WIN_SKIP_STATE)
            if (!have_now && m >= (uint32_t)S0) continue;
            const IT *src = image + g * BP + m;
            OT *o = outc + (int64_t)c * P;
            const int64_t kstride = (int64_t)D * P;
            int32_t k = 0;
            if constexpr (WIN_U > 1) {                   // WIN_U image words in flight before the first store of the batch
                for (; k + WIN_U <= K; k += WIN_U) {
                    IT v[WIN_U];
#pragma unroll
                    for (int u = 0; u < WIN_U; u++) v[u] = src[k + u];
#pragma unroll
                    for (int u = 0; u < WIN_U; u++)
                        if (in_batch) MGX_WIN_STORE((OT)v[u], o + (int64_t)(k + u) * kstride);
                }
            }
            for (; k < K; k++)
                if (in_batch) MGX_WIN_STORE((OT)src[k], o + (int64_t)k * kstride);
        }
        return;
    }
    // A lane's element schedule (which (grid, column) its j-th store carries) is the same for every block: the column map is
    // looked up once per element and reused for the wave's blocks k = wave, wave + 4, ... (image[... + k]: consecutive words)
    const int64_t block_stride = (int64_t)plan.pitch * D;
    OT *out0 = ring + ((int64_t)wave * plan.pitch + g0) * D;
    // MGX_WIN_SKIP_STATE=1 (experiment, OFF): a ring written ahead of the counter would leave the state columns alone -- the
    // step that reaches a block writes them, so the zeros here are bytes written twice, 3 % of a config-5 fleet step.  Measured
    // 10-25 % SLOWER (profiles/r04/exp_fleet_state_holes_ab.txt: 31 -> 34 us per fleet step at K = 16, 29 -> 36 us at K = 32):
    // a 48-byte hole per row turns 1.25 of its 9.75 lines into partial-line writes, and those cost far more than the bytes
    // they save.  (The state blocks start on even columns in both flat orders: a 16-byte pair is all state or all window.)
#ifndef MGX_WIN_SKIP_STATE
#define MGX_WIN_SKIP_STATE 0
#endif
    const bool skip_state = !have_now && MGX_WIN_SKIP_STATE;
    if (wide) {                                          // element pair f, f + 1 = 2 lane + 128 j -> (row, column)
        int32_t r = 2 * lane / D, c = 2 * lane - r * D;
        for (int32_t f = 2 * lane; f < total; f += 128) {
            const uint32_t m0 = map[c], m1 = map[c + 1];               // D even, c even: the pair never straddles two rows
            if (!(skip_state && m0 >= (uint32_t)S0)) {
                const IT *s0 = image + r * BP + m0 + wave;
                const IT *s1 = image + r * BP + m1 + wave;
                OT *out = out0 + f;
                const int64_t kw_stride = KW * block_stride;
                int32_t k = wave;
                if constexpr (WIN_U > 1) {               // WIN_U pairs of image words in flight before the first store of the batch
                    for (; k + (WIN_U - 1) * KW < K; k += WIN_U * KW) {
                        IT a0[WIN_U], a1[WIN_U];
#pragma unroll
                        for (int u = 0; u < WIN_U; u++) { a0[u] = s0[u * KW]; a1[u] = s1[u * KW]; }
#pragma unroll
                        for (int u = 0; u < WIN_U; u++) {
                            vec2 v2;
                            v2.x = (OT)a0[u]; v2.y = (OT)a1[u];
                            MGX_WIN_STORE(v2, reinterpret_cast<vec2 *>(out + u * kw_stride));
                        }
                        s0 += WIN_U * KW; s1 += WIN_U * KW; out += WIN_U * kw_stride;
                    }
                }
                for (; k < K; k += KW) {
                    vec2 v2;
                    v2.x = (OT)*s0; v2.y = (OT)*s1;
                    MGX_WIN_STORE(v2, reinterpret_cast<vec2 *>(out));
                    s0 += KW; s1 += KW; out += kw_stride;
                }
            }
            c += 128;
            while (c >= D) { c -= D; r++; }
        }
    } else {
        int32_t r = lane / D, c = lane - r * D;
        for (int32_t f = lane; f < total; f += 64) {
            const uint32_t m0 = map[c];
            if (!(skip_state && m0 >= (uint32_t)S0)) {
                const IT *s0 = image + r * BP + m0 + wave;
                OT *out = out0 + f;
                for (int32_t k = wave; k < K; k += KW) {
                    MGX_WIN_STORE((OT)*s0, out);
                    s0 += KW; out += KW * block_stride;
                }
            }
            c += 64;
            while (c >= D) { c -= D; r++; }
        }
    }
}

template <int F, typename OT>
__global__ __launch_bounds__(OBS_P1_THREADS) void obs_windows_k_kernel(const KArgs a, const WindowsKPlan plan, int32_t t,
                                                                      OT *__restrict__ ring)
{
    t = resolve_t_obs(a, t);
    constexpr int NSTATE = 4 * ((F & F_GENSET) != 0) + 2 * ((F & F_BATTERY) != 0);
    extern __shared__ double image[];
    const int64_t group = (int64_t)plan.group0 + blockIdx.x;
    double now[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (plan.with_state && (int)threadIdx.x < plan.group) {           // the q == 0 lanes: one per grid of the group
        const int64_t i = group * plan.group + threadIdx.x, ic = i < a.N ? i : group * plan.group;
        Params p; State s;
        load_state<F>(a.c, ic, true, s);
        load_params<F>(a.c, ic, p);
        observe_state_cols<F>(a, p, s, now, 0);
    }
    windows_body<(F & F_GRID) != 0, OT>(a, plan, t, ring, group, NSTATE, plan.with_state != 0, now, image);
}

// Restarted grids in a prefetched ring (mgx_patch_windows): the window columns of blocks first..K-1 of `ring` are recomputed
// for the grids with mask[i] != 0 -- their series rows have just been replaced (mgx_reset_grids*), so the rows the refill
// wrote for them belong to the old episode.  Block k of the ring is the row of counter value t + (k - first).  A wave takes the
// masked grids of its 64 one after the other: it reads the grid's rows t .. t + (K - first) + H - 1 ONCE (lanes = (row,
// component) pairs: scattered 8-byte reads, but each series value only once -- per output column they would be read and
// normalised K times over), keeps their normalised forms in LDS (unclipped for the "current value" position, clipped for
// forecast positions, as the refill kernel does), and writes the blocks' rows with lanes = columns (coalesced).  State
// columns are left alone: the restart does not touch the state, the step has already patched the block it reached.
constexpr int PATCH_MAX_ROWS = 512;     // (K - first) + H rows the LDS image of one grid can hold

template <bool GRID, typename OT>
__global__ __launch_bounds__(64) void patch_windows_kernel(const KArgs a, const uint8_t *__restrict__ mask, int32_t t, int32_t K,
                                                           int32_t first, int32_t pitch, OT *__restrict__ ring,
                                                           uint8_t *__restrict__ acc)
{
    constexpr int NCOMP = GRID ? 6 : 2;
    extern __shared__ double patch_lds[];                  // [2][NCOMP][R]: unclipped, clipped
    const int lane = threadIdx.x;
    const int64_t N = a.N;
    const int64_t i0 = (int64_t)blockIdx.x * 64;
    const int64_t i = i0 + lane;
    const bool mine = i < N && mask[i] != 0;
    if (acc && mine) acc[i] = 1;                           // grids that restarted since the ring ahead was launched
    unsigned long long todo = __ballot(mine);
    const int32_t W = 1 + a.H, D = a.obs_dim, R = (K - first) + a.H;
    double *nu = patch_lds, *nc = patch_lds + NCOMP * R;
    while (todo) {
        const int j = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int64_t g = i0 + j;
        __syncthreads();                                   // the previous grid's image has been consumed
        for (int32_t idx = lane; idx < NCOMP * R; idx += 64) {
            const int comp = idx / R, r = idx - comp * R;
            double lo, hi;
            if (comp == 0) { lo = a.c.load_lo[g]; hi = a.c.load_hi[g]; }
            else if (comp == 1) { lo = a.c.pv_lo[g]; hi = a.c.pv_hi[g]; }
            else { lo = a.c.grid_lo[(comp - 2) * N + g]; hi = a.c.grid_hi[(comp - 2) * N + g]; }
            const int32_t row = t + r + (a.ep_off ? a.ep_off[g] : 0);      // in-place episodes: the grid's own series row
            const bool in = row < a.T;
            const double v = series_component(a.c, N, comp, (int64_t)((in ? row : a.T - 1) & a.row_mask), g, a.pm_pitch);
            const double fill = (hi + lo) / 2, sp = space_spread(lo, hi);
            nu[idx] = obs_series_value(v, in, false, lo, hi, fill, sp);
            nc[idx] = obs_series_value(v, in, true, lo, hi, fill, sp);
        }
        __syncthreads();
        OT *save = a.final_obs ? (OT *)a.final_obs + g * D : nullptr;   // mgx_set_final_obs: the row the restart is about to replace
        for (int32_t col = lane; col < D; col += 64) {
            uint32_t comp, h;                              // which series component / horizon step this column shows
            decode_obs_col(a, GRID, col, W, comp, h);
            OT *cur = ring + ((int64_t)first * pitch + g) * D + col;
            if (save) save[col] = *cur;                    // (state columns too: the whole pre-restart observation)
            if (comp == 0xffffu) continue;                 // state columns
            const double *src = (h == 0 ? nu : nc) + comp * R + h;
            for (int32_t k = first; k < K; k++) ring[((int64_t)k * pitch + g) * D + col] = (OT)src[k - first];
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// Discrete action expansion: PriorityListAlgo._populate_action (priority_list.py:69-167).
// ------------------------------------------------------------------------------------------------------
template <int F>
__global__ __launch_bounds__(BLOCK) void expand_kernel(const KArgs a, const PLWords tab, const int32_t *__restrict__ action_id,
                                                       int32_t t, double *__restrict__ control, uint32_t *__restrict__ violations)
{
    t = resolve_t(a, t);
    constexpr int A = 2 * ((F & F_GENSET) != 0) + ((F & F_BATTERY) != 0) + ((F & F_GRID) != 0);
    const int64_t i = (int64_t)a.g0 + (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= a.g1) return;
    const int64_t N = a.N;
    Params p; State s; Inputs in;
    load_state<F>(a.c, i, false, s);
    load_params<F>(a.c, i, p);
    const int64_t tr = series_row(a, i, t);
    if (factorised(a.c)) {
        GridFactors f;
        load_factors<F>(a.c, i, f);
        fact_series<F>(a.c, N, i, tr, f, in, a.pm_pitch);
    } else {
        in.load = a.c.load_ts[ts_index(a.pm_pitch, N, tr, i, (F & F_GRID) ? 6 : 2)];
        in.pv = a.c.pv_ts[ts_index(a.pm_pitch, N, tr, i, (F & F_GRID) ? 6 : 2)];
        in.g_stat = 1.0;
        if constexpr (F & F_GRID) in.g_stat = a.c.grid_ts[grid_ts_index(a.pm_pitch, N, tr, 3, i)];
    }
    double q_unused;
    uint32_t xv = 0u;
    populate_core<F, true>(p, s, pl_select(tab, action_id[i]), in, q_unused, 0.0 + -1 * in.load, in.pv, false, &xv);
    if (violations) violations[i] = xv;              // where the reference's _populate_action asserts (priority_list.py:73-154)
    double *c = control + i * A;
    int k = 0;
    if constexpr (F & F_GENSET) { c[k] = in.a_goal; c[k + 1] = in.a_gen; k += 2; }
    if constexpr (F & F_BATTERY) { c[k++] = in.a_bat; }
    if constexpr (F & F_GRID) { c[k++] = in.a_grid; }
}

// the series part of a step's inputs at series row `row` of grid i (pm: grid-major copies during in-place episodes)
template <int F>
__device__ __forceinline__ void load_series_row(const mgx_columns &c, int64_t N, int64_t i, int64_t row, Inputs &in, int32_t pm)
{
    // two explicit forms writing the same fields (a stride selected per launch put `in` into scratch memory: 36 B per lane)
    if (pm) {
        constexpr int C = (F & F_GRID) ? 6 : 2;
        const int64_t e = (i * pm + row) * C;
        in.load = c.load_ts[e];
        in.pv = c.pv_ts[e];
        in.g_stat = 1.0;
        if constexpr (F & F_GRID) {
            const double *g = c.grid_ts + e;
            in.g_pimp = g[0]; in.g_pexp = g[1]; in.g_co2 = g[2]; in.g_stat = g[3];
        }
    } else {
        in.load = c.load_ts[row * N + i];
        in.pv = c.pv_ts[row * N + i];
        in.g_stat = 1.0;
        if constexpr (F & F_GRID) {
            const double *g = c.grid_ts + (row * 4) * N + i;
            in.g_pimp = g[0]; in.g_pexp = g[N]; in.g_co2 = g[2 * N]; in.g_stat = g[3 * N];
        }
    }
}

template <int F>
__device__ __forceinline__ void load_series_at(const double *__restrict__ lts, const double *__restrict__ pts,
                                               const double *__restrict__ gts, int64_t N, int64_t i, int64_t off,
                                               Inputs &in)
{
    in.load = lts[off];
    in.pv = pts[off];
    in.g_stat = 1.0;
    if constexpr (F & F_GRID) {
        const double *g = gts + (4 * off - 3 * i);
        in.g_pimp = g[0]; in.g_pexp = g[N]; in.g_co2 = g[2 * N]; in.g_stat = g[3 * N];
    }
}

// DiscreteMicrogridEnv.step in ONE launch (discrete.py:109-143): expand the priority list of every grid into its
// control and run Microgrid.run(control, normalized=False) on it, without the control ever leaving registers.
// (body shared by step_discrete_kernel and fleet_step_kernel)
template <int F, bool EP = false>
__device__ __forceinline__ void step_discrete_body(const KArgs &a, const PLWords &tab, const int32_t *__restrict__ action_id,
                                                   int32_t t, double *__restrict__ control, double *__restrict__ reward,
                                                   uint8_t *__restrict__ done, void *__restrict__ obs,
                                                   double *__restrict__ log, int64_t i, double *tile = nullptr)
{
    constexpr int A = 2 * ((F & F_GENSET) != 0) + ((F & F_BATTERY) != 0) + ((F & F_GRID) != 0);
    const int64_t N = a.N;
    Params p; State s; Inputs in; Outputs o; Derived d;
    const int32_t id = action_id[i];
    int32_t off = 0;
    if constexpr (EP) off = a.ep_off[i];
    const int64_t tr = EP ? episode_row(a, t, off) : (int64_t)(t & a.row_mask);
    if (factorised(a.c)) {
        GridFactors f;
        load_factors<F>(a.c, i, f);
        fact_series<F>(a.c, N, i, tr, f, in, EP ? a.pm_pitch : 0);
    } else {
        load_series_row<F>(a.c, N, i, tr, in, EP ? a.pm_pitch : 0);
    }
    load_state<F>(a.c, i, log != nullptr, s);
    load_params<F>(a.c, i, p);
    derive<F>(p, d);
    const bool gen_instant = genset_wave_is_instant<F>(p, s);
    double bat_q;
    uint32_t xv = 0u;                                // states in which _populate_action asserts (priority_list.py:73-154): into the
    populate_core<F, true>(p, s, pl_select(tab, id), in, bat_q, 0.0 + -1 * in.load, in.pv, false, &xv);   // log's violations column
    if (control) {                                   // optional copy of the expanded control (_get_action's value)
        double *c = control + i * A;
        int k = 0;
        if constexpr (F & F_GENSET) { c[k] = in.a_goal; c[k + 1] = in.a_gen; k += 2; }
        if constexpr (F & F_BATTERY) { c[k++] = in.a_bat; }
        if constexpr (F & F_GRID) { c[k++] = in.a_grid; }
    }
    step_core<F, true>(p, d, s, in, false, true, gen_instant, o, bat_q);
    o.violations |= xv;
    store_state<F>(a.c, i, s);
    reward[i] = shaped_reward<F>(a.shaper, o);
    const uint8_t dn = done_at(a, i, t);
    if (done) done[i] = dn;
    if (log) store_log<F>(log + i, N, o, s.status);
    if constexpr (EP) off = episode_tail<F>(a, i, t, off, dn != 0, p, s);
    if (obs) store_step_obs<F>(a, obs, i, t + 1 + off, p, s, EP ? a.pm_pitch : 0, tile);
}

// Dry run of one discrete step (mgx_check_discrete): the expansion and the step on a register copy of the state; only the mask
// leaves the kernel -- the expansion's assert bit if it has one (the reference raises there and never steps), else the step's.
template <int F>
__global__ __launch_bounds__(BLOCK) void check_discrete_kernel(const KArgs a, const PLWords tab, const int32_t *__restrict__ action_id,
                                                               int32_t t, uint32_t *__restrict__ violations)
{
    t = resolve_t(a, t);
    const int64_t i = (int64_t)a.g0 + (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= a.g1) return;
    const int64_t N = a.N;
    Params p; State s; Inputs in; Outputs o; Derived d;
    const int64_t tr = series_row(a, i, t);
    if (factorised(a.c)) {
        GridFactors f;
        load_factors<F>(a.c, i, f);
        fact_series<F>(a.c, N, i, tr, f, in, a.pm_pitch);
    } else {
        load_series_row<F>(a.c, N, i, tr, in, a.pm_pitch);
    }
    load_state<F>(a.c, i, true, s);
    load_params<F>(a.c, i, p);
    derive<F>(p, d);
    double bat_q;
    uint32_t xv = 0u;
    populate_core<F, true>(p, s, pl_select(tab, action_id[i]), in, bat_q, 0.0 + -1 * in.load, in.pv, false, &xv);
    step_core<F, true>(p, d, s, in, false, false, false, o, bat_q);
    violations[i] = xv ? xv : o.violations;
}

template <int F, bool EP = false>
__global__ __launch_bounds__(BLOCK) void step_discrete_kernel(const KArgs a, const PLWords tab,
                                                              const int32_t *__restrict__ action_id, int32_t t,
                                                              double *__restrict__ control, double *__restrict__ reward,
                                                              uint8_t *__restrict__ done, void *__restrict__ obs,
                                                              double *__restrict__ log)
{
    t = resolve_t(a, t);
    extern __shared__ __attribute__((aligned(16))) double row_tiles[];      // (dynamic: see step_kernel)
    const int64_t i = (int64_t)a.g0 + (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < a.g1) step_discrete_body<F, EP>(a, tab, action_id, t, control, reward, done, obs, log, i,
                                            row_tiles + (threadIdx.x >> 6) * (64 * ROW_TILE_MAX_D));
    advance_counter_in_kernel(a, 1);
}

// ------------------------------------------------------------------------------------------------------
// A heterogeneous fleet in ONE launch (mgx_fleet_step): up to MGX_FLEET_MAX batches of different layouts, each with its own
// columns / actions / outputs / step counter, laid end to end over the workgroups.  The table travels in the kernarg
// segment (scalar loads); a workgroup finds its batch with a few scalar compares and jumps -- wave-uniformly -- to that
// layout's specialisation of the step.  Three 33 000-grid batches cost one ~6 us launch instead of three ~5 us ones.
// ------------------------------------------------------------------------------------------------------
constexpr int MGX_FLEET_MAX = 5;
// What changes from step to step travels by value (small: one scalar-load round at kernel start); the big, rarely changing
// KArgs of every batch are read from DEVICE memory (the handle's own copy, refreshed by the host when it changes).  A table
// of whole KArgs in the kernarg segment measured 57 us per launch: the kernarg buffer lives in host memory and the chain
// of dependent scalar loads (which batch? -> its layout -> its columns) paid a host round trip per link.
struct FleetArgs {
    const KArgs *k[MGX_FLEET_MAX];               // device copies (mgx_handle::d_kargs)
    const PLWords *tab[MGX_FLEET_MAX];           // discrete items: the priority-list table (device copy), else NULL
    const void *actions[MGX_FLEET_MAX];          // continuous control [N, A] -- or the int32 priority-list ids [N] of a discrete item
    double *reward[MGX_FLEET_MAX];
    uint8_t *done[MGX_FLEET_MAX];
    void *obs[MGX_FLEET_MAX];
    double *log[MGX_FLEET_MAX];
    int32_t t[MGX_FLEET_MAX], flags[MGX_FLEET_MAX], block0[MGX_FLEET_MAX];   // block0: first workgroup of the batch
    int32_t n, normalized;
};

// Window chunks riding along with a fleet step: workgroups behind the step's own.  While the steps walk an observation
// ring of K blocks, the ring of the NEXT K counter values is due (obs_windows_k_kernel's job); as one launch per K steps it
// is a 150 us burst of pure writes between latency-bound step kernels.  Cut into K - 1 chunks of the batch's 16-grid
// groups, one chunk per step, the same bytes move at a constant rate in the shadow of the step kernels' latency.
struct FleetWin {
    const KArgs *k[MGX_FLEET_MAX];
    void *ring[MGX_FLEET_MAX];
    WindowsKPlan plan[MGX_FLEET_MAX];            // plan.group0 = first group of the chunk
    int32_t t[MGX_FLEET_MAX], block0[MGX_FLEET_MAX], kind[MGX_FLEET_MAX], nstate[MGX_FLEET_MAX];   // kind: bit 0 grid, bit 1 float rows
    int32_t n, first_block;                      // first_block = workgroups of the step part
};

// (its workgroups are BLOCK_FLEET = OBS_K_THREADS threads whatever BLOCK is: the window chunks' phase 2 needs that many)
constexpr int BLOCK_FLEET = OBS_K_THREADS;
static __global__ __launch_bounds__(BLOCK_FLEET) void fleet_step_kernel(const FleetArgs fa, const FleetWin fw)
{
    extern __shared__ double image[];
    if (fw.n > 0 && (int)blockIdx.x >= fw.first_block) {               // ---- window chunk workgroups
        const int b = (int)blockIdx.x - fw.first_block;
        const KArgs *kp = fw.k[0];
        void *ring = fw.ring[0];
        WindowsKPlan plan = fw.plan[0];
        int32_t t = fw.t[0], block0 = 0, kind = fw.kind[0], nstate = fw.nstate[0];
#pragma unroll
        for (int q = 1; q < MGX_FLEET_MAX; q++) {
            const bool mine = q < fw.n && b >= fw.block0[q];
            kp = mine ? fw.k[q] : kp; ring = mine ? fw.ring[q] : ring;
            plan.grid_col_base = mine ? fw.plan[q].grid_col_base : plan.grid_col_base;
            plan.group = mine ? fw.plan[q].group : plan.group; plan.K = mine ? fw.plan[q].K : plan.K;
            plan.rp = mine ? fw.plan[q].rp : plan.rp; plan.bp = mine ? fw.plan[q].bp : plan.bp;
            plan.group0 = mine ? fw.plan[q].group0 : plan.group0; plan.pitch = mine ? fw.plan[q].pitch : plan.pitch;
            plan.pairs = mine ? fw.plan[q].pairs : plan.pairs;
            t = mine ? fw.t[q] : t; block0 = mine ? fw.block0[q] : block0; kind = mine ? fw.kind[q] : kind;
            nstate = mine ? fw.nstate[q] : nstate;
        }
        const KArgs &a = *kp;
        const int64_t group = (int64_t)plan.group0 + (b - block0);
        const double none[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};     // a prefetch ahead of the counter: no state columns yet
        switch (kind) {
            case 0: windows_body<false, double>(a, plan, t, (double *)ring, group, nstate, false, none, image); break;
            case 1: windows_body<true, double>(a, plan, t, (double *)ring, group, nstate, false, none, image); break;
            case 2: windows_body<false, float>(a, plan, t, (float *)ring, group, nstate, false, none, image); break;
            default: windows_body<true, float>(a, plan, t, (float *)ring, group, nstate, false, none, image); break;
        }
        return;
    }
    // which batch owns this workgroup: selects over the (<= 6) kernarg entries, no run-time indexing (that would send the
    // struct to scratch)
    const KArgs *kp = fa.k[0];
    const PLWords *tp = fa.tab[0];
    const void *actions = fa.actions[0];
    double *reward = fa.reward[0]; uint8_t *done = fa.done[0]; void *obs = fa.obs[0]; double *log = fa.log[0];
    int32_t t = fa.t[0], flags = fa.flags[0], block0 = 0;
#pragma unroll
    for (int q = 1; q < MGX_FLEET_MAX; q++) {
        const bool mine = q < fa.n && (int)blockIdx.x >= fa.block0[q];
        kp = mine ? fa.k[q] : kp; tp = mine ? fa.tab[q] : tp;
        actions = mine ? fa.actions[q] : actions; reward = mine ? fa.reward[q] : reward;
        done = mine ? fa.done[q] : done; obs = mine ? fa.obs[q] : obs; log = mine ? fa.log[q] : log;
        t = mine ? fa.t[q] : t; flags = mine ? fa.flags[q] : flags; block0 = mine ? fa.block0[q] : block0;
    }
    const KArgs &a = *kp;                         // uniform address, read-only: scalar loads from HBM / L2
    const int64_t i = (int64_t)((int)blockIdx.x - block0) * BLOCK_FLEET + threadIdx.x;
    if (i >= a.N) return;
    if (tp != nullptr) {                          // a DiscreteMicrogridEnv batch: ids -> control -> run, in registers
        const PLWords &tab = *tp;
#define MGX_FLEET_CASE(FV) case FV: step_discrete_body<FV>(a, tab, (const int32_t *)actions, t, nullptr, reward, done, obs, log, i); break;
        switch (flags) {
            MGX_FLEET_CASE(0) MGX_FLEET_CASE(1) MGX_FLEET_CASE(2) MGX_FLEET_CASE(3) MGX_FLEET_CASE(4)
            MGX_FLEET_CASE(5) MGX_FLEET_CASE(6) MGX_FLEET_CASE(7) MGX_FLEET_CASE(14)
            default: step_discrete_body<15>(a, tab, (const int32_t *)actions, t, nullptr, reward, done, obs, log, i); break;
        }
#undef MGX_FLEET_CASE
        return;
    }
#define MGX_FLEET_CASE(FV) case FV: step_body<FV>(a, actions, t, fa.normalized, reward, done, obs, log, i); break;
    switch (flags) {
        MGX_FLEET_CASE(0) MGX_FLEET_CASE(1) MGX_FLEET_CASE(2) MGX_FLEET_CASE(3) MGX_FLEET_CASE(4)
        MGX_FLEET_CASE(5) MGX_FLEET_CASE(6) MGX_FLEET_CASE(7) MGX_FLEET_CASE(14)
        default: step_body<15>(a, actions, t, fa.normalized, reward, done, obs, log, i); break;
    }
#undef MGX_FLEET_CASE
}

// The same step part with every batch's KArgs BY VALUE in the kernarg segment and the batch picked by blockIdx.y -- a wave
// knows which bucket it serves before any load returns, so the bucket's KArgs, buffers and counter are ONE round of scalar loads
// at a computed offset of the kernarg segment (what a single batch's step_kernel pays for its by-value KArgs) instead of two
// dependent ones (the selects over fa.k[], then *kp out of device memory): 8.6 -> 6.4 us for the kernel alone on one
// 99 999-grid bucket (profiles/r05/exp_fleet_vs_env.txt), and every round trip saved counts double beside a ring refill.
// (Round 2's table of whole KArgs in the kernarg segment lost because the bucket was FOUND by loads: which batch? -> its
// layout -> its columns, a chain.)  Launched on a (max workgroups of a bucket, buckets) grid when no window chunks ride along.
struct FleetHead {                               // what picks the code path: one 32-byte scalar load
    const PLWords *tab;                          // discrete items: the priority-list table (device copy), else NULL
    int64_t n_grids;
    int32_t t, flags, pad0, pad1;
};
struct FleetBucket {
    FleetHead hd;
    KArgs k;
    const void *actions;                         // continuous control [N, A] -- or the int32 priority-list ids [N] of a discrete item
    double *reward; uint8_t *done; void *obs; double *log;
};
struct FleetArgsV {
    int32_t n, normalized, pad0, pad1;
    FleetBucket b[MGX_FLEET_MAX];
};
static_assert(sizeof(FleetArgsV) <= 3840, "the kernarg segment holds 4 KB");

// (the kernel's head -- which bucket, which specialisation -- and its two switches are shared by the plain form and the form that also
//  serves buckets with several modules of a kind, fleet_step_kernel_vm below the general kernels)
// Which specialisation: ONE 32-byte scalar load; the bucket's other fields are read THROUGH a reference into the kernarg segment
// by the case that uses them (scalar loads of invariant memory the compiler places and re-issues as it likes; a copy of the
// whole bucket in front of the switch made it fetch every field any case needs -- 158 dwords through ~100 SGPRs: nine
// load-wait-spill rounds before the first vector load -- and a copy inside every case still spilled SGPRs).
#define MGX_FLEET_V_HEAD \
    typedef const char __attribute__((address_space(4))) *kernarg_bytes; \
    const kernarg_bytes mine = (kernarg_bytes)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(FleetArgsV, b) \
                               + (size_t)blockIdx.y * sizeof(FleetBucket); \
    typedef uint32_t u32x8 __attribute__((ext_vector_type(8))); \
    static_assert(sizeof(FleetHead) == 32 && offsetof(FleetBucket, hd) == 0, "the head is one s_load_dwordx8"); \
    const u32x8 h8 = *reinterpret_cast<const u32x8 __attribute__((address_space(4))) *>(mine); \
    FleetHead hd; \
    hd.tab = reinterpret_cast<const PLWords *>((uint64_t)h8[0] | ((uint64_t)h8[1] << 32)); \
    hd.n_grids = (int64_t)((uint64_t)h8[2] | ((uint64_t)h8[3] << 32)); \
    hd.t = (int32_t)h8[4]; hd.flags = (int32_t)h8[5]; \
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; \
    if (i >= hd.n_grids) return; \
    const int normalized = fa.normalized; \
    const FleetBucket &B = *reinterpret_cast<const FleetBucket *>((const char *)mine);

#define MGX_FLEET_V_DISCRETE(FV) case FV: { step_discrete_body<FV>(B.k, *hd.tab, (const int32_t *)B.actions, hd.t, nullptr, B.reward, B.done, B.obs, B.log, i); } break;
#define MGX_FLEET_V_STEP(FV) case FV: { step_body<FV>(B.k, B.actions, hd.t, normalized, B.reward, B.done, B.obs, B.log, i); } break;
#define MGX_FLEET_V_SWITCHES                                                                                                          \
    if (hd.tab != nullptr) {                      /* a DiscreteMicrogridEnv batch: ids -> control -> run, in registers */            \
        switch (hd.flags) {                                                                                                           \
            MGX_FLEET_V_DISCRETE(0) MGX_FLEET_V_DISCRETE(1) MGX_FLEET_V_DISCRETE(2) MGX_FLEET_V_DISCRETE(3) MGX_FLEET_V_DISCRETE(4)   \
            MGX_FLEET_V_DISCRETE(5) MGX_FLEET_V_DISCRETE(6) MGX_FLEET_V_DISCRETE(7) MGX_FLEET_V_DISCRETE(14)                          \
            default: { step_discrete_body<15>(B.k, *hd.tab, (const int32_t *)B.actions, hd.t, nullptr, B.reward, B.done, B.obs, B.log, i); } break; \
        }                                                                                                                             \
        return;                                                                                                                       \
    }                                                                                                                                 \
    switch (hd.flags) {                                                                                                               \
        MGX_FLEET_V_STEP(0) MGX_FLEET_V_STEP(1) MGX_FLEET_V_STEP(2) MGX_FLEET_V_STEP(3) MGX_FLEET_V_STEP(4)                           \
        MGX_FLEET_V_STEP(5) MGX_FLEET_V_STEP(6) MGX_FLEET_V_STEP(7) MGX_FLEET_V_STEP(14)                                              \
        default: { step_body<15>(B.k, B.actions, hd.t, normalized, B.reward, B.done, B.obs, B.log, i); } break;                       \
    }

static __global__ __launch_bounds__(BLOCK) void fleet_step_kernel_v(const FleetArgsV fa)
{
    MGX_FLEET_V_HEAD
    MGX_FLEET_V_SWITCHES
}

// ------------------------------------------------------------------------------------------------------
// Discrete rollout: K fused steps whose control is expanded ON DEVICE from a priority-list id -- per step
// (ids [K, N], one byte each: a DiscreteMicrogridEnv roll-out) or constant per grid (ids [N]: RuleBasedControl.run,
// algos/rbc/rbc.py:64-93).  No action stream at all: per step only the series rows are read.
// ------------------------------------------------------------------------------------------------------
// FACT: factorised series (mgx_columns.base_load): nothing but the id bytes (PER_STEP) is read per step.
template <int F, int U, bool PER_STEP, bool RICH, bool FACT>
__global__ __launch_bounds__(BLOCK_K) void rollout_kernel(const KArgs a, const PLWords tab, const uint8_t *__restrict__ ids,
                                                          int32_t t0, int32_t K, const FusedOut out_rt, int32_t gpb)
{
    FusedOut out = out_rt;
    if constexpr (!RICH) { out.log = nullptr; out.status_trace = nullptr; }
    const int32_t K_launch = K;          // what the host asked for (the counter always moves by this much)
    t0 = resolve_t(a, t0);
    K = resolve_k(a, t0, K);
    const int64_t i = (int64_t)a.g0 + (int64_t)blockIdx.x * gpb + threadIdx.x;
    const bool active = (int32_t)threadIdx.x < gpb && i < a.g1;
    if constexpr (!FACT) { if (!active) return; }
    const int64_t N = a.N;
    Params p; State s; Derived d;
    const bool want_soc = (out.soc_trace != nullptr) || (out.log != nullptr);
    bool gen_instant = false;
    int32_t k_done = 0;
    // PER_STEP = false (one fixed list per grid: RuleBasedControl): `word` is loop-invariant and the list decoding of
    // populate_core is hoisted out of the step loop by the compiler
    uint32_t word = 0u;
    if (active) {
        load_state<F>(a.c, i, out.log != nullptr, s);
        load_params<F>(a.c, i, p);
        derive<F>(p, d);
        gen_instant = genset_wave_is_instant<F>(p, s);
        k_done = (a.grid_final ? a.grid_final[i] : a.final_step) - 1 - t0;
        word = PER_STEP ? 0u : pl_select(tab, ids[i]);
    }
    double ret = 0.0;

    // one step of this lane's grid: series values complete in `in` (GI: compile-time form of the wave-uniform gen_instant;
    // HOT: the launch writes exactly reward + SoC per step, unshaped -- no per-step tests of the output pointers, row bases in
    // SGPRs + a 32-bit lane offset)
    const bool hot = !RICH && out.reward != nullptr && out.done == nullptr && (!(F & F_BATTERY) || out.soc_trace != nullptr) &&
                     a.shaper == MGX_SHAPER_NONE;
    const uint32_t i32 = (uint32_t)i;
    auto consume = [&](auto gi_tag, auto hot_tag, Inputs &in, int32_t k, int64_t off) __attribute__((always_inline)) {
        constexpr bool GI = decltype(gi_tag)::value, HOT = decltype(hot_tag)::value;
        double bat_q;
        uint32_t xv = 0u;                // RICH: the expansion's assert mask joins the log's violations column
        populate_core<F, RICH>(p, s, word, in, bat_q, 0.0 + -1 * in.load, in.pv, GI, &xv);
        Outputs o;
        step_core<F, true>(p, d, s, in, false, HOT ? (F & F_BATTERY) != 0 : want_soc, GI, o, bat_q);
        if constexpr (RICH) o.violations |= xv;
        if constexpr (HOT) {
            const double r = o.reward;
            (out.reward + (int64_t)k * N)[i32] = r;
            if constexpr (F & F_BATTERY) (out.soc_trace + (int64_t)k * N)[i32] = s.soc;
            ret += r;
            return;
        }
        const double r = shaped_reward<F>(a.shaper, o);
        if (out.reward) out.reward[off] = r;
        if (out.done) store_done(a, out.done, off, i, k, k >= k_done);
        if constexpr (F & F_BATTERY) { if (out.soc_trace) out.soc_trace[off] = s.soc; }
        if constexpr (F & F_GENSET) { if (out.status_trace) out.status_trace[off] = s.status; }
        if (out.log) store_log<F>(out.log + (off - i) * a.log_dim + i, N, o, s.status);
        ret += r;
    };
    // The step loop exists in four forms, specialised at COMPILE time on the wave-uniform `gen_instant` (in the instant form the
    // genset's status is its goal, its limits under a fixed list are loop-invariant (hoisted), and the FSM is gone) and on `hot`.
    auto specialised = [&](auto fn) __attribute__((always_inline)) {
        if constexpr ((F & F_GENSET) != 0) {
            if (gen_instant) { if (hot) fn(std::true_type{}, std::true_type{}); else fn(std::true_type{}, std::false_type{}); }
            else { if (hot) fn(std::false_type{}, std::true_type{}); else fn(std::false_type{}, std::false_type{}); }
        } else {
            if (hot) fn(std::false_type{}, std::true_type{}); else fn(std::false_type{}, std::false_type{});
        }
    };
    if constexpr (FACT) {
        // the barriers of the LDS chunks stay at kernel scope (every wave of the workgroup meets the same ones); only the
        // loop over a chunk's steps is specialised
        __shared__ double base_lds[((F & F_GRID) ? 3 : 2) * FACT_ROWS * PP];
        static_assert(FACT_ROWS % U == 0, "ring slots are addressed by k % U across LDS chunks");
        GridFactors f;
        f.lr = 0.0; f.pr = 0.0; f.lp = 0u; f.pp = 0u; f.cp = 0u; f.pat = 0u;
        OutageWords ow;
        ow.cur = 0; ow.nxt = 0; ow.wi = 0;
        static_assert(U <= 8, "the id ring is one 64-bit register");
        uint64_t idq = 0;                 // PER_STEP: the id bytes of the U ring slots (an array would live in scratch memory)
        if (active) {
            load_factors<F>(a.c, i, f);
            if constexpr (F & F_GRID) outage_init(a.c, N, i, a.T, t0, ow);
            if constexpr (PER_STEP) {
#pragma unroll
                for (int u = 0; u < U; u++)
                    if (u < K) idq |= (uint64_t)(ids + (int64_t)u * N)[i32] << (8 * u);
            }
        }
        int64_t off = i;
        auto run_chunk = [&](auto gi_tag, auto hot_tag, int32_t kb, int32_t n) __attribute__((always_inline)) {
            BaseVals nb = read_base_row<F>(base_lds, 0, f);
            for (int32_t k0 = kb; k0 < kb + n; k0 += U) {
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int32_t k = k0 + u;
                    if (k < kb + n) {
                        if constexpr (PER_STEP) {
                            word = pl_select(tab, (int32_t)((idq >> (8 * u)) & 0xffu));
                            if (k + U < K)
                                idq = (idq & ~(0xffull << (8 * u))) | ((uint64_t)(ids + (int64_t)(k + U) * N)[i32] << (8 * u));
                        }
                        Inputs in;
                        in.load = fact_load(nb.load, f.lr);
                        in.pv = fact_pv(nb.pv, f.pr);
                        in.g_stat = 1.0;
                        if constexpr (F & F_GRID) {
                            in.g_pimp = tariff_price((int32_t)f.pat, t0 + k); in.g_pexp = 0.0;
                            in.g_co2 = nb.co2;
                            in.g_stat = outage_status(a.c, N, i, a.T, t0 + k, k == 0, ow);
                        }
                        const int32_t rn = (k + 1 - kb < n) ? k + 1 - kb : n - 1;      // next step's row (LDS, one step ahead)
                        nb = read_base_row<F>(base_lds, rn, f);
                        consume(gi_tag, hot_tag, in, k, off);
                        off += N;
                    }
                }
            }
        };
        for (int32_t kb = 0; kb < K; kb += FACT_ROWS) {          // K is uniform over the launch: the barriers are safe
            const int32_t n = K - kb < FACT_ROWS ? K - kb : FACT_ROWS;
            __syncthreads();                                     // the previous chunk's rows have been consumed
            stage_base_rows<F>(a.c, (int64_t)t0 + kb, n, base_lds, BLOCK_K);
            __syncthreads();
            if (active) specialised([&](auto gi_tag, auto hot_tag) __attribute__((always_inline)) { run_chunk(gi_tag, hot_tag, kb, n); });
        }
        if (!active) return;
    } else {
        specialised([&](auto gi_tag, auto hot_tag) __attribute__((always_inline)) {
            const double *__restrict__ lts = a.c.load_ts + (int64_t)t0 * N;
            const double *__restrict__ pts = a.c.pv_ts + (int64_t)t0 * N;
            const double *__restrict__ gts = (F & F_GRID) ? a.c.grid_ts + (int64_t)t0 * 4 * N : nullptr;
            Inputs ring[U];
            uint8_t idr[U];
#pragma unroll
            for (int u = 0; u < U; u++)
                if (u < K) {
                    load_series_at<F>(lts, pts, gts, N, i, (int64_t)u * N + i, ring[u]);
                    if constexpr (PER_STEP) idr[u] = ids[(int64_t)u * N + i];
                }

            int64_t off = i;
            for (int32_t k0 = 0; k0 < K; k0 += U) {
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int32_t k = k0 + u;
                    if (k < K) {
                        Inputs in = ring[u];
                        if constexpr (PER_STEP) word = pl_select(tab, idr[u]);
                        if (k + U < K) {
                            load_series_at<F>(lts, pts, gts, N, i, off + (int64_t)U * N, ring[u]);
                            if constexpr (PER_STEP) idr[u] = ids[off + (int64_t)U * N];
                        }
                        consume(gi_tag, hot_tag, in, k, off);
                        off += N;
                    }
                }
            }
        });
    }
    if constexpr (F & F_BATTERY) { if (!want_soc) s.soc = s.charge / p.bat_cmax; }
    store_state<F>(a.c, i, s);
    if (out.ret_acc) out.ret_acc[i] += ret;
    advance_counter_in_kernel(a, K_launch);
}

// ------------------------------------------------------------------------------------------------------
// General path: any number of modules per kind (n_load / n_pv != 1, or several gensets / batteries / grids: columns
// [n, N] instance-major, load / pv series [T, n, N], grid series [T, n_grid, 4, N]).  One lane per grid; the lists
// MicrogridStep sums live in LDS (StepLists).  Parity with the reference's multi-module microgrids, not speed.
// ------------------------------------------------------------------------------------------------------
constexpr int BLOCK_MULTI = 64;

__device__ __forceinline__ StepLists multi_lists(const KArgs &a, double *lds)
{
    const int cap = multi_list_capacity(a.n_load, a.n_pv, a.n_genset, a.n_battery, a.n_grid);
    StepLists L;
    L.stride = BLOCK_MULTI; L.n_prov = 0; L.n_absb = 0;
    L.prov = lds + threadIdx.x;
    L.absb = lds + (size_t)cap * BLOCK_MULTI + threadIdx.x;
    return L;
}

// noise: GaussianNoiseForecaster on the general path (round 6; forecaster.py:220-275): forecast value h > 0 of the window gets
// std * N(0, 1) -- std x (1 + log(1 + (h - 1))) with increase_uncertainty -- from the counter-based generator (seed; grid, component id,
// series row, h), BEFORE the clip to the bounds, as in window_finish<NOISE>; statistical parity only.
// (the noise arguments travel BY VALUE: a struct handed over by address lived in the private segment -- 48 B of scratch per lane, which
//  slowed every launch of step_multi_kernel by 7 %: tests/test_host_logic.py now refuses scratch on the general single-step kernels too)
template <typename OT>
__device__ __forceinline__ void observe_series_multi(const double *__restrict__ ts, int64_t row_stride, int32_t T, int32_t t, int32_t H,
                                                     double lo, double hi, OT *__restrict__ obs, int obs_stride = 1, bool noisy = false,
                                                     double noise_std = 0.0, uint64_t noise_seed = 0, int noise_increase = 0,
                                                     int64_t grid = 0, uint32_t comp = 0)
{
    const double fill = (hi + lo) / 2, sp = space_spread(lo, hi);
    for (int h = 0; h <= H; h++) {
        const bool in = t < T && t + h < T;
        double v = in ? ts[(int64_t)(t + h) * row_stride] : 0.0;
        if (noisy && in && h > 0) {
            const double sd = noise_increase ? noise_std * (1.0 + log(1.0 + (double)(h - 1))) : noise_std;
            v += sd * forecast_normal(noise_seed, grid, comp, t, h);
        }
        obs[h * obs_stride] = (OT)obs_series_value(v, in, h > 0, lo, hi, fill, sp);
    }
}

// the state columns of grid i as they stand in the columns: 4 per genset, then 2 per battery; value j at dst[j * stride]
template <int F, typename OT>
__device__ inline void observe_state_multi(const KArgs &a, int64_t i, OT *__restrict__ dst, int64_t stride)
{
    const int64_t N = a.N;
    int64_t k = 0;
    if constexpr (F & F_GENSET) {
        for (int j = 0; j < a.n_genset; j++) {
            const int64_t c = (int64_t)j * N + i;
            const uint32_t times = a.c.gen_times[c], st = a.c.gen_status[c];
            const double su = (double)(times & 0xff), wd = (double)((times >> 16) & 0xff);
            dst[(k++) * stride] = (OT)space_norm(0.0, 1.0, (double)(st & 0xff));
            dst[(k++) * stride] = (OT)space_norm(0.0, 1.0, (double)((st >> 8) & 0xff));
            dst[(k++) * stride] = (OT)space_norm(0.0, su, (double)((st >> 16) & 0xff));
            dst[(k++) * stride] = (OT)space_norm(0.0, wd, (double)(st >> 24));
        }
    }
    if constexpr (F & F_BATTERY) {
        for (int j = 0; j < a.n_battery; j++) {
            const int64_t c = (int64_t)j * N + i;
            const double cmin = a.c.bat_min_capacity[c], cmax = a.c.bat_max_capacity[c];
            dst[(k++) * stride] = (OT)space_norm(cmin / cmax, 1.0, a.c.soc[c]);
            dst[(k++) * stride] = (OT)space_norm(cmin, cmax, a.c.charge[c]);
        }
    }
}

// flat order: load windows, pv windows, gensets (4 columns each), batteries (2 each), grid windows (4 (1 + H) each).
// MGX_OBS_ROWS_STATE_ONLY (a.obs_state_only == 1): the row lives in a prefetched ring whose window columns
// obs_windows_k_multi_kernel has written already -- only the genset / battery columns are stored.
// NOISE: the noisy-forecaster form -- only observe_multi_kernel carries it (a noisy batch's step launches write no rows themselves, the host
// sends the rows' launch behind them: with the noise code inlined step_multi_kernel spilled 90 more SGPRs and ran 7 % slower for everyone)
template <int F, typename OT, bool NOISE = false>
__device__ inline void observe_row_multi(const KArgs &a, int64_t i, int32_t t, OT *__restrict__ obs_row)
{
    const int64_t N = a.N;
    const int W = 1 + a.H;
    int k = (a.n_load + a.n_pv) * W;
    observe_state_multi<F, OT>(a, i, obs_row + k, 1);
    if (a.obs_state_only) return;
    k = 0;
    // noisy forecasters: one std per module instance ([n, N] columns), a component id per (kind, instance, component) for the generator
    const uint64_t nseed = a.noise_seed;
    const int ninc = a.noise_increase;
    for (int j = 0; j < a.n_load; j++, k += W) {
        const bool noisy = NOISE && a.c.load_noise_std != nullptr;
        const double sd = noisy ? a.c.load_noise_std[(int64_t)j * N + i] : 0.0;
        observe_series_multi(a.c.load_ts + (int64_t)j * N + i, (int64_t)a.n_load * N, a.T, t, a.H, a.c.load_lo[(int64_t)j * N + i],
                             a.c.load_hi[(int64_t)j * N + i], obs_row + k, 1, noisy, sd, nseed, ninc, i, (uint32_t)j);
    }
    for (int j = 0; j < a.n_pv; j++, k += W) {
        const bool noisy = NOISE && a.c.pv_noise_std != nullptr;
        const double sd = noisy ? a.c.pv_noise_std[(int64_t)j * N + i] : 0.0;
        observe_series_multi(a.c.pv_ts + (int64_t)j * N + i, (int64_t)a.n_pv * N, a.T, t, a.H, a.c.pv_lo[(int64_t)j * N + i],
                             a.c.pv_hi[(int64_t)j * N + i], obs_row + k, 1, noisy, sd, nseed, ninc, i, (uint32_t)(MGX_MAX_MODULES + j));
    }
    k += 4 * a.n_genset + 2 * a.n_battery;
    if constexpr (F & F_GRID) {
        for (int j = 0; j < a.n_grid; j++, k += 4 * W)
            for (int cc = 0; cc < 4; cc++) {
                const int64_t c = ((int64_t)j * 4 + cc) * N + i;
                const bool noisy = NOISE && a.c.grid_noise_std != nullptr;
                const double sd = noisy ? a.c.grid_noise_std[(int64_t)j * N + i] : 0.0;
                observe_series_multi(a.c.grid_ts + c, (int64_t)a.n_grid * 4 * N, a.T, t, a.H, a.c.grid_lo[c], a.c.grid_hi[c],
                                     obs_row + k + cc, 4, noisy, sd, nseed, ninc, i, (uint32_t)(2 * MGX_MAX_MODULES + 4 * j + cc));
            }
    }
}

// One general-path step of grid i on the register form (at most MS modules of a kind), lock-step: what a bucket with several
// modules of a kind runs inside the fleet's launch (fleet_step_kernel_v<true>, round 6) -- the `small` arm of step_multi_kernel
// without per-grid episodes.
template <int F>
__device__ __forceinline__ void step_multi_small_body(const KArgs &a, const void *__restrict__ actions, int32_t t, int normalized,
                                                      double *__restrict__ reward, uint8_t *__restrict__ done, void *__restrict__ obs,
                                                      double *__restrict__ log, int64_t i)
{
    const int A = 2 * a.n_genset + a.n_battery + a.n_grid;
    Outputs o;
    MultiRegs R; MultiStepIn sin;
    load_multi_regs<F>(a, i, R);
    if (a.act_f32) load_multi_step_in<F>(a, (const float *)actions + i * A, i, t, sin);
    else load_multi_step_in<F>(a, (const double *)actions + i * A, i, t, sin);
    step_multi_small<F>(a, R, sin, i, normalized != 0, log ? log + i : nullptr, o);
    store_multi_state<F>(a, i, R);
    reward[i] = shaped_reward<F>(a.shaper, o);
    if (done) done[i] = done_at(a, i, t);
    if (obs) {
        if (a.obs_state_only == 1 && a.obs_colpitch) {            // the state columns of a COLUMN-major ring block: coalesced runs
            const int64_t P = a.obs_colpitch, k0 = (int64_t)(a.n_load + a.n_pv) * (1 + a.H);
            if (a.obs_f32) observe_state_multi<F>(a, i, (float *)obs + k0 * P + i, P);
            else observe_state_multi<F>(a, i, (double *)obs + k0 * P + i, P);
        } else if (a.obs_f32) observe_row_multi<F>(a, i, t + 1, (float *)obs + i * a.obs_dim);
        else observe_row_multi<F>(a, i, t + 1, (double *)obs + i * a.obs_dim);
    }
}

// fleet_step_kernel_v for a fleet that holds buckets with several modules of a kind (round 6): such a bucket (head.pad0 = 1; at most
// MS modules of a kind, continuous controls, lock-step) steps on the register form inside the SAME launch instead of in launches of
// its own beside it.  A kernel of its own: the general bodies would cost the plain fleet launch registers it does not need.
static __global__ __launch_bounds__(BLOCK) void fleet_step_kernel_vm(const FleetArgsV fa)
{
    MGX_FLEET_V_HEAD
    if (h8[6]) {
#define MGX_FLEET_V_MULTI(FV) case FV: { step_multi_small_body<FV>(B.k, B.actions, hd.t, normalized, B.reward, B.done, B.obs, B.log, i); } break;
        switch (hd.flags) {
            MGX_FLEET_V_MULTI(0) MGX_FLEET_V_MULTI(1) MGX_FLEET_V_MULTI(2) MGX_FLEET_V_MULTI(3) MGX_FLEET_V_MULTI(4)
            MGX_FLEET_V_MULTI(5) MGX_FLEET_V_MULTI(6) MGX_FLEET_V_MULTI(7) MGX_FLEET_V_MULTI(14)
            default: { step_multi_small_body<15>(B.k, B.actions, hd.t, normalized, B.reward, B.done, B.obs, B.log, i); } break;
        }
#undef MGX_FLEET_V_MULTI
        return;
    }
    MGX_FLEET_V_SWITCHES
}

// EP: in-place per-grid episodes (the lock-step form carries none of it, as step_kernel<F, EP>)
template <int F, bool EP = false>
__global__ __launch_bounds__(BLOCK_MULTI) void step_multi_kernel(const KArgs a, const void *__restrict__ actions, int32_t t,
                                                                 int normalized, double *__restrict__ reward,
                                                                 uint8_t *__restrict__ done, void *__restrict__ obs,
                                                                 double *__restrict__ log, int small)
{
    extern __shared__ double multi_lds[];
    t = resolve_t(a, t);
    const int64_t i = (int64_t)a.g0 + (int64_t)blockIdx.x * BLOCK_MULTI + threadIdx.x;
    if (i < a.g1) {
        const int A = 2 * a.n_genset + a.n_battery + a.n_grid;
        StepLists L = multi_lists(a, multi_lds);
        Outputs o;
        // in-place per-grid episodes (mgx_reset_episodes on the general path, round 6): the grid reads row counter + ep_off[i] of its
        // own [T, n, N] series (a per-lane gather); `tr` is the counter itself otherwise
        int32_t off = 0, tr = t;
        if constexpr (EP) { off = a.ep_off[i]; tr = (int32_t)episode_row(a, t, off); }
        if (small) {                                     // at most MS modules of a kind: everything requested up front, the sweep on registers
            MultiRegs R; MultiStepIn sin;
            load_multi_regs<F>(a, i, R);
            if (a.act_f32) load_multi_step_in<F>(a, (const float *)actions + i * A, i, tr, sin);
            else load_multi_step_in<F>(a, (const double *)actions + i * A, i, tr, sin);
            step_multi_small<F>(a, R, sin, i, normalized != 0, log ? log + i : nullptr, o);
            store_multi_state<F>(a, i, R);
        } else if (a.act_f32) step_multi_core<F>(a, (const float *)actions + i * A, i, tr, normalized != 0, L, log ? log + i : nullptr, o);
        else step_multi_core<F>(a, (const double *)actions + i * A, i, tr, normalized != 0, L, log ? log + i : nullptr, o);
        reward[i] = shaped_reward<F>(a.shaper, o);
        const uint8_t dn = done_at(a, i, t);
        if (done) done[i] = dn;
        if constexpr (EP) { off = episode_auto_restart(a, i, t, off, dn != 0); t += off; }  // (the observation: row counter + 1 + offset)
        if (obs) {
            if (a.obs_state_only == 1 && a.obs_colpitch) {            // the state columns of a COLUMN-major ring block: coalesced runs
                const int64_t P = a.obs_colpitch, k0 = (int64_t)(a.n_load + a.n_pv) * (1 + a.H);
                if (a.obs_f32) observe_state_multi<F>(a, i, (float *)obs + k0 * P + i, P);
                else observe_state_multi<F>(a, i, (double *)obs + k0 * P + i, P);
            } else if (a.obs_f32) observe_row_multi<F>(a, i, t + 1, (float *)obs + i * a.obs_dim);
            else observe_row_multi<F>(a, i, t + 1, (double *)obs + i * a.obs_dim);
        }
    }
    advance_counter_in_kernel(a, 1);
}

template <int F>
__global__ __launch_bounds__(BLOCK_MULTI) void observe_multi_kernel(const KArgs a, int32_t t, void *__restrict__ obs)
{
    t = resolve_t_obs(a, t);
    const int64_t i = (int64_t)a.g0 + (int64_t)blockIdx.x * BLOCK_MULTI + threadIdx.x;
    if (i >= a.g1) return;
    if (a.ep_off) t += a.ep_off[i];                                   // in-place episodes: the grid's own row
    if (a.obs_state_only == 1 && a.obs_colpitch) {                    // state columns of a column-major ring block (step_multi_kernel)
        const int64_t P = a.obs_colpitch, k0 = (int64_t)(a.n_load + a.n_pv) * (1 + a.H);
        if (a.obs_f32) observe_state_multi<F>(a, i, (float *)obs + k0 * P + i, P);
        else observe_state_multi<F>(a, i, (double *)obs + k0 * P + i, P);
        return;
    }
    const bool noisy = a.c.load_noise_std || a.c.pv_noise_std || a.c.grid_noise_std;
    if (noisy) {
        if (a.obs_f32) observe_row_multi<F, float, true>(a, i, t, (float *)obs + i * a.obs_dim);
        else observe_row_multi<F, double, true>(a, i, t, (double *)obs + i * a.obs_dim);
    } else if (a.obs_f32) observe_row_multi<F>(a, i, t, (float *)obs + i * a.obs_dim);
    else observe_row_multi<F>(a, i, t, (double *)obs + i * a.obs_dim);
}

// Window prefetch on the GENERAL path (any number of load / renewable / genset / battery / grid modules per microgrid,
// module_container.py:355-413): the LDS image of obs_windows_k_kernel with run-time component counts.  Per grid
//   [n_series][RP] forecast forms, [n_series][K] current-value forms, [n_state][K] state columns (entry 0 = the current state
//   for a launch at the counter, zeros ahead of it),  n_series = n_load + n_pv + 4 n_grid,  n_state = 4 n_genset + 2 n_battery,
// filled by windows_k_module per module instance (every series value read and normalised ONCE per launch) and written out as
// K row blocks through the same column -> image-offset map; row- or column-major blocks (KArgs.obs_colpitch).
template <int F, typename OT>
__global__ __launch_bounds__(OBS_P1_THREADS) void obs_windows_k_multi_kernel(const KArgs a, const WindowsKPlan plan, int32_t t,
                                                                            OT *__restrict__ ring)
{
    t = resolve_t_obs(a, t);
    extern __shared__ double image_raw[];
    typedef typename std::conditional<sizeof(OT) == 4, float, double>::type IT;       // float rows: a float image (windows_body)
    IT *image = reinterpret_cast<IT *>(image_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int32_t NT = (int32_t)blockDim.x;              // phase 1 on every thread of the launch, phase 2 on the first OBS_K_THREADS (windows_body)
    const int32_t G = plan.group, Q = NT / G, K = plan.K, RP = plan.rp, BP = plan.bp;
    const int32_t g = tid & (G - 1), q = tid / G;
    const int64_t group = (int64_t)plan.group0 + blockIdx.x, g0 = group * G, N = a.N;
    const int32_t W = 1 + a.H, D = a.obs_dim, R = K + a.H;
    const int32_t n_series = a.n_load + a.n_pv + 4 * a.n_grid, n_state = 4 * a.n_genset + 2 * a.n_battery;
    const int32_t NU0 = n_series * RP, S0 = NU0 + n_series * K;
    const int64_t i = g0 + g, ic = i < N ? i : g0;
    IT *blk = image + g * BP;
    uint32_t *map = reinterpret_cast<uint32_t *>(image + G * BP);       // [D]
    int32_t e = 0;
    {
        const double *ts = a.c.load_ts;
        const int64_t n = a.n_load;
        for (int32_t j = 0; j < a.n_load; j++, e++)
            windows_k_module<1, IT>([&](int32_t r, int) { return ts[((int64_t)r * n + j) * N + ic]; }, N, a.c.load_lo + (int64_t)j * N,
                                a.c.load_hi + (int64_t)j * N, a.T, t, R, K, ic, q, Q, blk + e * RP, blk + NU0 + e * K, RP, a.row_mask);
    }
    {
        const double *ts = a.c.pv_ts;
        const int64_t n = a.n_pv;
        for (int32_t j = 0; j < a.n_pv; j++, e++)
            windows_k_module<1, IT>([&](int32_t r, int) { return ts[((int64_t)r * n + j) * N + ic]; }, N, a.c.pv_lo + (int64_t)j * N,
                                a.c.pv_hi + (int64_t)j * N, a.T, t, R, K, ic, q, Q, blk + e * RP, blk + NU0 + e * K, RP, a.row_mask);
    }
    if constexpr (F & F_GRID) {
        const double *ts = a.c.grid_ts;
        const int64_t n = a.n_grid;
        for (int32_t j = 0; j < a.n_grid; j++, e += 4)
            windows_k_module<4, IT>([&](int32_t r, int cc) { return ts[(((int64_t)r * n + j) * 4 + cc) * N + ic]; }, N,
                                a.c.grid_lo + (int64_t)j * 4 * N, a.c.grid_hi + (int64_t)j * 4 * N, a.T, t, R, K, ic, q, Q,
                                blk + e * RP, blk + NU0 + e * K, RP, a.row_mask);
    }
    if (q == 0) {                                        // state strips (windows_body): one strip of zeros ahead of the counter
        const int32_t n_zero = plan.with_state ? n_state * K : K;
        for (int32_t j = 0; j < n_zero; j++) blk[S0 + j] = (IT)0.0;
        if (plan.with_state) observe_state_multi<F, IT>(a, ic, blk + S0, K);
    }
    const int32_t nw = (a.n_load + a.n_pv) * W;
    for (int32_t col = tid; col < D; col += NT) {                       // column -> offset of its k = 0 entry in a block
        uint32_t m;
        if (col < nw) {
            const int32_t s = col / W, h = col - s * W;
            m = h == 0 ? NU0 + s * K : s * RP + h;
        } else if (col < nw + n_state) {
            m = S0 + (plan.with_state ? (col - nw) * K : 0);
        } else {
            const int32_t c = col - nw - n_state, j = c / (4 * W), cj = c - j * 4 * W;
            const int32_t s = a.n_load + a.n_pv + 4 * j + (cj & 3), h = cj >> 2;
            m = h == 0 ? NU0 + s * K : s * RP + h;
        }
        map[col] = m;
    }
    __syncthreads();
    if (tid >= OBS_K_THREADS) return;                    // phase 2: the first OBS_K_THREADS threads (no barrier follows)
    const int32_t n_valid = (N - g0 < G) ? (int32_t)(N - g0) : G;
    const int32_t total = n_valid * D;
    typedef OT vec2 __attribute__((ext_vector_type(2)));
    constexpr int KW = OBS_K_THREADS / 64;
    if (a.obs_colpitch) {
        // COLUMN-major blocks (windows_body): value (block k, column c, grid g0 + g) at (k * D + c) * P + g0 + g -- the G grids
        // of a (k, c) pair are whole 128-byte lines (32 doubles: two, 32 floats: one).  A ring written ahead of the counter leaves the state columns to the
        // steps: here they are lines of their own, so not writing them costs nothing (7 % of a 162-column row).
        const int64_t P = a.obs_colpitch, kstride = (int64_t)D * P;
        if (plan.pairs && store_colmajor_packs<OT, IT>(image, BP, map, D, plan.with_state ? 0x7fffffff : S0, K, G, g0, N, P, ring, tid)) return;
        const int32_t Q2 = OBS_K_THREADS / G;
        const bool in_batch = i < N;
        OT *outc = ring + g0 + g;
        for (int32_t c = q; c < D; c += Q2) {
            const uint32_t m = map[c];
            if (!plan.with_state && m >= (uint32_t)S0) continue;
            const IT *src = image + g * BP + m;
            OT *o = outc + (int64_t)c * P;
            int32_t k = 0;
            if constexpr (WIN_U > 1) {
                for (; k + WIN_U <= K; k += WIN_U) {
                    IT v[WIN_U];
#pragma unroll
                    for (int u = 0; u < WIN_U; u++) v[u] = src[k + u];
#pragma unroll
                    for (int u = 0; u < WIN_U; u++)
                        if (in_batch) MGX_WIN_STORE((OT)v[u], o + (int64_t)(k + u) * kstride);
                }
            }
            for (; k < K; k++)
                if (in_batch) MGX_WIN_STORE((OT)src[k], o + (int64_t)k * kstride);
        }
        return;
    }
    const int64_t block_stride = (int64_t)plan.pitch * D;
    OT *out0 = ring + ((int64_t)wave * plan.pitch + g0) * D;
    // pairs of elements where they never straddle two rows (D even) and the ring is 16-byte aligned; else element by element
    if (!(D & 1) && (reinterpret_cast<uintptr_t>(ring) & (sizeof(vec2) - 1)) == 0) {
        int32_t r = 2 * lane / D, c = 2 * lane - r * D;
        for (int32_t f = 2 * lane; f < total; f += 128) {
            const IT *s0 = image + r * BP + map[c] + wave, *s1 = image + r * BP + map[c + 1] + wave;
            OT *out = out0 + f;
            const int64_t kw_stride = KW * block_stride;
            int32_t k = wave;
            if constexpr (WIN_U > 1) {                   // WIN_U pairs of image words in flight before the first store of the batch
                for (; k + (WIN_U - 1) * KW < K; k += WIN_U * KW) {
                    IT a0[WIN_U], a1[WIN_U];
#pragma unroll
                    for (int u = 0; u < WIN_U; u++) { a0[u] = s0[u * KW]; a1[u] = s1[u * KW]; }
#pragma unroll
                    for (int u = 0; u < WIN_U; u++) {
                        vec2 v2;
                        v2.x = (OT)a0[u]; v2.y = (OT)a1[u];
                        MGX_WIN_STORE(v2, reinterpret_cast<vec2 *>(out + u * kw_stride));
                    }
                    s0 += WIN_U * KW; s1 += WIN_U * KW; out += WIN_U * kw_stride;
                }
            }
            for (; k < K; k += KW) {
                vec2 v2;
                v2.x = (OT)*s0; v2.y = (OT)*s1;
                MGX_WIN_STORE(v2, reinterpret_cast<vec2 *>(out));
                s0 += KW; s1 += KW; out += kw_stride;
            }
            c += 128;
            while (c >= D) { c -= D; r++; }
        }
    } else {
        int32_t r = lane / D, c = lane - r * D;
        for (int32_t f = lane; f < total; f += 64) {
            const IT *s0 = image + r * BP + map[c] + wave;
            OT *out = out0 + f;
            for (int32_t k = wave; k < K; k += KW) {
                MGX_WIN_STORE((OT)*s0, out);
                s0 += KW; out += KW * block_stride;
            }
            c += 64;
            while (c >= D) { c -= D; r++; }
        }
    }
}

// dry run (mgx_check_step) on the general path: the violations of the controllable instances depend on their own
// state and request only, never on the other modules
template <int F>
__global__ __launch_bounds__(BLOCK_MULTI) void check_multi_kernel(const KArgs a, const void *__restrict__ actions, int32_t t,
                                                                  int normalized, uint32_t *__restrict__ violations)
{
    t = resolve_t(a, t);
    const int64_t i = (int64_t)a.g0 + (int64_t)blockIdx.x * BLOCK_MULTI + threadIdx.x;
    if (i >= a.g1) return;
    if (a.ep_off) t = (int32_t)episode_row(a, t, a.ep_off[i]);        // in-place episodes: the grid's own row
    const int64_t N = a.N;
    const int NG = a.n_genset, NB = a.n_battery, NR = a.n_grid, A = 2 * NG + NB + NR;
    auto ld = [&](int j) { return a.act_f32 ? (double)((const float *)actions)[i * A + j] : ((const double *)actions)[i * A + j]; };
    uint32_t viol = 0u;
    Inputs in; in.load = 0.0; in.pv = 0.0;
    Outputs oc;
    if constexpr (F & F_GENSET)
        for (int j = 0; j < NG; j++) {
            Params p; Derived d; State s;
            load_module_params<F_GENSET>(a.c, (int64_t)j * N + i, p); derive<F_GENSET>(p, d);
            s.status = a.c.gen_status[(int64_t)j * N + i];
            in.a_goal = ld(2 * j); in.a_gen = ld(2 * j + 1);
            step_core<F_GENSET>(p, d, s, in, normalized != 0, false, false, oc);
            viol |= oc.violations;
        }
    if constexpr (F & F_BATTERY)
        for (int j = 0; j < NB; j++) {
            Params p; Derived d; State s;
            load_module_params<F_BATTERY>(a.c, (int64_t)j * N + i, p); derive<F_BATTERY>(p, d);
            s.charge = a.c.charge[(int64_t)j * N + i]; s.soc = 0.0; s.status = 0u;
            in.a_bat = ld(2 * NG + j);
            step_core<F_BATTERY>(p, d, s, in, normalized != 0, false, false, oc);
            viol |= oc.violations;
        }
    if constexpr (F & F_GRID)
        for (int j = 0; j < NR; j++) {
            Params p; Derived d; State s;
            load_module_params<F_GRID>(a.c, (int64_t)j * N + i, p); derive<F_GRID>(p, d);
            s.charge = 0.0; s.soc = 0.0; s.status = 0u;
            in.g_pimp = 0.0; in.g_pexp = 0.0; in.g_co2 = 0.0;
            in.g_stat = a.c.grid_ts[(((int64_t)t * NR + j) * 4 + 3) * N + i];
            in.a_grid = ld(2 * NG + NB + j);
            step_core<F_GRID>(p, d, s, in, normalized != 0, false, false, oc);
            viol |= oc.violations;
        }
    violations[i] = viol;
}

// BaseMicrogridModule.sample_action(strict_bound=True) (base_module.py:326-356): the NORMALISED interval a module's action may be
// drawn from so that it satisfies the instantaneous bounds -- [normalize(-max_consumption), normalize(max_production)] of the
// battery (battery_module.py:283-291,332-338) and the grid (grid_module.py:125-132,314-320: limits x grid_status[t]) at the
// CURRENT state and row; a NaN bound is 0 (:349-355).  Genset columns get [0, 1] (the reference itself fails on them, SURVEY Q4:
// the host refuses such layouts before it gets here).  lo / hi: [N, A], the action layout.  Any multiplicity (columns [n, N]).
template <int F>
__global__ __launch_bounds__(BLOCK_MULTI) void action_bounds_kernel(const KArgs a, int32_t t, double *__restrict__ lo, double *__restrict__ hi)
{
    t = resolve_t(a, t);
    const int64_t i = (int64_t)a.g0 + (int64_t)blockIdx.x * BLOCK_MULTI + threadIdx.x;
    if (i >= a.g1) return;
    const int64_t N = a.N;
    const int NG = a.n_genset, NB = a.n_battery, NR = a.n_grid, A = 2 * NG + NB + NR;
    double *l = lo + i * A, *h = hi + i * A;
    auto unit = [](double v) { return v != v ? 0.0 : v; };                   // np.isnan(bound) -> 0
    for (int j = 0; j < 2 * NG; j++) { l[j] = 0.0; h[j] = 1.0; }
    if constexpr (F & F_BATTERY)
        for (int j = 0; j < NB; j++) {
            Params p; Derived d;
            load_module_params<F_BATTERY>(a.c, (int64_t)j * N + i, p); derive<F_BATTERY>(p, d);
            const double c = a.c.charge[(int64_t)j * N + i];
            l[2 * NG + j] = unit((-1 * battery_max_consumption(p, c) - d.bat_lo) / d.bat_sp);      // space.py:207-218
            h[2 * NG + j] = unit((battery_max_production(p, c) - d.bat_lo) / d.bat_sp);
        }
    if constexpr (F & F_GRID)
        for (int j = 0; j < NR; j++) {
            Params p; Derived d;
            load_module_params<F_GRID>(a.c, (int64_t)j * N + i, p); derive<F_GRID>(p, d);
            // one GridModule: whichever way the batch holds its series; several: [T, n_grid, 4, N] arrays
            const int64_t row = series_row(a, i, t);
            const double stat = NR == 1 ? series_component(a.c, N, 5, row, i, a.pm_pitch) : a.c.grid_ts[((row * NR + j) * 4 + 3) * N + i];
            l[2 * NG + NB + j] = unit((-1 * (p.grid_exp * stat) - d.grid_lo) / d.grid_sp);
            h[2 * NG + NB + j] = unit((p.grid_imp * stat - d.grid_lo) / d.grid_sp);
        }
}

// K consecutive steps of the general path in ONE launch (mgx_step_k / mgx_rollout_lists on layouts with several modules of a
// kind): a loop around step_multi_core -- the state columns are re-read every step (cache hits), nothing is kept in
// registers across steps.  With `lists` the control of every step is expanded on device from the grid's priority list
// (ids [K, N] with per_step, else one fixed list per grid = RuleBasedControl.run) and applied unnormalised.
// soc_trace / status_trace report battery 0 / genset 0.
constexpr int MGX_MAX_ACTIONS_MULTI = 4 * MGX_MAX_INSTANCES;

template <int F>
__global__ __launch_bounds__(BLOCK_MULTI) void step_k_multi_kernel(const KArgs a, const void *__restrict__ actions,
                                                                   const int32_t *__restrict__ lists, int32_t n_lists, int32_t list_len,
                                                                   const int32_t *__restrict__ ids, int per_step, int32_t t0, int32_t K,
                                                                   int normalized, const FusedOut out, int small)
{
    extern __shared__ double multi_lds[];
    const int32_t K_launch = K;
    t0 = resolve_t(a, t0);
    K = resolve_k(a, t0, K);
    const int64_t i = (int64_t)a.g0 + (int64_t)blockIdx.x * BLOCK_MULTI + threadIdx.x;
    if (i < a.g1) {
        const int64_t N = a.N;
        const int A = 2 * a.n_genset + a.n_battery + a.n_grid;
        StepLists L = multi_lists(a, multi_lds);
        const int32_t k_done = (a.grid_final ? a.grid_final[i] : a.final_step) - 1 - t0;
        double ret = 0.0;
        if (small && !lists) {
            // At most MS modules of a kind, continuous controls: parameters and state stay in REGISTERS for the K steps (loaded
            // once, the state written back once); per step only the controls and the series rows are read -- the NEXT step's are
            // requested before this step's sweep starts -- and the requested outputs written.
            MultiRegs R; MultiStepIn cur, nxt;
            load_multi_regs<F>(a, i, R);
            if (K > 0) {
                if (a.act_f32) load_multi_step_in<F>(a, (const float *)actions + (int64_t)i * A, i, t0, cur);
                else load_multi_step_in<F>(a, (const double *)actions + (int64_t)i * A, i, t0, cur);
            }
            for (int32_t k = 0; k < K; k++) {
                const int64_t off = (int64_t)k * N + i;
                const int32_t kn = k + 1 < K ? k + 1 : k;                // (the last step re-reads its own inputs: unconditional loads)
                const int64_t offn = (int64_t)kn * N + i;
                if (a.act_f32) load_multi_step_in<F>(a, (const float *)actions + offn * A, i, t0 + kn, nxt);
                else load_multi_step_in<F>(a, (const double *)actions + offn * A, i, t0 + kn, nxt);
                Outputs o;
                double *log = out.log ? out.log + (int64_t)k * a.log_dim * N + i : nullptr;
                step_multi_small<F>(a, R, cur, i, normalized != 0, log, o);
                const double r = shaped_reward<F>(a.shaper, o);
                if (out.reward) out.reward[off] = r;
                if (out.done) out.done[off] = (uint8_t)(k >= k_done);
                if constexpr (F & F_BATTERY) { if (out.soc_trace) out.soc_trace[off] = R.b_soc[0]; }
                if constexpr (F & F_GENSET) { if (out.status_trace) out.status_trace[off] = R.g_status[0]; }
                ret += r;
                cur = nxt;
            }
            store_multi_state<F>(a, i, R);
        } else
        for (int32_t k = 0; k < K; k++) {
            const int64_t off = (int64_t)k * N + i;
            Outputs o;
            double *log = out.log ? out.log + (int64_t)k * a.log_dim * N + i : nullptr;
            if (lists) {
                double ctrl[MGX_MAX_ACTIONS_MULTI];
                int32_t id = per_step ? ids[off] : ids[i];
                id = (id >= 0 && id < n_lists) ? id : 0;
                const uint32_t xv = populate_multi<F>(a, lists + (int64_t)id * list_len * 3, list_len, i, t0 + k, ctrl);
                step_multi_core<F>(a, (const double *)ctrl, i, t0 + k, false, L, log, o, xv);
            } else if (a.act_f32) {
                step_multi_core<F>(a, (const float *)actions + off * A, i, t0 + k, normalized != 0, L, log, o);
            } else {
                step_multi_core<F>(a, (const double *)actions + off * A, i, t0 + k, normalized != 0, L, log, o);
            }
            const double r = shaped_reward<F>(a.shaper, o);
            if (out.reward) out.reward[off] = r;
            if (out.done) out.done[off] = (uint8_t)(k >= k_done);
            if constexpr (F & F_BATTERY) { if (out.soc_trace) out.soc_trace[off] = a.c.soc[i]; }
            if constexpr (F & F_GENSET) { if (out.status_trace) out.status_trace[off] = a.c.gen_status[i]; }
            ret += r;
        }
        if (out.ret_acc) out.ret_acc[i] += ret;
    }
    advance_counter_in_kernel(a, K_launch);
}

// The register form of the K-step launch alone (layouts of at most MS modules of a kind, continuous controls): the same loop as the
// `small` arm of step_k_multi_kernel in a kernel of its own -- without the run-time-count arm and the priority-list arm beside it the
// loop keeps its pointers in SGPRs (the shared kernel reloaded ~100 spilled SGPRs per step through v_readlane).
// CNT: the instance counts, run-time (CountsRT: any small layout) or compile-time (CountsCT: the layouts of mgx_fused.hip part 5).
#ifndef MGX_M3_WAVES
#define MGX_M3_WAVES 2
#endif
#ifndef MGX_M3_PARK
#define MGX_M3_PARK 1
#endif
template <int F, class CNT = CountsRT, int M = MS>
__global__ __launch_bounds__(BLOCK_MULTI, (M >= 3 ? MGX_M3_WAVES : 1)) void step_k_multi_small_kernel(const KArgs a, const void *__restrict__ actions, int32_t t0, int32_t K,
                                                                         int normalized, const FusedOut out)
{
    const int32_t K_launch = K;
    t0 = resolve_t(a, t0);
    K = resolve_k(a, t0, K);
    const int64_t i = (int64_t)a.g0 + (int64_t)blockIdx.x * BLOCK_MULTI + threadIdx.x;
    // M = 3: the once-per-step parameters wait in LDS (park_multi_regs, mgx_core.hpp) -- two waves per SIMD instead of one
    constexpr bool PARK = M >= 3 && MGX_M3_PARK;
    static_assert(BLOCK_MULTI == PARK_STRIDE, "one LDS slot row per workgroup lane");
    __shared__ double park[PARK ? ParkSlots<CNT, M>::COUNT * PARK_STRIDE : 1];
    lds_double *pk = (lds_double *)park + threadIdx.x;
    if (i < a.g1) {
        const int64_t N = a.N;
        const int A = 2 * CNT::ng(a) + CNT::nb(a) + CNT::nr(a);
        const int32_t k_done = (a.grid_final ? a.grid_final[i] : a.final_step) - 1 - t0;
        double ret = 0.0;
        MultiRegsT<M> R; MultiStepInT<M> cur, nxt;
        load_multi_regs<F, CNT, M>(a, i, R);
        if constexpr (PARK) park_multi_regs<F, CNT, M>(a, R, pk);
        if (K > 0) {
            if (a.act_f32) load_multi_step_in<F, float, CNT, M>(a, (const float *)actions + (int64_t)i * A, i, t0, cur);
            else load_multi_step_in<F, double, CNT, M>(a, (const double *)actions + (int64_t)i * A, i, t0, cur);
        }
        // two steps per trip: the inputs of step k + 1 are requested into `nxt` before step k runs on `cur`, those of step k + 2 into
        // `cur` before step k + 1 runs on `nxt` -- no copy of 26 doubles per step between the two buffers
        auto fetch = [&](int32_t kk, MultiStepInT<M> &dst) __attribute__((always_inline)) {
            const int32_t kc = kk < K ? kk : K - 1;                  // (past the end: re-read the last step's inputs, unconditional loads)
            const int64_t offc = (int64_t)kc * N + i;
            if (a.act_f32) load_multi_step_in<F, float, CNT, M>(a, (const float *)actions + offc * A, i, t0 + kc, dst);
            else load_multi_step_in<F, double, CNT, M>(a, (const double *)actions + offc * A, i, t0 + kc, dst);
        };
        auto one_step = [&](int32_t k, const MultiStepInT<M> &in) __attribute__((always_inline)) {
            const int64_t off = (int64_t)k * N + i;
            Outputs o;
            double *log = out.log ? out.log + (int64_t)k * a.log_dim * N + i : nullptr;
            step_multi_small<F, CNT, M, PARK>(a, R, in, i, normalized != 0, log, o, pk);
            const double r = shaped_reward<F>(a.shaper, o);
            // (write-once [K, N] streams: non-temporal stores, as in the single-instance fused kernels)
            if (out.reward) __builtin_nontemporal_store(r, out.reward + off);
            if (out.done) out.done[off] = (uint8_t)(k >= k_done);
            if constexpr (F & F_BATTERY) { if (out.soc_trace) __builtin_nontemporal_store(R.b_soc[0], out.soc_trace + off); }
            if constexpr (F & F_GENSET) { if (out.status_trace) __builtin_nontemporal_store(R.g_status[0], out.status_trace + off); }
            ret += r;
        };
        int32_t k = 0;
        for (; k + 1 < K; k += 2) {
            fetch(k + 1, nxt);
            one_step(k, cur);
            fetch(k + 2, cur);
            one_step(k + 1, nxt);
        }
        if (k < K) one_step(k, cur);
        // (the state columns' addresses are formed again here from an index the compiler cannot see through: held across the loop since the
        //  prologue's loads they were the registers the three-of-a-kind form spilled)
        int64_t i_out = i;
        asm volatile("" : "+v"(i_out));
        store_multi_state<F, CNT, M>(a, i_out, R);
        if (out.ret_acc) out.ret_acc[i] += ret;
    }
    advance_counter_in_kernel(a, K_launch);
}

// mgx_step_lists on the register form (round 6, ABI minor 2): ONE discrete Gym step of a layout with at most MS modules of a kind --
// the grid's priority list packed into a word, walked out of registers (populate_multi_small), the step, the observation.  What
// DiscreteMicrogridEnv.step was as two launches (expand_multi_kernel -> control [N, A] -> step_multi_kernel).
template <int F>
__global__ __launch_bounds__(BLOCK_MULTI) void step_lists_small_kernel(const KArgs a, const int32_t *__restrict__ ids,
                                                                       const int32_t *__restrict__ lists, int32_t n_lists, int32_t list_len,
                                                                       int32_t t, double *__restrict__ control, double *__restrict__ reward,
                                                                       uint8_t *__restrict__ done, void *__restrict__ obs,
                                                                       double *__restrict__ log)
{
    t = resolve_t(a, t);
    const int64_t i = (int64_t)a.g0 + (int64_t)blockIdx.x * BLOCK_MULTI + threadIdx.x;
    if (i < a.g1) {
        const int NG = a.n_genset, NB = a.n_battery, NR = a.n_grid;
        Outputs o;
        MultiRegs R; MultiStepIn sin;
        load_multi_regs<F>(a, i, R);
        load_multi_series<F>(a, i, t, sin);
        int32_t id = ids[i];
        id = (id >= 0 && id < n_lists) ? id : 0;              // ids outside [0, n) fall back to list 0 (the reference raises)
        const uint64_t plw = pack_priority_list(lists + (int64_t)id * list_len * 3, list_len, NG, NB, NR);
        const uint32_t xv = populate_multi_small<F, CountsRT, MS>(a, R, plw, sin, nullptr, log != nullptr);
        if (control) {                                         // the expanded control, in mgx_expand_lists' column order
            double *c = control + i * (2 * NG + NB + NR);
#pragma unroll
            for (int j = 0; j < MS; j++) {
                if (j < NG) { c[2 * j] = sin.goal[j]; c[2 * j + 1] = sin.gen[j]; }
                if (j < NB) c[2 * NG + j] = sin.bat[j];
                if (j < NR) c[2 * NG + NB + j] = sin.grd[j];
            }
        }
        step_multi_small<F>(a, R, sin, i, false, log ? log + i : nullptr, o, nullptr, xv);
        store_multi_state<F>(a, i, R);
        reward[i] = shaped_reward<F>(a.shaper, o);
        if (done) done[i] = done_at(a, i, t);
        if (obs) {
            if (a.obs_state_only == 1 && a.obs_colpitch) {
                const int64_t P = a.obs_colpitch, k0 = (int64_t)(a.n_load + a.n_pv) * (1 + a.H);
                if (a.obs_f32) observe_state_multi<F>(a, i, (float *)obs + k0 * P + i, P);
                else observe_state_multi<F>(a, i, (double *)obs + k0 * P + i, P);
            } else if (a.obs_f32) observe_row_multi<F>(a, i, t + 1, (float *)obs + i * a.obs_dim);
            else observe_row_multi<F>(a, i, t + 1, (double *)obs + i * a.obs_dim);
        }
    }
    advance_counter_in_kernel(a, 1);
}

// mgx_rollout_lists on the register form (round 6): K fused steps whose controls come from a priority list over module instances --
// RuleBasedControl on a layout with several modules of a kind (rbc.py:64-93; ids [N]: one fixed list per grid) or a discrete roll-out
// (ids [K, N]).  The loop of step_k_multi_small_kernel with populate_multi_small in place of the action stream: per step only the
// series rows are read.  The list is packed into one 64-bit word (pack_priority_list): once per launch for fixed lists.
template <int F, class CNT, int M>
__global__ __launch_bounds__(BLOCK_MULTI, (M >= 3 ? MGX_M3_WAVES : 1)) void rollout_multi_small_kernel(
    const KArgs a, const int32_t *__restrict__ lists, int32_t n_lists, int32_t list_len, const int32_t *__restrict__ ids, int per_step,
    int32_t t0, int32_t K, const FusedOut out)
{
    static_assert(CNT::kNG + CNT::kNB + CNT::kNR_ <= PL_PACK_MAX, "a packed list holds every controllable module of the layout");
    const int32_t K_launch = K;
    t0 = resolve_t(a, t0);
    K = resolve_k(a, t0, K);
    const int64_t i = (int64_t)a.g0 + (int64_t)blockIdx.x * BLOCK_MULTI + threadIdx.x;
    constexpr bool PARK = M >= 3 && MGX_M3_PARK;
    __shared__ double park[PARK ? ParkSlots<CNT, M>::COUNT * PARK_STRIDE : 1];
    lds_double *pk = (lds_double *)park + threadIdx.x;
    if (i < a.g1) {
        const int64_t N = a.N;
        const int32_t k_done = (a.grid_final ? a.grid_final[i] : a.final_step) - 1 - t0;
        double ret = 0.0;
        MultiRegsT<M> R; MultiStepInT<M> cur, nxt;
        load_multi_regs<F, CNT, M>(a, i, R);
        if constexpr (PARK) park_multi_regs<F, CNT, M>(a, R, pk);
        auto list_word = [&](int32_t id) __attribute__((always_inline)) {
            id = (id >= 0 && id < n_lists) ? id : 0;            // ids outside [0, n) fall back to list 0 (the reference raises)
            return pack_priority_list(lists + (int64_t)id * list_len * 3, list_len, CNT::ng(a), CNT::nb(a), CNT::nr(a));
        };
        uint64_t plw = per_step ? 0ull : list_word(ids[i]);
        if (K > 0) load_multi_series<F, CNT, M>(a, i, t0, cur);
        auto fetch = [&](int32_t kk, MultiStepInT<M> &dst) __attribute__((always_inline)) {
            const int32_t kc = kk < K ? kk : K - 1;                  // (past the end: re-read the last step's rows, unconditional loads)
            load_multi_series<F, CNT, M>(a, i, t0 + kc, dst);
        };
        auto one_step = [&](int32_t k, MultiStepInT<M> &in) __attribute__((always_inline)) {
            const int64_t off = (int64_t)k * N + i;
            if (per_step) plw = list_word(ids[off]);
            const uint32_t xv = populate_multi_small<F, CNT, M, PARK>(a, R, plw, in, pk, out.log != nullptr);
            Outputs o;
            double *log = out.log ? out.log + (int64_t)k * a.log_dim * N + i : nullptr;
            step_multi_small<F, CNT, M, PARK>(a, R, in, i, false, log, o, pk, xv);
            const double r = shaped_reward<F>(a.shaper, o);
            if (out.reward) __builtin_nontemporal_store(r, out.reward + off);
            if (out.done) out.done[off] = (uint8_t)(k >= k_done);
            if constexpr (F & F_BATTERY) { if (out.soc_trace) __builtin_nontemporal_store(R.b_soc[0], out.soc_trace + off); }
            if constexpr (F & F_GENSET) { if (out.status_trace) __builtin_nontemporal_store(R.g_status[0], out.status_trace + off); }
            ret += r;
        };
        int32_t k = 0;
        for (; k + 1 < K; k += 2) {
            fetch(k + 1, nxt);
            one_step(k, cur);
            fetch(k + 2, cur);
            one_step(k + 1, nxt);
        }
        if (k < K) one_step(k, cur);
        int64_t i_out = i;
        asm volatile("" : "+v"(i_out));
        store_multi_state<F, CNT, M>(a, i_out, R);
        if (out.ret_acc) out.ret_acc[i] += ret;
    }
    advance_counter_in_kernel(a, K_launch);
}

// mgx_expand_lists / mgx_expand_discrete on the general path: lists [n_lists, list_len, 3] in device memory
template <int F>
__global__ __launch_bounds__(BLOCK_MULTI) void expand_multi_kernel(const KArgs a, const int32_t *__restrict__ lists, int32_t n_lists,
                                                                   int32_t list_len, const int32_t *__restrict__ action_id,
                                                                   int32_t t, double *__restrict__ control,
                                                                   uint32_t *__restrict__ violations)
{
    t = resolve_t(a, t);
    const int64_t i = (int64_t)a.g0 + (int64_t)blockIdx.x * BLOCK_MULTI + threadIdx.x;
    if (i >= a.g1) return;
    const int A = 2 * a.n_genset + a.n_battery + a.n_grid;
    if (a.ep_off) t = (int32_t)episode_row(a, t, a.ep_off[i]);        // in-place episodes: the grid's own row
    int32_t id = action_id[i];
    id = (id >= 0 && id < n_lists) ? id : 0;              // ids outside [0, n) fall back to list 0 (the reference raises)
    const uint32_t xv = populate_multi<F>(a, lists + (int64_t)id * list_len * 3, list_len, i, t, control + i * A);
    if (violations) violations[i] = xv;
}

// device-resident step counter (hipGraph-replayable stepping): counter[0] = t, counter[1] = overrun flag
static __global__ void set_counter_kernel(int32_t *counter, int32_t t) { counter[0] = t; counter[1] = 0; counter[2] = 0; }

// ------------------------------------------------------------------------------------------------------
// Metrics: deterministic column sums  sums[m] = sum_i values[m, i].
// Stage 1: every block folds a fixed slice of column m (lane-strided running sums, then a 64-lane
// wavefront shuffle tree, then the 4 wave results through LDS).  Stage 2: one block per column folds the
// per-block partials the same way.  No atomics: the order is fixed by (N, grid size) alone.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

__device__ __forceinline__ double block_sum(double v, double *lds)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 0; w < BLOCK / 64; w++) r += lds[w];
    }
    __syncthreads();
    return r;      // valid in thread 0
}

static __global__ __launch_bounds__(BLOCK) void colsum_stage1(const double *__restrict__ values, int64_t N, int32_t per_block,
                                                       double *__restrict__ partial)
{
    __shared__ double lds[BLOCK / 64];
    const int m = blockIdx.y;
    const int64_t begin = (int64_t)blockIdx.x * per_block;
    int64_t end = begin + per_block; if (end > N) end = N;
    const double *col = values + (int64_t)m * N;
    double acc = 0.0;
    for (int64_t i = begin + threadIdx.x; i < end; i += BLOCK) acc += col[i];
    const double r = block_sum(acc, lds);
    if (threadIdx.x == 0) partial[(int64_t)m * gridDim.x + blockIdx.x] = r;
}

static __global__ __launch_bounds__(BLOCK) void colsum_stage2(const double *__restrict__ partial, int32_t n_partial,
                                                       double *__restrict__ sums)
{
    __shared__ double lds[BLOCK / 64];
    const int m = blockIdx.x;
    double acc = 0.0;
    for (int i = threadIdx.x; i < n_partial; i += BLOCK) acc += partial[(int64_t)m * n_partial + i];
    const double r = block_sum(acc, lds);
    if (threadIdx.x == 0) sums[m] = r;
}

// ------------------------------------------------------------------------------------------------------
// Per-grid episode windows (mgx_reset_windows).  Every reference Microgrid owns its step counter and draws its own
// trajectory at reset (microgrid.py:205-225, base_module.py:65-77,292-296, trajectory/stochastic.py:15-30).  The batch
// keeps ONE counter: at reset the rows [start_i, start_i + R) of every grid's series are gathered into window buffers
// [R, N] that the step kernels then walk from row 0 -- coalesced, like the full series.  Rows beyond the end of the
// series receive the forecaster's padding value (lo + hi) / 2 (forecaster.py:95,120-137), so the observation kernels
// need no per-grid series length.  One lane per grid: its reads are one line per row (once per episode), the writes
// are coalesced.
// ------------------------------------------------------------------------------------------------------
// [T, PP] base table -> profile-major [PP, pitch] (mgx_reset_episodes: in-place episodes read one row per lane)
static __global__ __launch_bounds__(BLOCK) void profile_major_kernel(const double *__restrict__ src, double *__restrict__ dst,
                                                                     int32_t T, int32_t pitch)
{
    const int64_t e = (int64_t)blockIdx.x * BLOCK + threadIdx.x;          // element of the destination
    if (e >= (int64_t)PP * pitch) return;
    const int32_t p = (int32_t)(e / pitch), row = (int32_t)(e - (int64_t)p * pitch);
    dst[e] = row < T ? src[(int64_t)row * PP + p] : 0.0;
}

// [T, Cs, N] series -> the grid-major copy [N, pitch, C] at component offset c0 (mgx_reset_episodes on [T, N] series: every
// lane reads its own row; the C values of a row are adjacent, consecutive rows of a grid share lines).  One workgroup per
// 32 x 32 (row, grid) tile of one source component.  A one-off pass per reset: the strided writes do not matter.
static __global__ __launch_bounds__(256) void grid_major_kernel(const double *__restrict__ src, double *__restrict__ dst, int64_t N,
                                                                int32_t T, int32_t Cs, int32_t pitch, int32_t C, int32_t c0)
{
    __shared__ double tile[32][33];
    const int32_t tx = threadIdx.x & 31, ty = threadIdx.x >> 5;        // 32 x 8
    const int64_t i0 = (int64_t)blockIdx.x * 32;
    const int32_t t0 = (int32_t)blockIdx.y * 32, cs = (int32_t)blockIdx.z;
    for (int32_t r = ty; r < 32; r += 8) {                             // read: lanes along the grids
        const int32_t t = t0 + r;
        const int64_t i = i0 + tx;
        tile[r][tx] = (t < T && i < N) ? src[((int64_t)t * Cs + cs) * N + i] : 0.0;
    }
    __syncthreads();
    for (int32_t g = ty; g < 32; g += 8) {                             // write: lanes along the rows
        const int64_t i = i0 + g;
        const int32_t t = t0 + tx;
        if (i < N && t < pitch) dst[(i * pitch + t) * C + c0 + cs] = tile[tx][g];
    }
}

struct GatherArgs {
    const double *load_ts, *pv_ts, *grid_ts;
    const double *load_lo, *load_hi, *pv_lo, *pv_hi, *grid_lo, *grid_hi;
    double *load_w, *pv_w, *grid_w;
    const int32_t *start, *length;
    int32_t *final_rel;
    int32_t N, T, rows, max_length, lo, hi;
    const uint8_t *mask;      // NULL = every grid; else only grids with mask[i] != 0 (a partial reset)
    int32_t row0, row_mask;   // destination row of source row start_i + r: (row0 + r) & row_mask (linear windows: 0, -1)
    mgx_columns fc;           // factorised source (fc.base_load != NULL): the rows are formed from the factors
    int32_t has_grid;
    // mgx_reset_grids_random: start / length are DRAWN here (trajectory/stochastic.py:9-30 per grid) instead of read
    int32_t draw, fixed_length;
    uint64_t seed;
    int32_t *start_io, *length_io, *t0_io;     // optional [N]: what the restarted grids got
    int32_t *ep_off;          // in-place episodes (mgx_reset_episodes): the row offset is all a (re)start writes; rows == 0
};

// the episode of grid i: start row and length (given, or drawn), clamped into the env's window; bookkeeping outputs
__device__ __forceinline__ void gather_episode(const GatherArgs &g, int64_t i, int32_t &s, int32_t &len)
{
    if (g.draw) {
        episode_draw(g.seed, i, g.row0, g.fixed_length, g.lo, g.hi, s, len);
    } else {
        s = g.start[i];
        len = g.length ? g.length[i] : g.max_length;
    }
    episode_clamp(g.lo, g.hi, g.max_length, s, len);
    if (g.final_rel) g.final_rel[i] = g.row0 + len;                    // counter value at which the episode has run its length
    if (g.start_io) g.start_io[i] = s;
    if (g.length_io) g.length_io[i] = len;
    if (g.t0_io) g.t0_io[i] = g.row0;
    if (g.ep_off) g.ep_off[i] = s - g.row0;                            // in-place episodes: no rows to copy
}

// one lane per grid walks its rows (a wave moves 512 contiguous bytes per row when all of its grids take part).  For the
// sparse restarts of mgx_reset_grids* a variant whose 64 lanes share the rows of each restarting grid was measured SLOWER
// (36 vs 20 us per step at N = 100 000, one grid in 168 restarting: its accesses are 8 bytes per line)
static __global__ __launch_bounds__(BLOCK) void gather_windows_kernel(const GatherArgs g)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= g.N || (g.mask && !g.mask[i])) return;
    int32_t s, len;
    gather_episode(g, i, s, len);
    if (g.ep_off) return;
    // 8 rows per round, every load of the round in flight before its first store (the compiler cannot prove that window and
    // series buffers do not alias: row by row the loop is one memory round trip per row -- 20 us when a few lanes restart)
    const int64_t N = g.N;
    const double fl = (g.load_lo && g.load_hi) ? (g.load_hi[i] + g.load_lo[i]) / 2 : 0.0;
    const double fp = (g.pv_lo && g.pv_hi) ? (g.pv_hi[i] + g.pv_lo[i]) / 2 : 0.0;
    const bool fact = factorised(g.fc);
    const bool has_grid = g.has_grid != 0;
    double fg[4] = {0.0, 0.0, 0.0, 0.0};
    if (has_grid && g.grid_lo && g.grid_hi) {
#pragma unroll
        for (int c = 0; c < 4; c++) fg[c] = (g.grid_hi[c * N + i] + g.grid_lo[c * N + i]) / 2;
    }
    constexpr int RB = 8;
    for (int32_t r0 = 0; r0 < g.rows; r0 += RB) {
        double vl[RB], vp[RB], vg[RB][4];
#pragma unroll
        for (int u = 0; u < RB; u++) {
            const int64_t row = (int64_t)s + r0 + u;
            const int64_t rc = row < g.T ? row : (int64_t)g.T - 1;
            if (fact) {
                vl[u] = series_component(g.fc, N, 0, rc, i); vp[u] = series_component(g.fc, N, 1, rc, i);
                if (has_grid) {
#pragma unroll
                    for (int c = 0; c < 4; c++) vg[u][c] = series_component(g.fc, N, 2 + c, rc, i);
                }
            } else {
                vl[u] = g.load_ts[rc * N + i]; vp[u] = g.pv_ts[rc * N + i];
                if (has_grid) {
#pragma unroll
                    for (int c = 0; c < 4; c++) vg[u][c] = g.grid_ts[(rc * 4 + c) * N + i];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < RB; u++) {
            const int32_t r = r0 + u;
            if (r < g.rows) {
                const bool in = (int64_t)s + r < g.T;
                const int64_t dst = (g.row0 + r) & g.row_mask;
                g.load_w[dst * N + i] = in ? vl[u] : fl;
                g.pv_w[dst * N + i] = in ? vp[u] : fp;
                if (has_grid) {
#pragma unroll
                    for (int c = 0; c < 4; c++) g.grid_w[(dst * 4 + c) * N + i] = in ? vg[u][c] : fg[c];
                }
            }
        }
    }
}

// The same for layouts with several modules of a kind (mgx_reset_windows on the general path): series [T, C, N] with
// C = n_load / n_pv / 4 n_grid components, bounds [C, N], window buffers [rows, C, N].  One lane per grid, component by
// component, 8 rows in flight; once per episode.
__device__ __forceinline__ void gather_component_rows(const double *__restrict__ src, const double *__restrict__ lo,
                                                      const double *__restrict__ hi, double *__restrict__ dst, int32_t C, int64_t N,
                                                      int32_t T, int32_t rows, int32_t s, int64_t i)
{
    constexpr int RB = 8;
    for (int32_t c = 0; c < C; c++) {
        const double pad = (lo && hi) ? (hi[(int64_t)c * N + i] + lo[(int64_t)c * N + i]) / 2 : 0.0;   // forecaster.py:95,120-137
        for (int32_t r0 = 0; r0 < rows; r0 += RB) {
            double v[RB];
#pragma unroll
            for (int u = 0; u < RB; u++) {
                const int64_t row = (int64_t)s + r0 + u, rc = row < T ? row : (int64_t)T - 1;
                v[u] = src[(rc * C + c) * N + i];
            }
#pragma unroll
            for (int u = 0; u < RB; u++) {
                const int32_t r = r0 + u;
                if (r < rows) dst[((int64_t)r * C + c) * N + i] = ((int64_t)s + r < T) ? v[u] : pad;
            }
        }
    }
}

static __global__ __launch_bounds__(BLOCK) void gather_windows_multi_kernel(const GatherArgs g, int32_t n_load, int32_t n_pv, int32_t n_grid)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= g.N) return;
    int32_t s, len;
    gather_episode(g, i, s, len);
    gather_component_rows(g.load_ts, g.load_lo, g.load_hi, g.load_w, n_load, g.N, g.T, g.rows, s, i);
    gather_component_rows(g.pv_ts, g.pv_lo, g.pv_hi, g.pv_w, n_pv, g.N, g.T, g.rows, s, i);
    if (n_grid > 0) gather_component_rows(g.grid_ts, g.grid_lo, g.grid_hi, g.grid_w, 4 * n_grid, g.N, g.T, g.rows, s, i);
}


// ------------------------------------------------------------------------------------------------------
// Series synthesis (mgx_synthesize_series): MicrogridGenerator's time series for N grids, written at HBM speed.
//   load / pv     base profile x ratio, ratio = size / max(profile)      (_scale_ts 'max', MicrogridGenerator.py:137-147)
//   import price  tariff pattern 1 / 2 by hour of day                    (_get_electricity_tariff, :253-285)
//   co2           base co2 profile, verbatim                             (_get_co2_ts, :205-212)
//   grid status   weak-grid outages: 0 where a uniform draw of rows t .. t+duration-1 falls below outage_per_day / 24;
//                 the back-fill never reaches row 0                      (_generate_weak_grid_profile, :321-340)
// One lane per grid walks the rows from the last to the first (the outage back-fill looks forward in time); every row
// of every output is one coalesced store per wave.  Uniforms: Philox4x32-10 keyed by the seed, counter = (GLOBAL grid
// index, row), so a shard's draw does not depend on how the batch is split over ranks.
// ------------------------------------------------------------------------------------------------------

static __global__ __launch_bounds__(BLOCK) void synthesize_series_kernel(const mgx_synth a)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= a.n_grids) return;
    const int64_t N = a.n_grids;
    const int32_t T = a.n_steps;
    const bool series = a.load_ts != nullptr;              // load / pv rows wanted (else: only the outage words)
    int32_t lp = 0, pp = 0;
    double lr = 0.0, pr = 0.0;
    if (series) { lp = a.load_profile[i]; pp = a.pv_profile[i]; lr = a.load_ratio[i]; pr = a.pv_ratio[i]; }
    const bool grid = a.grid_ts != nullptr;
    int32_t cp = 0, pat = 0, dur = 1;
    double prob = 0.0;
    if (grid) { cp = a.co2_profile[i]; pat = a.tariff[i]; }
    const bool weak = (grid || a.outage_bits != nullptr) && a.outage_per_day != nullptr && a.weak[i] != 0;
    if (weak) {
        prob = a.outage_per_day[i] / 24;                                        // weak_grid_timeseries[i] < outage_per_day/24 (:332)
        dur = a.outage_duration ? a.outage_duration[i] : 1;
    }
    const int64_t gi = a.grid_index ? a.grid_index[i] : a.grid_index0 + i;      // the Philox counter: GLOBAL grid index
    // rows still covered by an outage that starts later: the extra draw of row T (the reference draws T + 1 values) first
    int32_t cover = 0;
    if (weak && synth_uniform(a.seed, gi, T) < prob) cover = dur - 1;
    uint64_t word = 0;                                                          // outage bits of rows [64 w, 64 w + 64)
    for (int32_t t = T - 1; t >= 0; t--) {
        if (series) {
            a.load_ts[(int64_t)t * N + i] = -1.0 * fabs(a.base_load[(int64_t)t * a.n_load_profiles + lp] * lr);   // stored sign
            a.pv_ts[(int64_t)t * N + i] = fabs(a.base_pv[(int64_t)t * a.n_pv_profiles + pp] * pr);
        }
        double status = 1.0;
        if (weak) {
            const bool own = synth_uniform(a.seed, gi, t) < prob;
            status = (own || (cover > 0 && t > 0)) ? 0.0 : 1.0;                 // "if i-j > 0": the back-fill spares row 0
            cover = own ? dur - 1 : (cover > 0 ? cover - 1 : 0);
        }
        if (grid) {
            double *g = a.grid_ts + (int64_t)t * 4 * N + i;
            g[0] = tariff_price(pat, t);
            g[N] = 0.0;                                                       // price_export = zeros (:264)
            g[2 * N] = a.base_co2[(int64_t)t * a.n_co2_profiles + cp];
            g[3 * N] = status;
        }
        if (a.outage_bits) {
            word |= (status == 0.0 ? 1ull : 0ull) << (t & 63);
            if ((t & 63) == 0) { a.outage_bits[(int64_t)(t >> 6) * N + i] = word; word = 0; }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// MicrogridGenerator's draws and sizing rules (mgx_generate_columns; include/mgx.h has the rule list with the reference lines).
// One lane per grid.  The only loop is the mean of the scaled load series: T products summed in numpy's pairwise order
// (DOUBLE_pairwise_sum: blocks of <= 128 with eight interleaved partial sums, halves split at multiples of 8), so that
// ceil(hours x mean) lands on the integer the reference's pandas / numpy arithmetic gives.
// ------------------------------------------------------------------------------------------------------
enum GenQuantity : int32_t {      // Philox counter word 2: one id per drawn quantity (the normals take 12 consecutive ids each)
    GQ_BIN = 0, GQ_SIZE_LOAD, GQ_LOAD_FILE, GQ_PV_PEN, GQ_BAT_HOURS, GQ_PV_FILE, GQ_WEAK, GQ_TARIFF, GQ_OUTAGE_DUR, GQ_CO2_FILE,
    GQ_SU, GQ_WD, GQ_SOC0_NORMAL = 16, GQ_OUTAGE_NORMAL = 32
};
constexpr uint64_t GEN_SEED_SALT = 0x9E3779B97F4A7C15ull;

__device__ __forceinline__ int32_t gen_randint(uint64_t seed, int64_t gi, int32_t q, int32_t lo, int32_t hi)
{
    const int32_t span = hi - lo;
    const int32_t k = (int32_t)floor(synth_uniform(seed, gi, q) * (double)span);
    return lo + (k < span - 1 ? k : span - 1);
}

// sum of 12 uniforms - 6: additions only, left to right
__device__ __forceinline__ double gen_normal(uint64_t seed, int64_t gi, int32_t q0)
{
    double acc = 0.0;
    for (int j = 0; j < 12; j++) acc += synth_uniform(seed, gi, q0 + j);
    return acc - 6.0;
}

// numpy's pairwise sum of col[k * stride] * ratio, k < n
__device__ inline double gen_pairwise_products(const double *__restrict__ col, int64_t stride, int32_t n, double ratio)
{
    if (n < 8) {
        double res = 0.0;
        for (int32_t k = 0; k < n; k++) res += col[(int64_t)k * stride] * ratio;
        return res;
    }
    if (n <= 128) {
        double r[8];
#pragma unroll
        for (int j = 0; j < 8; j++) r[j] = col[(int64_t)j * stride] * ratio;
        int32_t k;
        for (k = 8; k < n - (n % 8); k += 8)
#pragma unroll
            for (int j = 0; j < 8; j++) r[j] += col[(int64_t)(k + j) * stride] * ratio;
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; k < n; k++) res += col[(int64_t)k * stride] * ratio;
        return res;
    }
    int32_t n2 = n / 2;
    n2 -= n2 % 8;
    return gen_pairwise_products(col, stride, n2, ratio) + gen_pairwise_products(col + (int64_t)n2 * stride, stride, n - n2, ratio);
}

static __global__ __launch_bounds__(BLOCK) void generate_columns_kernel(const mgx_gen a)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= a.n_grids) return;
    const int64_t N = a.n_grids;
    const int64_t gi = a.grid_index ? a.grid_index[i] : a.grid_index0 + i;
    const uint64_t seed = a.seed ^ GEN_SEED_SALT;
    // the draws
    const double bin = synth_uniform(seed, gi, GQ_BIN);
    const int32_t size_load = gen_randint(seed, gi, GQ_SIZE_LOAD, 100, 100001);
    const int32_t lf = gen_randint(seed, gi, GQ_LOAD_FILE, 0, a.n_load_profiles);
    const int32_t pv_pen = gen_randint(seed, gi, GQ_PV_PEN, 30, 151);
    const int32_t hours = gen_randint(seed, gi, GQ_BAT_HOURS, 3, 6);
    const int32_t pf = gen_randint(seed, gi, GQ_PV_FILE, 0, a.n_pv_profiles);
    const int32_t weak = gen_randint(seed, gi, GQ_WEAK, 0, 2);
    const int32_t tariff = gen_randint(seed, gi, GQ_TARIFF, 1, 3);
    const int32_t odur = gen_randint(seed, gi, GQ_OUTAGE_DUR, 1, 8);
    const int32_t cf = a.n_co2_profiles > 0 ? gen_randint(seed, gi, GQ_CO2_FILE, 0, a.n_co2_profiles) : 0;
    const int32_t su = a.mixed_timers ? gen_randint(seed, gi, GQ_SU, 0, 4) : 0;
    const int32_t wd = a.mixed_timers ? gen_randint(seed, gi, GQ_WD, 0, 4) : 0;
    const double soc_n = gen_normal(seed, gi, GQ_SOC0_NORMAL), out_n = gen_normal(seed, gi, GQ_OUTAGE_NORMAL);
    // the rules (generator.derive has the same lines in numpy)
    const double load_ratio = (double)size_load / a.load_max[lf];                 // _scale_ts 'max' (:137-147)
    const double load_peak = a.load_max[lf] * load_ratio;                         // max of the scaled series
    const double pv_size = load_peak * ((double)pv_pen / 100);                    // _size_mg (:357)
    const double pv_ratio = pv_size / a.pv_max[pf];
    const double mean_load = gen_pairwise_products(a.base_load + lf, a.n_load_profiles, a.n_mean_rows, load_ratio) / (double)a.n_mean_rows;
    const double cap = ceil((double)hours * mean_load);                          // _size_battery (:382-386)
    const double power = ceil(cap / 4);                                          // _get_battery (:230-243), duration 4
    const double soc0 = fmin(fmax(soc_n, 0.2), 1.0);                             // min(max(randn, soc_min), soc_max)
    const double rated = ceil(load_peak / 0.9);                                  // _size_genset (:372-379)
    const double grid_power = floor(load_peak * 2);                              // int(max(load.values) * 2) (:364)
    // architecture (:417-435, :535-538)
    bool genset = bin < 0.33 || bin >= 0.66;
    const bool grid = bin >= 0.33;
    genset = genset || (grid && weak != 0);
    if (a.arch) a.arch[i] = (uint8_t)(genset && grid ? 2 : (grid ? 1 : 0));
    if (a.load_profile) a.load_profile[i] = (uint8_t)lf;
    if (a.pv_profile) a.pv_profile[i] = (uint8_t)pf;
    if (a.co2_profile) a.co2_profile[i] = (uint8_t)cf;
    if (a.tariff) a.tariff[i] = (uint8_t)tariff;
    if (a.weak) a.weak[i] = weak;
    if (a.outage_duration) a.outage_duration[i] = odur;
    if (a.outage_per_day) a.outage_per_day[i] = out_n * 3 / 4 + 0.25;            // _get_grid (:291)
    if (a.load_ratio) a.load_ratio[i] = load_ratio;
    if (a.pv_ratio) a.pv_ratio[i] = pv_ratio;
    if (a.load_lo) a.load_lo[i] = -(a.load_bound_max[lf] * load_ratio);          // stored sign: load <= 0 (bounds :81-88)
    if (a.load_hi) a.load_hi[i] = 0.0;
    if (a.pv_lo) a.pv_lo[i] = 0.0;
    if (a.pv_hi) a.pv_hi[i] = a.pv_bound_max[pf] * pv_ratio;
    // (the status component: 1 = never out; the caller lowers it where the outage words say otherwise)
    if (a.grid_lo) { a.grid_lo[i] = a.tariff_min[tariff]; a.grid_lo[N + i] = 0.0; a.grid_lo[2 * N + i] = a.co2_min ? a.co2_min[cf] : 0.0; a.grid_lo[3 * N + i] = 1.0; }
    if (a.grid_hi) { a.grid_hi[i] = a.tariff_max[tariff]; a.grid_hi[N + i] = 0.0; a.grid_hi[2 * N + i] = a.co2_max ? a.co2_max[cf] : 0.0; a.grid_hi[3 * N + i] = 1.0; }
    if (a.bat_max_capacity) a.bat_max_capacity[i] = cap;
    if (a.bat_min_capacity) a.bat_min_capacity[i] = cap * 0.2;                   // get_battery_module: capacity * soc_min
    if (a.bat_max_charge) a.bat_max_charge[i] = power;
    if (a.bat_max_discharge) a.bat_max_discharge[i] = power;
    if (a.soc) a.soc[i] = soc0;
    if (a.charge) a.charge[i] = soc0 * cap;                                      // battery_module.py:96-106
    if (a.gen_running_min) a.gen_running_min[i] = 0.05 * rated;                  // get_genset_module: p_min * rated_power
    if (a.gen_running_max) a.gen_running_max[i] = 0.9 * rated;
    if (a.gen_times) a.gen_times[i] = (uint32_t)su | ((uint32_t)wd << 16);
    if (a.gen_status) a.gen_status[i] = 1u | (1u << 8) | ((uint32_t)wd << 24);   // initially on: steps_until_down = wind_down_time
    if (a.grid_max_import) a.grid_max_import[i] = grid_power;
    if (a.grid_max_export) a.grid_max_export[i] = grid_power;
    if (a.d_bin_rand) a.d_bin_rand[i] = bin;
    if (a.d_soc0_normal) a.d_soc0_normal[i] = soc_n;
    if (a.d_outage_normal) a.d_outage_normal[i] = out_n;
    if (a.d_size_load) a.d_size_load[i] = size_load;
    if (a.d_pv_pen) a.d_pv_pen[i] = pv_pen;
    if (a.d_bat_hours) a.d_bat_hours[i] = hours;
    if (a.d_su) a.d_su[i] = su;
    if (a.d_wd) a.d_wd[i] = wd;
}

// ------------------------------------------------------------------------------------------------------
// The zero-copy observation contract (mgx_normalise_series): the series normalised ONCE, grid-major, so that the window
// columns of grid i at step t are the contiguous slice n[i, t : t + 1 + H] -- a strided view instead of D values rewritten
// per step.  A [T, N] -> [N, R] transpose: a workgroup owns 64 grids x TR rows; wave w reads rows w, w + 4, ... (lane =
// grid: one 512-byte segment), the normalised values cross an LDS tile (pitch 65: conflict-free both ways), and the
// workgroup writes TR * NC consecutive values per grid (lane = position along the row).  Rows >= T hold the forecaster's
// padding value.  Values outside [lo, hi] (where the reference's forecast clip would bite) are counted in `clipped`.
// ------------------------------------------------------------------------------------------------------
template <int NC, typename OT>
__global__ __launch_bounds__(256) void normalise_series_kernel(const KArgs a, int which /* 0 load, 1 pv, 2 grid */,
                                                               OT *__restrict__ out, int32_t R, int32_t TR,
                                                               int32_t *__restrict__ clipped)
{
    extern __shared__ double ntile[];                   // [NC][TR][65]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t N = a.N;
    const int64_t g0 = (int64_t)blockIdx.x * 64;
    const int32_t r0 = (int32_t)blockIdx.y * TR;
    const int64_t i = g0 + lane, ic = i < N ? i : N - 1;
    const double *lo_col = which == 0 ? a.c.load_lo : (which == 1 ? a.c.pv_lo : a.c.grid_lo);
    const double *hi_col = which == 0 ? a.c.load_hi : (which == 1 ? a.c.pv_hi : a.c.grid_hi);
    double lo[NC], hi[NC], sp[NC], zf[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        lo[c] = lo_col[c * N + ic]; hi[c] = hi_col[c * N + ic];
        sp[c] = space_spread(lo[c], hi[c]);
        zf[c] = ((hi[c] + lo[c]) / 2 - lo[c]) / sp[c];                     // a row beyond the series (forecaster.py:95,120-137)
    }
    int32_t n_clip = 0;
    const int comp0 = which == 2 ? 2 : which;
    for (int32_t rr = wave; rr < TR; rr += 4) {
        const int32_t row = r0 + rr;
        if (row >= R) break;
        const bool in = row < a.T;
#pragma unroll
        for (int c = 0; c < NC; c++) {
            double z = zf[c];
            if (in) {
                const double x = series_component(a.c, N, comp0 + c, (int64_t)(row & a.row_mask), ic);
                n_clip += (x < lo[c] || x > hi[c]) ? 1 : 0;
                z = (x - lo[c]) / sp[c];
            }
            ntile[(c * TR + rr) * 65 + lane] = z;
        }
    }
    if (clipped && n_clip && i < N) atomicAdd(clipped, n_clip);
    __syncthreads();
    const int32_t rows = (R - r0 < TR) ? R - r0 : TR;
    const int32_t per_grid = rows * NC;                  // consecutive output values of one grid in this tile
    for (int32_t g = wave; g < 64 && g0 + g < N; g += 4) {
        OT *dst = out + ((g0 + g) * (int64_t)R + r0) * NC;
        for (int32_t e = lane; e < per_grid; e += 64) {
            const int32_t rr = e / NC, c = e - rr * NC;
            dst[e] = (OT)ntile[(c * TR + rr) * 65 + g];
        }
    }
}

}  // namespace mgx
