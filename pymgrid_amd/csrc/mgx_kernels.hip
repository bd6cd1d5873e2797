// mgx_kernels.hip -- HIP kernels of the batched microgrid-step engine, written for gfx950 (MI355X, CDNA4).
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
#include "mgx_core.hpp"

#include <type_traits>

namespace mgx {

#ifndef MGX_BLOCK
#define MGX_BLOCK 256
#endif
constexpr int BLOCK = MGX_BLOCK;
#ifndef MGX_RING
#define MGX_RING 4          // register-ring depth of the fused kernel (steps of loads in flight)
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
// body of one step of grid i (shared by step_kernel and fleet_step_kernel)
template <int F>
__device__ __forceinline__ void step_body(const KArgs &a, const void *__restrict__ actions, int32_t t, int normalized,
                                          double *__restrict__ reward, uint8_t *__restrict__ done, void *__restrict__ obs,
                                          double *__restrict__ log, int64_t i)
{
    // all loads first (independent, one latency round), then the arithmetic
    Params p; State s; Inputs in; Outputs o; Derived d;
    if (a.act_f32) load_inputs<F>(a.c, (const float *)actions, a.N, i, t, in);
    else load_inputs<F>(a.c, (const double *)actions, a.N, i, t, in);
    load_state<F>(a.c, i, log != nullptr, s);
    load_params<F>(a.c, i, p);
    derive<F>(p, d);
    const bool gen_instant = genset_wave_is_instant<F>(p, s);

    step_core<F>(p, d, s, in, normalized != 0, true, gen_instant, o);

    store_state<F>(a.c, i, s);
    reward[i] = shaped_reward<F>(a.shaper, o);
    // _done(): t >= final_step - 1, evaluated before the counter moves (base_timeseries_module.py:124-125)
    if (done) done[i] = done_at(a, i, t);
    if (log) store_log<F>(log + i, a.N, o, s.status);
    // post-step observation (base.py:205-209): without a forecaster the whole 8..12-value row is stored here; with
    // one (H > 0) the host launches obs_rows_wave_kernel behind this kernel and passes obs == nullptr
    if (obs) {
        if (a.obs_state_only) {                 // the window columns of this row were prefetched (obs_windows_k_kernel)
            if (a.obs_f32) observe_state_cols<F>(a, p, s, (float *)obs + i * a.obs_dim);
            else observe_state_cols<F>(a, p, s, (double *)obs + i * a.obs_dim);
        } else if (a.obs_f32) observe_row_h0<F>(a, i, t + 1, p, s, (float *)obs + i * a.obs_dim);
        else observe_row_h0<F>(a, i, t + 1, p, s, (double *)obs + i * a.obs_dim);
    }
}

template <int F>
__global__ __launch_bounds__(BLOCK) void step_kernel(const KArgs a, const void *__restrict__ actions, int32_t t,
                                                     int normalized, double *__restrict__ reward,
                                                     uint8_t *__restrict__ done, void *__restrict__ obs,
                                                     double *__restrict__ log)
{
    t = resolve_t(a, t);
    const int64_t i = (int64_t)a.g0 + (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < a.g1) step_body<F>(a, actions, t, normalized, reward, done, obs, log, i);
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
    if (a.act_f32) load_inputs<F>(a.c, (const float *)actions, a.N, i, t, in);
    else load_inputs<F>(a.c, (const double *)actions, a.N, i, t, in);
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

// RICH = false: the launch writes neither log rows nor the status trace -- compiled out, together with every value only
// they consume (balance sums, co2, the violations mask): the lean form is the hot one (reward / done / SoC streams).
template <int F, int U, typename AT, bool RICH>
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
    if ((int32_t)threadIdx.x >= gpb || i >= a.g1) return;
    const int64_t N = a.N;
    Params p; State s; Derived d;
    load_state<F>(a.c, i, out.log != nullptr, s);
    load_params<F>(a.c, i, p);
    derive<F>(p, d);
    // series bases moved to row t0 once (scalar), so row k of this launch is base + k*N
    const double *__restrict__ lts = a.c.load_ts + (int64_t)t0 * N;
    const double *__restrict__ pts = a.c.pv_ts + (int64_t)t0 * N;
    const double *__restrict__ gts = (F & F_GRID) ? a.c.grid_ts + (int64_t)t0 * 4 * N : nullptr;
    const bool norm = normalized != 0;
    const bool want_soc = (out.soc_trace != nullptr) || (out.log != nullptr);
    const bool gen_instant = genset_wave_is_instant<F>(p, s);
    const int32_t k_done = (a.grid_final ? a.grid_final[i] : a.final_step) - 1 - t0;      // done <=> k >= k_done
    double ret = 0.0;

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
                Outputs o;
                step_core<F>(p, d, s, in, norm, want_soc, gen_instant, o);
                const double r = shaped_reward<F>(a.shaper, o);
                if (out.reward) out.reward[off] = r;
                if (out.done) out.done[off] = (uint8_t)(k >= k_done);
                if constexpr (F & F_BATTERY) { if (out.soc_trace) out.soc_trace[off] = s.soc; }
                if constexpr (F & F_GENSET) { if (out.status_trace) out.status_trace[off] = s.status; }
                if (out.log) store_log<F>(out.log + (off - i) * a.log_dim + i, N, o, s.status);
                ret += r;
                off += N;
            }
        }
    }
    if constexpr (F & F_BATTERY) { if (!want_soc) s.soc = s.charge / p.bat_cmax; }
    store_state<F>(a.c, i, s);
    if (out.ret_acc) out.ret_acc[i] += ret;
    advance_counter_in_kernel(a, K_launch);
}

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
    if (a.obs_state_only) {
        if (a.obs_f32) observe_state_cols<F>(a, p, s, (float *)obs + i * a.obs_dim);
        else observe_state_cols<F>(a, p, s, (double *)obs + i * a.obs_dim);
    } else if (a.obs_f32) observe_row_h0<F>(a, i, t, p, s, (float *)obs + i * a.obs_dim);   // H == 0 only (host dispatches)
    else observe_row_h0<F>(a, i, t, p, s, (double *)obs + i * a.obs_dim);
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
    // wave-uniform: every slot's row exists, lane offsets fit 32-bit byte offsets
    const bool fast = t >= 0 && (int64_t)t + W_pad <= a.T && (int64_t)(4 * Q + 4) * N < (int64_t(1) << 28);
    if (fast) {
        WinBounds<1> bl, bp;
        WinBounds<(F & F_GRID) ? 4 : 1> bg;
        window_bounds<1>(a.c.load_lo, a.c.load_hi, N, ic, bl);
        window_bounds<1>(a.c.pv_lo, a.c.pv_hi, N, ic, bp);
        if constexpr (F & F_GRID) window_bounds<4>(a.c.grid_lo, a.c.grid_hi, N, ic, bg);
        for (int32_t hb = 0; hb < W; hb += slots) {      // one round for the usual 24 / 25-step windows
            double vl[OBS_JB][1], vp[OBS_JB][1], vg[OBS_JB][(F & F_GRID) ? 4 : 1];
            window_issue<1>(a.c.load_ts, N, N, t, hb, Q, (uint32_t)(q * N + ic), vl);     // all loads of the round in flight
            window_issue<1>(a.c.pv_ts, N, N, t, hb, Q, (uint32_t)(q * N + ic), vp);
            if constexpr (F & F_GRID) window_issue<4>(a.c.grid_ts, N, 4 * N, t, hb, Q, (uint32_t)(q * 4 * N + ic), vg);
            if (hb == 0) window_bounds_finish<1>(bl);
            window_finish<1, NOISE, OT>(vl, bl, W, t, hb, i, ic, q, Q, row, a.c.load_noise_std, 0u, a.noise_seed, a.noise_increase);
            if (hb == 0) window_bounds_finish<1>(bp);
            window_finish<1, NOISE, OT>(vp, bp, W, t, hb, i, ic, q, Q, row + W, a.c.pv_noise_std, 1u, a.noise_seed, a.noise_increase);
            if constexpr (F & F_GRID) {
                if (hb == 0) window_bounds_finish<4>(bg);
                window_finish<4, NOISE, OT>(vg, bg, W, t, hb, i, ic, q, Q, row + plan.grid_col_base, a.c.grid_noise_std, 2u,
                                        a.noise_seed, a.noise_increase);
            }
        }
    } else {
        observe_window_cols<1, NOISE, OT>(a.c.load_ts, N, N, a.c.load_lo, a.c.load_hi, a.T, t, W, i, ic, q, Q, row,
                                      a.c.load_noise_std, 0u, a.noise_seed, a.noise_increase);
        observe_window_cols<1, NOISE, OT>(a.c.pv_ts, N, N, a.c.pv_lo, a.c.pv_hi, a.T, t, W, i, ic, q, Q, row + W,
                                      a.c.pv_noise_std, 1u, a.noise_seed, a.noise_increase);
        if constexpr (F & F_GRID)
            observe_window_cols<4, NOISE, OT>(a.c.grid_ts, N, 4 * N, a.c.grid_lo, a.c.grid_hi, a.T, t, W, i, ic, q, Q,
                                          row + plan.grid_col_base, a.c.grid_noise_std, 2u, a.noise_seed, a.noise_increase);
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
constexpr int OBS_KJ = 2;                               // rows per thread and latency round (x 16 phases = 32 rows)
constexpr int OBS_K_THREADS = 256;

template <int NC>
__device__ __forceinline__ void windows_k_module(const double *__restrict__ ts, int64_t N, int64_t row_stride,
                                                 const double *__restrict__ lo_col, const double *__restrict__ hi_col,
                                                 int32_t T, int32_t t, int32_t R, int32_t K, int64_t ic, int32_t q, int32_t Q,
                                                 double *nc /* [NC][RP] of this grid */, double *nu /* [NC][K] */, int32_t RP)
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
            const int32_t rc = r < T ? (r < 0 ? 0 : r) : T - 1;
#pragma unroll
            for (int c = 0; c < NC; c++) v[jj][c] = ts[(int64_t)rc * row_stride + c * N + ic];
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
                    nc[c * RP + rr] = n_c;
                    if (rr < K) nu[c * K + rr] = n_u;
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
};

#ifdef MGX_WIN_PLAIN_STORES
#define MGX_WIN_STORE(v, p) (*(p) = (v))
#else
#define MGX_WIN_STORE(v, p) __builtin_nontemporal_store((v), (p))
#endif
// Body shared by obs_windows_k_kernel and the window part of fleet_step_kernel: workgroup `group` (16 grids) of the batch.
// GRID: the layout has a GridModule (6 instead of 2 series components).  `now` (meaningful in the q == 0 lanes): the state
// columns of the current state for block 0, or nullptr (a prefetch ahead of the counter: every state column is zero).
template <bool GRID, typename OT>
__device__ __forceinline__ void windows_body(const KArgs &a, const WindowsKPlan &plan, int32_t t, OT *__restrict__ ring,
                                             int64_t group, int32_t nstate, const double *now, double *image)
{
    constexpr int NCOMP = 2 + (GRID ? 4 : 0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int32_t G = plan.group, Q = OBS_K_THREADS / G, K = plan.K, RP = plan.rp, BP = plan.bp;
    const int32_t g = tid & (G - 1), q = tid / G;
    const int64_t g0 = group * G;
    const int64_t N = a.N;
    const int32_t W = 1 + a.H, D = a.obs_dim, R = K + a.H;
    const int64_t i = g0 + g, ic = i < N ? i : g0;
    const int32_t NU0 = NCOMP * RP, S0 = NU0 + NCOMP * K;
    double *blk = image + g * BP;
    uint32_t *map = reinterpret_cast<uint32_t *>(image + G * BP);       // [D]

    windows_k_module<1>(a.c.load_ts, N, N, a.c.load_lo, a.c.load_hi, a.T, t, R, K, ic, q, Q, blk, blk + NU0, RP);
    windows_k_module<1>(a.c.pv_ts, N, N, a.c.pv_lo, a.c.pv_hi, a.T, t, R, K, ic, q, Q, blk + RP, blk + NU0 + K, RP);
    if constexpr (GRID)
        windows_k_module<4>(a.c.grid_ts, N, 4 * N, a.c.grid_lo, a.c.grid_hi, a.T, t, R, K, ic, q, Q, blk + 2 * RP,
                            blk + NU0 + 2 * K, RP);
    if (q == 0) {                                        // state columns: the current state for block 0, zeros ahead
        for (int j = 0; j < nstate; j++) {
            blk[S0 + j * K] = now ? now[j] : 0.0;
            for (int32_t k = 1; k < K; k++) blk[S0 + j * K + k] = 0.0;
        }
    }
    for (int32_t col = tid; col < D; col += OBS_K_THREADS) {            // column -> offset of its k = 0 entry in a block
        uint32_t comp, h;
        if (col < W) { comp = 0; h = col; }
        else if (col < 2 * W) { comp = 1; h = col - W; }
        else if (col < plan.grid_col_base) { comp = 0xffffu; h = col - 2 * W; }
        else { comp = 2u + ((col - plan.grid_col_base) & 3); h = (col - plan.grid_col_base) >> 2; }
        map[col] = comp == 0xffffu ? S0 + h * K : (h == 0 ? NU0 + comp * K : comp * RP + h);
    }
    __syncthreads();
    const int32_t n_valid = (N - g0 < G) ? (int32_t)(N - g0) : G;
    const int32_t total = n_valid * D;                   // D is even (one load, one renewable module)
    typedef OT vec2 __attribute__((ext_vector_type(2)));
    const bool wide = (reinterpret_cast<uintptr_t>(ring) & (sizeof(vec2) - 1)) == 0;
    for (int32_t k = wave; k < K; k += OBS_K_THREADS / 64) {
        OT *out = ring + ((int64_t)k * N + g0) * D;
        const double *src = image + k;
        if (wide) {                                      // element pair f, f + 1 = 2 lane + 128 j -> (row, column)
            int32_t r = 2 * lane / D, c = 2 * lane - r * D;
            for (int32_t f = 2 * lane; f < total; f += 128) {
                vec2 v2;
                v2.x = (OT)src[r * BP + map[c]];
                v2.y = (OT)src[r * BP + map[c + 1]];     // D even, c even: the pair never straddles two rows
                MGX_WIN_STORE(v2, reinterpret_cast<vec2 *>(out + f));
                c += 128;
                while (c >= D) { c -= D; r++; }
            }
        } else {
            int32_t r = lane / D, c = lane - r * D;
            for (int32_t f = lane; f < total; f += 64) {
                MGX_WIN_STORE((OT)src[r * BP + map[c]], out + f);
                c += 64;
                while (c >= D) { c -= D; r++; }
            }
        }
    }
}

template <int F, typename OT>
__global__ __launch_bounds__(OBS_K_THREADS) void obs_windows_k_kernel(const KArgs a, const WindowsKPlan plan, int32_t t,
                                                                      OT *__restrict__ ring)
{
    t = resolve_t_obs(a, t);
    constexpr int NSTATE = 4 * ((F & F_GENSET) != 0) + 2 * ((F & F_BATTERY) != 0);
    extern __shared__ double image[];
    const int64_t group = (int64_t)plan.group0 + blockIdx.x;
    double now[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (plan.with_state && (int)threadIdx.x < plan.group) {           // the q == 0 lanes: one per grid of the group
        const int64_t i = group * plan.group + threadIdx.x, ic = i < a.N ? i : group * plan.group;
        Params p; State s;
        load_state<F>(a.c, ic, true, s);
        load_params<F>(a.c, ic, p);
        observe_state_cols<F>(a, p, s, now, 0);
    }
    windows_body<(F & F_GRID) != 0, OT>(a, plan, t, ring, group, NSTATE, plan.with_state ? now : nullptr, image);
}

// ------------------------------------------------------------------------------------------------------
// Discrete action expansion: PriorityListAlgo._populate_action (priority_list.py:69-167).
// ------------------------------------------------------------------------------------------------------
template <int F>
__global__ __launch_bounds__(BLOCK) void expand_kernel(const KArgs a, const PLWords tab, const int32_t *__restrict__ action_id,
                                                       int32_t t, double *__restrict__ control)
{
    t = resolve_t(a, t);
    constexpr int A = 2 * ((F & F_GENSET) != 0) + ((F & F_BATTERY) != 0) + ((F & F_GRID) != 0);
    const int64_t i = (int64_t)a.g0 + (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= a.g1) return;
    const int64_t N = a.N;
    Params p; State s; Inputs in;
    load_state<F>(a.c, i, false, s);
    load_params<F>(a.c, i, p);
    in.load = a.c.load_ts[(int64_t)t * N + i];
    in.pv = a.c.pv_ts[(int64_t)t * N + i];
    in.g_stat = 1.0;
    if constexpr (F & F_GRID) in.g_stat = a.c.grid_ts[((int64_t)t * 4 + 3) * N + i];
    double q_unused;
    populate_core<F>(p, s, pl_select(tab, action_id[i]), in, q_unused, 0.0 + -1 * in.load, in.pv);
    double *c = control + i * A;
    int k = 0;
    if constexpr (F & F_GENSET) { c[k] = in.a_goal; c[k + 1] = in.a_gen; k += 2; }
    if constexpr (F & F_BATTERY) { c[k++] = in.a_bat; }
    if constexpr (F & F_GRID) { c[k++] = in.a_grid; }
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
template <int F>
__device__ __forceinline__ void step_discrete_body(const KArgs &a, const PLWords &tab, const int32_t *__restrict__ action_id,
                                                   int32_t t, double *__restrict__ control, double *__restrict__ reward,
                                                   uint8_t *__restrict__ done, void *__restrict__ obs,
                                                   double *__restrict__ log, int64_t i)
{
    constexpr int A = 2 * ((F & F_GENSET) != 0) + ((F & F_BATTERY) != 0) + ((F & F_GRID) != 0);
    const int64_t N = a.N;
    Params p; State s; Inputs in; Outputs o; Derived d;
    const int32_t id = action_id[i];
    load_series_at<F>(a.c.load_ts + (int64_t)t * N, a.c.pv_ts + (int64_t)t * N,
                      (F & F_GRID) ? a.c.grid_ts + (int64_t)t * 4 * N : nullptr, N, i, i, in);
    load_state<F>(a.c, i, log != nullptr, s);
    load_params<F>(a.c, i, p);
    derive<F>(p, d);
    const bool gen_instant = genset_wave_is_instant<F>(p, s);
    double bat_q;
    populate_core<F>(p, s, pl_select(tab, id), in, bat_q, 0.0 + -1 * in.load, in.pv);
    if (control) {                                   // optional copy of the expanded control (_get_action's value)
        double *c = control + i * A;
        int k = 0;
        if constexpr (F & F_GENSET) { c[k] = in.a_goal; c[k + 1] = in.a_gen; k += 2; }
        if constexpr (F & F_BATTERY) { c[k++] = in.a_bat; }
        if constexpr (F & F_GRID) { c[k++] = in.a_grid; }
    }
    step_core<F, true>(p, d, s, in, false, true, gen_instant, o, bat_q);
    store_state<F>(a.c, i, s);
    reward[i] = shaped_reward<F>(a.shaper, o);
    if (done) done[i] = done_at(a, i, t);
    if (log) store_log<F>(log + i, N, o, s.status);
    if (obs) {
        if (a.obs_state_only) {                 // the window columns of this row were prefetched (obs_windows_k_kernel)
            if (a.obs_f32) observe_state_cols<F>(a, p, s, (float *)obs + i * a.obs_dim);
            else observe_state_cols<F>(a, p, s, (double *)obs + i * a.obs_dim);
        } else if (a.obs_f32) observe_row_h0<F>(a, i, t + 1, p, s, (float *)obs + i * a.obs_dim);
        else observe_row_h0<F>(a, i, t + 1, p, s, (double *)obs + i * a.obs_dim);
    }
}

template <int F>
__global__ __launch_bounds__(BLOCK) void step_discrete_kernel(const KArgs a, const PLWords tab,
                                                              const int32_t *__restrict__ action_id, int32_t t,
                                                              double *__restrict__ control, double *__restrict__ reward,
                                                              uint8_t *__restrict__ done, void *__restrict__ obs,
                                                              double *__restrict__ log)
{
    t = resolve_t(a, t);
    const int64_t i = (int64_t)a.g0 + (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < a.g1) step_discrete_body<F>(a, tab, action_id, t, control, reward, done, obs, log, i);
    advance_counter_in_kernel(a, 1);
}

// ------------------------------------------------------------------------------------------------------
// A heterogeneous fleet in ONE launch (mgx_fleet_step): up to MGX_FLEET_MAX batches of different layouts, each with its own
// columns / actions / outputs / step counter, laid end to end over the workgroups.  The table travels in the kernarg
// segment (scalar loads); a workgroup finds its batch with a few scalar compares and jumps -- wave-uniformly -- to that
// layout's specialisation of the step.  Three 33 000-grid batches cost one ~6 us launch instead of three ~5 us ones.
// ------------------------------------------------------------------------------------------------------
constexpr int MGX_FLEET_MAX = 6;
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

__global__ __launch_bounds__(BLOCK) void fleet_step_kernel(const FleetArgs fa, const FleetWin fw)
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
            plan.group0 = mine ? fw.plan[q].group0 : plan.group0;
            t = mine ? fw.t[q] : t; block0 = mine ? fw.block0[q] : block0; kind = mine ? fw.kind[q] : kind;
            nstate = mine ? fw.nstate[q] : nstate;
        }
        const KArgs &a = *kp;
        const int64_t group = (int64_t)plan.group0 + (b - block0);
        switch (kind) {
            case 0: windows_body<false, double>(a, plan, t, (double *)ring, group, nstate, nullptr, image); break;
            case 1: windows_body<true, double>(a, plan, t, (double *)ring, group, nstate, nullptr, image); break;
            case 2: windows_body<false, float>(a, plan, t, (float *)ring, group, nstate, nullptr, image); break;
            default: windows_body<true, float>(a, plan, t, (float *)ring, group, nstate, nullptr, image); break;
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
    const int64_t i = (int64_t)((int)blockIdx.x - block0) * BLOCK + threadIdx.x;
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

// ------------------------------------------------------------------------------------------------------
// Discrete rollout: K fused steps whose control is expanded ON DEVICE from a priority-list id -- per step
// (ids [K, N], one byte each: a DiscreteMicrogridEnv roll-out) or constant per grid (ids [N]: RuleBasedControl.run,
// algos/rbc/rbc.py:64-93).  No action stream at all: per step only the series rows are read.
// ------------------------------------------------------------------------------------------------------
template <int F, int U, bool PER_STEP, bool RICH>
__global__ __launch_bounds__(BLOCK_K) void rollout_kernel(const KArgs a, const PLWords tab, const uint8_t *__restrict__ ids,
                                                          int32_t t0, int32_t K, const FusedOut out_rt, int32_t gpb)
{
    FusedOut out = out_rt;
    if constexpr (!RICH) { out.log = nullptr; out.status_trace = nullptr; }
    const int32_t K_launch = K;          // what the host asked for (the counter always moves by this much)
    t0 = resolve_t(a, t0);
    K = resolve_k(a, t0, K);
    const int64_t i = (int64_t)a.g0 + (int64_t)blockIdx.x * gpb + threadIdx.x;
    if ((int32_t)threadIdx.x >= gpb || i >= a.g1) return;
    const int64_t N = a.N;
    Params p; State s; Derived d;
    load_state<F>(a.c, i, out.log != nullptr, s);
    load_params<F>(a.c, i, p);
    derive<F>(p, d);
    const double *__restrict__ lts = a.c.load_ts + (int64_t)t0 * N;
    const double *__restrict__ pts = a.c.pv_ts + (int64_t)t0 * N;
    const double *__restrict__ gts = (F & F_GRID) ? a.c.grid_ts + (int64_t)t0 * 4 * N : nullptr;
    const bool want_soc = (out.soc_trace != nullptr) || (out.log != nullptr);
    const bool gen_instant = genset_wave_is_instant<F>(p, s);
    const int32_t k_done = (a.grid_final ? a.grid_final[i] : a.final_step) - 1 - t0;
    // PER_STEP = false (one fixed list per grid: RuleBasedControl): `word` is loop-invariant and the list decoding of
    // populate_core is hoisted out of the step loop by the compiler
    uint32_t word = PER_STEP ? 0u : pl_select(tab, ids[i]);
    double ret = 0.0;

    // The step loop exists twice, specialised at COMPILE time on the wave-uniform `gen_instant`: in the instant form the
    // genset's status is its goal, its limits under a fixed list are loop-invariant (hoisted), and the FSM is gone.
    auto run = [&](auto gi_tag) __attribute__((always_inline)) {
        constexpr bool GI = decltype(gi_tag)::value;
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
                    double bat_q;
                    populate_core<F>(p, s, word, in, bat_q, 0.0 + -1 * in.load, in.pv, GI);
                    Outputs o;
                    step_core<F, true>(p, d, s, in, false, want_soc, GI, o, bat_q);
                    const double r = shaped_reward<F>(a.shaper, o);
                    if (out.reward) out.reward[off] = r;
                    if (out.done) out.done[off] = (uint8_t)(k >= k_done);
                    if constexpr (F & F_BATTERY) { if (out.soc_trace) out.soc_trace[off] = s.soc; }
                    if constexpr (F & F_GENSET) { if (out.status_trace) out.status_trace[off] = s.status; }
                    if (out.log) store_log<F>(out.log + (off - i) * a.log_dim + i, N, o, s.status);
                    ret += r;
                    off += N;
                }
            }
        }
    };
    if constexpr ((F & F_GENSET) != 0) {
        if (gen_instant) run(std::true_type{}); else run(std::false_type{});
    } else {
        run(std::false_type{});
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

template <typename OT>
__device__ inline void observe_series_multi(const double *__restrict__ ts, int64_t row_stride, int32_t T, int32_t t, int32_t H,
                                            double lo, double hi, OT *__restrict__ obs, int obs_stride = 1)
{
    const double fill = (hi + lo) / 2, sp = space_spread(lo, hi);
    for (int h = 0; h <= H; h++) {
        const bool in = t < T && t + h < T;
        const double v = in ? ts[(int64_t)(t + h) * row_stride] : 0.0;
        obs[h * obs_stride] = (OT)obs_series_value(v, in, h > 0, lo, hi, fill, sp);
    }
}

// flat order: load windows, pv windows, gensets (4 columns each), batteries (2 each), grid windows (4 (1 + H) each)
template <int F, typename OT>
__device__ inline void observe_row_multi(const KArgs &a, int64_t i, int32_t t, OT *__restrict__ obs_row)
{
    const int64_t N = a.N;
    const int W = 1 + a.H;
    int k = 0;
    for (int j = 0; j < a.n_load; j++, k += W)
        observe_series_multi(a.c.load_ts + (int64_t)j * N + i, (int64_t)a.n_load * N, a.T, t, a.H, a.c.load_lo[(int64_t)j * N + i],
                             a.c.load_hi[(int64_t)j * N + i], obs_row + k);
    for (int j = 0; j < a.n_pv; j++, k += W)
        observe_series_multi(a.c.pv_ts + (int64_t)j * N + i, (int64_t)a.n_pv * N, a.T, t, a.H, a.c.pv_lo[(int64_t)j * N + i],
                             a.c.pv_hi[(int64_t)j * N + i], obs_row + k);
    if constexpr (F & F_GENSET) {
        for (int j = 0; j < a.n_genset; j++) {
            const int64_t c = (int64_t)j * N + i;
            const uint32_t times = a.c.gen_times[c], st = a.c.gen_status[c];
            const double su = (double)(times & 0xff), wd = (double)((times >> 16) & 0xff);
            obs_row[k++] = (OT)space_norm(0.0, 1.0, (double)(st & 0xff));
            obs_row[k++] = (OT)space_norm(0.0, 1.0, (double)((st >> 8) & 0xff));
            obs_row[k++] = (OT)space_norm(0.0, su, (double)((st >> 16) & 0xff));
            obs_row[k++] = (OT)space_norm(0.0, wd, (double)(st >> 24));
        }
    }
    if constexpr (F & F_BATTERY) {
        for (int j = 0; j < a.n_battery; j++) {
            const int64_t c = (int64_t)j * N + i;
            const double cmin = a.c.bat_min_capacity[c], cmax = a.c.bat_max_capacity[c];
            obs_row[k++] = (OT)space_norm(cmin / cmax, 1.0, a.c.soc[c]);
            obs_row[k++] = (OT)space_norm(cmin, cmax, a.c.charge[c]);
        }
    }
    if constexpr (F & F_GRID) {
        for (int j = 0; j < a.n_grid; j++, k += 4 * W)
            for (int cc = 0; cc < 4; cc++) {
                const int64_t c = ((int64_t)j * 4 + cc) * N + i;
                observe_series_multi(a.c.grid_ts + c, (int64_t)a.n_grid * 4 * N, a.T, t, a.H, a.c.grid_lo[c], a.c.grid_hi[c],
                                     obs_row + k + cc, 4);
            }
    }
}

template <int F>
__global__ __launch_bounds__(BLOCK_MULTI) void step_multi_kernel(const KArgs a, const void *__restrict__ actions, int32_t t,
                                                                 int normalized, double *__restrict__ reward,
                                                                 uint8_t *__restrict__ done, void *__restrict__ obs,
                                                                 double *__restrict__ log)
{
    extern __shared__ double multi_lds[];
    t = resolve_t(a, t);
    const int64_t i = (int64_t)a.g0 + (int64_t)blockIdx.x * BLOCK_MULTI + threadIdx.x;
    if (i < a.g1) {
        const int A = 2 * a.n_genset + a.n_battery + a.n_grid;
        StepLists L = multi_lists(a, multi_lds);
        Outputs o;
        if (a.act_f32) step_multi_core<F>(a, (const float *)actions + i * A, i, t, normalized != 0, L, log ? log + i : nullptr, o);
        else step_multi_core<F>(a, (const double *)actions + i * A, i, t, normalized != 0, L, log ? log + i : nullptr, o);
        reward[i] = shaped_reward<F>(a.shaper, o);
        if (done) done[i] = done_at(a, i, t);
        if (obs) {
            if (a.obs_f32) observe_row_multi<F>(a, i, t + 1, (float *)obs + i * a.obs_dim);
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
    if (a.obs_f32) observe_row_multi<F>(a, i, t, (float *)obs + i * a.obs_dim);
    else observe_row_multi<F>(a, i, t, (double *)obs + i * a.obs_dim);
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
                                                                   int normalized, const FusedOut out)
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
        for (int32_t k = 0; k < K; k++) {
            const int64_t off = (int64_t)k * N + i;
            Outputs o;
            double *log = out.log ? out.log + (int64_t)k * a.log_dim * N + i : nullptr;
            if (lists) {
                double ctrl[MGX_MAX_ACTIONS_MULTI];
                int32_t id = per_step ? ids[off] : ids[i];
                id = (id >= 0 && id < n_lists) ? id : 0;
                populate_multi<F>(a, lists + (int64_t)id * list_len * 3, list_len, i, t0 + k, ctrl);
                step_multi_core<F>(a, (const double *)ctrl, i, t0 + k, false, L, log, o);
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

// mgx_expand_lists / mgx_expand_discrete on the general path: lists [n_lists, list_len, 3] in device memory
template <int F>
__global__ __launch_bounds__(BLOCK_MULTI) void expand_multi_kernel(const KArgs a, const int32_t *__restrict__ lists, int32_t n_lists,
                                                                   int32_t list_len, const int32_t *__restrict__ action_id,
                                                                   int32_t t, double *__restrict__ control)
{
    t = resolve_t(a, t);
    const int64_t i = (int64_t)a.g0 + (int64_t)blockIdx.x * BLOCK_MULTI + threadIdx.x;
    if (i >= a.g1) return;
    const int A = 2 * a.n_genset + a.n_battery + a.n_grid;
    int32_t id = action_id[i];
    id = (id >= 0 && id < n_lists) ? id : 0;              // ids outside [0, n) fall back to list 0 (the reference raises)
    populate_multi<F>(a, lists + (int64_t)id * list_len * 3, list_len, i, t, control + i * A);
}

// one wave that idles for `ticks` of the 100 MHz real-time counter: mgx_fork staggers the shard streams with it
__global__ void stagger_kernel(int64_t ticks)
{
    const int64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

// device-resident step counter (hipGraph-replayable stepping): counter[0] = t, counter[1] = overrun flag
__global__ void set_counter_kernel(int32_t *counter, int32_t t) { counter[0] = t; counter[1] = 0; counter[2] = 0; }

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

__global__ __launch_bounds__(BLOCK) void colsum_stage1(const double *__restrict__ values, int64_t N, int32_t per_block,
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

__global__ __launch_bounds__(BLOCK) void colsum_stage2(const double *__restrict__ partial, int32_t n_partial,
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
struct GatherArgs {
    const double *load_ts, *pv_ts, *grid_ts;
    const double *load_lo, *load_hi, *pv_lo, *pv_hi, *grid_lo, *grid_hi;
    double *load_w, *pv_w, *grid_w;
    const int32_t *start, *length;
    int32_t *final_rel;
    int32_t N, T, rows, max_length, lo, hi;
};

__global__ __launch_bounds__(BLOCK) void gather_windows_kernel(const GatherArgs g)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= g.N) return;
    const int64_t N = g.N;
    int32_t s = g.start[i];
    s = s < g.lo ? g.lo : (s > g.hi - 1 ? g.hi - 1 : s);            // a start outside the env's window is clamped into it
    int32_t len = g.length ? g.length[i] : g.max_length;
    const int32_t room = g.hi - s;
    len = len < 1 ? 1 : len;
    len = len > g.max_length ? g.max_length : len;
    len = len > room ? room : len;                                     // the episode ends at the env's final step at the latest
    if (g.final_rel) g.final_rel[i] = len;
    const double fl = (g.load_lo && g.load_hi) ? (g.load_hi[i] + g.load_lo[i]) / 2 : 0.0;
    const double fp = (g.pv_lo && g.pv_hi) ? (g.pv_hi[i] + g.pv_lo[i]) / 2 : 0.0;
    double fg[4] = {0.0, 0.0, 0.0, 0.0};
    if (g.grid_ts && g.grid_lo && g.grid_hi) {
#pragma unroll
        for (int c = 0; c < 4; c++) fg[c] = (g.grid_hi[c * N + i] + g.grid_lo[c * N + i]) / 2;
    }
    for (int32_t r = 0; r < g.rows; r++) {
        const int64_t row = (int64_t)s + r;
        const bool in = row < g.T;
        const int64_t src = (in ? row : (int64_t)g.T - 1) * N + i;
        const double vl = g.load_ts[src], vp = g.pv_ts[src];
        g.load_w[(int64_t)r * N + i] = in ? vl : fl;
        g.pv_w[(int64_t)r * N + i] = in ? vp : fp;
        if (g.grid_ts) {
            const int64_t sg = (in ? row : (int64_t)g.T - 1) * 4 * N + i;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const double v = g.grid_ts[sg + c * N];
                g.grid_w[((int64_t)r * 4 + c) * N + i] = in ? v : fg[c];
            }
        }
    }
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
__device__ __forceinline__ double synth_uniform(uint64_t seed, int64_t grid, int32_t row)
{
    uint32_t r[4];
    philox4x32_10((uint32_t)grid, (uint32_t)((uint64_t)grid >> 32), (uint32_t)row, 0x5eedu, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    return (double)(((uint64_t)r[0] << 21) ^ (r[1] >> 11)) * (1.0 / 9007199254740992.0);      // 53 bits, [0, 1)
}

// MicrogridGenerator._get_electricity_tariff (:253-285)
__device__ __forceinline__ double tariff_price(int32_t pattern, int32_t row)
{
    const int32_t h = row % 24;
    if (pattern == 1) return (h >= 12 && h < 18) ? 0.59 : ((h < 8 || h >= 21) ? 0.22 : 0.29);
    if (pattern == 2) return ((h >= 0 && h < 5) || (h >= 14 && h < 17)) ? 0.08 : 0.11;
    return 0.0;
}

__global__ __launch_bounds__(BLOCK) void synthesize_series_kernel(const mgx_synth a)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= a.n_grids) return;
    const int64_t N = a.n_grids;
    const int32_t T = a.n_steps;
    const int32_t lp = a.load_profile[i], pp = a.pv_profile[i];
    const double lr = a.load_ratio[i], pr = a.pv_ratio[i];
    const bool grid = a.grid_ts != nullptr;
    int32_t cp = 0, pat = 0, dur = 1;
    double prob = 0.0;
    if (grid) {
        cp = a.co2_profile[i]; pat = a.tariff[i];
        prob = a.outage_per_day ? a.outage_per_day[i] / 24 : 0.0;           // weak_grid_timeseries[i] < outage_per_day/24 (:332)
        dur = a.outage_duration ? a.outage_duration[i] : 1;
    }
    const bool weak = grid && a.outage_per_day != nullptr && a.weak[i] != 0;
    const int64_t gi = a.grid_index ? a.grid_index[i] : a.grid_index0 + i;      // the Philox counter: GLOBAL grid index
    // rows still covered by an outage that starts later: the extra draw of row T (the reference draws T + 1 values) first
    int32_t cover = 0;
    if (weak && synth_uniform(a.seed, gi, T) < prob) cover = dur - 1;
    for (int32_t t = T - 1; t >= 0; t--) {
        a.load_ts[(int64_t)t * N + i] = -1.0 * fabs(a.base_load[(int64_t)t * a.n_load_profiles + lp] * lr);   // stored sign
        a.pv_ts[(int64_t)t * N + i] = fabs(a.base_pv[(int64_t)t * a.n_pv_profiles + pp] * pr);
        if (grid) {
            double status = 1.0;
            if (weak) {
                const bool own = synth_uniform(a.seed, gi, t) < prob;
                status = (own || (cover > 0 && t > 0)) ? 0.0 : 1.0;             // "if i-j > 0": the back-fill spares row 0
                cover = own ? dur - 1 : (cover > 0 ? cover - 1 : 0);
            }
            double *g = a.grid_ts + (int64_t)t * 4 * N + i;
            g[0] = tariff_price(pat, t);
            g[N] = 0.0;                                                       // price_export = zeros (:264)
            g[2 * N] = a.base_co2[(int64_t)t * a.n_co2_profiles + cp];
            g[3 * N] = status;
        }
    }
}

}  // namespace mgx

// ======================================================================================================
// Host side: the C ABI (include/mgx.h)
// ======================================================================================================
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using namespace mgx;

#define MGX_MAX_SHARDS 8

struct mgx_handle {
    KArgs k;
    mgx_layout layout;
    int32_t window_lo, window_hi;   // episode window given at create: trajectories must stay inside it
    bool multi;             // n_load != 1, n_pv != 1 or several gensets / batteries / grids: general (slow) kernels
    size_t multi_lds;       // LDS bytes of a general-kernel workgroup (the MicrogridStep lists)
    std::vector<std::string> log_names;
    int32_t *d_lists;       // general path: device copy of the priority lists handed to mgx_expand_discrete as a host table
    std::vector<int32_t> lists_uploaded;
    int32_t *d_counter;     // device step counter + overrun flag (used when k.t_dev != NULL)
    int32_t n_cu;           // compute units of the device (workgroup balancing of the fused kernels)
    int32_t flags;          // F
    int32_t t;              // current step
    int32_t action_dim;
    int device;
    double *scratch;        // [64 * MAX_PARTIAL] column-sum partials
    // shards (mgx_set_shards): stepping launches are split into n_shards contiguous grid ranges, one internal stream each
    int32_t n_shards;                        // 1 = off
    int32_t shard_lo[MGX_MAX_SHARDS + 1];    // range j = [shard_lo[j], shard_lo[j + 1])
    hipStream_t shard_stream[MGX_MAX_SHARDS];
    hipEvent_t shard_event[MGX_MAX_SHARDS];
    hipEvent_t fork_event;
    double stagger_us;                       // mgx_fork delays shard j by j / S of this (0: off)
    hipStream_t counter_stream;              // device-counter mode: the stream of the last call that touched the counter
    KArgs *d_kargs;                          // device copy of `k` for fleet_step_kernel, refreshed when `k` changed
    KArgs k_uploaded;
    bool k_uploaded_valid;
    PLWords *d_table;                        // device copy of the priority-list table of a discrete fleet item
    PLWords table_uploaded;
    bool table_uploaded_valid;
    hipStream_t prefetch_stream;             // mgx_observe_windows_ahead: the window prefetch overlaps the steps
    hipEvent_t prefetch_gate, prefetch_done;
    bool prefetch_pending;
    // per-grid episode windows (mgx_reset_windows): the full series are remembered here while the handle steps over the
    // caller's window buffers
    bool windowed;
    const double *full_load_ts, *full_pv_ts, *full_grid_ts;
    int32_t full_T, full_final, full_initial, full_window_lo, full_window_hi;
};

namespace {

constexpr int MAX_PARTIAL = 1024;
constexpr int MAX_METRICS = 64;

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int hip_fail(hipError_t e, const char *what)
{
    return fail(MGX_ERR_DEVICE, "%s: %s", what, hipGetErrorString(e));
}

const char *const kCommonNames[] = {
    "reward", "fixed_provided", "fixed_absorbed", "controllable_provided", "controllable_absorbed",
    "overall_provided", "overall_absorbed", "load_met", "renewable_used", "curtailment", "loss_load",
    "overgeneration", "unbalanced_reward"};
const char *const kGensetNames[] = {"genset_production", "genset_co2_production", "genset_reward", "genset_status"};
const char *const kBatteryNames[] = {"discharge_amount", "charge_amount", "battery_reward", "soc_pre", "charge_pre"};
const char *const kGridNames[] = {"grid_import", "grid_export", "grid_co2_production", "grid_reward"};

inline unsigned blocks_for(int64_t n) { return (unsigned)((n + BLOCK - 1) / BLOCK); }

// kernel argument for "the current step": the host's counter, or 0 (= offset to the device counter)
inline int32_t t_arg(const mgx_handle *h) { return h->k.t_dev ? 0 : h->t; }
inline bool dev_counter(const mgx_handle *h) { return h->k.t_dev != nullptr; }
// rows a stepping call may consume: the series -- or, during a per-grid-window episode, the longest episode (the window
// buffers hold horizon + 1 further rows, but those are forecast rows: stepping into them is stepping past the end)
inline int32_t step_limit(const mgx_handle *h) { return h->windowed ? h->layout.final_step : h->k.T; }
inline void advance(mgx_handle *h, int32_t k, hipStream_t st)
{
    h->counter_stream = st;      // device-counter mode: the stepping kernel itself advanced the counter (on this stream)
    h->t += k;
}

// The handle's device for the duration of a scope: streams, events and buffers the library creates lazily must live on
// the device the batch is on, whatever the caller's current device happens to be.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int device)
    {
        if (hipGetDevice(&prev) == hipSuccess && prev != device) switched = hipSetDevice(device) == hipSuccess;
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};

// One stepping call = one launch per shard.  fn(kargs with [g0, g1) set, stream of that shard).
template <class Fn>
inline void for_each_shard(const mgx_handle *h, hipStream_t user, Fn fn)
{
    KArgs k = h->k;
    if (h->n_shards <= 1) { k.g0 = 0; k.g1 = k.N; fn(k, user); return; }
    for (int j = 0; j < h->n_shards; j++) {
        k.g0 = h->shard_lo[j]; k.g1 = h->shard_lo[j + 1];
        if (k.g1 > k.g0) fn(k, h->shard_stream[j]);
    }
}

// dispatch a kernel template on the runtime layout flags
#define MGX_DISPATCH_F(flags, CALL)                     \
    switch (flags) {                                    \
        case 0: { constexpr int F = 0; CALL; } break;   \
        case 1: { constexpr int F = 1; CALL; } break;   \
        case 2: { constexpr int F = 2; CALL; } break;   \
        case 3: { constexpr int F = 3; CALL; } break;   \
        case 4: { constexpr int F = 4; CALL; } break;   \
        case 5: { constexpr int F = 5; CALL; } break;   \
        case 6: { constexpr int F = 6; CALL; } break;   \
        case 7: { constexpr int F = 7; CALL; } break;   \
        case 14: { constexpr int F = 14; CALL; } break; \
        default: { constexpr int F = 15; CALL; } break; \
    }

}  // namespace

// Grids per workgroup of the fused kernels.  Those kernels are bound by what the BUSIEST compute unit has to stream
// (measured: N = 100 000 in 391 workgroups of 256 leaves 135 CUs with two workgroups and 121 with one: 0.60 of peak;
// 131 072 grids = exactly two per CU: 0.64).  Pick the multiple of 16 grids (one 128-byte line of doubles, so every
// workgroup's rows stay line-aligned) in [192, 256] that minimises  ceil(workgroups / CUs) * grids_per_workgroup
// (smaller workgroups measured slower at equal cost: more, emptier waves).
static int32_t fused_grids_per_block(const mgx_handle *h, int64_t N)
{
    const int cus = h->n_cu > 0 ? h->n_cu : 256;
    int32_t best = BLOCK_K;
    int64_t best_cost = -1;
    for (int32_t g = BLOCK_K; g >= 192 && g >= BLOCK_K - 64; g -= 16) {
        const int64_t blocks = (N + g - 1) / g;
        const int64_t cost = ((blocks + cus - 1) / cus) * g;          // grids streamed by the busiest CU
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = g; }
    }
    return best;
}

template <int F, bool NOISE, typename OT>
static void launch_obs_rows_as(const KArgs &k, const WindowPlan &plan, int32_t t, void *obs, unsigned blocks, size_t lds, hipStream_t st)
{
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void *)obs_rows_wave_kernel<F, NOISE, OT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    obs_rows_wave_kernel<F, NOISE, OT><<<blocks, 64, lds, st>>>(k, plan, t, (OT *)obs);
}

template <int F>
static void launch_obs_rows(const KArgs &k, const WindowPlan &plan, int32_t t, void *obs, unsigned blocks, size_t lds,
                            hipStream_t st)
{
    const bool noise = k.c.load_noise_std || k.c.pv_noise_std || k.c.grid_noise_std;
    if (noise) {
        if (k.obs_f32) launch_obs_rows_as<F, true, float>(k, plan, t, obs, blocks, lds, st);
        else launch_obs_rows_as<F, true, double>(k, plan, t, obs, blocks, lds, st);
    } else {
        if (k.obs_f32) launch_obs_rows_as<F, false, float>(k, plan, t, obs, blocks, lds, st);
        else launch_obs_rows_as<F, false, double>(k, plan, t, obs, blocks, lds, st);
    }
}

static inline unsigned multi_blocks(int64_t n) { return (unsigned)((n + BLOCK_MULTI - 1) / BLOCK_MULTI); }

// observation of the state at series index t into obs [N, D]
static int launch_observe(const mgx_handle *h, int32_t t, void *obs, hipStream_t st)
{
    if (h->multi) {
        MGX_DISPATCH_F(h->flags, (observe_multi_kernel<F><<<multi_blocks(h->k.N), BLOCK_MULTI, 0, st>>>(h->k, t, obs)));
        return MGX_OK;
    }
    if (h->k.H == 0 || h->k.obs_state_only) {
        MGX_DISPATCH_F(h->flags, (observe_kernel<F><<<blocks_for(h->k.N), BLOCK, 0, st>>>(h->k, t, obs)));
        return MGX_OK;
    }
    const int32_t W = 1 + h->k.H, D = h->k.obs_dim;
    WindowPlan plan;
    plan.grid_col_base = 2 * W + 4 * h->layout.has_genset + 2 * h->layout.has_battery;
    plan.ld = D | 1;
    plan.group = 16;                                   // grids per wave tile; halve while a tile would not fit the LDS
    const size_t esz = h->k.obs_f32 ? sizeof(float) : sizeof(double);
    while (plan.group > 1 && (size_t)plan.group * plan.ld * esz > 160 * 1024) plan.group /= 2;
    const size_t lds = ((size_t)plan.group * plan.ld * esz + 7) & ~(size_t)7;
    if (lds > 160 * 1024)
        return fail(MGX_ERR_UNSUPPORTED, "observation rows of %d values do not fit the 160 KiB LDS tile (horizon too large)", D);
    const unsigned blocks = (unsigned)(((int64_t)h->k.N + plan.group - 1) / plan.group);
    MGX_DISPATCH_F(h->flags, (launch_obs_rows<F>(h->k, plan, t, obs, blocks, lds, st)));
    return MGX_OK;
}

extern "C" {

int mgx_abi_version(void) { return MGX_ABI_VERSION; }

const char *mgx_last_error(void) { return g_err; }

int mgx_create(const mgx_layout *L, const mgx_columns *C, mgx_handle **out)
{
    g_err[0] = 0;
    if (!L || !C || !out) return fail(MGX_ERR_INVALID, "mgx_create: NULL argument");
    *out = nullptr;
    if (L->struct_size != (int32_t)sizeof(mgx_layout) || C->struct_size != (int32_t)sizeof(mgx_columns))
        return fail(MGX_ERR_INVALID, "mgx_create: struct_size mismatch (ABI %d): layout %d vs %zu, columns %d vs %zu",
                    MGX_ABI_VERSION, L->struct_size, sizeof(mgx_layout), C->struct_size, sizeof(mgx_columns));
    if (L->n_grids <= 0 || L->n_steps <= 0 || L->horizon < 0)
        return fail(MGX_ERR_INVALID, "mgx_create: need n_grids > 0, n_steps > 0, horizon >= 0");
    if ((L->has_genset | L->has_battery | L->has_grid | L->grid_before_battery) & ~1)
        return fail(MGX_ERR_INVALID, "mgx_create: has_genset / has_battery / has_grid / grid_before_battery must be 0 or 1");
    if (L->n_load < 0 || L->n_pv < 0 || L->n_load > MGX_MAX_MODULES || L->n_pv > MGX_MAX_MODULES)
        return fail(MGX_ERR_UNSUPPORTED, "mgx_create: at most %d load and %d renewable modules per grid (got n_load=%d "
                                         "n_pv=%d)", MGX_MAX_MODULES, MGX_MAX_MODULES, L->n_load, L->n_pv);
    const int32_t n_genset = L->n_genset > 0 ? L->n_genset : L->has_genset, n_battery = L->n_battery > 0 ? L->n_battery : L->has_battery,
                  n_grid = L->n_grid > 0 ? L->n_grid : L->has_grid;
    if (L->n_genset < 0 || L->n_battery < 0 || L->n_grid < 0 || n_genset > MGX_MAX_INSTANCES || n_battery > MGX_MAX_INSTANCES ||
        n_grid > MGX_MAX_INSTANCES)
        return fail(MGX_ERR_UNSUPPORTED, "mgx_create: at most %d gensets, batteries and grids per microgrid (got %d, %d, %d)",
                    MGX_MAX_INSTANCES, L->n_genset, L->n_battery, L->n_grid);
    if ((n_genset > 0) != (L->has_genset != 0) || (n_battery > 0) != (L->has_battery != 0) || (n_grid > 0) != (L->has_grid != 0))
        return fail(MGX_ERR_INVALID, "mgx_create: n_genset / n_battery / n_grid contradict has_genset / has_battery / has_grid");
    const int32_t final_step = L->final_step <= 0 ? L->n_steps : L->final_step;   // base_timeseries_module.py:321-326
    if (final_step > L->n_steps) return fail(MGX_ERR_INVALID, "mgx_create: final_step %d > n_steps %d", final_step, L->n_steps);
    if (L->initial_step < 0 || L->initial_step >= final_step)
        return fail(MGX_ERR_INVALID, "mgx_create: final_step value must be greater than initial_step");
#define NEED(cond, ptr) if ((cond) && !(C->ptr)) return fail(MGX_ERR_INVALID, "mgx_create: column " #ptr " is NULL")
    NEED(L->n_load > 0, load_ts); NEED(L->n_pv > 0, pv_ts); NEED(true, loss_load_cost); NEED(true, overgeneration_cost);
    NEED(L->has_battery, bat_min_capacity); NEED(L->has_battery, bat_max_capacity); NEED(L->has_battery, bat_max_charge);
    NEED(L->has_battery, bat_max_discharge); NEED(L->has_battery, bat_efficiency); NEED(L->has_battery, bat_cost_cycle);
    NEED(L->has_battery, charge); NEED(L->has_battery, soc);
    NEED(L->has_genset, gen_running_min); NEED(L->has_genset, gen_running_max); NEED(L->has_genset, gen_cost);
    NEED(L->has_genset, gen_co2_per_unit); NEED(L->has_genset, gen_cost_per_unit_co2); NEED(L->has_genset, gen_times);
    NEED(L->has_genset, gen_status);
    NEED(L->has_grid, grid_max_import); NEED(L->has_grid, grid_max_export); NEED(L->has_grid, grid_cost_per_unit_co2);
    NEED(L->has_grid, grid_ts);
#undef NEED
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(MGX_ERR_DEVICE, "mgx_create: no HIP device available (%s) -- this engine has no CPU path",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    mgx_handle *h = new (std::nothrow) mgx_handle();
    if (!h) return fail(MGX_ERR_INVALID, "mgx_create: out of host memory");
    if ((e = hipGetDevice(&h->device)) != hipSuccess) { delete h; return hip_fail(e, "hipGetDevice"); }
    h->n_cu = 0;
    (void)hipDeviceGetAttribute(&h->n_cu, hipDeviceAttributeMultiprocessorCount, h->device);
    h->layout = *L;
    h->layout.final_step = final_step;
    h->k.c = *C;
    h->k.N = L->n_grids; h->k.T = L->n_steps; h->k.H = L->horizon; h->k.final_step = final_step;
    h->flags = (L->has_genset ? F_GENSET : 0) | (L->has_battery ? F_BATTERY : 0) | (L->has_grid ? F_GRID : 0) |
               ((L->grid_before_battery && L->has_battery && L->has_grid) ? F_GRID_FIRST : 0);
    h->layout.n_genset = n_genset; h->layout.n_battery = n_battery; h->layout.n_grid = n_grid;
    h->action_dim = 2 * n_genset + n_battery + n_grid;
    const int w = 1 + L->horizon;
    h->k.obs_dim = (L->n_load + L->n_pv) * w + 4 * n_genset + 2 * n_battery + 4 * w * n_grid;
    h->k.n_load = L->n_load; h->k.n_pv = L->n_pv;
    h->k.n_genset = n_genset; h->k.n_battery = n_battery; h->k.n_grid = n_grid;
    h->multi = (L->n_load != 1 || L->n_pv != 1 || n_genset > 1 || n_battery > 1 || n_grid > 1);
    h->multi_lds = 2 * (size_t)multi_list_capacity(L->n_load, L->n_pv, n_genset, n_battery, n_grid) * BLOCK_MULTI * sizeof(double);
    h->k.log_dim = LC_COMMON_END + LC_GENSET_N * n_genset + LC_BATTERY_N * n_battery + LC_GRID_N * n_grid + 1;
    for (int c = 0; c < LC_COMMON_END; c++) h->log_names.push_back(kCommonNames[c]);
    auto add_block = [&](const char *const *names, int n_cols, int n_inst) {       // instance 0: plain names, j > 0: name[j]
        for (int j = 0; j < n_inst; j++)
            for (int c = 0; c < n_cols; c++)
                h->log_names.push_back(j == 0 ? std::string(names[c]) : std::string(names[c]) + "[" + std::to_string(j) + "]");
    };
    add_block(kGensetNames, LC_GENSET_N, n_genset); add_block(kBatteryNames, LC_BATTERY_N, n_battery);
    add_block(kGridNames, LC_GRID_N, n_grid);
    h->log_names.push_back("violations");
    h->d_lists = nullptr;
    h->t = L->initial_step;
    h->k.shaper = MGX_SHAPER_NONE;
    h->k.noise_seed = 0; h->k.noise_increase = 0; h->k.obs_f32 = 0; h->k.obs_state_only = 0; h->k.act_f32 = 0;
    h->window_lo = L->initial_step; h->window_hi = final_step;
    h->k.g0 = 0; h->k.g1 = L->n_grids; h->k.grid_final = nullptr;
    h->n_shards = 1; h->shard_lo[0] = 0; h->shard_lo[1] = L->n_grids;
    for (int j = 0; j < MGX_MAX_SHARDS; j++) { h->shard_stream[j] = nullptr; h->shard_event[j] = nullptr; }
    h->fork_event = nullptr; h->counter_stream = nullptr;
    { const char *e = getenv("MGX_FORK_STAGGER_US"); h->stagger_us = e ? atof(e) : 0.0; }
    h->d_kargs = nullptr; h->k_uploaded_valid = false;
    h->d_table = nullptr; h->table_uploaded_valid = false;
    h->prefetch_stream = nullptr; h->prefetch_gate = nullptr; h->prefetch_done = nullptr; h->prefetch_pending = false;
    h->windowed = false;
    if ((e = hipMalloc((void **)&h->scratch, sizeof(double) * MAX_METRICS * MAX_PARTIAL)) != hipSuccess) {
        delete h;
        return hip_fail(e, "hipMalloc(scratch)");
    }
    h->d_counter = nullptr;
    h->k.t_dev = nullptr;
    if ((e = hipMalloc((void **)&h->d_counter, 4 * sizeof(int32_t))) != hipSuccess) {
        (void)hipFree(h->scratch); delete h;
        return hip_fail(e, "hipMalloc(counter)");
    }
    *out = h;
    return MGX_OK;
}

void mgx_destroy(mgx_handle *h)
{
    if (!h) return;
    for (int j = 0; j < MGX_MAX_SHARDS; j++) {
        if (h->shard_stream[j]) { (void)hipStreamSynchronize(h->shard_stream[j]); (void)hipStreamDestroy(h->shard_stream[j]); }
        if (h->shard_event[j]) (void)hipEventDestroy(h->shard_event[j]);
    }
    if (h->fork_event) (void)hipEventDestroy(h->fork_event);
    if (h->prefetch_stream) { (void)hipStreamSynchronize(h->prefetch_stream); (void)hipStreamDestroy(h->prefetch_stream); }
    if (h->d_kargs) (void)hipFree(h->d_kargs);
    if (h->d_table) (void)hipFree(h->d_table);
    if (h->d_lists) (void)hipFree(h->d_lists);
    if (h->prefetch_gate) (void)hipEventDestroy(h->prefetch_gate);
    if (h->prefetch_done) (void)hipEventDestroy(h->prefetch_done);
    if (h->scratch) (void)hipFree(h->scratch);
    if (h->d_counter) (void)hipFree(h->d_counter);
    delete h;
}

int32_t mgx_action_dim(const mgx_handle *h) { return h ? h->action_dim : -1; }
int32_t mgx_obs_dim(const mgx_handle *h) { return h ? h->k.obs_dim : -1; }
int32_t mgx_log_dim(const mgx_handle *h) { return h ? h->k.log_dim : -1; }
int32_t mgx_current_step(const mgx_handle *h)
{
    if (!h) return -1;
    if (h->k.t_dev) {                    // device-counter mode: the truth lives on the device (blocking read)
        int32_t c[2] = {0, 0};
        // the stream of the last call that touched the counter may be a non-blocking side stream (torch streams are):
        // the null-stream copy below would not wait for it
        if (hipStreamSynchronize(h->counter_stream) != hipSuccess) return -1;
        if (hipMemcpy(c, h->d_counter, sizeof(c), hipMemcpyDeviceToHost) != hipSuccess) return -1;
        return c[0];
    }
    return h->t;
}

int mgx_use_device_counter(mgx_handle *h, int enable, mgx_stream stream)
{
    g_err[0] = 0;
    if (!h) return fail(MGX_ERR_INVALID, "mgx_use_device_counter: NULL handle");
    hipStream_t st = (hipStream_t)stream;
    if (enable && h->n_shards > 1)
        return fail(MGX_ERR_UNSUPPORTED, "mgx_use_device_counter: not offered while the handle steps in shards (mgx_set_shards)");
    h->counter_stream = st;
    if (enable) {
        set_counter_kernel<<<1, 1, 0, st>>>(h->d_counter, h->t);
        h->k.t_dev = h->d_counter;
    } else if (h->k.t_dev) {
        int32_t c[2] = {0, 0};
        hipError_t e = hipStreamSynchronize(st);
        if (e == hipSuccess) e = hipMemcpy(c, h->d_counter, sizeof(c), hipMemcpyDeviceToHost);
        if (e != hipSuccess) return hip_fail(e, "mgx_use_device_counter: reading the counter back");
        h->t = c[0];
        h->k.t_dev = nullptr;
        if (c[1]) return fail(MGX_ERR_RANGE, "a replayed step ran past the end of the time series (length %d)", h->k.T);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MGX_OK : hip_fail(e, "mgx_use_device_counter");
}

const char *mgx_log_name(const mgx_handle *h, int32_t col)
{
    if (!h || col < 0 || col >= h->k.log_dim) return nullptr;
    return h->log_names[col].c_str();
}

static int need_obs_bounds(const mgx_handle *h, const char *who)
{
    const mgx_columns &c = h->k.c;
    if ((h->layout.n_load > 0 && (!c.load_lo || !c.load_hi)) || (h->layout.n_pv > 0 && (!c.pv_lo || !c.pv_hi)) ||
        (h->layout.has_grid && (!c.grid_lo || !c.grid_hi)))
        return fail(MGX_ERR_INVALID, "%s: observations requested but the *_lo / *_hi bound columns are NULL", who);
    return MGX_OK;
}

int mgx_set_obs_format(mgx_handle *h, int32_t format)
{
    g_err[0] = 0;
    if (!h) return fail(MGX_ERR_INVALID, "mgx_set_obs_format: NULL handle");
    if (format != MGX_OBS_F64 && format != MGX_OBS_F32)
        return fail(MGX_ERR_INVALID, "mgx_set_obs_format: unknown format %d", format);
    h->k.obs_f32 = format == MGX_OBS_F32;
    return MGX_OK;
}

int mgx_set_action_format(mgx_handle *h, int32_t format)
{
    g_err[0] = 0;
    if (!h) return fail(MGX_ERR_INVALID, "mgx_set_action_format: NULL handle");
    if (format != MGX_ACT_F64 && format != MGX_ACT_F32)
        return fail(MGX_ERR_INVALID, "mgx_set_action_format: unknown format %d", format);
    h->k.act_f32 = format == MGX_ACT_F32;
    return MGX_OK;
}

int mgx_set_obs_mode(mgx_handle *h, int32_t mode)
{
    g_err[0] = 0;
    if (!h) return fail(MGX_ERR_INVALID, "mgx_set_obs_mode: NULL handle");
    if (mode != MGX_OBS_ROWS_FULL && mode != MGX_OBS_ROWS_STATE_ONLY)
        return fail(MGX_ERR_INVALID, "mgx_set_obs_mode: unknown mode %d", mode);
    if (mode == MGX_OBS_ROWS_STATE_ONLY && h->multi)
        return fail(MGX_ERR_UNSUPPORTED, "mgx_set_obs_mode: state-only rows need exactly one load and one renewable module per grid");
    h->k.obs_state_only = mode == MGX_OBS_ROWS_STATE_ONLY;
    return MGX_OK;
}

// shape of a window prefetch: LDS image plan + bytes; n_groups = 16-grid groups of the batch
static int windows_plan(const mgx_handle *h, int32_t ahead, int32_t K, const void *ring, const char *who, WindowsKPlan *plan,
                        size_t *lds_out, int32_t *n_groups)
{
    if (!h || !ring) return fail(MGX_ERR_INVALID, "%s: NULL argument", who);
    if (K < 1 || K > 4096) return fail(MGX_ERR_INVALID, "%s: K = %d outside [1, 4096]", who, K);
    if (ahead < 0) return fail(MGX_ERR_INVALID, "%s: ahead = %d is negative", who, ahead);
    if (h->multi) return fail(MGX_ERR_UNSUPPORTED, "%s: needs exactly one module of every kind per grid", who);
    if (h->k.c.load_noise_std || h->k.c.pv_noise_std || h->k.c.grid_noise_std)
        return fail(MGX_ERR_UNSUPPORTED, "%s: forecast noise depends on (step, horizon index), windows cannot be shared", who);
    if (int rc = need_obs_bounds(h, who)) return rc;
    if (!dev_counter(h) && ahead == 0 && h->t > h->k.T)
        return fail(MGX_ERR_RANGE, "%s: step %d is outside the time series (length %d)", who, h->t, h->k.T);
    const int32_t W = 1 + h->k.H, R = K + h->k.H, ncomp = 2 + 4 * h->layout.has_grid;
    plan->grid_col_base = 2 * W + 4 * h->layout.has_genset + 2 * h->layout.has_battery;
    plan->K = K;
    plan->rp = R;
    plan->bp = (ncomp * (R + K) + 6 * K) | 1;
    static const int group_env = [] { const char *e = getenv("MGX_WIN_GROUP"); return e ? atoi(e) : 0; }();   // experiment knob
    plan->group = (group_env == 8 || group_env == 4 || group_env == 16) ? group_env : 16;
    plan->with_state = ahead == 0;
    plan->group0 = 0;
    auto lds_of = [&](int32_t g) { return (size_t)g * plan->bp * sizeof(double) + (size_t)h->k.obs_dim * sizeof(uint32_t); };
    while (plan->group > 1 && lds_of(plan->group) > 160 * 1024) plan->group /= 2;
    const size_t lds = (lds_of(plan->group) + 7) & ~(size_t)7;
    if (lds > 160 * 1024)
        return fail(MGX_ERR_UNSUPPORTED, "%s: K + horizon = %d rows do not fit the 160 KiB LDS", who, R);
    *lds_out = lds;
    *n_groups = (int32_t)(((int64_t)h->k.N + plan->group - 1) / plan->group);
    return MGX_OK;
}

// groups [chunk * per, (chunk + 1) * per) of n_groups, per = ceil(n_groups / n_chunks)
static void chunk_range(int32_t n_groups, int32_t chunk, int32_t n_chunks, int32_t *first, int32_t *count)
{
    if (n_chunks <= 1) { *first = 0; *count = n_groups; return; }
    const int32_t per = (n_groups + n_chunks - 1) / n_chunks;
    const int64_t lo = (int64_t)chunk * per, hi = lo + per;
    *first = (int32_t)(lo < n_groups ? lo : n_groups);
    *count = (int32_t)((hi < n_groups ? hi : n_groups) - *first);
}

static int launch_windows(mgx_handle *h, int32_t ahead, int32_t K, void *ring, hipStream_t st, const char *who,
                          int32_t chunk = 0, int32_t n_chunks = 1)
{
    WindowsKPlan plan;
    size_t lds;
    int32_t n_groups, first, count;
    if (int rc = windows_plan(h, ahead, K, ring, who, &plan, &lds, &n_groups)) return rc;
    chunk_range(n_groups, chunk, n_chunks, &first, &count);
    if (count <= 0) return MGX_OK;
    plan.group0 = first;
    const unsigned blocks = (unsigned)count;
    const int32_t t = t_arg(h) + ahead;
    if (h->k.obs_f32) {
        MGX_DISPATCH_F(h->flags, ((lds > 64 * 1024 ? (void)hipFuncSetAttribute((const void *)obs_windows_k_kernel<F, float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) : (void)0),
                                  obs_windows_k_kernel<F, float><<<blocks, OBS_K_THREADS, lds, st>>>(h->k, plan, t, (float *)ring)));
    } else {
        MGX_DISPATCH_F(h->flags, ((lds > 64 * 1024 ? (void)hipFuncSetAttribute((const void *)obs_windows_k_kernel<F, double>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) : (void)0),
                                  obs_windows_k_kernel<F, double><<<blocks, OBS_K_THREADS, lds, st>>>(h->k, plan, t, (double *)ring)));
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MGX_OK : hip_fail(e, "obs_windows_k_kernel launch");
}

int mgx_observe_windows(mgx_handle *h, int32_t K, void *ring, mgx_stream stream)
{
    g_err[0] = 0;
    // a prefetch still in flight may be writing this very ring (a reset in the middle of an episode: the ring being refilled
    // now can be the one the last mgx_observe_windows_ahead targets): its stale rows must not land on top of the new ones
    if (h && h->prefetch_pending) { if (int rc = mgx_prefetch_wait(h, stream)) return rc; }
    return launch_windows(h, 0, K, ring, (hipStream_t)stream, "mgx_observe_windows");
}

int mgx_observe_windows_ahead(mgx_handle *h, int32_t ahead, int32_t K, void *ring, mgx_stream stream)
{
    g_err[0] = 0;
    if (!h) return fail(MGX_ERR_INVALID, "mgx_observe_windows_ahead: NULL handle");
    if (ahead < 1) return fail(MGX_ERR_INVALID, "mgx_observe_windows_ahead: ahead must be >= 1 (mgx_observe_windows is the ahead = 0 form)");
    if (dev_counter(h)) return fail(MGX_ERR_UNSUPPORTED, "mgx_observe_windows_ahead: not offered in device-counter mode (the prefetch "
                                                         "stream would race with the kernels that move the counter)");
    hipError_t e = hipSuccess;
    DeviceGuard on_device(h->device);
    if (!h->prefetch_stream) {
        // (a low- or high-priority prefetch stream is slower: 31.4 / 33.5 vs 29.6 us per config-5 fleet step)
        e = hipStreamCreateWithFlags(&h->prefetch_stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&h->prefetch_gate, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&h->prefetch_done, hipEventDisableTiming);
        if (e != hipSuccess) return hip_fail(e, "mgx_observe_windows_ahead: creating the prefetch stream");
    }
    // readers of the ring's previous contents were queued on `stream`: the prefetch starts behind them
    e = hipEventRecord(h->prefetch_gate, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(h->prefetch_stream, h->prefetch_gate, 0);
    if (e != hipSuccess) return hip_fail(e, "mgx_observe_windows_ahead: ordering behind the caller's stream");
    if (int rc = launch_windows(h, ahead, K, ring, h->prefetch_stream, "mgx_observe_windows_ahead")) return rc;
    e = hipEventRecord(h->prefetch_done, h->prefetch_stream);
    h->prefetch_pending = true;
    return e == hipSuccess ? MGX_OK : hip_fail(e, "mgx_observe_windows_ahead: recording the completion event");
}

int mgx_prefetch_wait(mgx_handle *h, mgx_stream stream)
{
    g_err[0] = 0;
    if (!h) return fail(MGX_ERR_INVALID, "mgx_prefetch_wait: NULL handle");
    if (!h->prefetch_pending) return MGX_OK;
    hipError_t e = hipStreamWaitEvent((hipStream_t)stream, h->prefetch_done, 0);
    return e == hipSuccess ? MGX_OK : hip_fail(e, "mgx_prefetch_wait");
}

int mgx_observe(mgx_handle *h, void *obs, mgx_stream stream)
{
    g_err[0] = 0;
    if (!h || !obs) return fail(MGX_ERR_INVALID, "mgx_observe: NULL argument");
    if (int rc = need_obs_bounds(h, "mgx_observe")) return rc;
    if (int rc = launch_observe(h, t_arg(h), obs, (hipStream_t)stream)) return rc;
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MGX_OK : hip_fail(e, "observe launch");
}

int mgx_set_window(mgx_handle *h, int32_t initial_step, int32_t final_step)
{
    g_err[0] = 0;
    if (!h) return fail(MGX_ERR_INVALID, "mgx_set_window: NULL handle");
    if (initial_step < h->window_lo)
        return fail(MGX_ERR_INVALID, "trajectory_func returned initial_step value (%d) less than env's initial step: (%d)",
                    initial_step, h->window_lo);
    if (final_step > h->window_hi)
        return fail(MGX_ERR_INVALID, "trajectory_func returned final_step value (%d) greater than env's final step: (%d)",
                    final_step, h->window_hi);
    if (initial_step >= final_step)
        return fail(MGX_ERR_INVALID, "trajectory_func returned values (%d, %d) such that initial_step was greater than "
                                     "or equal to final_step.", initial_step, final_step);
    h->layout.initial_step = initial_step;
    h->layout.final_step = final_step;
    h->k.final_step = final_step;
    return MGX_OK;
}

int mgx_set_reward_shaper(mgx_handle *h, int32_t shaper)
{
    g_err[0] = 0;
    if (!h) return fail(MGX_ERR_INVALID, "mgx_set_reward_shaper: NULL handle");
    if (shaper < MGX_SHAPER_NONE || shaper > MGX_SHAPER_BATTERY_DISCHARGE)
        return fail(MGX_ERR_INVALID, "mgx_set_reward_shaper: unknown shaper %d", shaper);
    h->k.shaper = shaper;
    return MGX_OK;
}

int mgx_set_forecast_noise(mgx_handle *h, uint64_t seed, int increase_uncertainty)
{
    g_err[0] = 0;
    if (!h) return fail(MGX_ERR_INVALID, "mgx_set_forecast_noise: NULL handle");
    h->k.noise_seed = seed;
    h->k.noise_increase = increase_uncertainty ? 1 : 0;
    return MGX_OK;
}

// device copy of the handle's KArgs (fleet_step_kernel, step_dk_kernel): uploaded on `st` when it changed
static int sync_device_kargs(mgx_handle *h, hipStream_t st, const char *who)
{
    if (h->k_uploaded_valid && memcmp(&h->k, &h->k_uploaded, sizeof(KArgs)) == 0) return MGX_OK;
    DeviceGuard on_device(h->device);
    hipError_t e = hipSuccess;
    if (!h->d_kargs) e = hipMalloc((void **)&h->d_kargs, sizeof(KArgs));
    if (e == hipSuccess) e = hipMemcpyAsync(h->d_kargs, &h->k, sizeof(KArgs), hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return hip_fail(e, who);
    memcpy(&h->k_uploaded, &h->k, sizeof(KArgs));
    h->k_uploaded_valid = true;
    return MGX_OK;
}

// back to the full series after a per-grid-window episode
static void leave_windows(mgx_handle *h)
{
    if (!h->windowed) return;
    h->k.c.load_ts = h->full_load_ts; h->k.c.pv_ts = h->full_pv_ts; h->k.c.grid_ts = h->full_grid_ts;
    h->k.T = h->full_T; h->k.final_step = h->full_final; h->k.grid_final = nullptr;
    h->layout.n_steps = h->full_T; h->layout.final_step = h->full_final; h->layout.initial_step = h->full_initial;
    h->window_lo = h->full_window_lo; h->window_hi = h->full_window_hi;
    h->windowed = false;
}

int mgx_reset(mgx_handle *h, int32_t initial_step, void *obs, mgx_stream stream)
{
    g_err[0] = 0;
    if (!h) return fail(MGX_ERR_INVALID, "mgx_reset: NULL handle");
    leave_windows(h);
    const int32_t t0 = initial_step >= 0 ? initial_step : h->layout.initial_step;
    if (t0 >= h->layout.final_step)
        return fail(MGX_ERR_INVALID, "mgx_reset: initial_step %d must be below final_step %d", t0, h->layout.final_step);
    h->t = t0;                       // base_module.py:292-296 -- nothing else is restored
    if (h->k.t_dev) { set_counter_kernel<<<1, 1, 0, (hipStream_t)stream>>>(h->d_counter, t0); h->counter_stream = (hipStream_t)stream; }
    return obs ? mgx_observe(h, obs, stream) : MGX_OK;
}

int mgx_reset_windows(mgx_handle *h, const int32_t *start, const int32_t *length, int32_t max_length, double *load_w,
                      double *pv_w, double *grid_w, int32_t *final_rel, void *obs, mgx_stream stream)
{
    g_err[0] = 0;
    if (!h || !start || !load_w || !pv_w) return fail(MGX_ERR_INVALID, "mgx_reset_windows: NULL argument");
    if (h->multi) return fail(MGX_ERR_UNSUPPORTED, "mgx_reset_windows: needs exactly one module of every kind per grid");
    if (h->layout.has_grid && !grid_w) return fail(MGX_ERR_INVALID, "mgx_reset_windows: grid_w is NULL but the layout has a GridModule");
    if (length && !final_rel) return fail(MGX_ERR_INVALID, "mgx_reset_windows: per-grid lengths need the final_rel buffer");
    if (h->k.t_dev) return fail(MGX_ERR_UNSUPPORTED, "mgx_reset_windows: not offered in device-counter mode");
    if (h->n_shards > 1) return fail(MGX_ERR_UNSUPPORTED, "mgx_reset_windows: not offered while the handle steps in shards");
    if (!h->windowed) {
        h->full_load_ts = h->k.c.load_ts; h->full_pv_ts = h->k.c.pv_ts; h->full_grid_ts = h->k.c.grid_ts;
        h->full_T = h->k.T; h->full_final = h->layout.final_step; h->full_initial = h->layout.initial_step;
        h->full_window_lo = h->window_lo; h->full_window_hi = h->window_hi;
    }
    if (max_length < 1 || max_length > h->full_window_hi - h->full_window_lo)
        return fail(MGX_ERR_INVALID, "Cannot create a trajectory of length %d between initial_step (%d) and final_step (%d)",
                    max_length, h->full_window_lo, h->full_window_hi);
    const int32_t rows = max_length + h->k.H + 1;
    hipStream_t st = (hipStream_t)stream;
    GatherArgs g;
    g.load_ts = h->full_load_ts; g.pv_ts = h->full_pv_ts; g.grid_ts = h->layout.has_grid ? h->full_grid_ts : nullptr;
    g.load_lo = h->k.c.load_lo; g.load_hi = h->k.c.load_hi; g.pv_lo = h->k.c.pv_lo; g.pv_hi = h->k.c.pv_hi;
    g.grid_lo = h->k.c.grid_lo; g.grid_hi = h->k.c.grid_hi;
    g.load_w = load_w; g.pv_w = pv_w; g.grid_w = grid_w;
    g.start = start; g.length = length; g.final_rel = final_rel;
    g.N = h->k.N; g.T = h->full_T; g.rows = rows; g.max_length = max_length;
    g.lo = h->full_window_lo; g.hi = h->full_window_hi;
    gather_windows_kernel<<<blocks_for(h->k.N), BLOCK, 0, st>>>(g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "gather_windows_kernel launch");
    h->k.c.load_ts = load_w; h->k.c.pv_ts = pv_w; if (h->layout.has_grid) h->k.c.grid_ts = grid_w;
    h->k.T = rows; h->k.final_step = max_length; h->k.grid_final = length ? final_rel : nullptr;
    h->layout.n_steps = rows; h->layout.final_step = max_length; h->layout.initial_step = 0;
    h->window_lo = 0; h->window_hi = max_length;
    h->windowed = true;
    h->t = 0;
    return obs ? mgx_observe(h, obs, stream) : MGX_OK;
}

// ---- shards -------------------------------------------------------------------------------------------------
int mgx_set_shards(mgx_handle *h, int32_t n_shards)
{
    g_err[0] = 0;
    if (!h) return fail(MGX_ERR_INVALID, "mgx_set_shards: NULL handle");
    if (n_shards < 1 || n_shards > MGX_MAX_SHARDS) return fail(MGX_ERR_INVALID, "mgx_set_shards: n_shards must be in [1, %d]", MGX_MAX_SHARDS);
    if (n_shards > 1 && h->k.t_dev) return fail(MGX_ERR_UNSUPPORTED, "mgx_set_shards: not offered in device-counter mode");
    if (n_shards > 1 && h->windowed) return fail(MGX_ERR_UNSUPPORTED, "mgx_set_shards: not offered during a per-grid-window episode");
    DeviceGuard on_device(h->device);
    for (int j = 0; j < h->n_shards && h->n_shards > 1; j++)          // work still queued on the old shard streams
        if (h->shard_stream[j]) (void)hipStreamSynchronize(h->shard_stream[j]);
    hipError_t e = hipSuccess;
    if (n_shards > 1) {
        if (!h->fork_event) e = hipEventCreateWithFlags(&h->fork_event, hipEventDisableTiming);
        for (int j = 0; j < n_shards && e == hipSuccess; j++) {
            if (!h->shard_stream[j]) e = hipStreamCreateWithFlags(&h->shard_stream[j], hipStreamNonBlocking);
            if (e == hipSuccess && !h->shard_event[j]) e = hipEventCreateWithFlags(&h->shard_event[j], hipEventDisableTiming);
        }
        if (e != hipSuccess) return hip_fail(e, "mgx_set_shards: creating streams / events");
    }
    // contiguous ranges whose bounds are multiples of 256 grids (rows of every stream stay line-aligned per shard)
    const int64_t N = h->k.N;
    int64_t per = ((N + n_shards - 1) / n_shards + 255) / 256 * 256;
    for (int j = 0; j <= n_shards; j++) { const int64_t b = (int64_t)j * per; h->shard_lo[j] = (int32_t)(b < N ? b : N); }
    h->shard_lo[n_shards] = (int32_t)N;
    h->n_shards = n_shards;
    return MGX_OK;
}

int mgx_fork(mgx_handle *h, mgx_stream stream)
{
    g_err[0] = 0;
    if (!h) return fail(MGX_ERR_INVALID, "mgx_fork: NULL handle");
    if (h->n_shards <= 1) return MGX_OK;
    hipError_t e = hipEventRecord(h->fork_event, (hipStream_t)stream);
    for (int j = 0; j < h->n_shards && e == hipSuccess; j++) e = hipStreamWaitEvent(h->shard_stream[j], h->fork_event, 0);
    // Stagger: shard j starts j / S of `stagger_us` late, so that the launch boundaries of the shards -- which otherwise
    // all start together after a fork and, their kernels being equally long, STAY together -- interleave: one range's
    // ramp-up / tail then always falls into the others' steady state (mgx_set_shard_stagger; 0 = off).
    if (h->stagger_us > 0)
        for (int j = 1; j < h->n_shards && e == hipSuccess; j++) {
            stagger_kernel<<<1, 64, 0, h->shard_stream[j]>>>((int64_t)(h->stagger_us * 100.0 * j / h->n_shards));
            e = hipGetLastError();
        }
    return e == hipSuccess ? MGX_OK : hip_fail(e, "mgx_fork");
}

int mgx_join(mgx_handle *h, mgx_stream stream)
{
    g_err[0] = 0;
    if (!h) return fail(MGX_ERR_INVALID, "mgx_join: NULL handle");
    if (h->n_shards <= 1) return MGX_OK;
    hipError_t e = hipSuccess;
    for (int j = 0; j < h->n_shards && e == hipSuccess; j++) {
        e = hipEventRecord(h->shard_event[j], h->shard_stream[j]);
        if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)stream, h->shard_event[j], 0);
    }
    return e == hipSuccess ? MGX_OK : hip_fail(e, "mgx_join");
}

void *mgx_shard_stream(mgx_handle *h, int32_t shard)
{
    if (!h || shard < 0 || shard >= h->n_shards || h->n_shards <= 1) return nullptr;
    return (void *)h->shard_stream[shard];
}

// ---- single steps ----------------------------------------------------------------------------------------------
// one Microgrid.run of every grid: the launches of mgx_step without its argument checks
static int step_once(mgx_handle *h, const void *actions, int normalized, double *reward, uint8_t *done, void *obs, double *log,
                     hipStream_t st)
{
    if (h->multi) {
        for_each_shard(h, st, [&](const KArgs &k, hipStream_t s) {
            MGX_DISPATCH_F(h->flags, (step_multi_kernel<F><<<multi_blocks(k.g1 - k.g0), BLOCK_MULTI, h->multi_lds, s>>>(
                                          k, actions, t_arg(h), normalized, reward, done, obs, log)));
        });
        hipError_t em = hipGetLastError();
        if (em != hipSuccess) return hip_fail(em, "step_multi_kernel launch");
        advance(h, 1, st);
        return MGX_OK;
    }
    void *obs_inline = (obs && (h->k.H == 0 || h->k.obs_state_only)) ? obs : nullptr;
    for_each_shard(h, st, [&](const KArgs &k, hipStream_t s) {
        MGX_DISPATCH_F(h->flags, (step_kernel<F><<<blocks_for(k.g1 - k.g0), BLOCK, 0, s>>>(k, actions, t_arg(h), normalized, reward,
                                                                                       done, obs_inline, log)));
    });
    if (obs && !obs_inline) { if (int rc = launch_observe(h, dev_counter(h) ? 0 : h->t + 1, obs, st)) return rc; }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "step_kernel launch");
    advance(h, 1, st);
    return MGX_OK;
}

static int check_step_args(const mgx_handle *h, const void *actions, const double *reward, const void *obs, int32_t K, const char *who)
{
    if (!h || !reward || (h->action_dim > 0 && !actions)) return fail(MGX_ERR_INVALID, "%s: NULL argument", who);
    if (K <= 0) return fail(MGX_ERR_INVALID, "%s: K must be positive", who);
    if (!dev_counter(h) && (h->t < 0 || (int64_t)h->t + K > step_limit(h)))
        return K == 1 ? fail(MGX_ERR_RANGE, "%s: step %d is outside the time series (length %d)", who, h->t, step_limit(h))
                      : fail(MGX_ERR_RANGE, "%s: steps [%d, %d) leave the time series (length %d)", who, h->t, h->t + K, step_limit(h));
    if (obs) {
        if (int rc = need_obs_bounds(h, who)) return rc;
        if (h->n_shards > 1 && !h->multi && h->k.H > 0 && !h->k.obs_state_only)
            return fail(MGX_ERR_UNSUPPORTED, "%s: observation rows with a forecast horizon are not written per shard; "
                                             "mgx_join, mgx_observe on your stream, mgx_fork", who);
    }
    return MGX_OK;
}

int mgx_step(mgx_handle *h, const void *actions, int normalized, double *reward, uint8_t *done, void *obs,
             double *log, mgx_stream stream)
{
    g_err[0] = 0;
    if (int rc = check_step_args(h, actions, reward, obs, 1, "mgx_step")) return rc;
    return step_once(h, actions, normalized, reward, done, obs, log, (hipStream_t)stream);
}

int mgx_check_step(mgx_handle *h, const void *actions, int normalized, uint32_t *violations, mgx_stream stream)
{
    g_err[0] = 0;
    if (!h || !violations || (h->action_dim > 0 && !actions)) return fail(MGX_ERR_INVALID, "mgx_check_step: NULL argument");
    if (!dev_counter(h) && (h->t < 0 || h->t >= step_limit(h)))
        return fail(MGX_ERR_RANGE, "mgx_check_step: step %d is outside the time series (length %d)", h->t, step_limit(h));
    for_each_shard(h, (hipStream_t)stream, [&](const KArgs &k, hipStream_t s) {
        if (h->multi) {
            MGX_DISPATCH_F(h->flags, (check_multi_kernel<F><<<multi_blocks(k.g1 - k.g0), BLOCK_MULTI, 0, s>>>(k, actions, t_arg(h), normalized, violations)));
        } else {
            MGX_DISPATCH_F(h->flags, (check_kernel<F><<<blocks_for(k.g1 - k.g0), BLOCK, 0, s>>>(k, actions, t_arg(h), normalized, violations)));
        }
    });
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MGX_OK : hip_fail(e, "check_kernel launch");
}

int mgx_step_many(mgx_handle *h, const void *actions, int32_t K, int normalized, double *reward, uint8_t *done, void *obs,
                  double *log, mgx_stream stream)
{
    g_err[0] = 0;
    if (int rc = check_step_args(h, actions, reward, obs, K, "mgx_step_many")) return rc;
    const int64_t N = h->k.N;
    const size_t act_row = (size_t)N * h->action_dim * (h->k.act_f32 ? sizeof(float) : sizeof(double));
    const size_t obs_row = (size_t)N * h->k.obs_dim * (h->k.obs_f32 ? sizeof(float) : sizeof(double));
    for (int32_t k = 0; k < K; k++) {
        if (int rc = step_once(h, actions ? (const char *)actions + k * act_row : nullptr, normalized, reward + k * N,
                               done ? done + k * N : nullptr, obs ? (char *)obs + k * obs_row : nullptr,
                               log ? log + (int64_t)k * h->k.log_dim * N : nullptr, (hipStream_t)stream))
            return rc;
    }
    return MGX_OK;
}

int mgx_step_k(mgx_handle *h, const void *actions, int32_t K, int normalized, double *reward, uint8_t *done,
               double *soc_trace, uint32_t *status_trace, double *ret_acc, double *log, mgx_stream stream)
{
    g_err[0] = 0;
    if (!h || (h->action_dim > 0 && !actions)) return fail(MGX_ERR_INVALID, "mgx_step_k: NULL argument");
    if (K <= 0) return fail(MGX_ERR_INVALID, "mgx_step_k: K must be positive");
    if (!dev_counter(h) && (h->t < 0 || (int64_t)h->t + K > step_limit(h)))
        return fail(MGX_ERR_RANGE, "mgx_step_k: steps [%d, %d) leave the time series (length %d)", h->t, h->t + K, step_limit(h));
    hipStream_t st = (hipStream_t)stream;
    const FusedOut fo{reward, done, soc_trace, status_trace, ret_acc, log};
    if (h->multi) {                                       // general path: the K-step loop around the general step
        for_each_shard(h, st, [&](const KArgs &k, hipStream_t s) {
            MGX_DISPATCH_F(h->flags, (step_k_multi_kernel<F><<<multi_blocks(k.g1 - k.g0), BLOCK_MULTI, h->multi_lds, s>>>(
                                          k, actions, nullptr, 0, 0, nullptr, 0, t_arg(h), K, normalized, fo)));
        });
        hipError_t em = hipGetLastError();
        if (em != hipSuccess) return hip_fail(em, "step_k_multi_kernel launch");
        advance(h, K, st);
        return MGX_OK;
    }
    for_each_shard(h, st, [&](const KArgs &k, hipStream_t s) {
        const int32_t gpb = fused_grids_per_block(h, k.g1 - k.g0);
        const unsigned blocks = (unsigned)((k.g1 - k.g0 + gpb - 1) / gpb);
        const bool rich = log != nullptr || status_trace != nullptr;
        if (k.act_f32) {
            if (rich) { MGX_DISPATCH_F(h->flags, (step_k_kernel<F, MGX_RING, float, true><<<blocks, BLOCK_K, 0, s>>>(
                                                      k, (const float *)actions, t_arg(h), K, normalized, fo, gpb))); }
            else { MGX_DISPATCH_F(h->flags, (step_k_kernel<F, MGX_RING, float, false><<<blocks, BLOCK_K, 0, s>>>(
                                                 k, (const float *)actions, t_arg(h), K, normalized, fo, gpb))); }
        } else {
            if (rich) { MGX_DISPATCH_F(h->flags, (step_k_kernel<F, MGX_RING, double, true><<<blocks, BLOCK_K, 0, s>>>(
                                                      k, (const double *)actions, t_arg(h), K, normalized, fo, gpb))); }
            else { MGX_DISPATCH_F(h->flags, (step_k_kernel<F, MGX_RING, double, false><<<blocks, BLOCK_K, 0, s>>>(
                                                 k, (const double *)actions, t_arg(h), K, normalized, fo, gpb))); }
        }
    });
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "step_k_kernel launch");
    advance(h, K, st);
    return MGX_OK;
}

static int encode_table(const mgx_handle *h, const int32_t *table, int32_t n_actions, PLWords *tab, const char *who)
{
    if (n_actions <= 0 || n_actions > 12) return fail(MGX_ERR_INVALID, "%s: n_actions must be in [1, 12]", who);
    memset(tab, 0, sizeof(*tab));
    tab->n_actions = n_actions;
    for (int i = 0; i < n_actions; i++)
        for (int k = 0; k < 3; k++) {
            const int32_t mod = table[(i * 3 + k) * 2], act = table[(i * 3 + k) * 2 + 1];
            if (mod == -1) continue;
            if (mod < 0 || mod > 2 || act < 0 || act > 1) return fail(MGX_ERR_INVALID, "%s: bad table entry (%d, %d)", who, mod, act);
            if (mod == 0 && !h->layout.has_genset) return fail(MGX_ERR_INVALID, "%s: table names a genset, layout has none", who);
            if (mod == 1 && !h->layout.has_battery) return fail(MGX_ERR_INVALID, "%s: table names a battery, layout has none", who);
            if (mod == 2 && !h->layout.has_grid) return fail(MGX_ERR_INVALID, "%s: table names a grid, layout has none", who);
            for (int k2 = 0; k2 < k; k2++)
                if (table[(i * 3 + k2) * 2] == mod)
                    return fail(MGX_ERR_INVALID, "%s: list %d names module %d twice (priority lists hold each module once, "
                                                 "priority_list.py:40-48)", who, i, mod);
            tab->w[i] |= ((uint32_t)mod | ((uint32_t)act << 2) | 8u) << (4 * k);
        }
    return MGX_OK;
}

static int launch_expand_lists(mgx_handle *h, const int32_t *action_id, const int32_t *d_lists, int32_t n_lists, int32_t list_len,
                               double *control, hipStream_t st)
{
    for_each_shard(h, st, [&](const KArgs &k, hipStream_t s) {
        MGX_DISPATCH_F(h->flags, (expand_multi_kernel<F><<<multi_blocks(k.g1 - k.g0), BLOCK_MULTI, 0, s>>>(k, d_lists, n_lists, list_len,
                                                                                                       action_id, t_arg(h), control)));
    });
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MGX_OK : hip_fail(e, "expand_multi_kernel launch");
}

int mgx_expand_discrete(mgx_handle *h, const int32_t *action_id, const int32_t *table, int32_t n_actions,
                        double *control, mgx_stream stream)
{
    g_err[0] = 0;
    if (!h || !action_id || !table || !control) return fail(MGX_ERR_INVALID, "mgx_expand_discrete: NULL argument");
    if (!dev_counter(h) && (h->t < 0 || h->t >= step_limit(h)))
        return fail(MGX_ERR_RANGE, "mgx_expand_discrete: step %d is outside the time series (length %d)", h->t, step_limit(h));
    PLWords tab;
    if (int rc = encode_table(h, table, n_actions, &tab, "mgx_expand_discrete")) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (h->multi) {                                       // general path: the table as device lists of (kind, instance 0, action)
        if (h->layout.n_genset > 1 || h->layout.n_battery > 1 || h->layout.n_grid > 1)
            return fail(MGX_ERR_UNSUPPORTED, "mgx_expand_discrete: the layout has several gensets / batteries / grids, its priority "
                                             "lists name module instances: use mgx_expand_lists");
        std::vector<int32_t> lists((size_t)n_actions * 9);
        for (int i = 0; i < n_actions * 3; i++) {
            lists[3 * i] = table[2 * i]; lists[3 * i + 1] = 0; lists[3 * i + 2] = table[2 * i + 1];
        }
        if (lists != h->lists_uploaded) {
            DeviceGuard on_device(h->device);
            hipError_t e = hipSuccess;
            if (!h->d_lists) e = hipMalloc((void **)&h->d_lists, 12 * 9 * sizeof(int32_t));
            // the previous table may still be read by a launch in flight on another stream: settle before overwriting
            if (e == hipSuccess && !h->lists_uploaded.empty()) e = hipDeviceSynchronize();
            if (e == hipSuccess) e = hipMemcpyAsync(h->d_lists, lists.data(), lists.size() * sizeof(int32_t), hipMemcpyHostToDevice, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);     // `lists` is a host temporary
            if (e != hipSuccess) return hip_fail(e, "mgx_expand_discrete: uploading the priority lists");
            h->lists_uploaded = lists;
        }
        return launch_expand_lists(h, action_id, h->d_lists, n_actions, 3, control, st);
    }
    for_each_shard(h, st, [&](const KArgs &k, hipStream_t s) {
        MGX_DISPATCH_F(h->flags, (expand_kernel<F><<<blocks_for(k.g1 - k.g0), BLOCK, 0, s>>>(k, tab, action_id, t_arg(h), control)));
    });
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MGX_OK : hip_fail(e, "expand_kernel launch");
}

int mgx_expand_lists(mgx_handle *h, const int32_t *action_id, const int32_t *lists, int32_t n_lists, int32_t list_len,
                     double *control, mgx_stream stream)
{
    g_err[0] = 0;
    if (!h || !action_id || !lists || !control) return fail(MGX_ERR_INVALID, "mgx_expand_lists: NULL argument");
    if (n_lists <= 0 || list_len <= 0 || list_len > 3 * MGX_MAX_INSTANCES)
        return fail(MGX_ERR_INVALID, "mgx_expand_lists: need n_lists > 0 and list_len in [1, %d]", 3 * MGX_MAX_INSTANCES);
    if (!dev_counter(h) && (h->t < 0 || h->t >= step_limit(h)))
        return fail(MGX_ERR_RANGE, "mgx_expand_lists: step %d is outside the time series (length %d)", h->t, step_limit(h));
    return launch_expand_lists(h, action_id, lists, n_lists, list_len, control, (hipStream_t)stream);
}

int mgx_step_discrete(mgx_handle *h, const int32_t *action_id, const int32_t *table, int32_t n_actions, double *control,
                      double *reward, uint8_t *done, void *obs, double *log, mgx_stream stream)
{
    g_err[0] = 0;
    if (!h || !action_id || !table || !reward) return fail(MGX_ERR_INVALID, "mgx_step_discrete: NULL argument");
    if (h->multi) return fail(MGX_ERR_UNSUPPORTED, "mgx_step_discrete: needs exactly one module of every kind per "
                                                    "grid; use mgx_expand_discrete / mgx_expand_lists + mgx_step");
    if (int rc = check_step_args(h, action_id, reward, obs, 1, "mgx_step_discrete")) return rc;
    PLWords tab;
    if (int rc = encode_table(h, table, n_actions, &tab, "mgx_step_discrete")) return rc;
    hipStream_t st = (hipStream_t)stream;
    void *obs_inline = (obs && (h->k.H == 0 || h->k.obs_state_only)) ? obs : nullptr;
    for_each_shard(h, st, [&](const KArgs &k, hipStream_t s) {
        MGX_DISPATCH_F(h->flags, (step_discrete_kernel<F><<<blocks_for(k.g1 - k.g0), BLOCK, 0, s>>>(k, tab, action_id, t_arg(h), control,
                                                                                                reward, done, obs_inline, log)));
    });
    if (obs && !obs_inline) { if (int rc = launch_observe(h, dev_counter(h) ? 0 : h->t + 1, obs, st)) return rc; }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "step_discrete_kernel launch");
    advance(h, 1, st);
    return MGX_OK;
}

int mgx_rollout_discrete(mgx_handle *h, const uint8_t *action_id, int per_step, const int32_t *table, int32_t n_actions,
                         int32_t K, double *reward, uint8_t *done, double *soc_trace, uint32_t *status_trace,
                         double *ret_acc, double *log, mgx_stream stream)
{
    g_err[0] = 0;
    if (!h || !action_id || !table) return fail(MGX_ERR_INVALID, "mgx_rollout_discrete: NULL argument");
    if (K <= 0) return fail(MGX_ERR_INVALID, "mgx_rollout_discrete: K must be positive");
    if (h->multi) return fail(MGX_ERR_UNSUPPORTED, "mgx_rollout_discrete: needs exactly one module of every kind "
                                                    "per grid; use mgx_expand_discrete / mgx_expand_lists + mgx_step");
    if (!dev_counter(h) && (h->t < 0 || (int64_t)h->t + K > step_limit(h)))
        return fail(MGX_ERR_RANGE, "mgx_rollout_discrete: steps [%d, %d) leave the time series (length %d)", h->t, h->t + K, step_limit(h));
    PLWords tab;
    if (int rc = encode_table(h, table, n_actions, &tab, "mgx_rollout_discrete")) return rc;
    const FusedOut fo{reward, done, soc_trace, status_trace, ret_acc, log};
    hipStream_t st = (hipStream_t)stream;
    static const int gpb_env = [] { const char *e = getenv("MGX_GPB_ROLLOUT"); return e ? atoi(e) : 0; }();   // experiment knob
    for_each_shard(h, st, [&](const KArgs &k, hipStream_t s) {
        const int32_t gpb = gpb_env > 0 ? gpb_env : fused_grids_per_block(h, k.g1 - k.g0);
        const unsigned blocks = (unsigned)((k.g1 - k.g0 + gpb - 1) / gpb);
        const bool rich = log != nullptr || status_trace != nullptr;
#define MGX_ROLLOUT(PS, RC) MGX_DISPATCH_F(h->flags, (rollout_kernel<F, (F & F_GRID) ? 4 : MGX_RING_ROLLOUT, PS, RC><<<blocks, BLOCK_K, 0, s>>>( \
                                                          k, tab, action_id, t_arg(h), K, fo, gpb)))
        if (per_step) { if (rich) { MGX_ROLLOUT(true, true); } else { MGX_ROLLOUT(true, false); } }
        else { if (rich) { MGX_ROLLOUT(false, true); } else { MGX_ROLLOUT(false, false); } }
#undef MGX_ROLLOUT
    });
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "rollout_kernel launch");
    advance(h, K, st);
    return MGX_OK;
}

int mgx_rollout_lists(mgx_handle *h, const int32_t *action_id, int per_step, const int32_t *lists, int32_t n_lists, int32_t list_len,
                      int32_t K, double *reward, uint8_t *done, double *soc_trace, uint32_t *status_trace, double *ret_acc,
                      double *log, mgx_stream stream)
{
    g_err[0] = 0;
    if (!h || !action_id || !lists) return fail(MGX_ERR_INVALID, "mgx_rollout_lists: NULL argument");
    if (K <= 0) return fail(MGX_ERR_INVALID, "mgx_rollout_lists: K must be positive");
    if (n_lists <= 0 || list_len <= 0 || list_len > 3 * MGX_MAX_INSTANCES)
        return fail(MGX_ERR_INVALID, "mgx_rollout_lists: need n_lists > 0 and list_len in [1, %d]", 3 * MGX_MAX_INSTANCES);
    if (!dev_counter(h) && (h->t < 0 || (int64_t)h->t + K > step_limit(h)))
        return fail(MGX_ERR_RANGE, "mgx_rollout_lists: steps [%d, %d) leave the time series (length %d)", h->t, h->t + K, step_limit(h));
    hipStream_t st = (hipStream_t)stream;
    const FusedOut fo{reward, done, soc_trace, status_trace, ret_acc, log};
    for_each_shard(h, st, [&](const KArgs &k, hipStream_t s) {
        MGX_DISPATCH_F(h->flags, (step_k_multi_kernel<F><<<multi_blocks(k.g1 - k.g0), BLOCK_MULTI, h->multi_lds, s>>>(
                                      k, nullptr, lists, n_lists, list_len, action_id, per_step, t_arg(h), K, 0, fo)));
    });
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "step_k_multi_kernel launch");
    advance(h, K, st);
    return MGX_OK;
}

// ---- fleets: several batches (one per layout) stepped by ONE call ---------------------------------------------------
int mgx_fleet_step(const mgx_fleet_item *items, int32_t n, int normalized, mgx_stream stream)
{
    g_err[0] = 0;
    if (!items || n <= 0) return fail(MGX_ERR_INVALID, "mgx_fleet_step: no items");
    hipStream_t st = (hipStream_t)stream;
    for (int32_t j = 0; j < n; j++) {                       // all checks first: a fleet step is all or nothing
        const mgx_fleet_item &it = items[j];
        if (it.struct_size != (int32_t)sizeof(mgx_fleet_item))
            return fail(MGX_ERR_INVALID, "mgx_fleet_step: item %d struct_size %d vs %zu", j, it.struct_size, sizeof(mgx_fleet_item));
        if (!it.handle) return fail(MGX_ERR_INVALID, "mgx_fleet_step: item %d has no handle", j);
        if (it.action_id && !it.table) return fail(MGX_ERR_INVALID, "mgx_fleet_step: item %d: NULL table", j);
        if (int rc = check_step_args(it.handle, it.action_id ? (const void *)it.action_id : it.actions, it.reward, it.obs, 1,
                                     "mgx_fleet_step")) return rc;
        if (it.refill_ring) {
            if (it.refill_K < 1 || it.refill_ahead < 0 || it.refill_chunks < 0 || it.refill_chunk < 0 ||
                (it.refill_chunks > 0 && (it.refill_chunk >= it.refill_chunks || it.refill_ahead < 1)))
                return fail(MGX_ERR_INVALID, "mgx_fleet_step: item %d: bad refill_K / refill_ahead / refill_chunk(s)", j);
            WindowsKPlan plan; size_t lds; int32_t ng;
            if (int rc = windows_plan(it.handle, it.refill_ahead, it.refill_K, it.refill_ring, "mgx_fleet_step", &plan, &lds, &ng)) return rc;
        }
    }
    // one launch for all batches (continuous and discrete items alike) unless an item needs a kernel of its own
    bool fusable = true;
    for (int32_t j = 0; j < n && fusable; j++) {
        const mgx_fleet_item &it = items[j];
        const mgx_handle *h = it.handle;
        fusable = !h->multi && !dev_counter(h) && h->n_shards <= 1 && !(it.obs && h->k.H > 0 && !h->k.obs_state_only);
        for (int32_t q = 0; q < j && fusable; q++) fusable = items[q].handle != it.handle;      // a batch steps once per call
    }
    bool chunk_done[64];                                    // window chunks that rode along with the step launch
    for (int32_t j = 0; j < n && j < 64; j++) chunk_done[j] = false;
    for (int32_t j = 0; j < n; j++)
        if (items[j].wait_prefetch) { if (int rc = mgx_prefetch_wait(items[j].handle, stream)) return rc; }
    if (fusable) {
        for (int32_t j = 0; j < n; j++) {                 // device copies of the batches' KArgs: uploaded when they changed
            if (int rc = sync_device_kargs(items[j].handle, st, "mgx_fleet_step: uploading the layout table")) return rc;
            if (!items[j].action_id) continue;            // discrete item: its priority-list table, too
            mgx_handle *h = items[j].handle;
            PLWords tab;
            if (int rc = encode_table(h, items[j].table, items[j].n_actions, &tab, "mgx_fleet_step")) return rc;
            if (h->table_uploaded_valid && memcmp(&tab, &h->table_uploaded, sizeof(PLWords)) == 0) continue;
            hipError_t e = hipSuccess;
            DeviceGuard on_device(h->device);
            if (!h->d_table) e = hipMalloc((void **)&h->d_table, sizeof(PLWords));
            if (e == hipSuccess) e = hipMemcpyAsync(h->d_table, &tab, sizeof(PLWords), hipMemcpyHostToDevice, st);
            if (e != hipSuccess) return hip_fail(e, "mgx_fleet_step: uploading the priority-list table");
            memcpy(&h->table_uploaded, &tab, sizeof(PLWords));
            h->table_uploaded_valid = true;
        }
        for (int32_t j0 = 0; j0 < n; j0 += MGX_FLEET_MAX) {
            FleetArgs fa;
            FleetWin fw;
            memset(&fa, 0, sizeof(fa));
            memset(&fw, 0, sizeof(fw));
            fa.n = n - j0 < MGX_FLEET_MAX ? n - j0 : MGX_FLEET_MAX;
            fa.normalized = normalized;
            int32_t blocks = 0, wblocks = 0;
            size_t lds_max = 0;
            for (int32_t q = 0; q < fa.n; q++) {
                const mgx_fleet_item &it = items[j0 + q];
                mgx_handle *h = it.handle;
                fa.k[q] = h->d_kargs;
                fa.tab[q] = it.action_id ? h->d_table : nullptr;
                fa.actions[q] = it.action_id ? (const void *)it.action_id : it.actions; fa.reward[q] = it.reward; fa.done[q] = it.done; fa.obs[q] = it.obs; fa.log[q] = it.log;
                fa.t[q] = h->t; fa.flags[q] = h->flags; fa.block0[q] = blocks;
                blocks += (int32_t)blocks_for(h->k.N);
                if (it.refill_ring && it.refill_chunks > 0 && j0 + q < 64) {        // this step's share of the next ring
                    WindowsKPlan plan; size_t lds; int32_t ng, first, count;
                    (void)windows_plan(h, it.refill_ahead, it.refill_K, it.refill_ring, "mgx_fleet_step", &plan, &lds, &ng);
                    if (lds > 64 * 1024) continue;                                   // launched on its own below
                    chunk_range(ng, it.refill_chunk, it.refill_chunks, &first, &count);
                    chunk_done[j0 + q] = true;
                    if (count <= 0) continue;
                    const int w = fw.n++;
                    plan.group0 = first;
                    fw.k[w] = h->d_kargs; fw.ring[w] = it.refill_ring; fw.plan[w] = plan;
                    fw.t[w] = h->t + 1 + it.refill_ahead;
                    fw.block0[w] = wblocks;
                    fw.kind[w] = (h->layout.has_grid ? 1 : 0) | (h->k.obs_f32 ? 2 : 0);
                    fw.nstate[w] = 4 * h->layout.has_genset + 2 * h->layout.has_battery;
                    wblocks += count;
                    if (lds > lds_max) lds_max = lds;
                }
            }
            fw.first_block = blocks;
            fleet_step_kernel<<<(unsigned)(blocks + wblocks), BLOCK, lds_max, st>>>(fa, fw);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return hip_fail(e, "fleet_step_kernel launch");
        }
        for (int32_t j = 0; j < n; j++) advance(items[j].handle, 1, st);
    }
    for (int32_t j = 0; j < n && !fusable; j++) {
        const mgx_fleet_item &it = items[j];
        int rc;
        if (it.action_id)
            rc = mgx_step_discrete(it.handle, it.action_id, it.table, it.n_actions, nullptr, it.reward, it.done, it.obs, it.log, stream);
        else
            rc = step_once(it.handle, it.actions, normalized, it.reward, it.done, it.obs, it.log, st);
        if (rc) return rc;
    }
    for (int32_t j = 0; j < n; j++) {                       // window prefetch that did not ride along with the step launch
        const mgx_fleet_item &it = items[j];
        if (!it.refill_ring || (j < 64 && chunk_done[j])) continue;
        int rc;
        if (it.refill_chunks > 0)                           // a chunk, on the caller's stream (the counter has advanced: ahead as given)
            rc = launch_windows(it.handle, it.refill_ahead, it.refill_K, it.refill_ring, st, "mgx_fleet_step", it.refill_chunk, it.refill_chunks);
        else
            rc = it.refill_ahead > 0 ? mgx_observe_windows_ahead(it.handle, it.refill_ahead, it.refill_K, it.refill_ring, stream)
                                     : mgx_observe_windows(it.handle, it.refill_K, it.refill_ring, stream);
        if (rc) return rc;
    }
    return MGX_OK;
}

int mgx_synthesize_series(const mgx_synth *a, mgx_stream stream)
{
    g_err[0] = 0;
    if (!a) return fail(MGX_ERR_INVALID, "mgx_synthesize_series: NULL argument");
    if (a->struct_size != (int32_t)sizeof(mgx_synth))
        return fail(MGX_ERR_INVALID, "mgx_synthesize_series: struct_size %d vs %zu (ABI %d)", a->struct_size, sizeof(mgx_synth), MGX_ABI_VERSION);
    if (a->n_grids <= 0 || a->n_steps <= 0 || a->n_load_profiles <= 0 || a->n_pv_profiles <= 0)
        return fail(MGX_ERR_INVALID, "mgx_synthesize_series: need n_grids, n_steps, n_load_profiles, n_pv_profiles > 0");
    if (!a->base_load || !a->base_pv || !a->load_profile || !a->pv_profile || !a->load_ratio || !a->pv_ratio || !a->load_ts || !a->pv_ts)
        return fail(MGX_ERR_INVALID, "mgx_synthesize_series: NULL load / pv argument");
    if (a->grid_ts && (!a->base_co2 || !a->co2_profile || !a->tariff || a->n_co2_profiles <= 0))
        return fail(MGX_ERR_INVALID, "mgx_synthesize_series: grid_ts requested without base_co2 / co2_profile / tariff");
    if (a->grid_ts && a->outage_per_day && (!a->weak || !a->outage_duration))
        return fail(MGX_ERR_INVALID, "mgx_synthesize_series: outage_per_day given without weak / outage_duration");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(MGX_ERR_DEVICE, "mgx_synthesize_series: no HIP device available -- this engine has no CPU path");
    synthesize_series_kernel<<<blocks_for(a->n_grids), BLOCK, 0, (hipStream_t)stream>>>(*a);
    e = hipGetLastError();
    return e == hipSuccess ? MGX_OK : hip_fail(e, "synthesize_series_kernel launch");
}

int mgx_metrics(mgx_handle *h, const double *values, int32_t M, double *sums, mgx_stream stream)
{
    g_err[0] = 0;
    if (!h || !values || !sums) return fail(MGX_ERR_INVALID, "mgx_metrics: NULL argument");
    if (M <= 0 || M > MAX_METRICS) return fail(MGX_ERR_INVALID, "mgx_metrics: M must be in [1, %d]", MAX_METRICS);
    const int64_t N = h->k.N;
    // fixed slice per block: >= 2048 grids, at most MAX_PARTIAL blocks
    int64_t per_block = 2048;
    while ((N + per_block - 1) / per_block > MAX_PARTIAL) per_block *= 2;
    const unsigned nb = (unsigned)((N + per_block - 1) / per_block);
    hipStream_t st = (hipStream_t)stream;
    colsum_stage1<<<dim3(nb, (unsigned)M), BLOCK, 0, st>>>(values, N, (int32_t)per_block, h->scratch);
    colsum_stage2<<<(unsigned)M, BLOCK, 0, st>>>(h->scratch, (int32_t)nb, sums);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MGX_OK : hip_fail(e, "colsum launch");
}

}  // extern "C"
