// mgx_abi.hip -- host side of libmgx.so: the C ABI declared in include/mgx.h (handles, argument checks, launch shapes,
// shard / prefetch streams) over the kernels of mgx_kernels.hpp and the per-grid device arithmetic of mgx_core.hpp.
// Build (pymgrid_amd/_lib.py: build()): this file + the slices of mgx_fused.hip (the K-step kernels), compiled in parallel,
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -c mgx_abi.hip ; ... -c -DMGX_FUSED_PART=p mgx_fused.hip (p = 0..4)
//   hipcc --offload-arch=gfx950 -fPIC -shared *.o -o libmgx.so
#include "mgx_kernels.hpp"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

using namespace mgx;

#define MGX_MAX_SHARDS 8

struct mgx_handle {
    KArgs k;
    mgx_layout layout;
    int32_t window_lo, window_hi;   // episode window given at create: trajectories must stay inside it
    bool multi;             // n_load != 1, n_pv != 1 or several gensets / batteries / grids: general (slow) kernels
    int32_t ring_pitch;     // rows between the blocks of an observation ring (mgx_set_ring_pitch; default N)
    size_t multi_lds;       // LDS bytes of a general-kernel workgroup (the MicrogridStep lists)
    int multi_small;        // at most MS modules of every kind per grid: the general steps run their register form (step_multi_small)
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
    int32_t launch_threads;                  // shards 1.. issued by the device's launch workers: 0 never, 1 inside mgx_step_many, 2 every single step
    hipStream_t counter_stream;              // device-counter mode: the stream of the last call that touched the counter
    KArgs *d_kargs;                          // device copy of `k` for fleet_step_kernel, refreshed when `k` changed
    KArgs k_uploaded;
    bool k_uploaded_valid;
    PLWords *d_table;                        // device copy of the priority-list table of a discrete fleet item
    PLWords table_uploaded;
    bool table_uploaded_valid;
    // mgx_env_bind: the rotating outputs / observation rings of a Gym loop, walked by mgx_env_step itself
    bool env_bound;
    int32_t env_n_slots, env_next, env_ring_K, env_ring_idx, env_ring_pos, env_n_actions;
    mgx_env_slot env_slots[MGX_ENV_MAX_SLOTS];
    char *env_rings[3];
    int32_t env_table[12 * 3 * 2];
    hipStream_t prefetch_stream;             // mgx_observe_windows_ahead: the window prefetch overlaps the steps
    bool prefetch_pooled;                    // ... on the per-device pooled stream (MGX_TUNE_PREFETCH_POOL): not destroyed with the handle
    hipEvent_t prefetch_gate, prefetch_done;
    bool prefetch_pending;
    // per-grid episode windows (mgx_reset_windows): the full series are remembered here while the handle steps over the
    // caller's window buffers
    bool windowed;
    bool rolling;                            // mgx_reset_windows_rolling: ring buffers, partial resets (mgx_reset_grids)
    bool inplace;                            // mgx_reset_episodes: rolling episodes on the series themselves (no window buffers)
    double *pm_tables;                       // ... and the profile-major copies of the base tables they read ([3][PP][pm_pitch], lazily)
    double *gm_tables;                       // ... or, for [T, N] series, their grid-major copy [N][gm_pitch][2 or 6] (load, pv, grid x 4)
    int32_t gm_pitch;
    int32_t pm_pitch;
    int32_t rolling_max_length;
    double *roll_load_w, *roll_pv_w, *roll_grid_w;
    int32_t *roll_final;
    const double *full_load_ts, *full_pv_ts, *full_grid_ts;
    mgx_columns full_c;                      // the columns as given at create (a factorised batch steps over materialised
                                             // window buffers during a per-grid-window episode: k.c.base_load is NULL then)
    int32_t full_T, full_final, full_initial, full_window_lo, full_window_hi;
};

namespace {

constexpr int MAX_PARTIAL = 1024;
constexpr int MAX_METRICS = 64;

thread_local char g_err[512] = "";

// Shard streams are a per-DEVICE pool shared by every handle of the process (created on first use, alive until exit).  The
// runtime maps HIP streams onto a handful of hardware queues: with streams of its own per handle, the second handle's pair
// landed on ONE queue and its two "concurrent" launches ran one after the other (measured: 79 instead of 64 us per round for
// the second engine of a process).  Handles that step in shards at the same time share these streams (in-order per stream).
constexpr int MGX_MAX_DEVICES = 16;
hipStream_t g_shard_streams[MGX_MAX_DEVICES][MGX_MAX_SHARDS] = {};
std::mutex g_shard_streams_lock;          // handles live on different host threads (one thread per handle): creation is guarded
// MGX_TUNE_PREFETCH_POOL = 1: one prefetch stream per device shared by every handle -- the ring refills of a fleet's buckets then run one
// after the other instead of side by side (each alone has the whole memory system; a fleet step orders them with one gate event)
hipStream_t g_prefetch_streams[MGX_MAX_DEVICES] = {};

// ---- tunables (mgx_set_tunable): process-wide launch-shape knobs, defaults = what the measurements of DESIGN.md chose -------------
// The library reads NOTHING from the environment: a consumer sees and pins every knob through the ABI.
std::atomic<int64_t> g_tune[MGX_TUNE_COUNT_];
const int64_t kTuneDefault[MGX_TUNE_COUNT_] = {
    /* MGX_TUNE_WIN_THREADS     */ 0,      // 0 = automatic (256 ahead of the counter, 1 024 where a reset waits)
    /* MGX_TUNE_WIN_GROUP       */ 0,      // 0 = automatic (16 / 32 grids per refill workgroup, windows_plan)
    /* MGX_TUNE_WIN_PAIRS       */ -1,     // -1 = automatic (pairs of adjacent grids per lane for column-major blocks)
    /* MGX_TUNE_WIN_MIN_LDS     */ -1,     // -1 = automatic (81 KB ahead of the counter: one refill workgroup per CU)
    /* MGX_TUNE_PREFETCH_POOL   */ 0,
    /* MGX_TUNE_MULTI_GENERIC   */ 0,
    /* MGX_TUNE_MULTI_SMALL_OWN */ 1,
    /* MGX_TUNE_GRID_MAJOR_COPY */ 1,
    /* MGX_TUNE_FLEET_BYVALUE   */ 1,
    /* MGX_TUNE_LAUNCH_THREADS  */ 1,      // 0 / 1 / 2: see mgx_set_launch_threads
    /* MGX_TUNE_MULTI_STATIC    */ 1,
};
struct TuneInit { TuneInit() { for (int j = 0; j < MGX_TUNE_COUNT_; j++) g_tune[j].store(kTuneDefault[j], std::memory_order_relaxed); } } g_tune_init;
inline int64_t tune(int id) { return g_tune[id].load(std::memory_order_relaxed); }

// ---- launch workers ------------------------------------------------------------------------------------------------------------
// A single-step launch cannot overlap its predecessor (the next step needs this one's state), and ONE host thread issues a launch
// every ~4.3 us (profiles/r02/exp_sharded_single_steps.txt): S shard launches per env-step from one thread made the Gym cadence S
// times slower.  With shards on (mgx_set_shards) shard j >= 1 of a single-step call is therefore issued by a resident host thread
// of its own -- one per (device, shard), shared by all handles like the shard streams -- while the caller issues shard 0: the
// two dependent launch chains then advance side by side, one chain's launch boundary under the other's memory round trip.
// Hand-over by two counters the parties spin on (a worker goes to sleep on a condition variable after SPIN_US idle microseconds);
// the call returns when every shard's launches have been ISSUED, so "asynchronous on its streams, issued when the call returns" holds.
struct LaunchWorker {
    std::atomic<uint64_t> posted{0}, done{0};
    std::atomic<int> asleep{0};
    void (*fn)(void *, int) = nullptr;        // fn(ctx, shard)
    void *ctx = nullptr;
    int shard = 0, device = 0;
    hipError_t err = hipSuccess;              // first launch error of the last job (hipGetLastError is per host thread)
    std::mutex sleep_lock, caller_lock;       // caller_lock: one job at a time (handles on different host threads share a worker)
    std::condition_variable wake;
    std::thread th;
    static constexpr int SPIN_US = 500;

    void run()
    {
        (void)hipSetDevice(device);
        uint64_t seen = 0;
        for (;;) {
            auto idle_since = std::chrono::steady_clock::now();
            unsigned spins = 0;
            while (posted.load(std::memory_order_acquire) == seen) {
                __builtin_ia32_pause();
                if ((++spins & 1023) == 0 &&
                    std::chrono::steady_clock::now() - idle_since > std::chrono::microseconds(SPIN_US)) {
                    std::unique_lock<std::mutex> lk(sleep_lock);
                    asleep.store(1, std::memory_order_seq_cst);
                    wake.wait(lk, [&] { return posted.load(std::memory_order_seq_cst) != seen; });
                    asleep.store(0, std::memory_order_seq_cst);
                }
            }
            (void)hipGetLastError();
            fn(ctx, shard);
            err = hipGetLastError();
            seen += 1;
            done.store(seen, std::memory_order_release);
        }
    }
    void post(void (*f)(void *, int), void *c, int j)
    {
        fn = f; ctx = c; shard = j;
        posted.store(posted.load(std::memory_order_relaxed) + 1, std::memory_order_seq_cst);
        if (asleep.load(std::memory_order_seq_cst)) {
            std::lock_guard<std::mutex> lk(sleep_lock);
            wake.notify_one();
        }
    }
    hipError_t wait()
    {
        const uint64_t want = posted.load(std::memory_order_relaxed);
        while (done.load(std::memory_order_acquire) != want) __builtin_ia32_pause();
        return err;
    }
};
LaunchWorker *g_workers[MGX_MAX_DEVICES][MGX_MAX_SHARDS] = {};     // never destroyed: the threads live until the process ends
thread_local hipError_t g_worker_err = hipSuccess;                  // a worker's launch error, handed to the calling thread

LaunchWorker *launch_worker(int device, int shard)
{
    if (device < 0 || device >= MGX_MAX_DEVICES || shard < 1 || shard >= MGX_MAX_SHARDS) return nullptr;
    std::lock_guard<std::mutex> guard(g_shard_streams_lock);
    LaunchWorker *&w = g_workers[device][shard];
    if (!w) {
        w = new (std::nothrow) LaunchWorker();
        if (!w) return nullptr;
        w->device = device;
        w->th = std::thread([w] { w->run(); });
        w->th.detach();
    }
    return w;
}

// the launch error of a stepping call: the calling thread's, or the first one a launch worker met
inline hipError_t launch_error()
{
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = g_worker_err;
    g_worker_err = hipSuccess;
    return e;
}

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
inline int32_t step_limit(const mgx_handle *h)
{
    if (h->rolling) return INT32_MAX / 2;            // rolling windows: every grid ends (and restarts) on its own
    return h->windowed ? h->layout.final_step : h->k.T;
}
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

// The same for the single-step calls: with shards on, shard j >= 1 is issued by launch worker j of the device (see LaunchWorker)
// while the calling thread issues shard 0; returns when all of them are issued.  fn must be safe to run on several threads at once.
template <class Fn>
inline void for_each_shard_threaded(const mgx_handle *h, hipStream_t user, Fn fn, int32_t min_mode = 2)
{
    if (h->n_shards <= 1 || h->launch_threads < min_mode) { for_each_shard(h, user, fn); return; }
    struct Job { const mgx_handle *h; Fn *fn; };
    Job job{h, &fn};
    auto thunk = [](void *ctx, int j) {
        const Job *jb = (const Job *)ctx;
        KArgs k = jb->h->k;
        k.g0 = jb->h->shard_lo[j]; k.g1 = jb->h->shard_lo[j + 1];
        (*jb->fn)(k, jb->h->shard_stream[j]);
    };
    LaunchWorker *ws[MGX_MAX_SHARDS] = {};
    for (int j = 1; j < h->n_shards; j++) {
        if (h->shard_lo[j + 1] <= h->shard_lo[j]) continue;
        ws[j] = launch_worker(h->device, j);
        if (!ws[j]) { thunk(&job, j); continue; }          // (no worker: this thread issues the shard)
        ws[j]->caller_lock.lock();
        ws[j]->post(thunk, &job, j);
    }
    if (h->shard_lo[1] > h->shard_lo[0]) thunk(&job, 0);
    for (int j = 1; j < h->n_shards; j++) {
        if (!ws[j]) continue;
        const hipError_t e = ws[j]->wait();
        ws[j]->caller_lock.unlock();
        if (e != hipSuccess && g_worker_err == hipSuccess) g_worker_err = e;
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
    for (int32_t g = BLOCK_K; g >= BLOCK_K * 3 / 4; g -= 16) {
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

// the opt-in to more than 64 KB of dynamic LDS is a driver call: made once per kernel and device, not per launch
template <int F, typename OT>
static void launch_windows_kernel(const KArgs &k, const WindowsKPlan &plan, int32_t t, void *ring, unsigned blocks, size_t lds, hipStream_t st)
{
    static bool opted_in[MGX_MAX_DEVICES] = {};
    if (lds > 64 * 1024) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev < 0 || dev >= MGX_MAX_DEVICES || !opted_in[dev]) {
            (void)hipFuncSetAttribute((const void *)obs_windows_k_kernel<F, OT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (dev >= 0 && dev < MGX_MAX_DEVICES) opted_in[dev] = true;
        }
    }
    // Threads per refill workgroup (phase 1; phase 2 is always the first 256).  A refill that runs ALONE (the one a reset waits
    // for: with_state) takes all 1 024 -- 12 % faster (profiles/r05/exp_refill_threads.txt); one written AHEAD, beside the step
    // launches, stays at 256: the faster it runs the harder it leans on the memory system and the more the steps beside it pay
    // (config-5 fleet step 16.5 -> 17.2 us with 1 024).  mgx_set_tunable(MGX_TUNE_WIN_THREADS, 256 / 512 / 1024) pins it for both.
    const unsigned forced = (unsigned)tune(MGX_TUNE_WIN_THREADS);
    const unsigned threads = forced ? forced : (plan.with_state ? (unsigned)OBS_P1_THREADS : (unsigned)OBS_K_THREADS);
    obs_windows_k_kernel<F, OT><<<blocks, threads, lds, st>>>(k, plan, t, (OT *)ring);
}

template <int F, typename OT>
static void launch_windows_multi_kernel(const KArgs &k, const WindowsKPlan &plan, int32_t t, void *ring, unsigned blocks, size_t lds,
                                        hipStream_t st)
{
    static bool opted_in[MGX_MAX_DEVICES] = {};
    if (lds > 64 * 1024) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev < 0 || dev >= MGX_MAX_DEVICES || !opted_in[dev]) {
            (void)hipFuncSetAttribute((const void *)obs_windows_k_multi_kernel<F, OT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (dev >= 0 && dev < MGX_MAX_DEVICES) opted_in[dev] = true;
        }
    }
    // 1 024 threads for phase 1 here, ahead or not: on the general path the refill -- not the chain of step launches -- sets the
    // pace of a Gym step with rows (1.15 ms per ring of 32 blocks against 0.3 ms of steps).  MGX_TUNE_WIN_THREADS overrides.
    const unsigned forced = (unsigned)tune(MGX_TUNE_WIN_THREADS);
    obs_windows_k_multi_kernel<F, OT><<<blocks, forced ? forced : (unsigned)OBS_P1_THREADS, lds, st>>>(k, plan, t, (OT *)ring);
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
    const int32_t D = h->k.obs_dim;
    WindowPlan plan;
    plan.grid_col_base = h->k.col_grid;
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
int mgx_abi_minor(void) { return MGX_ABI_MINOR; }

int mgx_set_tunable(int32_t id, int64_t value)
{
    g_err[0] = 0;
    if (id < 0 || id >= MGX_TUNE_COUNT_) return fail(MGX_ERR_INVALID, "mgx_set_tunable: unknown tunable %d", id);
    bool ok = true;
    switch (id) {
        case MGX_TUNE_WIN_THREADS: ok = value == 0 || value == 256 || value == 512 || value == 1024; break;
        case MGX_TUNE_WIN_GROUP: ok = value >= 0 && value <= 64; break;
        case MGX_TUNE_WIN_PAIRS: ok = value >= -1 && value <= 1; break;
        case MGX_TUNE_WIN_MIN_LDS: ok = value >= -1 && value <= 160 * 1024; break;
        case MGX_TUNE_LAUNCH_THREADS: ok = value >= 0 && value <= 2; break;
        default: ok = value == 0 || value == 1; break;
    }
    if (!ok) return fail(MGX_ERR_INVALID, "mgx_set_tunable: value %lld is not valid for tunable %d", (long long)value, id);
    g_tune[id].store(value, std::memory_order_relaxed);
    return MGX_OK;
}

int mgx_get_tunable(int32_t id, int64_t *value, int64_t *default_value)
{
    g_err[0] = 0;
    if (id < 0 || id >= MGX_TUNE_COUNT_) return fail(MGX_ERR_INVALID, "mgx_get_tunable: unknown tunable %d", id);
    if (value) *value = tune(id);
    if (default_value) *default_value = kTuneDefault[id];
    return MGX_OK;
}

int mgx_set_launch_threads(mgx_handle *h, int32_t enable)
{
    g_err[0] = 0;
    if (!h) return fail(MGX_ERR_INVALID, "mgx_set_launch_threads: NULL handle");
    if (enable < 0 || enable > 2) return fail(MGX_ERR_INVALID, "mgx_set_launch_threads: mode must be 0, 1 or 2");
    h->launch_threads = enable;
    return MGX_OK;
}

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
    if (L->flat_order != MGX_FLAT_MODULE && L->flat_order != MGX_FLAT_GYM)
        return fail(MGX_ERR_INVALID, "mgx_create: unknown flat_order %d", L->flat_order);
    if (L->flat_order != MGX_FLAT_MODULE && (L->n_load != 1 || L->n_pv != 1 || n_genset > 1 || n_battery > 1 || n_grid > 1))
        return fail(MGX_ERR_UNSUPPORTED, "mgx_create: flat_order is offered for one module of every kind per grid (permute the rows "
                                         "of a multi-module layout on your side)");
    const int32_t final_step = L->final_step <= 0 ? L->n_steps : L->final_step;   // base_timeseries_module.py:321-326
    if (final_step > L->n_steps) return fail(MGX_ERR_INVALID, "mgx_create: final_step %d > n_steps %d", final_step, L->n_steps);
    if (L->initial_step < 0 || L->initial_step >= final_step)
        return fail(MGX_ERR_INVALID, "mgx_create: final_step value must be greater than initial_step");
#define NEED(cond, ptr) if ((cond) && !(C->ptr)) return fail(MGX_ERR_INVALID, "mgx_create: column " #ptr " is NULL")
    const bool fact = C->base_load != nullptr;           // factorised series: the [T, N] arrays are optional
    if (fact && (L->n_load != 1 || L->n_pv != 1 || n_genset > 1 || n_battery > 1 || n_grid > 1))
        return fail(MGX_ERR_UNSUPPORTED, "mgx_create: factorised series need exactly one module of every kind per grid (the "
                                         "general kernels read materialised series)");
    NEED(L->n_load > 0 && !fact, load_ts); NEED(L->n_pv > 0 && !fact, pv_ts); NEED(true, loss_load_cost); NEED(true, overgeneration_cost);
    NEED(fact, base_pv); NEED(fact, load_profile); NEED(fact, pv_profile); NEED(fact, load_ratio); NEED(fact, pv_ratio);
    NEED(fact && L->has_grid, base_co2); NEED(fact && L->has_grid, co2_profile); NEED(fact && L->has_grid, tariff);
    NEED(L->has_battery, bat_min_capacity); NEED(L->has_battery, bat_max_capacity); NEED(L->has_battery, bat_max_charge);
    NEED(L->has_battery, bat_max_discharge); NEED(L->has_battery, bat_efficiency); NEED(L->has_battery, bat_cost_cycle);
    NEED(L->has_battery, charge); NEED(L->has_battery, soc);
    NEED(L->has_genset, gen_running_min); NEED(L->has_genset, gen_running_max); NEED(L->has_genset, gen_cost);
    NEED(L->has_genset, gen_co2_per_unit); NEED(L->has_genset, gen_cost_per_unit_co2); NEED(L->has_genset, gen_times);
    NEED(L->has_genset, gen_status);
    NEED(L->has_grid, grid_max_import); NEED(L->has_grid, grid_max_export); NEED(L->has_grid, grid_cost_per_unit_co2);
    NEED(L->has_grid && !fact, grid_ts);
#undef NEED
    if (C->uniform_mask && (L->n_load != 1 || L->n_pv != 1 || n_genset > 1 || n_battery > 1 || n_grid > 1))
        return fail(MGX_ERR_UNSUPPORTED, "mgx_create: uniform_mask is offered for one module of every kind per grid");
    if (C->uniform_mask >> (MGX_U_OVERGENERATION_COST + 1))
        return fail(MGX_ERR_INVALID, "mgx_create: uniform_mask 0x%x has bits beyond MGX_U_OVERGENERATION_COST", C->uniform_mask);

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
    {   // column bases of the module blocks inside a flat observation row
        const int32_t g4 = 4 * n_genset, b2 = 2 * n_battery, r4 = 4 * w * n_grid;
        if (L->flat_order == MGX_FLAT_GYM) {               // battery, genset, grid, load, pv
            h->k.col_bat = 0; h->k.col_gen = b2; h->k.col_grid = b2 + g4; h->k.col_load = b2 + g4 + r4; h->k.col_pv = h->k.col_load + w * L->n_load;
        } else {                                            // load, pv, genset, battery, grid
            h->k.col_load = 0; h->k.col_pv = w * L->n_load; h->k.col_gen = w * (L->n_load + L->n_pv); h->k.col_bat = h->k.col_gen + g4;
            h->k.col_grid = h->k.col_bat + b2;
        }
    }
    h->multi_lds = 2 * (size_t)multi_list_capacity(L->n_load, L->n_pv, n_genset, n_battery, n_grid) * BLOCK_MULTI * sizeof(double);
    {   // MGX_TUNE_MULTI_GENERIC = 1 (tests): every general layout on the run-time-count form, whatever its size
        const bool generic = tune(MGX_TUNE_MULTI_GENERIC) != 0;
        h->multi_small = (!generic && multi_is_small(L->n_load, L->n_pv, n_genset, n_battery, n_grid)) ? 1 : 0;
    }
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
    h->ring_pitch = L->n_grids;
    h->t = L->initial_step;
    h->k.shaper = MGX_SHAPER_NONE;
    h->k.noise_seed = 0; h->k.noise_increase = 0; h->k.obs_f32 = 0; h->k.obs_state_only = 0; h->k.act_f32 = 0;
    h->k.done_bits = 0; h->k.obs_colpitch = 0;
    h->full_c = *C;
    h->window_lo = L->initial_step; h->window_hi = final_step;
    h->k.g0 = 0; h->k.g1 = L->n_grids; h->k.grid_final = nullptr;
    h->n_shards = 1; h->shard_lo[0] = 0; h->shard_lo[1] = L->n_grids;
    for (int j = 0; j < MGX_MAX_SHARDS; j++) { h->shard_stream[j] = nullptr; h->shard_event[j] = nullptr; }
    h->fork_event = nullptr; h->counter_stream = nullptr;
    h->launch_threads = (int32_t)tune(MGX_TUNE_LAUNCH_THREADS);
    h->d_kargs = nullptr; h->k_uploaded_valid = false;
    h->d_table = nullptr; h->table_uploaded_valid = false;
    h->env_bound = false; h->env_n_slots = 0; h->env_next = 0; h->env_ring_K = 0; h->env_ring_idx = 0; h->env_ring_pos = 0; h->env_n_actions = 0;
    h->prefetch_stream = nullptr; h->prefetch_gate = nullptr; h->prefetch_done = nullptr; h->prefetch_pending = false;
    h->prefetch_pooled = false;
    h->windowed = false; h->rolling = false;
    h->k.row_mask = -1;
    h->k.ep_off = nullptr; h->k.ep_final = nullptr; h->k.ar_mode = 0; h->k.ar_fixed_length = 0; h->k.ar_lo = 0; h->k.ar_hi = 0;
    h->k.ar_max_length = 0; h->k.ar_seed = 0; h->k.ar_start_io = nullptr; h->k.ar_length_io = nullptr; h->k.ar_t0_io = nullptr;
    h->k.final_obs = nullptr; h->k.pm_pitch = 0;
    h->inplace = false; h->pm_tables = nullptr; h->pm_pitch = 0; h->gm_tables = nullptr; h->gm_pitch = 0;
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
        if (h->shard_stream[j]) (void)hipStreamSynchronize(h->shard_stream[j]);      // (pooled: not destroyed with the handle)
        if (h->shard_event[j]) (void)hipEventDestroy(h->shard_event[j]);
    }
    if (h->fork_event) (void)hipEventDestroy(h->fork_event);
    if (h->prefetch_stream) { (void)hipStreamSynchronize(h->prefetch_stream); if (!h->prefetch_pooled) (void)hipStreamDestroy(h->prefetch_stream); }
    if (h->pm_tables) (void)hipFree(h->pm_tables);
    if (h->gm_tables) (void)hipFree(h->gm_tables);
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
    if (enable && (h->windowed || h->rolling))
        return fail(MGX_ERR_UNSUPPORTED, "mgx_use_device_counter: not offered during a per-grid-window episode (restarts, ring "
                                         "patches and episode ends are placed by the host's counter)");
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
    if (mode != MGX_OBS_ROWS_FULL && mode != MGX_OBS_ROWS_STATE_ONLY && mode != MGX_OBS_ROWS_STATE_COMPACT)
        return fail(MGX_ERR_INVALID, "mgx_set_obs_mode: unknown mode %d", mode);
    if (mode == MGX_OBS_ROWS_STATE_COMPACT && h->multi)
        return fail(MGX_ERR_UNSUPPORTED, "mgx_set_obs_mode: compact state rows need exactly one module of every kind per grid");
    h->k.obs_state_only = mode == MGX_OBS_ROWS_STATE_ONLY ? 1 : (mode == MGX_OBS_ROWS_STATE_COMPACT ? 2 : 0);
    return MGX_OK;
}

int mgx_set_done_format(mgx_handle *h, int32_t format)
{
    g_err[0] = 0;
    if (!h) return fail(MGX_ERR_INVALID, "mgx_set_done_format: NULL handle");
    if (format != MGX_DONE_U8 && format != MGX_DONE_BITS)
        return fail(MGX_ERR_INVALID, "mgx_set_done_format: unknown format %d", format);
    if (format == MGX_DONE_BITS && h->multi)
        return fail(MGX_ERR_UNSUPPORTED, "mgx_set_done_format: bit sets need exactly one module of every kind per grid");
    h->k.done_bits = format == MGX_DONE_BITS;
    return MGX_OK;
}

// shape of a window prefetch: LDS image plan + bytes; n_groups = 16-grid groups of the batch
static int windows_plan(const mgx_handle *h, int32_t ahead, int32_t K, const void *ring, const char *who, WindowsKPlan *plan,
                        size_t *lds_out, int32_t *n_groups)
{
    if (!h || !ring) return fail(MGX_ERR_INVALID, "%s: NULL argument", who);
    if (K < 1 || K > 4096) return fail(MGX_ERR_INVALID, "%s: K = %d outside [1, 4096]", who, K);
    if (ahead < 0) return fail(MGX_ERR_INVALID, "%s: ahead = %d is negative", who, ahead);
    if (h->multi && (h->rolling || h->inplace || factorised(h->k.c)))
        return fail(MGX_ERR_UNSUPPORTED, "%s: with several modules of a kind per grid the window prefetch is offered for lock-step "
                                         "counters (mgx_reset, mgx_reset_windows) over [T, n, N] series", who);
    if (h->k.c.load_noise_std || h->k.c.pv_noise_std || h->k.c.grid_noise_std)
        return fail(MGX_ERR_UNSUPPORTED, "%s: forecast noise depends on (step, horizon index), windows cannot be shared", who);
    if (h->k.obs_state_only == 2)
        return fail(MGX_ERR_UNSUPPORTED, "%s: the handle writes compact state rows (MGX_OBS_ROWS_STATE_COMPACT): the windows are "
                                         "views of mgx_normalise_series' output, there are no rings to fill", who);
    if (int rc = need_obs_bounds(h, who)) return rc;
    if (!dev_counter(h) && ahead == 0 && !h->inplace && h->t > h->k.T)       // (in place the counter never ends: per-grid rows)
        return fail(MGX_ERR_RANGE, "%s: step %d is outside the time series (length %d)", who, h->t, h->k.T);
    // series components and state columns per grid (general path: per module instance, module_container.py:355-413)
    const int32_t R = K + h->k.H, ncomp = h->multi ? h->k.n_load + h->k.n_pv + 4 * h->k.n_grid : 2 + 4 * h->layout.has_grid;
    const int32_t nstate = h->multi ? 4 * h->k.n_genset + 2 * h->k.n_battery : 6;
    plan->grid_col_base = h->k.col_grid;
    plan->K = K;
    plan->rp = R;
    plan->bp = (ncomp * (R + K) + (ahead == 0 ? nstate : 1) * K) | 1;      // ahead of the counter: ONE strip of zeros for all state columns
    const int group_env = (int)tune(MGX_TUNE_WIN_GROUP);
    // Float rows keep a FLOAT image (windows_body): half the LDS per grid, and a refill ahead of the counter has ONE strip of zeros
    // for all state columns.  Grids per workgroup (halved below while the image does not fit): 32 for float rows -- 32 x 4 bytes are
    // what a whole 128-byte line of a column-major block needs -- and for column-major blocks of doubles on the general path (two
    // lines per run; 144 KB at K = 32), 16 for everything else in doubles.  us per 100 000-grid Gym step with rows, 16 -> 32 grids
    // (profiles/r05/exp_refill_group32.txt, exp_fleet_group_ab.txt): general path 40.3 -> 37.9 (f64 columns), 32.9 -> 22.7 (f32
    // columns); single env f64 columns 33.0 -> 31.4 but the config-5 FLEET 24.5 -> 25.2 (a refill that is faster alone costs the
    // step launches beside it more: hence 16); f32 columns 23.8 -> 17.9, fleet 22.0 -> 15.5; row-major f64 32.5 -> 34.5.
    const bool float_image = h->k.obs_f32;
    plan->group = (group_env == 8 || group_env == 4 || group_env == 16 || group_env == 32 || group_env == 64) ? group_env
                  : ((float_image || (h->multi && h->k.obs_colpitch)) ? 32 : 16);
    // Column-major blocks: a lane stores a PAIR of adjacent grids per instruction (16-byte stores of doubles, 8-byte stores of floats:
    // half the store instructions for the same lines).  us per 100 000-grid Gym step with rows, off -> on
    // (profiles/r05/exp_refill_col_pairs_matrix.txt): single env f64 31.3 -> 28.8, general path f64 37.9 -> 33.6, config-5 fleet f64
    // 23.1 -> 21.7, f32 15.5 -> 14.9; float rows of single envs unchanged within the spread.  MGX_TUNE_WIN_PAIRS = 0 / 1 overrides.
    const int pairs_env = (int)tune(MGX_TUNE_WIN_PAIRS);
    plan->pairs = pairs_env >= 0 ? pairs_env : (h->k.obs_colpitch ? 1 : 0);
    plan->with_state = ahead == 0;
    plan->group0 = 0;
    plan->pitch = h->ring_pitch;
    const size_t img_esz = float_image ? sizeof(float) : sizeof(double);
    auto lds_of = [&](int32_t g) { return (size_t)g * plan->bp * img_esz + (size_t)h->k.obs_dim * sizeof(uint32_t); };
    while (plan->group > 1 && lds_of(plan->group) > 160 * 1024) plan->group /= 2;
    size_t lds = (lds_of(plan->group) + 7) & ~(size_t)7;
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
    if (ahead > 0 && n_chunks <= 1) {                 // a whole ring written ahead of the counter, beside the step launches
        // Refill workgroups per CU.  A refill runs BESIDE the step launches of the rows it is not needed for yet, and how hard it
        // leans on the memory system decides what those steps cost: with two refill workgroups per CU (K = 16: 55 KB of LDS each)
        // a 100 000-grid fleet step took 60 us while the refills ran, with one per CU (K = 32: 92 KB) 25 us -- at the same
        // 5.3 TB/s of row writes (profiles/r04/fleet_timeline_K16.txt, _K32.txt).  Asking for more than half of the 160 KB keeps
        // it at one workgroup per CU whatever K is.  MGX_TUNE_WIN_MIN_LDS overrides (bytes; 0 = exactly what the image needs).
        const long min_lds_env = (long)tune(MGX_TUNE_WIN_MIN_LDS);
        const size_t min_lds = min_lds_env >= 0 ? (size_t)min_lds_env : (size_t)(ahead > 0 ? 81 * 1024 : 0);
        if (lds < min_lds && min_lds <= 160 * 1024) lds = min_lds;
    }

    const unsigned blocks = (unsigned)count;
    const int32_t t = t_arg(h) + ahead;
    if (h->multi) {                                   // general path: run-time component counts (obs_windows_k_multi_kernel)
        if (h->k.obs_f32) {
            MGX_DISPATCH_F(h->flags, (launch_windows_multi_kernel<F, float>(h->k, plan, t, ring, blocks, lds, st)));
        } else {
            MGX_DISPATCH_F(h->flags, (launch_windows_multi_kernel<F, double>(h->k, plan, t, ring, blocks, lds, st)));
        }
        hipError_t em = hipGetLastError();
        return em == hipSuccess ? MGX_OK : hip_fail(em, "obs_windows_k_multi_kernel launch");
    }
    if (h->k.obs_f32) {
        MGX_DISPATCH_F(h->flags, (launch_windows_kernel<F, float>(h->k, plan, t, ring, blocks, lds, st)));
    } else {
        MGX_DISPATCH_F(h->flags, (launch_windows_kernel<F, double>(h->k, plan, t, ring, blocks, lds, st)));
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MGX_OK : hip_fail(e, "obs_windows_k_kernel launch");
}

int mgx_patch_windows(mgx_handle *h, const uint8_t *mask, int32_t K, void *ring, int32_t first_block, int32_t ahead,
                      uint8_t *restarted, mgx_stream stream)
{
    g_err[0] = 0;
    if (!h || !mask || !ring) return fail(MGX_ERR_INVALID, "mgx_patch_windows: NULL argument");
    if (K < 1 || first_block < 0 || first_block > K) return fail(MGX_ERR_INVALID, "mgx_patch_windows: first_block %d outside [0, K = %d]", first_block, K);
    if (ahead < 0) return fail(MGX_ERR_INVALID, "mgx_patch_windows: ahead = %d is negative", ahead);
    if (h->multi) return fail(MGX_ERR_UNSUPPORTED, "mgx_patch_windows: needs exactly one module of every kind per grid");
    if (h->k.obs_colpitch) return fail(MGX_ERR_UNSUPPORTED, "mgx_patch_windows: not offered with column-major ring blocks");
    if (dev_counter(h)) return fail(MGX_ERR_UNSUPPORTED, "mgx_patch_windows: not offered in device-counter mode");
    if (int rc = need_obs_bounds(h, "mgx_patch_windows")) return rc;
    if (first_block == K) return MGX_OK;
    const unsigned blocks = (unsigned)((h->k.N + 63) / 64);
    hipStream_t st = (hipStream_t)stream;
    const int32_t rows = (K - first_block) + h->k.H;
    if (rows > PATCH_MAX_ROWS) return fail(MGX_ERR_UNSUPPORTED, "mgx_patch_windows: (K - first_block) + horizon = %d rows exceed %d", rows, PATCH_MAX_ROWS);
    const size_t lds = 2 * (size_t)(h->layout.has_grid ? 6 : 2) * rows * sizeof(double);
    if (h->layout.has_grid) {
        if (h->k.obs_f32) patch_windows_kernel<true, float><<<blocks, 64, lds, st>>>(h->k, mask, h->t + ahead, K, first_block, h->ring_pitch, (float *)ring, restarted);
        else patch_windows_kernel<true, double><<<blocks, 64, lds, st>>>(h->k, mask, h->t + ahead, K, first_block, h->ring_pitch, (double *)ring, restarted);
    } else {
        if (h->k.obs_f32) patch_windows_kernel<false, float><<<blocks, 64, lds, st>>>(h->k, mask, h->t + ahead, K, first_block, h->ring_pitch, (float *)ring, restarted);
        else patch_windows_kernel<false, double><<<blocks, 64, lds, st>>>(h->k, mask, h->t + ahead, K, first_block, h->ring_pitch, (double *)ring, restarted);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MGX_OK : hip_fail(e, "patch_windows_kernel launch");
}

int mgx_set_ring_pitch(mgx_handle *h, int32_t rows)
{
    g_err[0] = 0;
    if (!h) return fail(MGX_ERR_INVALID, "mgx_set_ring_pitch: NULL handle");
    if (rows < h->k.N) return fail(MGX_ERR_INVALID, "mgx_set_ring_pitch: %d rows per block, the batch has %d grids", rows, h->k.N);
    if (h->k.obs_colpitch && rows % 32) return fail(MGX_ERR_INVALID, "mgx_set_ring_pitch: column-major blocks need a pitch that is a multiple of 32");
    h->ring_pitch = rows;
    if (h->k.obs_colpitch) h->k.obs_colpitch = rows;
    return MGX_OK;
}

int mgx_set_ring_layout(mgx_handle *h, int32_t layout)
{
    g_err[0] = 0;
    if (!h) return fail(MGX_ERR_INVALID, "mgx_set_ring_layout: NULL handle");
    if (layout != MGX_RING_ROWS && layout != MGX_RING_COLUMNS) return fail(MGX_ERR_INVALID, "mgx_set_ring_layout: unknown layout %d", layout);
    if (layout == MGX_RING_COLUMNS) {
        if (h->windowed || h->rolling || h->inplace)
            return fail(MGX_ERR_UNSUPPORTED, "mgx_set_ring_layout: column-major blocks are offered for lock-step episodes (restarts patch row-major rings)");
        // 32 grids = one 128-byte line of a float column (16 of a double column): every (block, column) run of a refill workgroup is
        // then whole lines in either observation format
        if (h->ring_pitch % 32) return fail(MGX_ERR_INVALID, "mgx_set_ring_layout: the ring pitch (%d) must be a multiple of 32 first (mgx_set_ring_pitch)", h->ring_pitch);
    }
    h->k.obs_colpitch = layout == MGX_RING_COLUMNS ? h->ring_pitch : 0;
    return MGX_OK;
}

int mgx_observe_windows(mgx_handle *h, int32_t K, void *ring, mgx_stream stream)
{
    g_err[0] = 0;
    // a prefetch still in flight may be writing this very ring (a reset in the middle of an episode: the ring being refilled
    // now can be the one the last mgx_observe_windows_ahead targets): its stale rows must not land on top of the new ones
    if (h && h->prefetch_pending) { if (int rc = mgx_prefetch_wait(h, stream)) return rc; }
    return launch_windows(h, 0, K, ring, (hipStream_t)stream, "mgx_observe_windows");
}

// the handle's prefetch stream and events, created on first use
static int ensure_prefetch_stream(mgx_handle *h, const char *who)
{
    if (h->prefetch_stream) return MGX_OK;
    hipError_t e = hipSuccess;
    const bool pool = tune(MGX_TUNE_PREFETCH_POOL) != 0;
    if (pool && h->device >= 0 && h->device < MGX_MAX_DEVICES) {
        std::lock_guard<std::mutex> guard(g_shard_streams_lock);
        hipStream_t &pooled = g_prefetch_streams[h->device];
        if (!pooled) e = hipStreamCreateWithFlags(&pooled, hipStreamNonBlocking);
        h->prefetch_stream = pooled;
        h->prefetch_pooled = true;
    } else {
        // (a low- or high-priority prefetch stream is slower: 31.4 / 33.5 vs 29.6 us per config-5 fleet step)
        // (a CU-masked prefetch stream -- hipExtStreamCreateWithCUMask, 25..75 % of every XCD's CUs -- is 2-3x slower:
        //  profiles/r03/exp_fleet_cu_mask_reverted.txt)
        e = hipStreamCreateWithFlags(&h->prefetch_stream, hipStreamNonBlocking);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->prefetch_gate, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->prefetch_done, hipEventDisableTiming);
    return e == hipSuccess ? MGX_OK : hip_fail(e, who);
}

// gate != nullptr: an event already recorded on the caller's stream (a fleet step records ONE for all of its buckets)
static int observe_windows_ahead(mgx_handle *h, int32_t ahead, int32_t K, void *ring, hipStream_t stream, hipEvent_t gate,
                                 hipStream_t *gated_stream)
{
    if (!h) return fail(MGX_ERR_INVALID, "mgx_observe_windows_ahead: NULL handle");
    if (ahead < 1) return fail(MGX_ERR_INVALID, "mgx_observe_windows_ahead: ahead must be >= 1 (mgx_observe_windows is the ahead = 0 form)");
    if (dev_counter(h)) return fail(MGX_ERR_UNSUPPORTED, "mgx_observe_windows_ahead: not offered in device-counter mode (the prefetch "
                                                         "stream would race with the kernels that move the counter)");
    hipError_t e = hipSuccess;
    DeviceGuard on_device(h->device);
    if (int rc = ensure_prefetch_stream(h, "mgx_observe_windows_ahead: creating the prefetch stream")) return rc;
    // readers of the ring's previous contents were queued on `stream`: the prefetch starts behind them
    if (!gate) {
        e = hipEventRecord(h->prefetch_gate, stream);
        gate = h->prefetch_gate;
    }
    if (e == hipSuccess && !(gated_stream && *gated_stream == h->prefetch_stream))     // (a pooled stream is gated once per fleet step)
        e = hipStreamWaitEvent(h->prefetch_stream, gate, 0);
    if (gated_stream) *gated_stream = h->prefetch_stream;
    if (e != hipSuccess) return hip_fail(e, "mgx_observe_windows_ahead: ordering behind the caller's stream");
    if (int rc = launch_windows(h, ahead, K, ring, h->prefetch_stream, "mgx_observe_windows_ahead")) return rc;
    e = hipEventRecord(h->prefetch_done, h->prefetch_stream);
    h->prefetch_pending = true;
    return e == hipSuccess ? MGX_OK : hip_fail(e, "mgx_observe_windows_ahead: recording the completion event");
}

int mgx_observe_windows_ahead(mgx_handle *h, int32_t ahead, int32_t K, void *ring, mgx_stream stream)
{
    g_err[0] = 0;
    return observe_windows_ahead(h, ahead, K, ring, (hipStream_t)stream, nullptr, nullptr);
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

// in-place episode state off: offsets, restart switches, final rows, and the base tables back to the caller's [T, PP] arrays
static void leave_inplace(mgx_handle *h)
{
    h->inplace = false;
    h->k.ep_off = nullptr; h->k.ep_final = nullptr; h->k.ar_mode = 0; h->k.final_obs = nullptr;
    h->k.ar_start_io = nullptr; h->k.ar_length_io = nullptr; h->k.ar_t0_io = nullptr;
    if (h->k.pm_pitch) {
        if (factorised(h->full_c)) {
            h->k.c.base_load = h->full_c.base_load; h->k.c.base_pv = h->full_c.base_pv; h->k.c.base_co2 = h->full_c.base_co2;
        } else {                                          // the caller's [T, N] arrays instead of the grid-major copies
            h->k.c.load_ts = h->full_load_ts; h->k.c.pv_ts = h->full_pv_ts; h->k.c.grid_ts = h->full_grid_ts;
        }
        h->k.pm_pitch = 0;
    }
}

// back to the full series after a per-grid-window episode
static void leave_windows(mgx_handle *h)
{
    if (!h->windowed) return;
    h->k.c.load_ts = h->full_load_ts; h->k.c.pv_ts = h->full_pv_ts; h->k.c.grid_ts = h->full_grid_ts;
    h->k.c.base_load = h->full_c.base_load;              // a factorised batch is factorised again
    h->k.T = h->full_T; h->k.final_step = h->full_final; h->k.grid_final = nullptr;
    h->layout.n_steps = h->full_T; h->layout.final_step = h->full_final; h->layout.initial_step = h->full_initial;
    h->window_lo = h->full_window_lo; h->window_hi = h->full_window_hi;
    h->windowed = false; h->rolling = false;
    h->k.row_mask = -1;
    leave_inplace(h);
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
    if (h->multi && factorised(h->windowed ? h->full_c : h->k.c))
        return fail(MGX_ERR_UNSUPPORTED, "mgx_reset_windows: several modules of a kind per grid need [T, n, N] series arrays");
    if (h->k.obs_colpitch) return fail(MGX_ERR_UNSUPPORTED, "mgx_reset_windows: not offered with column-major ring blocks (mgx_set_ring_layout)");
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
    g.mask = nullptr; g.row0 = 0; g.row_mask = -1;
    g.fc = h->full_c; g.has_grid = h->layout.has_grid;
    g.draw = 0; g.fixed_length = 0; g.seed = 0; g.start_io = nullptr; g.length_io = nullptr; g.t0_io = nullptr; g.ep_off = nullptr;
    if (h->multi)                                        // [T, n, N] series, window buffers [rows, n, N] ([rows, n_grid, 4, N])
        gather_windows_multi_kernel<<<blocks_for(h->k.N), BLOCK, 0, st>>>(g, h->k.n_load, h->k.n_pv, h->layout.has_grid ? h->k.n_grid : 0);
    else
        gather_windows_kernel<<<blocks_for(h->k.N), BLOCK, 0, st>>>(g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "gather_windows_kernel launch");
    h->rolling = false; h->k.row_mask = -1;
    leave_inplace(h);
    h->k.c.base_load = nullptr;                          // the episode steps over the (materialised) window buffers
    h->k.c.load_ts = load_w; h->k.c.pv_ts = pv_w; if (h->layout.has_grid) h->k.c.grid_ts = grid_w;
    h->k.T = rows; h->k.final_step = max_length; h->k.grid_final = length ? final_rel : nullptr;
    h->layout.n_steps = rows; h->layout.final_step = max_length; h->layout.initial_step = 0;
    h->window_lo = 0; h->window_hi = max_length;
    h->windowed = true;
    h->t = 0;
    return obs ? mgx_observe(h, obs, stream) : MGX_OK;
}

// ---- rolling per-grid windows: every grid restarts on its own ----------------------------------------------------------
static void rolling_gather_args(const mgx_handle *h, GatherArgs *g)
{
    g->load_ts = h->full_load_ts; g->pv_ts = h->full_pv_ts; g->grid_ts = h->layout.has_grid ? h->full_grid_ts : nullptr;
    g->load_lo = h->k.c.load_lo; g->load_hi = h->k.c.load_hi; g->pv_lo = h->k.c.pv_lo; g->pv_hi = h->k.c.pv_hi;
    g->grid_lo = h->k.c.grid_lo; g->grid_hi = h->k.c.grid_hi;
    g->load_w = h->roll_load_w; g->pv_w = h->roll_pv_w; g->grid_w = h->roll_grid_w;
    g->final_rel = h->roll_final;
    g->N = h->k.N; g->T = h->full_T; g->rows = h->rolling_max_length + h->k.H + 1; g->max_length = h->rolling_max_length;
    g->lo = h->full_window_lo; g->hi = h->full_window_hi;
    g->row_mask = h->k.row_mask;
    g->fc = h->full_c; g->has_grid = h->layout.has_grid;
    g->draw = 0; g->fixed_length = 0; g->seed = 0; g->start_io = nullptr; g->length_io = nullptr; g->t0_io = nullptr;
    g->ep_off = h->inplace ? h->k.ep_off : nullptr;     // in place: a (re)start writes the grid's row offset and episode end, no rows
}

int mgx_reset_windows_rolling(mgx_handle *h, const int32_t *start, const int32_t *length, int32_t max_length, int32_t ring_rows,
                              double *load_w, double *pv_w, double *grid_w, int32_t *final_abs, void *obs, mgx_stream stream)
{
    g_err[0] = 0;
    if (!h || !start || !load_w || !pv_w || !final_abs) return fail(MGX_ERR_INVALID, "mgx_reset_windows_rolling: NULL argument");
    if (h->multi) return fail(MGX_ERR_UNSUPPORTED, "mgx_reset_windows_rolling: needs exactly one module of every kind per grid");
    if (h->k.obs_colpitch) return fail(MGX_ERR_UNSUPPORTED, "mgx_reset_windows_rolling: not offered with column-major ring blocks (mgx_set_ring_layout)");
    if (h->layout.has_grid && !grid_w) return fail(MGX_ERR_INVALID, "mgx_reset_windows_rolling: grid_w is NULL but the layout has a GridModule");
    if (h->k.t_dev) return fail(MGX_ERR_UNSUPPORTED, "mgx_reset_windows_rolling: not offered in device-counter mode");
    if (h->n_shards > 1) return fail(MGX_ERR_UNSUPPORTED, "mgx_reset_windows_rolling: not offered while the handle steps in shards");
    if (!h->windowed) {
        h->full_load_ts = h->k.c.load_ts; h->full_pv_ts = h->k.c.pv_ts; h->full_grid_ts = h->k.c.grid_ts;
        h->full_T = h->k.T; h->full_final = h->layout.final_step; h->full_initial = h->layout.initial_step;
        h->full_window_lo = h->window_lo; h->full_window_hi = h->window_hi;
    }
    if (max_length < 1 || max_length > h->full_window_hi - h->full_window_lo)
        return fail(MGX_ERR_INVALID, "Cannot create a trajectory of length %d between initial_step (%d) and final_step (%d)",
                    max_length, h->full_window_lo, h->full_window_hi);
    if (ring_rows < max_length + h->k.H + 1 || (ring_rows & (ring_rows - 1)) != 0)
        return fail(MGX_ERR_INVALID, "mgx_reset_windows_rolling: ring_rows = %d must be a power of two >= max_length + horizon + 1 = %d",
                    ring_rows, max_length + h->k.H + 1);
    h->rolling_max_length = max_length;
    h->roll_load_w = load_w; h->roll_pv_w = pv_w; h->roll_grid_w = grid_w; h->roll_final = final_abs;
    h->k.row_mask = ring_rows - 1;
    leave_inplace(h);
    GatherArgs g;
    rolling_gather_args(h, &g);
    g.start = start; g.length = length; g.mask = nullptr; g.row0 = 0;
    hipStream_t st = (hipStream_t)stream;
    gather_windows_kernel<<<blocks_for(h->k.N), BLOCK, 0, st>>>(g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "gather_windows_kernel launch");
    h->k.c.base_load = nullptr;                          // the episodes step over the (materialised) window rings
    h->k.c.load_ts = load_w; h->k.c.pv_ts = pv_w; if (h->layout.has_grid) h->k.c.grid_ts = grid_w;
    h->k.T = INT32_MAX / 2; h->k.final_step = INT32_MAX / 2; h->k.grid_final = final_abs;
    h->layout.n_steps = ring_rows; h->layout.final_step = INT32_MAX / 2; h->layout.initial_step = 0;
    h->window_lo = 0; h->window_hi = INT32_MAX / 2;
    h->windowed = true; h->rolling = true;
    h->t = 0;
    return obs ? mgx_observe(h, obs, stream) : MGX_OK;
}

int mgx_reset_episodes(mgx_handle *h, const int32_t *start, const int32_t *length, int32_t max_length, int32_t *row_off,
                       int32_t *final_abs, void *obs, mgx_stream stream)
{
    g_err[0] = 0;
    if (!h || !start || !row_off || !final_abs) return fail(MGX_ERR_INVALID, "mgx_reset_episodes: NULL argument");
    // several modules of a kind (round 6): single steps of the general kernels read the grid's own rows of the [T, n, N] series (a per-lane
    // gather: no grid-major copy); rows per step only -- the ring patches (mgx_patch_windows) are single-instance
    if (h->multi && h->k.obs_state_only == 1)
        return fail(MGX_ERR_UNSUPPORTED, "mgx_reset_episodes: with several modules of a kind the observation rows are written per step "
                                         "(no rings: mgx_set_obs_mode(MGX_OBS_ROWS_FULL))");
    if (h->k.obs_colpitch) return fail(MGX_ERR_UNSUPPORTED, "mgx_reset_episodes: not offered with column-major ring blocks (mgx_set_ring_layout)");
    if (h->k.t_dev) return fail(MGX_ERR_UNSUPPORTED, "mgx_reset_episodes: not offered in device-counter mode");
    if (h->n_shards > 1) return fail(MGX_ERR_UNSUPPORTED, "mgx_reset_episodes: not offered while the handle steps in shards");
    {   // every argument is checked before the handle leaves the mode it is in
        const int32_t lo = h->windowed ? h->full_window_lo : h->window_lo, hi = h->windowed ? h->full_window_hi : h->window_hi;
        if (max_length < 1 || max_length > hi - lo)
            return fail(MGX_ERR_INVALID, "Cannot create a trajectory of length %d between initial_step (%d) and final_step (%d)",
                        max_length, lo, hi);
    }
    if (h->prefetch_pending) { if (int rc = mgx_prefetch_wait(h, stream)) return rc; }
    leave_windows(h);                                   // back to the full (factorised) series, whatever mode the handle was in
    h->full_load_ts = h->k.c.load_ts; h->full_pv_ts = h->k.c.pv_ts; h->full_grid_ts = h->k.c.grid_ts;
    h->full_T = h->k.T; h->full_final = h->layout.final_step; h->full_initial = h->layout.initial_step;
    h->full_window_lo = h->window_lo; h->full_window_hi = h->window_hi;
    // Factorised series: profile-major copies of the base tables (1.7 MB for a year of hourly rows) -- in this mode every lane reads
    // its own row.  [T, N] arrays are read where they lie: a lane takes 8 bytes of its own row's line (a gather, 64 lines per wave
    // and component instead of 4; still no window buffers to copy and restart into).
    if (factorised(h->k.c)) {
        const int32_t pitch = h->full_T;
        const bool co2 = h->full_c.base_co2 != nullptr;
        DeviceGuard on_device(h->device);
        hipError_t e = hipSuccess;
        if (!h->pm_tables || h->pm_pitch != pitch) {
            if (h->pm_tables) (void)hipFree(h->pm_tables);
            h->pm_tables = nullptr;
            e = hipMalloc((void **)&h->pm_tables, (size_t)3 * MGX_PROFILE_PITCH * pitch * sizeof(double));
            if (e != hipSuccess) return hip_fail(e, "mgx_reset_episodes: allocating the profile-major base tables");
            h->pm_pitch = pitch;
        }
        const size_t one = (size_t)MGX_PROFILE_PITCH * pitch;
        const unsigned blocks = (unsigned)blocks_for((int64_t)one);
        profile_major_kernel<<<blocks, BLOCK, 0, (hipStream_t)stream>>>(h->full_c.base_load, h->pm_tables, h->full_T, pitch);
        profile_major_kernel<<<blocks, BLOCK, 0, (hipStream_t)stream>>>(h->full_c.base_pv, h->pm_tables + one, h->full_T, pitch);
        if (co2) profile_major_kernel<<<blocks, BLOCK, 0, (hipStream_t)stream>>>(h->full_c.base_co2, h->pm_tables + 2 * one, h->full_T, pitch);
        e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "profile_major_kernel launch");
        h->k.c.base_load = h->pm_tables; h->k.c.base_pv = h->pm_tables + one;
        if (co2) h->k.c.base_co2 = h->pm_tables + 2 * one;
        h->k.pm_pitch = pitch;
    } else {
        // [T, N] series: grid-major copies [N, pitch] (as much memory again as the series; without it -- allocation refused --
        // the lanes gather their rows out of the [T, N] arrays: 64 lines per wave and component, same values)
        const int32_t pitch = (h->full_T + 15) & ~15;
        const int64_t N = h->k.N;
        const int ncomp = 2 + 4 * h->layout.has_grid;
        DeviceGuard on_device(h->device);
        const bool want_gm = tune(MGX_TUNE_GRID_MAJOR_COPY) != 0 && !h->multi;   // (0, tests: take the gather path although the copy would fit)
        if (want_gm && (!h->gm_tables || h->gm_pitch != pitch)) {
            if (h->gm_tables) (void)hipFree(h->gm_tables);
            h->gm_tables = nullptr; h->gm_pitch = 0;
            if (hipMalloc((void **)&h->gm_tables, (size_t)N * ncomp * pitch * sizeof(double)) == hipSuccess) h->gm_pitch = pitch;
            else { h->gm_tables = nullptr; (void)hipGetLastError(); }
        }
        if (want_gm && h->gm_tables) {
            const dim3 tiles((unsigned)((N + 31) / 32), (unsigned)((pitch + 31) / 32), 1);
            hipStream_t s = (hipStream_t)stream;
            grid_major_kernel<<<tiles, 256, 0, s>>>(h->full_load_ts, h->gm_tables, N, h->full_T, 1, pitch, ncomp, 0);
            grid_major_kernel<<<tiles, 256, 0, s>>>(h->full_pv_ts, h->gm_tables, N, h->full_T, 1, pitch, ncomp, 1);
            if (h->layout.has_grid)
                grid_major_kernel<<<dim3(tiles.x, tiles.y, 4), 256, 0, s>>>(h->full_grid_ts, h->gm_tables, N, h->full_T, 4, pitch, ncomp, 2);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return hip_fail(e, "grid_major_kernel launch");
            h->k.c.load_ts = h->gm_tables; h->k.c.pv_ts = h->gm_tables + 1;
            if (h->layout.has_grid) h->k.c.grid_ts = h->gm_tables + 2;
            h->k.pm_pitch = pitch;
        }
    }
    h->rolling_max_length = max_length;
    h->roll_load_w = nullptr; h->roll_pv_w = nullptr; h->roll_grid_w = nullptr; h->roll_final = final_abs;
    h->k.row_mask = -1;
    h->k.ep_off = row_off; h->k.ep_final = final_abs; h->k.grid_final = final_abs;
    h->k.ar_mode = 0; h->k.final_obs = nullptr;
    h->inplace = true;
    GatherArgs g;
    rolling_gather_args(h, &g);
    g.rows = 0;
    g.start = start; g.length = length; g.mask = nullptr; g.row0 = 0;
    gather_windows_kernel<<<blocks_for(h->k.N), BLOCK, 0, (hipStream_t)stream>>>(g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { leave_inplace(h); h->k.grid_final = nullptr; return hip_fail(e, "gather_windows_kernel launch"); }
    // k.T stays the series length (rows beyond it are padding, per grid); the counter itself never ends
    h->k.final_step = INT32_MAX / 2;
    h->layout.final_step = INT32_MAX / 2; h->layout.initial_step = 0;
    h->window_lo = 0; h->window_hi = INT32_MAX / 2;
    h->windowed = true; h->rolling = true;
    h->t = 0;
    return obs ? mgx_observe(h, obs, stream) : MGX_OK;
}

int mgx_set_auto_reset(mgx_handle *h, int32_t enable, uint64_t seed, int32_t fixed_length, int32_t *start_io, int32_t *length_io,
                       int32_t *t0_io)
{
    g_err[0] = 0;
    if (!h) return fail(MGX_ERR_INVALID, "mgx_set_auto_reset: NULL handle");
    if (!h->inplace) return fail(MGX_ERR_INVALID, "mgx_set_auto_reset: the handle is not stepping in-place episodes (mgx_reset_episodes)");
    if (!enable) {
        h->k.ar_mode = 0; h->k.ar_start_io = nullptr; h->k.ar_length_io = nullptr; h->k.ar_t0_io = nullptr;
        return MGX_OK;
    }
    if (fixed_length < 0 || fixed_length > h->rolling_max_length)
        return fail(MGX_ERR_INVALID, "mgx_set_auto_reset: fixed_length %d outside [0, max_length = %d]", fixed_length, h->rolling_max_length);
    if (fixed_length == 0 && h->rolling_max_length < h->full_window_hi - h->full_window_lo)
        return fail(MGX_ERR_INVALID, "mgx_set_auto_reset: StochasticTrajectory draws need max_length = the whole window (%d < %d)",
                    h->rolling_max_length, h->full_window_hi - h->full_window_lo);
    h->k.ar_mode = 1; h->k.ar_seed = seed; h->k.ar_fixed_length = fixed_length;
    h->k.ar_lo = h->full_window_lo; h->k.ar_hi = h->full_window_hi; h->k.ar_max_length = h->rolling_max_length;
    h->k.ar_start_io = start_io; h->k.ar_length_io = length_io; h->k.ar_t0_io = t0_io;
    return MGX_OK;
}

int mgx_set_final_obs(mgx_handle *h, void *final_obs)
{
    g_err[0] = 0;
    if (!h) return fail(MGX_ERR_INVALID, "mgx_set_final_obs: NULL handle");
    if (final_obs && !h->inplace)
        return fail(MGX_ERR_INVALID, "mgx_set_final_obs: the handle is not stepping in-place episodes (mgx_reset_episodes)");
    if (final_obs && h->multi)
        return fail(MGX_ERR_UNSUPPORTED, "mgx_set_final_obs: needs exactly one module of every kind per grid");
    h->k.final_obs = final_obs;
    return MGX_OK;
}

int mgx_reset_grids_random(mgx_handle *h, const uint8_t *mask, uint64_t seed, int32_t fixed_length, int32_t *start_io,
                           int32_t *length_io, int32_t *t0_io, mgx_stream stream)
{
    g_err[0] = 0;
    if (!h || !mask) return fail(MGX_ERR_INVALID, "mgx_reset_grids_random: NULL argument");
    if (!h->rolling) return fail(MGX_ERR_INVALID, "mgx_reset_grids_random: the handle is not in rolling-window mode");
    if (dev_counter(h)) return fail(MGX_ERR_UNSUPPORTED, "mgx_reset_grids_random: not offered in device-counter mode");
    if (fixed_length < 0 || fixed_length > h->rolling_max_length)
        return fail(MGX_ERR_INVALID, "mgx_reset_grids_random: fixed_length %d outside [0, max_length = %d]", fixed_length, h->rolling_max_length);
    if (fixed_length == 0 && h->rolling_max_length < h->full_window_hi - h->full_window_lo)
        return fail(MGX_ERR_INVALID, "mgx_reset_grids_random: StochasticTrajectory draws need rings for the whole window "
                                     "(max_length %d < %d)", h->rolling_max_length, h->full_window_hi - h->full_window_lo);
    if (h->t > INT32_MAX / 4) return fail(MGX_ERR_RANGE, "mgx_reset_grids_random: the shared step counter is about to overflow");
    GatherArgs g;
    rolling_gather_args(h, &g);
    g.start = nullptr; g.length = nullptr; g.mask = mask; g.row0 = h->t;
    g.draw = 1; g.fixed_length = fixed_length; g.seed = seed; g.start_io = start_io; g.length_io = length_io; g.t0_io = t0_io;
    gather_windows_kernel<<<blocks_for(h->k.N), BLOCK, 0, (hipStream_t)stream>>>(g);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MGX_OK : hip_fail(e, "gather_windows_kernel launch");
}

int mgx_reset_grids(mgx_handle *h, const uint8_t *mask, const int32_t *start, const int32_t *length, mgx_stream stream)
{
    g_err[0] = 0;
    if (!h || !mask || !start) return fail(MGX_ERR_INVALID, "mgx_reset_grids: NULL argument");
    if (!h->rolling) return fail(MGX_ERR_INVALID, "mgx_reset_grids: the handle is not in rolling-window mode (mgx_reset_windows_rolling)");
    if (dev_counter(h)) return fail(MGX_ERR_UNSUPPORTED, "mgx_reset_grids: not offered in device-counter mode");
    if (h->t > INT32_MAX / 4) return fail(MGX_ERR_RANGE, "mgx_reset_grids: the shared step counter is about to overflow; start over "
                                                         "with mgx_reset_windows_rolling");
    GatherArgs g;
    rolling_gather_args(h, &g);
    g.start = start; g.length = length; g.mask = mask; g.row0 = h->t;
    gather_windows_kernel<<<blocks_for(h->k.N), BLOCK, 0, (hipStream_t)stream>>>(g);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MGX_OK : hip_fail(e, "gather_windows_kernel launch");
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
            if (!h->shard_stream[j]) {
                if (h->device < 0 || h->device >= MGX_MAX_DEVICES) return fail(MGX_ERR_UNSUPPORTED, "mgx_set_shards: device index %d", h->device);
                std::lock_guard<std::mutex> guard(g_shard_streams_lock);
                hipStream_t &pooled = g_shard_streams[h->device][j];
                if (!pooled) e = hipStreamCreateWithFlags(&pooled, hipStreamNonBlocking);
                h->shard_stream[j] = pooled;
            }
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

// ---- a single step during in-place episodes -----------------------------------------------------------------------
// The step kernel restarts finished grids itself (mgx_set_auto_reset) and writes the observation before the restart
// (mgx_set_final_obs) -- as long as observation rows are written inline (no forecast horizon, or state columns only).  With
// a horizon the rows come from obs_rows_wave_kernel behind the step, which reads the offsets the step left: when the
// pre-restart rows are wanted too, the restart is taken out of the step kernel and issued between the two observation passes.
struct EpisodeStep {
    KArgs k;              // what the step kernel gets
    bool restart_after;   // the restart follows the step as a launch of its own (between the two observation passes)
};

static int episode_step_begin(mgx_handle *h, const uint8_t *done, void *obs, void *obs_inline, EpisodeStep *ep, const char *who)
{
    ep->k = h->k; ep->k.g0 = 0; ep->k.g1 = h->k.N;
    ep->restart_after = false;
    const bool rows_behind = obs && !obs_inline;
    if (ep->k.final_obs && !obs)
        return fail(MGX_ERR_INVALID, "%s: mgx_set_final_obs is set but the step writes no observation", who);
    if (ep->k.obs_state_only == 1) ep->k.final_obs = nullptr;    // rings: mgx_patch_windows saves the rows of the restarted grids
    if (rows_behind && ep->k.final_obs) {
        if (ep->k.ar_mode && !done)
            return fail(MGX_ERR_INVALID, "%s: with a forecast horizon the observation before an automatic restart needs the `done` flags", who);
        ep->restart_after = ep->k.ar_mode != 0;
        ep->k.ar_mode = 0;
        ep->k.final_obs = nullptr;                        // (written by the first observation pass below)
    }
    return MGX_OK;
}

static int episode_step_end(mgx_handle *h, const EpisodeStep &ep, const uint8_t *done, void *obs, void *obs_inline, hipStream_t st)
{
    if (!obs || obs_inline) return MGX_OK;
    if (h->k.final_obs) { if (int rc = launch_observe(h, h->t + 1, h->k.final_obs, st)) return rc; }
    if (ep.restart_after) {
        GatherArgs g;
        rolling_gather_args(h, &g);
        g.rows = 0;
        g.start = nullptr; g.length = nullptr; g.mask = done; g.row0 = h->t + 1;
        g.draw = 1; g.fixed_length = h->k.ar_fixed_length; g.seed = h->k.ar_seed;
        g.start_io = h->k.ar_start_io; g.length_io = h->k.ar_length_io; g.t0_io = h->k.ar_t0_io;
        gather_windows_kernel<<<blocks_for(h->k.N), BLOCK, 0, st>>>(g);
    }
    return launch_observe(h, h->t + 1, obs, st);
}

// LDS of a step_kernel / step_discrete_kernel launch: the wave-private row tiles (store_step_obs) -- only launches that write whole
// H = 0 rows ask for them; state-only rows, no rows and the final-observation rows of in-place episodes leave occupancy alone
static inline size_t row_tile_lds(const KArgs &k, const void *obs_inline)
{
    return step_rows_tiled(k, obs_inline) ? sizeof(double) * (BLOCK / 64) * 64 * ROW_TILE_MAX_D : 0;
}

// ---- single steps ----------------------------------------------------------------------------------------------
// one Microgrid.run of every grid: the launches of mgx_step without its argument checks
static int step_once(mgx_handle *h, const void *actions, int normalized, double *reward, uint8_t *done, void *obs, double *log,
                     hipStream_t st)
{
    if (h->multi) {
        // noisy forecasters: the rows come from observe_multi_kernel behind the step (the step kernel carries no noise code)
        const bool rows_behind = obs && !h->k.obs_state_only && (h->k.c.load_noise_std || h->k.c.pv_noise_std || h->k.c.grid_noise_std);
        void *obs_rows = obs;
        if (rows_behind) {
            if (h->n_shards > 1) return fail(MGX_ERR_UNSUPPORTED, "mgx_step: noisy observation rows are not written per shard");
            obs = nullptr;
        }
        for_each_shard_threaded(h, st, [&](const KArgs &k, hipStream_t s) {
            if (h->inplace && k.ep_off) {                 // in-place episodes: the EP form (the grid's own rows, restarts in the kernel)
                MGX_DISPATCH_F(h->flags, (step_multi_kernel<F, true><<<multi_blocks(k.g1 - k.g0), BLOCK_MULTI, h->multi_lds, s>>>(
                                              k, actions, t_arg(h), normalized, reward, done, obs, log, h->multi_small)));
            } else {
                MGX_DISPATCH_F(h->flags, (step_multi_kernel<F><<<multi_blocks(k.g1 - k.g0), BLOCK_MULTI, h->multi_lds, s>>>(
                                              k, actions, t_arg(h), normalized, reward, done, obs, log, h->multi_small)));
            }
        });
        if (rows_behind) { if (int rc = launch_observe(h, dev_counter(h) ? 0 : h->t + 1, obs_rows, st)) return rc; }
        hipError_t em = launch_error();
        if (em != hipSuccess) return hip_fail(em, "step_multi_kernel launch");
        advance(h, 1, st);
        return MGX_OK;
    }
    void *obs_inline = (obs && (h->k.H == 0 || h->k.obs_state_only)) ? obs : nullptr;
    if (h->inplace) {                                     // in-place episodes: the EP form of the kernel (no shards in this mode)
        EpisodeStep ep;
        if (int rc = episode_step_begin(h, done, obs, obs_inline, &ep, "mgx_step")) return rc;
        MGX_DISPATCH_F(h->flags, (step_kernel<F, true><<<blocks_for(ep.k.N), BLOCK, row_tile_lds(ep.k, obs_inline), st>>>(ep.k, actions, h->t, normalized, reward, done,
                                                                                       obs_inline, log)));
        if (int rc = episode_step_end(h, ep, done, obs, obs_inline, st)) return rc;
        hipError_t ee = hipGetLastError();
        if (ee != hipSuccess) return hip_fail(ee, "step_kernel launch");
        advance(h, 1, st);
        return MGX_OK;
    }
    for_each_shard_threaded(h, st, [&](const KArgs &k, hipStream_t s) {
        MGX_DISPATCH_F(h->flags, (step_kernel<F><<<blocks_for(k.g1 - k.g0), BLOCK, row_tile_lds(k, obs_inline), s>>>(k, actions, t_arg(h), normalized, reward,
                                                                                       done, obs_inline, log)));
    });
    if (obs && !obs_inline) { if (int rc = launch_observe(h, dev_counter(h) ? 0 : h->t + 1, obs, st)) return rc; }
    hipError_t e = launch_error();
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

int mgx_action_bounds(mgx_handle *h, double *lo, double *hi, mgx_stream stream)
{
    g_err[0] = 0;
    if (!h || !lo || !hi) return fail(MGX_ERR_INVALID, "mgx_action_bounds: NULL argument");
    if (!dev_counter(h) && (h->t < 0 || h->t >= step_limit(h)))
        return fail(MGX_ERR_RANGE, "mgx_action_bounds: step %d is outside the time series (length %d)", h->t, step_limit(h));
    KArgs k = h->k; k.g0 = 0; k.g1 = k.N;
    MGX_DISPATCH_F(h->flags, (action_bounds_kernel<F><<<multi_blocks(k.N), BLOCK_MULTI, 0, (hipStream_t)stream>>>(k, t_arg(h), lo, hi)));
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MGX_OK : hip_fail(e, "action_bounds_kernel launch");
}

int mgx_step_many(mgx_handle *h, const void *actions, int32_t K, int normalized, double *reward, uint8_t *done, void *obs,
                  double *log, mgx_stream stream)
{
    g_err[0] = 0;
    if (int rc = check_step_args(h, actions, reward, obs, K, "mgx_step_many")) return rc;
    const int64_t N = h->k.N;
    const size_t act_row = (size_t)N * h->action_dim * (h->k.act_f32 ? sizeof(float) : sizeof(double));
    const size_t obs_row = (size_t)N * h->k.obs_dim * (h->k.obs_f32 ? sizeof(float) : sizeof(double));
    void *obs_inline = (obs && (h->k.H == 0 || h->k.obs_state_only)) ? obs : nullptr;
    if (h->n_shards > 1 && h->launch_threads && !h->multi && !h->inplace && (!obs || obs_inline)) {
        // every shard's K dependent launches are issued back to back by a thread of their own: no hand-over between the steps
        const int32_t t0 = t_arg(h);
        for_each_shard_threaded(h, (hipStream_t)stream, [&](const KArgs &kk, hipStream_t s) {
            for (int32_t k = 0; k < K; k++) {
                MGX_DISPATCH_F(h->flags, (step_kernel<F><<<blocks_for(kk.g1 - kk.g0), BLOCK, row_tile_lds(kk, obs_inline), s>>>(
                                              kk, actions ? (const char *)actions + k * act_row : nullptr, t0 + k, normalized, reward + k * N,
                                              done ? done + k * N : nullptr, obs_inline ? (char *)obs_inline + k * obs_row : nullptr,
                                              log ? log + (int64_t)k * h->k.log_dim * N : nullptr)));
            }
        }, 1);
        hipError_t e = launch_error();
        if (e != hipSuccess) return hip_fail(e, "step_kernel launch");
        advance(h, K, (hipStream_t)stream);
        return MGX_OK;
    }
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
    if (h->rolling) return fail(MGX_ERR_UNSUPPORTED, "mgx_step_k: rolling windows take single steps (grids restart between steps)");
    if (!dev_counter(h) && (h->t < 0 || (int64_t)h->t + K > step_limit(h)))
        return fail(MGX_ERR_RANGE, "mgx_step_k: steps [%d, %d) leave the time series (length %d)", h->t, h->t + K, step_limit(h));
    hipStream_t st = (hipStream_t)stream;
    const FusedOut fo{reward, done, soc_trace, status_trace, ret_acc, log};
    if (h->multi) {                                       // general path: the K-step loop around the general step
        if (h->k.done_bits && done) return fail(MGX_ERR_UNSUPPORTED, "mgx_step_k: the general kernels write `done` as bytes");
        const bool own_kernel = tune(MGX_TUNE_MULTI_SMALL_OWN) != 0;
        const bool static_counts = tune(MGX_TUNE_MULTI_STATIC) != 0;
        for_each_shard(h, st, [&](const KArgs &k, hipStream_t s) {
            if (own_kernel && static_counts && k.n_load >= 1 && k.n_pv >= 1) {   // ... with compile-time instance counts where the layout has them
                // (also layouts with THREE modules of a kind: beyond the run-time-count register form, h->multi_small == 0)
                MultiStaticLaunch L{h->flags, k.n_genset, k.n_battery, k.n_grid, k.n_load, k.n_pv, multi_blocks(k.g1 - k.g0), s, &k, actions,
                                    t_arg(h), K, normalized, fo};
                if (launch_step_k_multi_static(L)) return;
            }
            if (h->multi_small && own_kernel) {           // at most MS modules of a kind: the register loop in a kernel of its own
                MGX_DISPATCH_F(h->flags, (step_k_multi_small_kernel<F><<<multi_blocks(k.g1 - k.g0), BLOCK_MULTI, 0, s>>>(
                                              k, actions, t_arg(h), K, normalized, fo)));
            } else {
                MGX_DISPATCH_F(h->flags, (step_k_multi_kernel<F><<<multi_blocks(k.g1 - k.g0), BLOCK_MULTI, h->multi_lds, s>>>(
                                              k, actions, nullptr, 0, 0, nullptr, 0, t_arg(h), K, normalized, fo, h->multi_small)));
            }
        });
        hipError_t em = hipGetLastError();
        if (em != hipSuccess) return hip_fail(em, "step_k_multi_kernel launch");
        advance(h, K, st);
        return MGX_OK;
    }
    const bool fact = factorised(h->k.c);
    for_each_shard(h, st, [&](const KArgs &k, hipStream_t s) {
        FusedLaunch L;
        L.flags = h->flags; L.act_f32 = k.act_f32 != 0; L.rich = log != nullptr || status_trace != nullptr; L.fact = fact;
        L.per_step = false;
        L.gpb = fused_grids_per_block(h, k.g1 - k.g0);
        L.blocks = (unsigned)((k.g1 - k.g0 + L.gpb - 1) / L.gpb);
        L.stream = s; L.k = &k; L.actions = actions; L.tab = nullptr; L.ids = nullptr;
        L.t = t_arg(h); L.K = K; L.normalized = normalized; L.out = fo;
        (void)(launch_step_k_p0(L) || launch_step_k_p1(L) || launch_step_k_p2(L) || launch_step_k_p3(L) || launch_step_k_p4(L));
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
                               double *control, uint32_t *violations, hipStream_t st)
{
    for_each_shard(h, st, [&](const KArgs &k, hipStream_t s) {
        MGX_DISPATCH_F(h->flags, (expand_multi_kernel<F><<<multi_blocks(k.g1 - k.g0), BLOCK_MULTI, 0, s>>>(k, d_lists, n_lists, list_len,
                                                                                                       action_id, t_arg(h), control,
                                                                                                       violations)));
    });
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MGX_OK : hip_fail(e, "expand_multi_kernel launch");
}

int mgx_expand_discrete(mgx_handle *h, const int32_t *action_id, const int32_t *table, int32_t n_actions,
                        double *control, uint32_t *violations, mgx_stream stream)
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
        return launch_expand_lists(h, action_id, h->d_lists, n_actions, 3, control, violations, st);
    }
    for_each_shard(h, st, [&](const KArgs &k, hipStream_t s) {
        MGX_DISPATCH_F(h->flags, (expand_kernel<F><<<blocks_for(k.g1 - k.g0), BLOCK, 0, s>>>(k, tab, action_id, t_arg(h), control, violations)));
    });
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MGX_OK : hip_fail(e, "expand_kernel launch");
}

int mgx_expand_lists(mgx_handle *h, const int32_t *action_id, const int32_t *lists, int32_t n_lists, int32_t list_len,
                     double *control, uint32_t *violations, mgx_stream stream)
{
    g_err[0] = 0;
    if (!h || !action_id || !lists || !control) return fail(MGX_ERR_INVALID, "mgx_expand_lists: NULL argument");
    if (n_lists <= 0 || list_len <= 0 || list_len > 3 * MGX_MAX_INSTANCES)
        return fail(MGX_ERR_INVALID, "mgx_expand_lists: need n_lists > 0 and list_len in [1, %d]", 3 * MGX_MAX_INSTANCES);
    if (!dev_counter(h) && (h->t < 0 || h->t >= step_limit(h)))
        return fail(MGX_ERR_RANGE, "mgx_expand_lists: step %d is outside the time series (length %d)", h->t, step_limit(h));
    return launch_expand_lists(h, action_id, lists, n_lists, list_len, control, violations, (hipStream_t)stream);
}

int mgx_step_lists(mgx_handle *h, const int32_t *action_id, const int32_t *lists, int32_t n_lists, int32_t list_len,
                   double *control, double *reward, uint8_t *done, void *obs, double *log, mgx_stream stream)
{
    g_err[0] = 0;
    if (!h || !action_id || !lists) return fail(MGX_ERR_INVALID, "mgx_step_lists: NULL argument");
    if (n_lists <= 0 || list_len <= 0 || list_len > 3 * MGX_MAX_INSTANCES)
        return fail(MGX_ERR_INVALID, "mgx_step_lists: need n_lists > 0 and list_len in [1, %d]", 3 * MGX_MAX_INSTANCES);
    if (int rc = check_step_args(h, action_id, reward, obs, 1, "mgx_step_lists")) return rc;
    hipStream_t st = (hipStream_t)stream;
    const bool noisy_rows = obs && !h->k.obs_state_only && (h->k.c.load_noise_std || h->k.c.pv_noise_std || h->k.c.grid_noise_std);
    const bool one_launch = h->multi && h->multi_small && !h->inplace && !h->rolling && !noisy_rows && tune(MGX_TUNE_MULTI_SMALL_OWN) != 0 &&
                            h->k.n_load >= 1 && h->k.n_pv >= 1;
    if (!one_launch) {                                    // any other layout: the control passes through the caller's buffer
        if (!control) return fail(MGX_ERR_INVALID, "mgx_step_lists: this layout steps in two launches and needs the control buffer [N, A]");
        if (int rc = launch_expand_lists(h, action_id, lists, n_lists, list_len, control, nullptr, st)) return rc;
        return step_once(h, control, 0, reward, done, obs, log, st);
    }
    for_each_shard_threaded(h, st, [&](const KArgs &k, hipStream_t s) {
        MGX_DISPATCH_F(h->flags, (step_lists_small_kernel<F><<<multi_blocks(k.g1 - k.g0), BLOCK_MULTI, 0, s>>>(
                                      k, action_id, lists, n_lists, list_len, t_arg(h), control, reward, done, obs, log)));
    });
    hipError_t e = launch_error();
    if (e != hipSuccess) return hip_fail(e, "step_lists_small_kernel launch");
    advance(h, 1, st);
    return MGX_OK;
}

int mgx_check_discrete(mgx_handle *h, const int32_t *action_id, const int32_t *table, int32_t n_actions, uint32_t *violations,
                       mgx_stream stream)
{
    g_err[0] = 0;
    if (!h || !action_id || !table || !violations) return fail(MGX_ERR_INVALID, "mgx_check_discrete: NULL argument");
    if (h->multi) return fail(MGX_ERR_UNSUPPORTED, "mgx_check_discrete: needs exactly one module of every kind per grid; use "
                                                    "mgx_expand_discrete / mgx_expand_lists with `violations`, then mgx_check_step");
    if (!dev_counter(h) && (h->t < 0 || h->t >= step_limit(h)))
        return fail(MGX_ERR_RANGE, "mgx_check_discrete: step %d is outside the time series (length %d)", h->t, step_limit(h));
    PLWords tab;
    if (int rc = encode_table(h, table, n_actions, &tab, "mgx_check_discrete")) return rc;
    for_each_shard(h, (hipStream_t)stream, [&](const KArgs &k, hipStream_t s) {
        MGX_DISPATCH_F(h->flags, (check_discrete_kernel<F><<<blocks_for(k.g1 - k.g0), BLOCK, 0, s>>>(k, tab, action_id, t_arg(h), violations)));
    });
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MGX_OK : hip_fail(e, "check_discrete_kernel launch");
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
    if (h->inplace) {
        EpisodeStep ep;
        if (int rc = episode_step_begin(h, done, obs, obs_inline, &ep, "mgx_step_discrete")) return rc;
        MGX_DISPATCH_F(h->flags, (step_discrete_kernel<F, true><<<blocks_for(ep.k.N), BLOCK, row_tile_lds(ep.k, obs_inline), st>>>(ep.k, tab, action_id, h->t, control, reward,
                                                                                                done, obs_inline, log)));
        if (int rc = episode_step_end(h, ep, done, obs, obs_inline, st)) return rc;
        hipError_t ee = hipGetLastError();
        if (ee != hipSuccess) return hip_fail(ee, "step_discrete_kernel launch");
        advance(h, 1, st);
        return MGX_OK;
    }
    for_each_shard_threaded(h, st, [&](const KArgs &k, hipStream_t s) {
        MGX_DISPATCH_F(h->flags, (step_discrete_kernel<F><<<blocks_for(k.g1 - k.g0), BLOCK, row_tile_lds(k, obs_inline), s>>>(k, tab, action_id, t_arg(h), control,
                                                                                                reward, done, obs_inline, log)));
    });
    if (obs && !obs_inline) { if (int rc = launch_observe(h, dev_counter(h) ? 0 : h->t + 1, obs, st)) return rc; }
    hipError_t e = launch_error();
    if (e != hipSuccess) return hip_fail(e, "step_discrete_kernel launch");
    advance(h, 1, st);
    return MGX_OK;
}

// ---- the Gym step without per-step bookkeeping on the caller's side (mgx_env_*) ---------------------------------
int mgx_env_bind(mgx_handle *h, const mgx_env_plan *plan)
{
    g_err[0] = 0;
    if (!h) return fail(MGX_ERR_INVALID, "mgx_env_bind: NULL handle");
    if (!plan) { h->env_bound = false; return MGX_OK; }
    if (plan->struct_size != (int32_t)sizeof(mgx_env_plan))
        return fail(MGX_ERR_INVALID, "mgx_env_bind: struct_size %d vs %zu (ABI %d)", plan->struct_size, sizeof(mgx_env_plan), MGX_ABI_VERSION);
    if (plan->n_slots < 1 || plan->n_slots > MGX_ENV_MAX_SLOTS || !plan->slots)
        return fail(MGX_ERR_INVALID, "mgx_env_bind: n_slots must be in [1, %d] with a slot array", MGX_ENV_MAX_SLOTS);
    for (int32_t j = 0; j < plan->n_slots; j++) {
        if (!plan->slots[j].reward) return fail(MGX_ERR_INVALID, "mgx_env_bind: slot %d has no reward buffer", j);
        if ((h->windowed || h->inplace) && !plan->slots[j].done)      // per-grid episodes end per grid: the flags must go somewhere
            return fail(MGX_ERR_INVALID, "mgx_env_bind: slot %d has no done buffer (the handle steps per-grid episodes)", j);
    }
    if (plan->ring_K < 0) return fail(MGX_ERR_INVALID, "mgx_env_bind: ring_K must be >= 0");
    if (plan->ring_K > 0) {
        if (h->k.obs_state_only != 1)
            return fail(MGX_ERR_INVALID, "mgx_env_bind: rings need the handle in MGX_OBS_ROWS_STATE_ONLY mode (mgx_set_obs_mode)");
        if (!plan->rings[0] || !plan->rings[1] || !plan->rings[2]) return fail(MGX_ERR_INVALID, "mgx_env_bind: three rings are required");
        if (h->rolling || h->windowed || h->inplace)
            return fail(MGX_ERR_UNSUPPORTED, "mgx_env_bind: rings are walked for lock-step episodes (restarted grids are patched in by the caller)");
    }
    if (plan->n_actions < 0 || plan->n_actions > 12 || (plan->n_actions > 0 && !plan->table))
        return fail(MGX_ERR_INVALID, "mgx_env_bind: a priority-list table holds 1..12 lists");
    if (plan->n_actions > 0) {
        PLWords tab;
        if (int rc = encode_table(h, plan->table, plan->n_actions, &tab, "mgx_env_bind")) return rc;
        memcpy(h->env_table, plan->table, sizeof(int32_t) * 6 * (size_t)plan->n_actions);
    }
    h->env_n_actions = plan->n_actions;
    h->env_n_slots = plan->n_slots;
    memcpy(h->env_slots, plan->slots, sizeof(mgx_env_slot) * (size_t)plan->n_slots);
    h->env_ring_K = plan->ring_K;
    for (int r = 0; r < 3; r++) h->env_rings[r] = (char *)plan->rings[r];
    h->env_next = 0; h->env_ring_idx = 0; h->env_ring_pos = 0;
    h->env_bound = true;
    return MGX_OK;
}

int mgx_env_seek(mgx_handle *h, int32_t next_slot, int32_t ring_idx, int32_t ring_pos)
{
    g_err[0] = 0;
    if (!h || !h->env_bound) return fail(MGX_ERR_INVALID, "mgx_env_seek: no plan is bound (mgx_env_bind)");
    if (next_slot < 0 || next_slot >= h->env_n_slots) return fail(MGX_ERR_INVALID, "mgx_env_seek: slot %d outside [0, %d)", next_slot, h->env_n_slots);
    if (h->env_ring_K > 0 && (ring_idx < 0 || ring_idx > 2 || ring_pos < 0 || ring_pos >= h->env_ring_K))
        return fail(MGX_ERR_INVALID, "mgx_env_seek: ring %d block %d outside 3 rings of %d blocks", ring_idx, ring_pos, h->env_ring_K);
    h->env_next = next_slot;
    if (h->env_ring_K > 0) { h->env_ring_idx = ring_idx; h->env_ring_pos = ring_pos; }
    return MGX_OK;
}

int mgx_env_position(const mgx_handle *h, int32_t *last_slot, int32_t *ring_idx, int32_t *ring_pos)
{
    g_err[0] = 0;
    if (!h || !h->env_bound) return fail(MGX_ERR_INVALID, "mgx_env_position: no plan is bound (mgx_env_bind)");
    if (last_slot) *last_slot = (h->env_next + h->env_n_slots - 1) % h->env_n_slots;
    if (ring_idx) *ring_idx = h->env_ring_idx;
    if (ring_pos) *ring_pos = h->env_ring_pos;
    return MGX_OK;
}

namespace {
// where the coming step's observation goes and what has to happen around it
struct EnvTarget { void *obs; bool enters_next_ring; };
inline EnvTarget env_target(const mgx_handle *h, const mgx_env_slot &sl)
{
    if (h->env_ring_K <= 0) return EnvTarget{sl.obs, false};
    const size_t block = (size_t)h->ring_pitch * (size_t)h->k.obs_dim * (h->k.obs_f32 ? sizeof(float) : sizeof(double));
    if (h->env_ring_pos + 1 < h->env_ring_K)
        return EnvTarget{h->env_rings[h->env_ring_idx] + (size_t)(h->env_ring_pos + 1) * block, false};
    return EnvTarget{h->env_rings[(h->env_ring_idx + 1) % 3], true};
}
// after a successful step: the slot moves on, the env moves to the block it just completed
inline int env_commit(mgx_handle *h, bool entered, mgx_stream stream)
{
    h->env_next = h->env_next + 1 < h->env_n_slots ? h->env_next + 1 : 0;
    if (h->env_ring_K <= 0) return MGX_OK;
    if (!entered) { h->env_ring_pos += 1; return MGX_OK; }
    h->env_ring_idx = (h->env_ring_idx + 1) % 3;
    h->env_ring_pos = 0;
    // the ring behind the one just entered: the windows of counter values t + K .. t + 2K - 1, written beside the next K steps
    return observe_windows_ahead(h, h->env_ring_K, h->env_ring_K, h->env_rings[(h->env_ring_idx + 1) % 3], (hipStream_t)stream, nullptr, nullptr);
}
}  // namespace

int mgx_env_step(mgx_handle *h, const void *actions, int normalized, mgx_stream stream)
{
    g_err[0] = 0;
    if (!h || !h->env_bound) return fail(MGX_ERR_INVALID, "mgx_env_step: no plan is bound (mgx_env_bind)");
    const mgx_env_slot &sl = h->env_slots[h->env_next];
    const EnvTarget tg = env_target(h, sl);
    if (int rc = check_step_args(h, actions, sl.reward, tg.obs, 1, "mgx_env_step")) return rc;
    if (tg.enters_next_ring) { if (int rc = mgx_prefetch_wait(h, stream)) return rc; }
    if (int rc = step_once(h, actions, normalized, sl.reward, sl.done, tg.obs, sl.log, (hipStream_t)stream)) return rc;
    return env_commit(h, tg.enters_next_ring, stream);
}

int mgx_env_step_discrete(mgx_handle *h, const int32_t *action_id, mgx_stream stream)
{
    g_err[0] = 0;
    if (!h || !h->env_bound) return fail(MGX_ERR_INVALID, "mgx_env_step_discrete: no plan is bound (mgx_env_bind)");
    if (h->env_n_actions <= 0) return fail(MGX_ERR_INVALID, "mgx_env_step_discrete: the bound plan holds no priority-list table");
    const mgx_env_slot &sl = h->env_slots[h->env_next];
    const EnvTarget tg = env_target(h, sl);
    if (!action_id) return fail(MGX_ERR_INVALID, "mgx_env_step_discrete: NULL argument");       // (nothing has moved yet, as in mgx_env_step)
    if (int rc = check_step_args(h, action_id, sl.reward, tg.obs, 1, "mgx_env_step_discrete")) return rc;
    if (tg.enters_next_ring) { if (int rc = mgx_prefetch_wait(h, stream)) return rc; }
    if (int rc = mgx_step_discrete(h, action_id, h->env_table, h->env_n_actions, nullptr, sl.reward, sl.done, tg.obs, sl.log, stream)) return rc;
    return env_commit(h, tg.enters_next_ring, stream);
}

int mgx_rollout_discrete(mgx_handle *h, const uint8_t *action_id, int per_step, const int32_t *table, int32_t n_actions,
                         int32_t K, double *reward, uint8_t *done, double *soc_trace, uint32_t *status_trace,
                         double *ret_acc, double *log, mgx_stream stream)
{
    g_err[0] = 0;
    if (!h || !action_id || !table) return fail(MGX_ERR_INVALID, "mgx_rollout_discrete: NULL argument");
    if (K <= 0) return fail(MGX_ERR_INVALID, "mgx_rollout_discrete: K must be positive");
    if (h->rolling) return fail(MGX_ERR_UNSUPPORTED, "mgx_rollout_discrete: rolling windows take single steps");
    if (h->multi) return fail(MGX_ERR_UNSUPPORTED, "mgx_rollout_discrete: needs exactly one module of every kind "
                                                    "per grid; use mgx_expand_discrete / mgx_expand_lists + mgx_step");
    if (!dev_counter(h) && (h->t < 0 || (int64_t)h->t + K > step_limit(h)))
        return fail(MGX_ERR_RANGE, "mgx_rollout_discrete: steps [%d, %d) leave the time series (length %d)", h->t, h->t + K, step_limit(h));
    PLWords tab;
    if (int rc = encode_table(h, table, n_actions, &tab, "mgx_rollout_discrete")) return rc;
    const FusedOut fo{reward, done, soc_trace, status_trace, ret_acc, log};
    hipStream_t st = (hipStream_t)stream;
    const bool fact = factorised(h->k.c);
    for_each_shard(h, st, [&](const KArgs &k, hipStream_t s) {
        FusedLaunch L;
        L.flags = h->flags; L.act_f32 = false; L.rich = log != nullptr || status_trace != nullptr; L.fact = fact;
        L.per_step = per_step != 0;
        L.gpb = fused_grids_per_block(h, k.g1 - k.g0);
        L.blocks = (unsigned)((k.g1 - k.g0 + L.gpb - 1) / L.gpb);
        L.stream = s; L.k = &k; L.actions = nullptr; L.tab = &tab; L.ids = action_id;
        L.t = t_arg(h); L.K = K; L.normalized = 0; L.out = fo;
        (void)(launch_rollout_p0(L) || launch_rollout_p1(L) || launch_rollout_p2(L) || launch_rollout_p3(L) || launch_rollout_p4(L));
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
    if (h->rolling) return fail(MGX_ERR_UNSUPPORTED, "mgx_rollout_lists: rolling windows take single steps");
    if (factorised(h->k.c)) return fail(MGX_ERR_UNSUPPORTED, "mgx_rollout_lists: the general kernels read materialised series; "
                                                             "use mgx_rollout_discrete on a factorised batch");
    if (h->k.done_bits && done) return fail(MGX_ERR_UNSUPPORTED, "mgx_rollout_lists: the general kernels write `done` as bytes");
    if (!dev_counter(h) && (h->t < 0 || (int64_t)h->t + K > step_limit(h)))
        return fail(MGX_ERR_RANGE, "mgx_rollout_lists: steps [%d, %d) leave the time series (length %d)", h->t, h->t + K, step_limit(h));
    hipStream_t st = (hipStream_t)stream;
    const FusedOut fo{reward, done, soc_trace, status_trace, ret_acc, log};
    const bool static_counts = tune(MGX_TUNE_MULTI_STATIC) != 0 && tune(MGX_TUNE_MULTI_SMALL_OWN) != 0;
    for_each_shard(h, st, [&](const KArgs &k, hipStream_t s) {
        if (static_counts && k.n_load >= 1 && k.n_pv >= 1) {     // the register form where the layout has a compile-time-count specialisation
            MultiStaticRollout L{h->flags, k.n_genset, k.n_battery, k.n_grid, k.n_load, k.n_pv, multi_blocks(k.g1 - k.g0), s, &k, lists,
                                 n_lists, list_len, action_id, per_step, t_arg(h), K, fo};
            if (launch_rollout_multi_static(L)) return;
        }
        if (h->multi_small && tune(MGX_TUNE_MULTI_SMALL_OWN) != 0) {     // any other layout of at most MS modules of a kind: run-time counts, same walk
            MGX_DISPATCH_F(h->flags, (rollout_multi_small_kernel<F, CountsRT, MS><<<multi_blocks(k.g1 - k.g0), BLOCK_MULTI, 0, s>>>(
                                          k, lists, n_lists, list_len, action_id, per_step, t_arg(h), K, fo)));
            return;
        }
        MGX_DISPATCH_F(h->flags, (step_k_multi_kernel<F><<<multi_blocks(k.g1 - k.g0), BLOCK_MULTI, h->multi_lds, s>>>(
                                      k, nullptr, lists, n_lists, list_len, action_id, per_step, t_arg(h), K, 0, fo, h->multi_small)));
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
    // One launch for all batches (continuous and discrete items alike) that can share it; an item that needs kernels of its own
    // (several modules of a kind, rolling windows, a device counter, shards, per-step rows with a horizon) is stepped beside it --
    // the others still share their launch (round 6: one such bucket used to send EVERY bucket of the fleet to its own launch).
    if (n > 64) return fail(MGX_ERR_INVALID, "mgx_fleet_step: at most 64 items per call");
    bool fuse[64];
    int32_t fz[64], nf = 0;                                 // the items that share the launch, in item order
    bool any_chunks = tune(MGX_TUNE_FLEET_BYVALUE) == 0;    // window chunks ride with the POINTER form of the kernel: single-instance layouts only
    for (int32_t j = 0; j < n; j++) any_chunks = any_chunks || (items[j].refill_ring && items[j].refill_chunks > 0);
    for (int32_t j = 0; j < n; j++) {
        const mgx_fleet_item &it = items[j];
        const mgx_handle *h = it.handle;
        // (a bucket with several modules of a kind shares the launch on the register form -- fleet_step_kernel_vm, round 6 -- when it
        //  holds at most MS of a kind, takes continuous controls, walks the series in lock-step and its rows carry no forecast noise)
        const bool multi_ok = h->multi && h->multi_small && !any_chunks && !it.action_id && !h->inplace && !h->windowed &&
                              !(it.obs && (h->k.c.load_noise_std || h->k.c.pv_noise_std || h->k.c.grid_noise_std));
        fuse[j] = (!h->multi || multi_ok) && !h->rolling && !dev_counter(h) && h->n_shards <= 1 && !(it.obs && h->k.H > 0 && !h->k.obs_state_only);
        for (int32_t q = 0; q < j; q++)
            if (items[q].handle == it.handle) return fail(MGX_ERR_INVALID, "mgx_fleet_step: item %d steps the batch of item %d again", j, q);
        if (fuse[j]) fz[nf++] = j;
    }
    bool chunk_done[64];                                    // window chunks that rode along with the step launch
    for (int32_t j = 0; j < n; j++) chunk_done[j] = false;
    for (int32_t j = 0; j < n; j++)
        if (items[j].wait_prefetch) { if (int rc = mgx_prefetch_wait(items[j].handle, stream)) return rc; }
    if (nf > 0) {
        for (int32_t z = 0; z < nf; z++) {                // device copies of the batches' KArgs: uploaded when they changed
            const int32_t j = fz[z];
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
        // MGX_TUNE_FLEET_BYVALUE = 0 keeps the pointer form of the kernel for every launch (A/B)
        const bool by_value = tune(MGX_TUNE_FLEET_BYVALUE) != 0;
        for (int32_t z0 = 0; z0 < nf; z0 += MGX_FLEET_MAX) {
            const int32_t nb = nf - z0 < MGX_FLEET_MAX ? nf - z0 : MGX_FLEET_MAX;
            bool chunks = !by_value;                       // window chunks riding along with this launch (refill="chunks")?
            for (int32_t q = 0; q < nb && !chunks; q++) chunks = items[fz[z0 + q]].refill_ring && items[fz[z0 + q]].refill_chunks > 0;
            bool any_multi = false;                        // (never with chunks: see multi_ok above)
            for (int32_t q = 0; q < nb; q++) any_multi = any_multi || items[fz[z0 + q]].handle->multi;
            if (!chunks) {
                // the buckets' KArgs by value, the bucket = blockIdx.y (fleet_step_kernel_v): unused buckets stay unwritten, never read
                FleetArgsV fv;
                fv.n = nb; fv.normalized = normalized; fv.pad0 = fv.pad1 = 0;
                int32_t most = 0;
                for (int32_t q = 0; q < nb; q++) {
                    const mgx_fleet_item &it = items[fz[z0 + q]];
                    const mgx_handle *h = it.handle;
                    FleetBucket &B = fv.b[q];
                    B.k = h->k;
                    B.hd.tab = it.action_id ? h->d_table : nullptr; B.hd.n_grids = h->k.N; B.hd.t = h->t; B.hd.flags = h->flags;
                    B.hd.pad0 = h->multi ? 1 : 0; B.hd.pad1 = 0;          // pad0: a bucket of the general path (fleet_step_kernel_vm)
                    B.actions = it.action_id ? (const void *)it.action_id : it.actions;
                    B.reward = it.reward; B.done = it.done; B.obs = it.obs; B.log = it.log;
                    const int32_t wg = (int32_t)blocks_for(h->k.N);
                    if (wg > most) most = wg;
                }
                if (any_multi) fleet_step_kernel_vm<<<dim3((unsigned)most, (unsigned)nb), BLOCK, 0, st>>>(fv);
                else fleet_step_kernel_v<<<dim3((unsigned)most, (unsigned)nb), BLOCK, 0, st>>>(fv);
                hipError_t ev = hipGetLastError();
                if (ev != hipSuccess) return hip_fail(ev, "fleet_step_kernel_v launch");
                continue;
            }
            FleetArgs fa;
            FleetWin fw;
            memset(&fa, 0, sizeof(fa));
            memset(&fw, 0, sizeof(fw));
            fa.n = nb;
            fa.normalized = normalized;
            int32_t blocks = 0, wblocks = 0;
            size_t lds_max = 0;
            for (int32_t q = 0; q < fa.n; q++) {
                const int32_t j = fz[z0 + q];
                const mgx_fleet_item &it = items[j];
                mgx_handle *h = it.handle;
                fa.k[q] = h->d_kargs;
                fa.tab[q] = it.action_id ? h->d_table : nullptr;
                fa.actions[q] = it.action_id ? (const void *)it.action_id : it.actions; fa.reward[q] = it.reward; fa.done[q] = it.done; fa.obs[q] = it.obs; fa.log[q] = it.log;
                fa.t[q] = h->t; fa.flags[q] = h->flags; fa.block0[q] = blocks;
                blocks += (int32_t)((h->k.N + BLOCK_FLEET - 1) / BLOCK_FLEET);
                if (it.refill_ring && it.refill_chunks > 0) {                        // this step's share of the next ring
                    WindowsKPlan plan; size_t lds; int32_t ng, first, count;
                    (void)windows_plan(h, it.refill_ahead, it.refill_K, it.refill_ring, "mgx_fleet_step", &plan, &lds, &ng);
                    if (lds > 64 * 1024) continue;                                   // launched on its own below
                    chunk_range(ng, it.refill_chunk, it.refill_chunks, &first, &count);
                    chunk_done[j] = true;
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
            fleet_step_kernel<<<(unsigned)(blocks + wblocks), BLOCK_FLEET, lds_max, st>>>(fa, fw);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return hip_fail(e, "fleet_step_kernel launch");
        }
        for (int32_t z = 0; z < nf; z++) advance(items[fz[z]].handle, 1, st);
    }
    for (int32_t j = 0; j < n; j++) {                     // the items with kernels of their own
        if (fuse[j]) continue;
        const mgx_fleet_item &it = items[j];
        int rc;
        if (it.action_id)
            rc = mgx_step_discrete(it.handle, it.action_id, it.table, it.n_actions, nullptr, it.reward, it.done, it.obs, it.log, stream);
        else
            rc = step_once(it.handle, it.actions, normalized, it.reward, it.done, it.obs, it.log, st);
        if (rc) return rc;
    }
    hipEvent_t fleet_gate = nullptr;                        // ONE gate event for every ring refill this fleet step starts
    hipStream_t gated = nullptr;
    for (int32_t j = 0; j < n; j++) {                       // window prefetch that did not ride along with the step launch
        const mgx_fleet_item &it = items[j];
        if (!it.refill_ring || chunk_done[j]) continue;
        int rc;
        if (it.refill_chunks > 0)                           // a chunk, on the caller's stream (the counter has advanced: ahead as given)
            rc = launch_windows(it.handle, it.refill_ahead, it.refill_K, it.refill_ring, st, "mgx_fleet_step", it.refill_chunk, it.refill_chunks);
        else if (it.refill_ahead > 0) {
            if (!fleet_gate) {
                DeviceGuard on_device(it.handle->device);
                if (int rc0 = ensure_prefetch_stream(it.handle, "mgx_fleet_step: creating the prefetch stream")) return rc0;
                hipError_t e = hipEventRecord(it.handle->prefetch_gate, st);
                if (e != hipSuccess) return hip_fail(e, "mgx_fleet_step: recording the refill gate");
                fleet_gate = it.handle->prefetch_gate;
            }
            rc = observe_windows_ahead(it.handle, it.refill_ahead, it.refill_K, it.refill_ring, st, fleet_gate, &gated);
        } else
            rc = mgx_observe_windows(it.handle, it.refill_K, it.refill_ring, stream);
        if (rc) return rc;
    }
    return MGX_OK;
}

int mgx_fleet_env_step(mgx_handle *const *handles, const void *const *actions, int32_t n, int normalized, mgx_stream stream)
{
    g_err[0] = 0;
    if (!handles || !actions) return fail(MGX_ERR_INVALID, "mgx_fleet_env_step: NULL argument");
    if (n < 1 || n > 64) return fail(MGX_ERR_INVALID, "mgx_fleet_env_step: n_handles = %d outside [1, 64]", n);
    mgx_fleet_item items[64];
    bool enters[64];
    for (int32_t j = 0; j < n; j++) {
        mgx_handle *h = handles[j];
        if (!h || !h->env_bound) return fail(MGX_ERR_INVALID, "mgx_fleet_env_step: handle %d has no plan bound (mgx_env_bind)", j);
        for (int32_t q = 0; q < j; q++)                     // one slot and one ring position per handle and step
            if (handles[q] == h) return fail(MGX_ERR_INVALID, "mgx_fleet_env_step: handle %d is handle %d again", j, q);
        const mgx_env_slot &sl = h->env_slots[h->env_next];
        const EnvTarget tg = env_target(h, sl);
        mgx_fleet_item &it = items[j];
        memset(&it, 0, sizeof(it));
        it.struct_size = (int32_t)sizeof(mgx_fleet_item);
        it.handle = h;
        if (h->env_n_actions > 0) { it.action_id = (const int32_t *)actions[j]; it.table = h->env_table; it.n_actions = h->env_n_actions; }
        else it.actions = actions[j];
        it.reward = sl.reward; it.done = sl.done; it.obs = tg.obs; it.log = sl.log;
        enters[j] = tg.enters_next_ring;
        if (tg.enters_next_ring) {                          // the step completes block 0 of the prefetched ring: wait for it, then have
            it.wait_prefetch = 1;                           // the ring behind it written ahead (env_commit's refill, issued by the fleet step)
            it.refill_ring = h->env_rings[(h->env_ring_idx + 2) % 3];
            it.refill_K = h->env_ring_K; it.refill_ahead = h->env_ring_K;
        }
    }
    if (int rc = mgx_fleet_step(items, n, normalized, stream)) return rc;
    for (int32_t j = 0; j < n; j++) {                       // the slots move on, the envs move to the blocks they just completed
        mgx_handle *h = handles[j];
        h->env_next = h->env_next + 1 < h->env_n_slots ? h->env_next + 1 : 0;
        if (h->env_ring_K <= 0) continue;
        if (!enters[j]) { h->env_ring_pos += 1; continue; }
        h->env_ring_idx = (h->env_ring_idx + 1) % 3;
        h->env_ring_pos = 0;
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
    if ((a->load_ts == nullptr) != (a->pv_ts == nullptr))
        return fail(MGX_ERR_INVALID, "mgx_synthesize_series: load_ts and pv_ts go together");
    if (!a->load_ts && !a->outage_bits) return fail(MGX_ERR_INVALID, "mgx_synthesize_series: nothing to write");
    if (!a->load_ts && a->grid_ts) return fail(MGX_ERR_INVALID, "mgx_synthesize_series: grid_ts without load_ts / pv_ts");
    if (a->load_ts && (!a->base_load || !a->base_pv || !a->load_profile || !a->pv_profile || !a->load_ratio || !a->pv_ratio))
        return fail(MGX_ERR_INVALID, "mgx_synthesize_series: NULL load / pv argument");
    if (a->grid_ts && (!a->base_co2 || !a->co2_profile || !a->tariff || a->n_co2_profiles <= 0))
        return fail(MGX_ERR_INVALID, "mgx_synthesize_series: grid_ts requested without base_co2 / co2_profile / tariff");
    if ((a->grid_ts || a->outage_bits) && a->outage_per_day && (!a->weak || !a->outage_duration))
        return fail(MGX_ERR_INVALID, "mgx_synthesize_series: outage_per_day given without weak / outage_duration");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(MGX_ERR_DEVICE, "mgx_synthesize_series: no HIP device available -- this engine has no CPU path");
    synthesize_series_kernel<<<blocks_for(a->n_grids), BLOCK, 0, (hipStream_t)stream>>>(*a);
    e = hipGetLastError();
    return e == hipSuccess ? MGX_OK : hip_fail(e, "synthesize_series_kernel launch");
}

int mgx_generate_columns(const mgx_gen *a, mgx_stream stream)
{
    g_err[0] = 0;
    if (!a) return fail(MGX_ERR_INVALID, "mgx_generate_columns: NULL argument");
    if (a->struct_size != (int32_t)sizeof(mgx_gen))
        return fail(MGX_ERR_INVALID, "mgx_generate_columns: struct_size %d vs %zu (ABI %d)", a->struct_size, sizeof(mgx_gen), MGX_ABI_VERSION);
    if (a->n_grids <= 0 || a->n_steps <= 0 || a->n_load_profiles <= 0 || a->n_pv_profiles <= 0 || a->n_co2_profiles < 0)
        return fail(MGX_ERR_INVALID, "mgx_generate_columns: need n_grids, n_steps, n_load_profiles, n_pv_profiles > 0");
    if (a->n_load_profiles > 255 || a->n_pv_profiles > 255 || a->n_co2_profiles > 255)
        return fail(MGX_ERR_INVALID, "mgx_generate_columns: profile ids are bytes");
    if (!a->base_load || !a->load_max || !a->pv_max || !a->load_bound_max || !a->pv_bound_max || a->n_mean_rows <= 0)
        return fail(MGX_ERR_INVALID, "mgx_generate_columns: NULL base_load / load_max / pv_max / *_bound_max, or n_mean_rows <= 0");
    if ((a->grid_lo || a->grid_hi) && a->n_co2_profiles > 0 && (!a->co2_min || !a->co2_max))
        return fail(MGX_ERR_INVALID, "mgx_generate_columns: grid bounds requested without co2_min / co2_max");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(MGX_ERR_DEVICE, "mgx_generate_columns: no HIP device available -- this engine has no CPU path");
    generate_columns_kernel<<<blocks_for(a->n_grids), BLOCK, 0, (hipStream_t)stream>>>(*a);
    e = hipGetLastError();
    return e == hipSuccess ? MGX_OK : hip_fail(e, "generate_columns_kernel launch");
}

int mgx_normalise_series(mgx_handle *h, void *load_n, void *pv_n, void *grid_n, int32_t *clipped, mgx_stream stream)
{
    g_err[0] = 0;
    if (!h || !load_n || !pv_n) return fail(MGX_ERR_INVALID, "mgx_normalise_series: NULL argument");
    if (h->multi) return fail(MGX_ERR_UNSUPPORTED, "mgx_normalise_series: needs exactly one module of every kind per grid");
    if (h->layout.has_grid && !grid_n) return fail(MGX_ERR_INVALID, "mgx_normalise_series: grid_n is NULL but the layout has a GridModule");
    if (h->rolling) return fail(MGX_ERR_UNSUPPORTED, "mgx_normalise_series: not offered for rolling windows (restarts rewrite series rows)");
    if (h->k.c.load_noise_std || h->k.c.pv_noise_std || h->k.c.grid_noise_std)
        return fail(MGX_ERR_UNSUPPORTED, "mgx_normalise_series: forecast noise depends on (step, horizon index): its windows are not "
                                         "slices of one series");
    if (int rc = need_obs_bounds(h, "mgx_normalise_series")) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int32_t R = h->k.T + h->k.H + 1;               // the window at counter value T (after the last step) is all padding
    const unsigned gx = (unsigned)((h->k.N + 63) / 64);
    auto launch = [&](int which, void *out, int nc) {
        const int32_t TR = nc == 1 ? 64 : 16;
        const dim3 grid(gx, (unsigned)((R + TR - 1) / TR));
        const size_t lds = (size_t)nc * TR * 65 * sizeof(double);
        if (nc == 1) {
            if (h->k.obs_f32) normalise_series_kernel<1, float><<<grid, 256, lds, st>>>(h->k, which, (float *)out, R, TR, clipped);
            else normalise_series_kernel<1, double><<<grid, 256, lds, st>>>(h->k, which, (double *)out, R, TR, clipped);
        } else {
            if (h->k.obs_f32) normalise_series_kernel<4, float><<<grid, 256, lds, st>>>(h->k, which, (float *)out, R, TR, clipped);
            else normalise_series_kernel<4, double><<<grid, 256, lds, st>>>(h->k, which, (double *)out, R, TR, clipped);
        }
    };
    launch(0, load_n, 1);
    launch(1, pv_n, 1);
    if (h->layout.has_grid) launch(2, grid_n, 4);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? MGX_OK : hip_fail(e, "normalise_series_kernel launch");
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
