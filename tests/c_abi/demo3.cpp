// Third C-ABI consumer (no Python, no torch): the GENERAL path -- microgrids with several modules of a kind (2 gensets + 2 batteries
// + 1 grid per microgrid: the reference's module container keeps a LIST per name, module_container.py:355-413) -- through mgx_step,
// mgx_step_k, mgx_expand_lists, mgx_step_lists (ABI minor 2) and mgx_rollout_lists, each checked bit for bit against the CPU oracle's
// multi-instance restatement (oracle/mgx_oracle.h: orc_mrun, orc_mpopulate_action -- TEST INFRASTRUCTURE; this file is built and run
// only by tests/test_c_abi_consumer.py).  Exit code 0 = identical.
//
// Reference interfaces exercised: Microgrid.run over module lists (microgrid/microgrid.py:227-325), DiscreteMicrogridEnv.step /
// _get_action with priority lists over module instances (envs/discrete/discrete.py:82-143, algos/priority_list/priority_list.py:15-167),
// RuleBasedControl.run (algos/rbc/rbc.py:64-93).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "mgx.h"
#include "mgx_oracle.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define MGX_CALL(x) do { int rc_ = (x); if (rc_ != MGX_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, mgx_last_error()); return 3; } } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static double uniform()
{
    rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
    return (double)((rng_state * 2685821657736338717ull) >> 11) / 9007199254740992.0;
}

template <typename T>
static T *to_device(const std::vector<T> &v)
{
    T *d = nullptr;
    if (hipMalloc((void **)&d, (v.empty() ? 1 : v.size()) * sizeof(T)) != hipSuccess) return nullptr;
    if (!v.empty() && hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return d;
}
template <typename T>
static std::vector<T> to_host(const T *d, size_t n)
{
    std::vector<T> v(n);
    if (hipMemcpy(v.data(), d, n * sizeof(T), hipMemcpyDeviceToHost) != hipSuccess) v.clear();
    return v;
}
template <typename T>
static T *dev_alloc(size_t n) { T *d = nullptr; return hipMalloc((void **)&d, (n ? n : 1) * sizeof(T)) == hipSuccess ? d : nullptr; }

constexpr int NG = 2, NB = 2, NR = 1, A = 2 * NG + NB + NR;

// columns [n, N] (instance-major), series [T, N] (one load, one pv), grid series [T, 4, N]
struct Batch {
    int N, T;
    std::vector<double> cmin, cmax, cch, cdis, eta, ccost, charge, soc;          // [NB, N]
    std::vector<double> gmin, gmax, gcost, gco2, gcco2;                           // [NG, N]
    std::vector<uint32_t> times, status;                                           // [NG, N]
    std::vector<double> imp, exp_, gridco2, llc, ogc, load, pv, grid, load_lo, load_hi, pv_lo, pv_hi, grid_lo, grid_hi;
    mgx_layout L;
    mgx_columns C;

    Batch(int n, int t) : N(n), T(t)
    {
        for (auto *v : {&cmin, &cmax, &cch, &cdis, &eta, &ccost, &charge, &soc}) v->assign((size_t)NB * N, 0.0);
        for (auto *v : {&gmin, &gmax, &gcost, &gco2, &gcco2}) v->assign((size_t)NG * N, 0.0);
        times.assign((size_t)NG * N, 0); status.assign((size_t)NG * N, 0);
        for (auto *v : {&imp, &exp_, &gridco2, &llc, &ogc, &load_lo, &load_hi, &pv_lo, &pv_hi}) v->assign(N, 0.0);
        load.assign((size_t)T * N, 0.0); pv.assign((size_t)T * N, 0.0); grid.assign((size_t)T * 4 * N, 0.0);
        grid_lo.assign(4 * (size_t)N, 1e300); grid_hi.assign(4 * (size_t)N, -1e300);
        for (int i = 0; i < N; i++) {
            for (int j = 0; j < NB; j++) {
                const size_t q = (size_t)j * N + i;
                cmax[q] = 60 + 150 * uniform(); cmin[q] = 0.2 * cmax[q]; cch[q] = cmax[q] / 4; cdis[q] = cmax[q] / 3;
                eta[q] = 0.8 + 0.2 * uniform(); ccost[q] = 0.02 + 0.01 * j;
                soc[q] = 0.3 + 0.6 * uniform(); charge[q] = soc[q] * cmax[q];
            }
            for (int j = 0; j < NG; j++) {
                const size_t q = (size_t)j * N + i;
                gmax[q] = 40 + 60 * uniform(); gmin[q] = ((i + j) % 5 == 0) ? 0.0 : 0.05 * gmax[q]; gcost[q] = 0.4 + 0.1 * j; gco2[q] = 2.0; gcco2[q] = 0.1;
                const uint32_t su = (uint32_t)(3 * uniform()), wd = (uint32_t)(3 * uniform());
                times[q] = su | (wd << 16);
                status[q] = ((i + j) & 1) ? (1u | (1u << 8) | (wd << 24)) : (su << 16);
            }
            imp[i] = 40 + 80 * uniform(); exp_[i] = 30 + 60 * uniform(); gridco2[i] = 0.1; llc[i] = 10.0; ogc[i] = 1.0;
            double llo = 0, lhi = 0, plo = 0, phi = 0;
            for (int r = 0; r < T; r++) {
                const double l = -(30 + 150 * uniform()), p = 70 * uniform() * (uniform() > 0.3);
                load[(size_t)r * N + i] = l; pv[(size_t)r * N + i] = p;
                llo = l < llo ? l : llo; lhi = l > lhi ? l : lhi; plo = p < plo ? p : plo; phi = p > phi ? p : phi;
                const double comp[4] = {0.1 + 0.3 * uniform(), 0.05 * uniform(), 0.2 + 0.3 * uniform(), uniform() > 0.15 ? 1.0 : 0.0};
                for (int c = 0; c < 4; c++) {
                    grid[((size_t)r * 4 + c) * N + i] = comp[c];
                    double &lo = grid_lo[(size_t)c * N + i], &hi = grid_hi[(size_t)c * N + i];
                    lo = comp[c] < lo ? comp[c] : lo; hi = comp[c] > hi ? comp[c] : hi;
                }
            }
            load_lo[i] = llo; load_hi[i] = lhi; pv_lo[i] = plo; pv_hi[i] = phi;
            grid_lo[3 * (size_t)N + i] = 0.0; grid_hi[3 * (size_t)N + i] = 1.0;
        }
        memset(&L, 0, sizeof(L));
        L.struct_size = (int32_t)sizeof(L); L.n_grids = N; L.n_steps = T; L.horizon = 0; L.initial_step = 0; L.final_step = T;
        L.has_genset = 1; L.has_battery = 1; L.has_grid = 1; L.n_load = 1; L.n_pv = 1;
        L.n_genset = NG; L.n_battery = NB; L.n_grid = NR;
        memset(&C, 0, sizeof(C));
        C.struct_size = (int32_t)sizeof(C);
    }

    bool upload()
    {
        C.bat_min_capacity = to_device(cmin); C.bat_max_capacity = to_device(cmax); C.bat_max_charge = to_device(cch);
        C.bat_max_discharge = to_device(cdis); C.bat_efficiency = to_device(eta); C.bat_cost_cycle = to_device(ccost);
        C.gen_running_min = to_device(gmin); C.gen_running_max = to_device(gmax); C.gen_cost = to_device(gcost);
        C.gen_co2_per_unit = to_device(gco2); C.gen_cost_per_unit_co2 = to_device(gcco2); C.gen_times = to_device(times);
        C.gen_status = to_device(status);
        C.grid_max_import = to_device(imp); C.grid_max_export = to_device(exp_); C.grid_cost_per_unit_co2 = to_device(gridco2);
        C.loss_load_cost = to_device(llc); C.overgeneration_cost = to_device(ogc);
        C.load_ts = to_device(load); C.pv_ts = to_device(pv); C.grid_ts = to_device(grid);
        C.load_lo = to_device(load_lo); C.load_hi = to_device(load_hi); C.pv_lo = to_device(pv_lo); C.pv_hi = to_device(pv_hi);
        C.grid_lo = to_device(grid_lo); C.grid_hi = to_device(grid_hi);
        C.charge = to_device(charge); C.soc = to_device(soc);
        return C.load_ts && C.grid_ts && C.charge && C.grid_hi && C.gen_status;
    }

    void oracle_grid(int i, orc_mgrid *g) const
    {
        memset(g, 0, sizeof(*g));
        orc_grid &b = g->base;
        b.has_genset = 1; b.has_battery = 1; b.has_grid = 1; b.n_load = 1; b.n_pv = 1; b.horizon = 0; b.T = T; b.final_step = T;
        b.loss_load_cost = llc[i]; b.overgeneration_cost = ogc[i];
        b.load_ts = load.data() + i; b.load_t_stride = N; b.pv_ts = pv.data() + i; b.pv_t_stride = N;
        b.load_lo = &load_lo[i]; b.load_hi = &load_hi[i]; b.pv_lo = &pv_lo[i]; b.pv_hi = &pv_hi[i];
        g->n_genset = NG; g->n_battery = NB; g->n_grid = NR;
        for (int j = 0; j < NG; j++) {
            const size_t q = (size_t)j * N + i;
            orc_grid &m = g->genset[j];
            m = b;
            m.gen_running_min = gmin[q]; m.gen_running_max = gmax[q]; m.gen_cost = gcost[q]; m.gen_co2_per_unit = gco2[q];
            m.gen_cost_per_unit_co2 = gcco2[q];
            m.gen_start_up_time = (int32_t)(times[q] & 0xff); m.gen_wind_down_time = (int32_t)(times[q] >> 16);
        }
        for (int j = 0; j < NB; j++) {
            const size_t q = (size_t)j * N + i;
            orc_grid &m = g->battery[j];
            m = b;
            m.bat_min_capacity = cmin[q]; m.bat_max_capacity = cmax[q]; m.bat_max_charge = cch[q]; m.bat_max_discharge = cdis[q];
            m.bat_efficiency = eta[q]; m.bat_cost_cycle = ccost[q];
        }
        orc_grid &r = g->grid[0];
        r = b;
        r.grid_max_import = imp[i]; r.grid_max_export = exp_[i]; r.grid_cost_per_unit_co2 = gridco2[i];
        r.grid_ts = grid.data() + i; r.grid_t_stride = 4 * (int64_t)N; r.grid_c_stride = N;
        for (int c = 0; c < 4; c++) { r.grid_lo[c] = grid_lo[(size_t)c * N + i]; r.grid_hi[c] = grid_hi[(size_t)c * N + i]; }
    }

    void oracle_state(int i, int t, orc_mstate *s) const
    {
        memset(s, 0, sizeof(*s));
        s->t = t;
        for (int j = 0; j < NB; j++) { s->battery[j].t = t; s->battery[j].charge = charge[(size_t)j * N + i]; s->battery[j].soc = soc[(size_t)j * N + i]; }
        for (int j = 0; j < NG; j++) {
            const uint32_t w = status[(size_t)j * N + i];
            s->genset[j].t = t;
            s->genset[j].gen_cur = w & 0xff; s->genset[j].gen_goal = (w >> 8) & 0xff; s->genset[j].gen_up = (w >> 16) & 0xff; s->genset[j].gen_down = w >> 24;
        }
    }
};

// device state == the oracle's states
static long state_mismatches(const Batch &b, const std::vector<orc_mstate> &os, const std::vector<uint8_t> &alive)
{
    const int N = b.N;
    const std::vector<double> ch = to_host(b.C.charge, (size_t)NB * N), so = to_host(b.C.soc, (size_t)NB * N);
    const std::vector<uint32_t> gs = to_host(b.C.gen_status, (size_t)NG * N);
    long bad = 0;
    for (int i = 0; i < N; i++) {
        if (!alive[i]) continue;
        for (int j = 0; j < NB; j++) bad += ch[(size_t)j * N + i] != os[i].battery[j].charge || so[(size_t)j * N + i] != os[i].battery[j].soc;
        for (int j = 0; j < NG; j++) {
            const orc_state &g = os[i].genset[j];
            bad += gs[(size_t)j * N + i] != ((uint32_t)g.gen_cur | ((uint32_t)g.gen_goal << 8) | ((uint32_t)g.gen_up << 16) | ((uint32_t)g.gen_down << 24));
        }
    }
    return bad;
}

int main()
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { fprintf(stderr, "no HIP device\n"); return 4; }
    if (mgx_abi_version() != MGX_ABI_VERSION || mgx_abi_minor() < 2) { fprintf(stderr, "ABI %d.%d\n", mgx_abi_version(), mgx_abi_minor()); return 5; }
    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));
    long bad = 0;

    const int N = 700, T = 60, K1 = 6, KK = 9, KD = 7, KR = 11;
    Batch b(N, T);
    if (!b.upload()) { fprintf(stderr, "device allocation failed\n"); return 2; }
    mgx_handle *h = nullptr;
    MGX_CALL(mgx_create(&b.L, &b.C, &h));
    if (mgx_action_dim(h) != A) { fprintf(stderr, "action dim %d vs %d\n", mgx_action_dim(h), A); return 5; }
    std::vector<orc_mgrid> og(N);
    std::vector<orc_mstate> os(N);
    for (int i = 0; i < N; i++) { b.oracle_grid(i, &og[i]); b.oracle_state(i, 0, &os[i]); }
    // a grid in whose state the reference gives up with an AssertionError (orc_mrun -> -3: a lossy battery one ulp over its capacity
    // asked to absorb) leaves the comparison, as in demo2.cpp
    std::vector<uint8_t> alive(N, 1);
    double *d_act = dev_alloc<double>((size_t)KK * N * A), *d_rew = dev_alloc<double>((size_t)KR * N), *d_ctrl = dev_alloc<double>((size_t)N * A);
    uint8_t *d_done = dev_alloc<uint8_t>((size_t)KR * N);
    MGX_CALL(mgx_reset(h, 0, nullptr, st));
    int t = 0;

    // (1) Microgrid.run over module lists: K1 single steps with normalised controls == orc_mrun
    for (int k = 0; k < K1; k++, t++) {
        std::vector<double> act((size_t)N * A);
        for (auto &v : act) v = uniform();
        HIP_OK(hipMemcpy(d_act, act.data(), act.size() * sizeof(double), hipMemcpyHostToDevice));
        MGX_CALL(mgx_step(h, d_act, 1, d_rew, d_done, nullptr, nullptr, st));
        HIP_OK(hipStreamSynchronize(st));
        const std::vector<double> rew = to_host(d_rew, N);
        for (int i = 0; i < N; i++) {
            orc_mstep_out o;
            const int rc = orc_mrun(&og[i], &os[i], &act[(size_t)i * A], 1, &o);
            if (rc == -3) { alive[i] = 0; continue; }
            bad += alive[i] && (rc != 0 || rew[i] != o.common.reward);
        }
    }
    bad += state_mismatches(b, os, alive);
    printf("single steps: %ld mismatches so far\n", bad);

    // (2) the same as ONE launch of KK steps (mgx_step_k), unnormalised controls
    {
        std::vector<double> act((size_t)KK * N * A);
        for (size_t q = 0; q < act.size(); q++) act[q] = (q % A < 2 * NG && (q % A) % 2 == 0) ? uniform() : 90 * uniform() - 30;
        HIP_OK(hipMemcpy(d_act, act.data(), act.size() * sizeof(double), hipMemcpyHostToDevice));
        MGX_CALL(mgx_step_k(h, d_act, KK, 0, d_rew, nullptr, nullptr, nullptr, nullptr, nullptr, st));
        HIP_OK(hipStreamSynchronize(st));
        const std::vector<double> rew = to_host(d_rew, (size_t)KK * N);
        for (int k = 0; k < KK; k++)
            for (int i = 0; i < N; i++) {
                orc_mstep_out o;
                const int rc = orc_mrun(&og[i], &os[i], &act[((size_t)k * N + i) * A], 0, &o);
                if (rc == -3) { alive[i] = 0; continue; }
                bad += alive[i] && (rc != 0 || rew[(size_t)k * N + i] != o.common.reward);
            }
        t += KK;
    }
    bad += state_mismatches(b, os, alive);
    printf("fused steps: %ld mismatches so far\n", bad);

    // priority lists over module instances: 12 hand-made lists (every module once, a duplicate and a padding element in between)
    const int NL = 12, LEN = NG + NB + NR + 2;
    std::vector<int32_t> lists((size_t)NL * LEN * 3, -1);
    std::vector<std::vector<orc_mpl_element>> olists(NL);
    for (int l = 0; l < NL; l++) {
        int mods[5][2] = {{0, 0}, {0, 1}, {1, 0}, {1, 1}, {2, 0}};
        for (int q = 4; q > 0; q--) { const int r = (int)(uniform() * (q + 1)); std::swap(mods[q][0], mods[r][0]); std::swap(mods[q][1], mods[r][1]); }
        int w = 0;
        auto put = [&](int kind, int inst, int act) {
            int32_t *e = &lists[((size_t)l * LEN + w++) * 3];
            e[0] = kind; e[1] = inst; e[2] = act;
        };
        for (int q = 0; q < 5; q++) {
            const int act = mods[q][0] == 0 ? (int)(2 * uniform()) : 0;
            bool seen = false;
            for (const auto &e : olists[l]) seen = seen || (e.kind == mods[q][0] && e.inst == mods[q][1]);
            put(mods[q][0], mods[q][1], act);
            if (!seen) olists[l].push_back({mods[q][0], mods[q][1], act});
            if (q == 1) put(mods[0][0], mods[0][1], 1);              // met again: skipped (priority_list.py:82-88)
            if (q == 2) w++;                                         // padding (-1)
        }
    }
    int32_t *d_lists = to_device(lists), *d_ids = dev_alloc<int32_t>((size_t)N);

    // (3) DiscreteMicrogridEnv.step: mgx_step_lists (one launch; the expanded control comes back too) == orc_mpopulate_action + orc_mrun,
    //     and mgx_expand_lists alone gives the same control
    for (int k = 0; k < KD; k++, t++) {
        std::vector<int32_t> ids(N);
        for (auto &v : ids) v = (int32_t)(uniform() * NL);
        HIP_OK(hipMemcpy(d_ids, ids.data(), ids.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        double *d_ctrl2 = dev_alloc<double>((size_t)N * A);
        MGX_CALL(mgx_expand_lists(h, d_ids, d_lists, NL, LEN, d_ctrl2, nullptr, st));
        MGX_CALL(mgx_step_lists(h, d_ids, d_lists, NL, LEN, d_ctrl, d_rew, d_done, nullptr, nullptr, st));
        HIP_OK(hipStreamSynchronize(st));
        const std::vector<double> rew = to_host(d_rew, N), ctrl = to_host(d_ctrl, (size_t)N * A), ctrl2 = to_host(d_ctrl2, (size_t)N * A);
        HIP_OK(hipFree(d_ctrl2));
        for (int i = 0; i < N; i++) {
            double oa[A];
            const auto &pl = olists[ids[i]];
            const int rp = orc_mpopulate_action(&og[i], &os[i], pl.data(), (int32_t)pl.size(), oa);
            orc_mstep_out o;
            const int rc = orc_mrun(&og[i], &os[i], oa, 0, &o);
            if (rp != 0 || rc == -3) { alive[i] = 0; continue; }
            if (!alive[i]) continue;
            for (int c = 0; c < A; c++) bad += ctrl[(size_t)i * A + c] != oa[c] || ctrl2[(size_t)i * A + c] != oa[c];
            bad += rc != 0 || rew[i] != o.common.reward;
        }
    }
    bad += state_mismatches(b, os, alive);
    printf("discrete steps: %ld mismatches so far\n", bad);

    // (4) RuleBasedControl.run: one fixed list per grid, KR steps in one launch (mgx_rollout_lists)
    {
        std::vector<int32_t> ids(N);
        for (auto &v : ids) v = (int32_t)(uniform() * NL);
        HIP_OK(hipMemcpy(d_ids, ids.data(), ids.size() * sizeof(int32_t), hipMemcpyHostToDevice));
        MGX_CALL(mgx_rollout_lists(h, d_ids, 0, d_lists, NL, LEN, KR, d_rew, d_done, nullptr, nullptr, nullptr, nullptr, st));
        HIP_OK(hipStreamSynchronize(st));
        const std::vector<double> rew = to_host(d_rew, (size_t)KR * N);
        for (int k = 0; k < KR; k++)
            for (int i = 0; i < N; i++) {
                double oa[A];
                const auto &pl = olists[ids[i]];
                const int rp = orc_mpopulate_action(&og[i], &os[i], pl.data(), (int32_t)pl.size(), oa);
                orc_mstep_out o;
                const int rc = orc_mrun(&og[i], &os[i], oa, 0, &o);
                if (rp != 0 || rc == -3) { alive[i] = 0; continue; }
                bad += alive[i] && (rc != 0 || rew[(size_t)k * N + i] != o.common.reward);
            }
        t += KR;
    }
    bad += state_mismatches(b, os, alive);
    int n_alive = 0;
    for (int i = 0; i < N; i++) n_alive += alive[i];
    if (mgx_current_step(h) != t) { fprintf(stderr, "counter %d vs %d\n", mgx_current_step(h), t); bad++; }
    mgx_destroy(h);
    printf("general path (%d gensets + %d batteries + %d grid, %d of %d grids compared): %ld mismatches\n", NG, NB, NR, n_alive, N, bad);
    return bad == 0 && n_alive > N * 9 / 10 ? 0 : 1;
}
