// A consumer of the C ABI that is neither Python nor torch: plain HIP runtime calls for memory, include/mgx.h for everything
// else.  Builds a small genset + battery + load + pv batch on the host, steps it through mgx_step (single steps, normalised
// random controls) and mgx_step_k (the same steps fused; then a twin batch whose series are FACTORISED: base profiles + a profile
// id and ratio per grid, mgx_columns.base_load), and checks every reward and the final state bit for bit against the
// CPU oracle (oracle/mgx_oracle.h -- TEST INFRASTRUCTURE; this file lives under tests/ and is only built and run by
// tests/test_c_abi_consumer.py).  Exit code 0 = identical.
//
// build: hipcc --offload-arch=gfx950 -O2 tests/c_abi/demo.cpp -Iinclude -Ioracle -Lpymgrid_amd -lmgx -Loracle/_build -lmgx_oracle
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

static uint64_t rng_state = 88172645463325252ull;
static double uniform()                                   // xorshift64*: deterministic host-side draws
{
    rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
    return (double)((rng_state * 2685821657736338717ull) >> 11) / 9007199254740992.0;
}

template <typename T>
static T *to_device(const std::vector<T> &v)
{
    T *d = nullptr;
    if (hipMalloc((void **)&d, v.size() * sizeof(T)) != hipSuccess) return nullptr;
    if (hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return d;
}

int main()
{
    const int N = 1000, T = 48, K = 40, A = 3;
    // host columns [N], series [T, N] (stored sign: load <= 0, pv >= 0)
    std::vector<double> cmin(N), cmax(N), cch(N), cdis(N), eta(N), ccost(N), gmin(N), gmax(N), gcost(N), gco2(N), gcco2(N),
        llc(N), ogc(N), charge(N), soc(N), load((size_t)T * N), pv((size_t)T * N);
    std::vector<uint32_t> times(N), status(N);
    for (int i = 0; i < N; i++) {
        cmax[i] = 80 + 200 * uniform(); cmin[i] = 0.2 * cmax[i]; cch[i] = cmax[i] / 4; cdis[i] = cmax[i] / 3;
        eta[i] = 0.8 + 0.2 * uniform(); ccost[i] = 0.02;
        gmax[i] = 60 + 100 * uniform(); gmin[i] = 0.05 * gmax[i]; gcost[i] = 0.4; gco2[i] = 2.0; gcco2[i] = 0.1;
        llc[i] = 10.0; ogc[i] = 1.0;
        soc[i] = 0.3 + 0.6 * uniform(); charge[i] = soc[i] * cmax[i];
        const uint32_t su = (uint32_t)(3 * uniform()), wd = (uint32_t)(3 * uniform());
        times[i] = su | (wd << 16);
        status[i] = (i & 1) ? (1u | (1u << 8) | (wd << 24)) : (su << 16);          // on / off, equilibrium counters
        for (int t = 0; t < T; t++) {
            load[(size_t)t * N + i] = -(20 + 100 * uniform());
            pv[(size_t)t * N + i] = 60 * uniform() * (uniform() > 0.3);
        }
    }
    std::vector<double> actions((size_t)K * N * A);
    for (auto &a : actions) a = uniform();

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { fprintf(stderr, "no HIP device\n"); return 4; }

    mgx_layout L;
    memset(&L, 0, sizeof(L));
    L.struct_size = (int32_t)sizeof(L); L.n_grids = N; L.n_steps = T; L.horizon = 0; L.initial_step = 0; L.final_step = T;
    L.has_genset = 1; L.has_battery = 1; L.has_grid = 0; L.n_load = 1; L.n_pv = 1;
    mgx_columns C;
    memset(&C, 0, sizeof(C));
    C.struct_size = (int32_t)sizeof(C);
    double *d_charge = to_device(charge), *d_soc = to_device(soc);
    uint32_t *d_status = to_device(status);
    C.bat_min_capacity = to_device(cmin); C.bat_max_capacity = to_device(cmax); C.bat_max_charge = to_device(cch);
    C.bat_max_discharge = to_device(cdis); C.bat_efficiency = to_device(eta); C.bat_cost_cycle = to_device(ccost);
    C.gen_running_min = to_device(gmin); C.gen_running_max = to_device(gmax); C.gen_cost = to_device(gcost);
    C.gen_co2_per_unit = to_device(gco2); C.gen_cost_per_unit_co2 = to_device(gcco2); C.gen_times = to_device(times);
    C.loss_load_cost = to_device(llc); C.overgeneration_cost = to_device(ogc);
    C.load_ts = to_device(load); C.pv_ts = to_device(pv);
    C.charge = d_charge; C.soc = d_soc; C.gen_status = d_status;
    double *d_actions = to_device(actions), *d_reward = nullptr;
    uint8_t *d_done = nullptr;
    HIP_OK(hipMalloc((void **)&d_reward, (size_t)K * N * sizeof(double)));
    HIP_OK(hipMalloc((void **)&d_done, (size_t)K * N));
    if (!C.load_ts || !C.pv_ts || !d_actions || !d_status) { fprintf(stderr, "device allocation failed\n"); return 2; }

    if (mgx_abi_version() != MGX_ABI_VERSION) { fprintf(stderr, "ABI %d vs header %d\n", mgx_abi_version(), MGX_ABI_VERSION); return 5; }
    mgx_handle *h = nullptr;
    MGX_CALL(mgx_create(&L, &C, &h));
    if (mgx_action_dim(h) != A) { fprintf(stderr, "action_dim %d\n", mgx_action_dim(h)); return 5; }
    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));

    // (1) K single steps
    for (int k = 0; k < K; k++)
        MGX_CALL(mgx_step(h, d_actions + (size_t)k * N * A, 1, d_reward + (size_t)k * N, d_done + (size_t)k * N, nullptr, nullptr, st));
    HIP_OK(hipStreamSynchronize(st));
    std::vector<double> r_step((size_t)K * N), ch_step(N);
    std::vector<uint8_t> done((size_t)K * N);
    HIP_OK(hipMemcpy(r_step.data(), d_reward, r_step.size() * sizeof(double), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(done.data(), d_done, done.size(), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(ch_step.data(), d_charge, N * sizeof(double), hipMemcpyDeviceToHost));
    if (mgx_current_step(h) != K) { fprintf(stderr, "counter %d\n", mgx_current_step(h)); return 5; }

    // (2) the same K steps in one fused launch, from the initial state
    HIP_OK(hipMemcpy(d_charge, charge.data(), N * sizeof(double), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_soc, soc.data(), N * sizeof(double), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_status, status.data(), N * sizeof(uint32_t), hipMemcpyHostToDevice));
    MGX_CALL(mgx_reset(h, 0, nullptr, st));
    MGX_CALL(mgx_step_k(h, d_actions, K, 1, d_reward, nullptr, nullptr, nullptr, nullptr, nullptr, st));
    HIP_OK(hipStreamSynchronize(st));
    std::vector<double> r_fused((size_t)K * N);
    HIP_OK(hipMemcpy(r_fused.data(), d_reward, r_fused.size() * sizeof(double), hipMemcpyDeviceToHost));

    // (3) a call that must fail cleanly: stepping past the series
    MGX_CALL(mgx_reset(h, T - 1, nullptr, st));
    MGX_CALL(mgx_step(h, d_actions, 1, d_reward, nullptr, nullptr, nullptr, st));
    if (mgx_step(h, d_actions, 1, d_reward, nullptr, nullptr, nullptr, st) != MGX_ERR_RANGE) { fprintf(stderr, "no MGX_ERR_RANGE\n"); return 5; }

    // (4) the same batch with FACTORISED series (mgx_columns.base_load ...): three base profiles [T, MGX_PROFILE_PITCH], a profile
    // id and a ratio per grid; no [T, N] arrays on the device.  K fused steps from the initial state, checked against the oracle on
    // the series the factors stand for (formed here with the same single multiply).
    std::vector<double> base_l((size_t)T * MGX_PROFILE_PITCH, 0.0), base_p((size_t)T * MGX_PROFILE_PITCH, 0.0), lr(N), pr(N);
    std::vector<uint8_t> lp(N), pp(N);
    for (int t = 0; t < T; t++)
        for (int c = 0; c < 3; c++) {
            base_l[(size_t)t * MGX_PROFILE_PITCH + c] = 0.2 + uniform();
            base_p[(size_t)t * MGX_PROFILE_PITCH + c] = uniform() * (uniform() > 0.4);
        }
    std::vector<double> load2((size_t)T * N), pv2((size_t)T * N);
    for (int i = 0; i < N; i++) {
        lp[i] = (uint8_t)(3 * uniform()); pp[i] = (uint8_t)(3 * uniform());
        lr[i] = 30 + 90 * uniform(); pr[i] = 50 * uniform();
        for (int t = 0; t < T; t++) {
            const double l = base_l[(size_t)t * MGX_PROFILE_PITCH + lp[i]] * lr[i], q = base_p[(size_t)t * MGX_PROFILE_PITCH + pp[i]] * pr[i];
            load2[(size_t)t * N + i] = -(l < 0 ? -l : l);
            pv2[(size_t)t * N + i] = q < 0 ? -q : q;
        }
    }
    mgx_columns C2 = C;
    C2.load_ts = nullptr; C2.pv_ts = nullptr;
    C2.base_load = to_device(base_l); C2.base_pv = to_device(base_p);
    C2.load_profile = to_device(lp); C2.pv_profile = to_device(pp);
    C2.load_ratio = to_device(lr); C2.pv_ratio = to_device(pr);
    if (!C2.base_load || !C2.base_pv || !C2.load_profile || !C2.pv_profile || !C2.load_ratio || !C2.pv_ratio) return 2;
    HIP_OK(hipMemcpy(d_charge, charge.data(), N * sizeof(double), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_soc, soc.data(), N * sizeof(double), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_status, status.data(), N * sizeof(uint32_t), hipMemcpyHostToDevice));
    mgx_handle *h2 = nullptr;
    MGX_CALL(mgx_create(&L, &C2, &h2));
    MGX_CALL(mgx_step_k(h2, d_actions, K, 1, d_reward, nullptr, nullptr, nullptr, nullptr, nullptr, st));
    HIP_OK(hipStreamSynchronize(st));
    std::vector<double> r_fact((size_t)K * N);
    HIP_OK(hipMemcpy(r_fact.data(), d_reward, r_fact.size() * sizeof(double), hipMemcpyDeviceToHost));
    // (5) per-grid episodes IN PLACE on the factorised batch (mgx_reset_episodes, ABI v6): grid i starts at its own row
    // start_i = i % (T - KE + 1) and walks KE rows of its own series; single steps from the initial state again
    const int KE = 6;
    std::vector<int32_t> ep_start(N);
    for (int i = 0; i < N; i++) ep_start[i] = i % (T - KE + 1);
    int32_t *d_start = to_device(ep_start), *d_off = nullptr, *d_fin = nullptr;
    uint8_t *d_edone = nullptr;
    HIP_OK(hipMalloc((void **)&d_off, N * sizeof(int32_t)));
    HIP_OK(hipMalloc((void **)&d_fin, N * sizeof(int32_t)));
    HIP_OK(hipMalloc((void **)&d_edone, (size_t)KE * N));
    HIP_OK(hipMemcpy(d_charge, charge.data(), N * sizeof(double), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_soc, soc.data(), N * sizeof(double), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_status, status.data(), N * sizeof(uint32_t), hipMemcpyHostToDevice));
    MGX_CALL(mgx_reset_episodes(h2, d_start, nullptr, KE, d_off, d_fin, nullptr, st));
    for (int k = 0; k < KE; k++)
        MGX_CALL(mgx_step(h2, d_actions + (size_t)k * N * A, 1, d_reward + (size_t)k * N, d_edone + (size_t)k * N, nullptr, nullptr, st));
    HIP_OK(hipStreamSynchronize(st));
    std::vector<double> r_ep((size_t)KE * N);
    std::vector<uint8_t> done_ep((size_t)KE * N);
    HIP_OK(hipMemcpy(r_ep.data(), d_reward, r_ep.size() * sizeof(double), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(done_ep.data(), d_edone, done_ep.size(), hipMemcpyDeviceToHost));
    if (mgx_step_k(h2, d_actions, 2, 1, d_reward, nullptr, nullptr, nullptr, nullptr, nullptr, st) != MGX_ERR_UNSUPPORTED) {
        fprintf(stderr, "fused launches must be refused during rolling episodes\n");
        return 7;
    }
    mgx_destroy(h2);
    // (6) the same episodes IN PLACE on the first batch's [T, N] arrays (round 4: the handle reads a grid-major copy it makes)
    HIP_OK(hipMemcpy(d_charge, charge.data(), N * sizeof(double), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_soc, soc.data(), N * sizeof(double), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_status, status.data(), N * sizeof(uint32_t), hipMemcpyHostToDevice));
    MGX_CALL(mgx_reset_episodes(h, d_start, nullptr, KE, d_off, d_fin, nullptr, st));
    for (int k = 0; k < KE; k++)
        MGX_CALL(mgx_step(h, d_actions + (size_t)k * N * A, 1, d_reward + (size_t)k * N, d_edone + (size_t)k * N, nullptr, nullptr, st));
    HIP_OK(hipStreamSynchronize(st));
    std::vector<double> r_epm((size_t)KE * N);
    std::vector<uint8_t> done_epm((size_t)KE * N);
    HIP_OK(hipMemcpy(r_epm.data(), d_reward, r_epm.size() * sizeof(double), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(done_epm.data(), d_edone, done_epm.size(), hipMemcpyDeviceToHost));

    // the oracle, one microgrid at a time
    long bad = 0;
    for (int i = 0; i < N; i++) {
        orc_grid g;
        memset(&g, 0, sizeof(g));
        g.has_genset = 1; g.has_battery = 1; g.n_load = 1; g.n_pv = 1; g.T = T; g.final_step = T;
        g.bat_min_capacity = cmin[i]; g.bat_max_capacity = cmax[i]; g.bat_max_charge = cch[i]; g.bat_max_discharge = cdis[i];
        g.bat_efficiency = eta[i]; g.bat_cost_cycle = ccost[i];
        g.gen_running_min = gmin[i]; g.gen_running_max = gmax[i]; g.gen_cost = gcost[i]; g.gen_co2_per_unit = gco2[i];
        g.gen_cost_per_unit_co2 = gcco2[i];
        g.gen_start_up_time = (int32_t)(times[i] & 0xff); g.gen_wind_down_time = (int32_t)(times[i] >> 16);
        g.loss_load_cost = llc[i]; g.overgeneration_cost = ogc[i];
        g.load_ts = load.data() + i; g.load_t_stride = N; g.pv_ts = pv.data() + i; g.pv_t_stride = N;
        orc_state s;
        memset(&s, 0, sizeof(s));
        s.charge = charge[i]; s.soc = soc[i];
        s.gen_cur = status[i] & 0xff; s.gen_goal = (status[i] >> 8) & 0xff; s.gen_up = (status[i] >> 16) & 0xff; s.gen_down = status[i] >> 24;
        for (int k = 0; k < K; k++) {
            orc_action a;
            memset(&a, 0, sizeof(a));
            const double *row = actions.data() + ((size_t)k * N + i) * A;
            a.genset[0] = row[0]; a.genset[1] = row[1]; a.battery = row[2];
            orc_step_out o;
            if (orc_run(&g, &s, &a, 1, &o) != 0) { fprintf(stderr, "oracle refused step %d of grid %d\n", k, i); return 6; }
            const size_t j = (size_t)k * N + i;
            bad += (o.reward != r_step[j]) + (o.reward != r_fused[j]) + ((uint8_t)o.done != done[j]);
        }
        bad += s.charge != ch_step[i];
        // ... and on the factorised batch's series
        g.load_ts = load2.data() + i; g.pv_ts = pv2.data() + i;
        memset(&s, 0, sizeof(s));
        s.charge = charge[i]; s.soc = soc[i];
        s.gen_cur = status[i] & 0xff; s.gen_goal = (status[i] >> 8) & 0xff; s.gen_up = (status[i] >> 16) & 0xff; s.gen_down = status[i] >> 24;
        for (int k = 0; k < K; k++) {
            orc_action a;
            memset(&a, 0, sizeof(a));
            const double *row = actions.data() + ((size_t)k * N + i) * A;
            a.genset[0] = row[0]; a.genset[1] = row[1]; a.battery = row[2];
            orc_step_out o;
            if (orc_run(&g, &s, &a, 1, &o) != 0) { fprintf(stderr, "oracle refused step %d of grid %d (factorised)\n", k, i); return 6; }
            bad += o.reward != r_fact[(size_t)k * N + i];
        }
        // ... and the in-place episode: the same microgrid started at its own row (the oracle's window = [start, start + KE))
        g.final_step = ep_start[i] + KE;
        memset(&s, 0, sizeof(s));
        s.t = ep_start[i];
        s.charge = charge[i]; s.soc = soc[i];
        s.gen_cur = status[i] & 0xff; s.gen_goal = (status[i] >> 8) & 0xff; s.gen_up = (status[i] >> 16) & 0xff; s.gen_down = status[i] >> 24;
        for (int k = 0; k < KE; k++) {
            orc_action a;
            memset(&a, 0, sizeof(a));
            const double *row = actions.data() + ((size_t)k * N + i) * A;
            a.genset[0] = row[0]; a.genset[1] = row[1]; a.battery = row[2];
            orc_step_out o;
            if (orc_run(&g, &s, &a, 1, &o) != 0) { fprintf(stderr, "oracle refused step %d of grid %d (episode)\n", k, i); return 6; }
            bad += (o.reward != r_ep[(size_t)k * N + i]) + ((uint8_t)o.done != done_ep[(size_t)k * N + i]);
        }
        // ... and the in-place episode on the [T, N] arrays of the first batch
        g.load_ts = load.data() + i; g.pv_ts = pv.data() + i;
        memset(&s, 0, sizeof(s));
        s.t = ep_start[i];
        s.charge = charge[i]; s.soc = soc[i];
        s.gen_cur = status[i] & 0xff; s.gen_goal = (status[i] >> 8) & 0xff; s.gen_up = (status[i] >> 16) & 0xff; s.gen_down = status[i] >> 24;
        for (int k = 0; k < KE; k++) {
            orc_action a;
            memset(&a, 0, sizeof(a));
            const double *row = actions.data() + ((size_t)k * N + i) * A;
            a.genset[0] = row[0]; a.genset[1] = row[1]; a.battery = row[2];
            orc_step_out o;
            if (orc_run(&g, &s, &a, 1, &o) != 0) { fprintf(stderr, "oracle refused step %d of grid %d (episode, [T, N])\n", k, i); return 6; }
            bad += (o.reward != r_epm[(size_t)k * N + i]) + ((uint8_t)o.done != done_epm[(size_t)k * N + i]);
        }
    }
    mgx_destroy(h);
    printf("c-abi consumer: %d grids x %d steps, single steps, one fused launch, one fused launch on factorised series and in-place "
           "per-grid episodes (factorised and [T, N] series) vs the CPU oracle: %ld mismatches\n", N, K, bad);
    return bad == 0 ? 0 : 1;
}
