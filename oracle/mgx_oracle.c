/*
 * mgx_oracle.c -- CPU ORACLE (test infrastructure; see mgx_oracle.h for the rules of use).
 *
 * Scalar, fp64, one microgrid at a time.  Every function restates a piece of the pure-Python
 * reference (Total-RD/pymgrid v1.2.2, paths relative to /root/reference/src/pymgrid) and cites it.
 * Compile with -ffp-contract=off: the reference never fuses a multiply-add.
 */
#include "mgx_oracle.h"

#include <math.h>
#include <stddef.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_MAX_ADDENDS 128

/* ------------------------------------------------------------------------------------------- */
/* numpy float64 add.reduce on a contiguous vector: DOUBLE_pairwise_sum (n < 8: running sum;
 * 8 <= n <= 128: eight partial sums combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the
 * tail).  MicrogridStep.balance calls np.sum on the provided/absorbed lists: utils/step.py:33-36. */
double orc_np_sum(const double *a, int32_t n)
{
    if (n < 8) {
        double res = 0.0;
        for (int32_t i = 0; i < n; i++) res += a[i];
        return res;
    }
    double r[8];
    int32_t i;
    for (i = 0; i < 8; i++) r[i] = a[i];
    for (i = 8; i < n - (n % 8); i += 8)
        for (int32_t j = 0; j < 8; j++) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
}

/* Accumulator mirroring MicrogridStep (utils/step.py:4-39). */
typedef struct {
    double provided[ORC_MAX_ADDENDS]; int32_t n_provided;
    double absorbed[ORC_MAX_ADDENDS]; int32_t n_absorbed;
    double reward;
    int32_t done;
} mstep;

static void mstep_append(mstep *m, double reward, int done, int as_source, double energy)
{
    /* step.py:13-31: reward +=, done |=, info['provided_energy'|'absorbed_energy'].append */
    m->reward += reward;
    if (done) m->done = 1;
    if (as_source) m->provided[m->n_provided++] = energy;
    else           m->absorbed[m->n_absorbed++] = energy;
}

/* ------------------------------------------------------------------------------------------- */
/* ModuleSpace, utils/space.py:184-231 */
static double space_spread(double low, double high)
{
    double s = high - low;          /* space.py:204 */
    if (s == 0.0) s = 1.0;          /* space.py:205 */
    return s;
}
static double space_denormalize(double low, double high, double v)
{
    return low + space_spread(low, high) * v;   /* space.py:224 */
}
static double space_normalize(double low, double high, double v)
{
    return (v - low) / space_spread(low, high); /* space.py:213 */
}

/* BaseTimeSeriesMicrogridModule._done, modules/base/timeseries/base_timeseries_module.py:124-125 */
static int ts_done(const orc_grid *g, int32_t t) { return t >= g->final_step - 1; }

/* ------------------------------------------------------------------------------------------- */
/* BatteryModule, modules/battery_module.py */
static double battery_max_production(const orc_grid *g, const orc_state *s)
{   /* :283-286 */
    double a = g->bat_max_discharge, b = s->charge - g->bat_min_capacity;
    return (b < a ? b : a) * g->bat_efficiency;      /* Python min(a, b): b if b < a else a */
}
static double battery_max_consumption(const orc_grid *g, const orc_state *s)
{   /* :288-291 */
    double a = g->bat_max_charge, b = g->bat_max_capacity - s->charge;
    return (b < a ? b : a) / g->bat_efficiency;
}
static double battery_transition(const orc_grid *g, double external)
{   /* default_transition_model :244-278 */
    if (external < 0) return external / g->bat_efficiency;
    return external * g->bat_efficiency;
}
static void battery_update_state(const orc_grid *g, orc_state *s, double energy_change)
{   /* _update_state :125-130 */
    s->charge += energy_change;
    if (s->charge < g->bat_min_capacity) s->charge = g->bat_min_capacity;
    s->soc = s->charge / g->bat_max_capacity;
}
/* BaseMicrogridModule.step for the battery: base_module.py:138-171 + as_source/as_sink :210-274
 * + BatteryModule.update :108-123 */
static int battery_step(const orc_grid *g, orc_state *s, double action, int normalized,
                        mstep *m, orc_step_out *out)
{
    double x = action;
    if (normalized) {
        double lo = -g->bat_max_discharge / g->bat_efficiency;   /* min_act :332-334 */
        double hi = g->bat_max_charge * g->bat_efficiency;       /* max_act :336-338 */
        x = space_denormalize(lo, hi, action);
    }
    out->soc_pre = s->soc;                /* state_dict() before the step, base_module.py:152 */
    out->charge_pre = s->charge;
    out->discharge_amount = 0.0;          /* _log: missing energy key -> 0.0, base_module.py:279-287 */
    out->charge_amount = 0.0;
    double internal, e;
    if (x < 0) {                          /* as_sink(-1.0*x), base_module.py:164-165, :262-274 */
        double ex = -1.0 * x, mc = battery_max_consumption(g, s);
        e = (ex > mc) ? mc : ex;
        if (!(e >= 0)) return -3;         /* `assert absorbed_energy >= 0`, base_module.py:272 */
        internal = battery_transition(g, e);
        battery_update_state(g, s, internal);
        out->charge_amount = e;
        out->battery_reward = -1.0 * (fabs(internal) * g->bat_cost_cycle);   /* get_cost :132-147 */
        mstep_append(m, out->battery_reward, 0, 0, e);
    } else {                              /* x > 0, or x == 0 and is_source: as_source(x) :210-226 */
        double mp = battery_max_production(g, s);
        if (x > mp) e = mp;
        else if (x < 0.0) e = 0.0;        /* min_production == 0, base_module.py:604-619 */
        else e = x;
        internal = battery_transition(g, -1.0 * e);
        if (!(internal <= 0)) return -3;  /* `assert internal_energy_change <= 0`, battery_module.py:114: max_production is
                                           * negative (charge below min_capacity) and the clip handed it on */
        battery_update_state(g, s, internal);
        out->discharge_amount = e;
        out->battery_reward = -1.0 * (fabs(internal) * g->bat_cost_cycle);
        mstep_append(m, out->battery_reward, 0, 1, e);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* GensetModule, modules/genset_module.py */
static void genset_reset_up_down_times(const orc_grid *g, orc_state *s)
{   /* _reset_up_down_times :216-227 (callers guarantee goal == cur) */
    if (s->gen_cur) { s->gen_up = 0; s->gen_down = g->gen_wind_down_time; }
    else            { s->gen_down = 0; s->gen_up = g->gen_start_up_time; }
}
static int genset_finish_in_progress_change(const orc_grid *g, orc_state *s)
{   /* :302-311 */
    if (s->gen_up == 0 && s->gen_goal == 1)   { s->gen_cur = 1; genset_reset_up_down_times(g, s); return 1; }
    if (s->gen_down == 0 && s->gen_goal == 0) { s->gen_cur = 0; genset_reset_up_down_times(g, s); return 1; }
    return 0;
}
static void genset_non_instantaneous_update(const orc_grid *g, orc_state *s, int32_t goal)
{   /* :327-346 */
    if (goal == s->gen_cur && s->gen_cur != s->gen_goal && !g->gen_no_abortion) {   /* abort: only if allow_abortion */
        s->gen_goal = goal;
        genset_reset_up_down_times(g, s);
    } else if (s->gen_cur == s->gen_goal && s->gen_goal != goal) {
        genset_reset_up_down_times(g, s);
        s->gen_goal = goal;
    }
    if (s->gen_goal != s->gen_cur) {          /* _update_up_down_times :229-233 */
        if (s->gen_goal == 0) s->gen_down -= 1;
        else                  s->gen_up -= 1;
    }
}
int32_t orc_genset_next_status(const orc_state *s, int32_t goal_status)
{   /* next_status :360-390 */
    if (goal_status) {
        if (s->gen_cur) return 1;
        return s->gen_up == 0 ? 1 : 0;
    }
    if (!s->gen_cur) return 0;
    return s->gen_down == 0 ? 0 : 1;
}
void orc_genset_update_status(const orc_grid *g, orc_state *s, double goal_status)
{   /* update_status :235-300 */
    int32_t goal = (goal_status > 0.5) ? 1 : 0;     /* Python round(): half-to-even, 0.5 -> 0 (:281) */
    if (goal == s->gen_cur && s->gen_cur == s->gen_goal) return;          /* :284-287 */
    /* :289-292 -- the new goal is taken if abortion is allowed or the change is instantaneous */
    int instant_up = g->gen_start_up_time == 0 && goal == 1;
    int instant_down = g->gen_wind_down_time == 0 && goal == 0;
    if (goal != s->gen_goal && (!g->gen_no_abortion || instant_up || instant_down)) s->gen_goal = goal;
    if (!genset_finish_in_progress_change(g, s))                            /* :294 */
        genset_non_instantaneous_update(g, s, goal);                       /* :296-297 */
}
static void genset_step(const orc_grid *g, orc_state *s, const double action[2], int normalized,
                        mstep *m, orc_step_out *out)
{
    /* GensetModule.step :146-149: status first (raw action[0]), then BaseMicrogridModule.step */
    orc_genset_update_status(g, s, action[0]);
    double x = action[1];
    if (normalized)   /* action space low [0,0], high [1, running_max] (:511-517); _energy_pos = 1 (:59) */
        x = space_denormalize(0.0, g->gen_running_max, action[1]);
    /* state_dict() is taken after update_status -> log shows the new status */
    out->gen_cur = s->gen_cur; out->gen_goal = s->gen_goal; out->gen_up = s->gen_up; out->gen_down = s->gen_down;
    /* genset is a source only: x > 0 or x == 0 -> as_source (base_module.py:161-171); x < 0 would assert */
    double mx = s->gen_cur * g->gen_running_max;    /* max_production :465-482 */
    double mn = s->gen_cur * g->gen_running_min;    /* min_production :484-501 */
    double e;
    if (x > mx) e = mx;                              /* base_module.py:213-224 */
    else if (x < mn) e = mn;
    else e = x;
    /* update :207-214, get_cost :188-205 */
    double co2 = g->gen_co2_per_unit * e;                                   /* get_co2 :151-166 */
    double cost = g->gen_cost * e + g->gen_cost_per_unit_co2 * co2;         /* _get_fuel_cost + get_co2_cost */
    out->genset_production = e;
    out->genset_co2_production = co2;
    out->genset_reward = -1.0 * cost;
    mstep_append(m, out->genset_reward, 0, 1, e);
}

/* ------------------------------------------------------------------------------------------- */
/* GridModule, modules/grid_module.py */
static double grid_comp(const orc_grid *g, int32_t t, int c)
{
    return g->grid_ts[(int64_t)t * g->grid_t_stride + (int64_t)c * g->grid_c_stride];
}
static void grid_step(const orc_grid *g, const orc_state *s, double action, int normalized,
                      mstep *m, orc_step_out *out)
{
    int32_t t = s->t;
    double x = action;
    if (normalized)   /* _get_bounds :125-132: min_act = -1*max_export, max_act = max_import */
        x = space_denormalize(-1 * g->grid_max_export, g->grid_max_import, action);
    double status = grid_comp(g, t, 3);            /* current_status :300-312 */
    int done = ts_done(g, t);
    out->grid_import = 0.0; out->grid_export = 0.0;
    if (x < 0) {                                   /* as_sink */
        double ex = -1.0 * x, mc = g->grid_max_export * status;            /* max_consumption :318-320 */
        double e = (ex > mc) ? mc : ex;
        double co2 = 0.0;                                                  /* get_co2_production :225-226 */
        double r = grid_comp(g, t, 1) * e + (-1.0 * g->grid_cost_per_unit_co2 * co2);   /* get_cost :169-171 */
        out->grid_export = e; out->grid_co2_production = co2; out->grid_reward = r;
        mstep_append(m, r, done, 0, e);
    } else {                                       /* as_source */
        double mp = g->grid_max_import * status;                           /* max_production :314-316 */
        double e;
        if (x > mp) e = mp; else if (x < 0.0) e = 0.0; else e = x;
        double co2 = e * grid_comp(g, t, 2);                               /* :221-224 */
        double r = -1 * grid_comp(g, t, 0) * e + (-1.0 * g->grid_cost_per_unit_co2 * co2);   /* :166-168, :176-197 */
        out->grid_import = e; out->grid_co2_production = co2; out->grid_reward = r;
        mstep_append(m, r, done, 1, e);
    }
}

/* ------------------------------------------------------------------------------------------- */
int orc_run(const orc_grid *g, orc_state *s, const orc_action *a, int normalized, orc_step_out *out)
{
    int32_t t = s->t;
    if (t < 0 || t >= g->T) return -2;
    if (g->n_load + g->n_pv + 6 > ORC_MAX_ADDENDS) return -2;
    memset(out, 0, sizeof(*out));
    mstep m; m.n_provided = 0; m.n_absorbed = 0; m.reward = 0.0; m.done = 0;
    int done_ts = ts_done(g, t);

    /* fixed modules: LoadModule.update, load_module.py:86-111   (microgrid.py:255-257) */
    for (int32_t j = 0; j < g->n_load; j++) {
        double cur = -1 * g->load_ts[(int64_t)t * g->load_t_stride + (int64_t)j * g->load_m_stride];
        out->load_met += cur;
        mstep_append(&m, 0.0, done_ts, 0, cur);
    }
    double fixed_provided = orc_np_sum(m.provided, m.n_provided);    /* microgrid.py:259 */
    double fixed_consumed = orc_np_sum(m.absorbed, m.n_absorbed);
    out->fixed_provided = fixed_provided; out->fixed_absorbed = fixed_consumed;

    /* controllable modules in container order: sources (genset) then source_and_sinks
     * (battery, grid)  -- module_container.py:355-413, microgrid.py:262-275 */
    if (g->has_genset)  genset_step(g, s, a->genset, normalized, &m, out);
    if (g->grid_before_battery && g->has_grid) grid_step(g, s, a->grid, normalized, &m, out);
    if (g->has_battery && battery_step(g, s, a->battery, normalized, &m, out) != 0) return -3;
    if (!g->grid_before_battery && g->has_grid) grid_step(g, s, a->grid, normalized, &m, out);

    double provided = orc_np_sum(m.provided, m.n_provided);          /* microgrid.py:277 */
    double consumed = orc_np_sum(m.absorbed, m.n_absorbed);
    double difference = provided - consumed;                         /* :278 */
    out->controllable_provided = provided - fixed_provided;          /* :281 */
    out->controllable_absorbed = consumed - fixed_consumed;

    /* flex modules: renewable(s) (flex sources) then unbalanced energy (flex source_and_sink) */
    if (difference > 0) {                                            /* :286-299 */
        double energy_excess = difference;
        for (int32_t j = 0; j < g->n_pv; j++) {
            double pv = g->pv_ts[(int64_t)t * g->pv_t_stride + (int64_t)j * g->pv_m_stride];
            /* not a sink: step(0.0) -> as_source(0.0) -> RenewableModule.update renewable_module.py:86-93 */
            out->curtailment += pv - 0.0;
            mstep_append(&m, 0.0, done_ts, 1, 0.0);
            energy_excess += 0.0;
        }
        /* unbalanced: max_consumption inf -> sink_amt = -1.0*excess -> as_sink(excess) */
        double sink_amt = -1.0 * energy_excess;
        double e = -1.0 * sink_amt;
        out->overgeneration = e; out->loss_load = 0.0;
        out->unbalanced_reward = -1.0 * (g->overgeneration_cost * e);   /* unbalanced_energy_module.py:28-70 */
        mstep_append(&m, out->unbalanced_reward, 0, 0, e);
    } else {                                                         /* :301-314 */
        double energy_needed = -difference;
        for (int32_t j = 0; j < g->n_pv; j++) {
            double pv = g->pv_ts[(int64_t)t * g->pv_t_stride + (int64_t)j * g->pv_m_stride];
            double amt = (pv < energy_needed) ? pv : energy_needed;
            /* as_source(amt): amt <= max_production and >= 0, so no clipping applies */
            out->renewable_used += amt;
            out->curtailment += pv - amt;
            mstep_append(&m, 0.0, done_ts, 1, amt);
            energy_needed -= amt;
        }
        double e = energy_needed;            /* unbalanced max_production inf -> source_amt = energy_needed */
        out->loss_load = e; out->overgeneration = 0.0;
        out->unbalanced_reward = -1.0 * (g->loss_load_cost * e);
        mstep_append(&m, out->unbalanced_reward, 0, 1, e);
    }

    out->overall_provided = orc_np_sum(m.provided, m.n_provided);    /* :316-317 */
    out->overall_absorbed = orc_np_sum(m.absorbed, m.n_absorbed);
    out->reward = m.reward;
    out->done = m.done;
    s->t = t + 1;                                                    /* every module._update_step */

    /* np.isclose(provided, consumed), microgrid.py:321-323 */
    if (!(fabs(out->overall_provided - out->overall_absorbed) <= 1e-8 + 1e-5 * fabs(out->overall_absorbed)))
        return -1;
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
int32_t orc_obs_dim(const orc_grid *g)
{
    int32_t w = 1 + g->horizon;
    return g->n_load * w + g->n_pv * w + 4 * g->has_genset + 2 * g->has_battery + 4 * w * g->has_grid;
}

/* one component of a time-series observation window: current_obs + forecast
 * (base_timeseries_module.py:103-140, forecaster.py:120-149,172-187,215-217) */
static void ts_window(const double *ts, int64_t t_stride, int32_t T, int32_t t, int32_t H,
                      double lo, double hi, double *obs, int32_t obs_stride)
{
    double fill = (hi + lo) / 2;                     /* Forecaster._fill_arr, forecaster.py:95 */
    for (int32_t h = 0; h <= H; h++) {
        double v;
        if (t < T && t + h < T) {
            v = ts[(int64_t)(t + h) * t_stride];
            if (h > 0) {                             /* _clip, forecaster.py:139-149 */
                if (v < lo) v = lo;
                if (v > hi) v = hi;
            }
        } else {
            v = fill;                                /* _pad / full_pad, forecaster.py:120-137 */
        }
        obs[(int64_t)h * obs_stride] = space_normalize(lo, hi, v);
    }
}

void orc_observe(const orc_grid *g, const orc_state *s, double *obs)
{
    int32_t H = g->horizon, w = 1 + H, k = 0;
    for (int32_t j = 0; j < g->n_load; j++, k += w)
        ts_window(g->load_ts + (int64_t)j * g->load_m_stride, g->load_t_stride, g->T, s->t, H,
                  g->load_lo[j], g->load_hi[j], obs + k, 1);
    for (int32_t j = 0; j < g->n_pv; j++, k += w)
        ts_window(g->pv_ts + (int64_t)j * g->pv_m_stride, g->pv_t_stride, g->T, s->t, H,
                  g->pv_lo[j], g->pv_hi[j], obs + k, 1);
    if (g->has_genset) {        /* min_obs/max_obs genset_module.py:503-509 */
        obs[k++] = space_normalize(0.0, 1.0, (double)s->gen_cur);
        obs[k++] = space_normalize(0.0, 1.0, (double)s->gen_goal);
        obs[k++] = space_normalize(0.0, (double)g->gen_start_up_time, (double)s->gen_up);
        obs[k++] = space_normalize(0.0, (double)g->gen_wind_down_time, (double)s->gen_down);
    }
    if (g->has_battery) {       /* battery_module.py:87,323-330: [soc, current_charge] */
        double min_soc = g->bat_min_capacity / g->bat_max_capacity;
        obs[k++] = space_normalize(min_soc, 1.0, s->soc);
        obs[k++] = space_normalize(g->bat_min_capacity, g->bat_max_capacity, s->charge);
    }
    if (g->has_grid) {          /* layout [c0..c3]_cur, [c0..c3]_+1, ...  base_timeseries_module.py:162-170 */
        for (int c = 0; c < 4; c++)
            ts_window(g->grid_ts + (int64_t)c * g->grid_c_stride, g->grid_t_stride, g->T, s->t, H,
                      g->grid_lo[c], g->grid_hi[c], obs + k + c, 4);
        k += 4 * w;
    }
}

/* ------------------------------------------------------------------------------------------- */
/* PriorityListAlgo._populate_action, algos/priority_list/priority_list.py:69-167.
 * Returns 0, or the line of the reference's `assert` that fails in this state (the reference raises AssertionError there and
 * returns no control): 73 `total_load >= 0 and renewable >= 0`, 121 `remaining_load <= 0.0`, 124 `module_max_consumption >= 0`
 * (a lossy battery whose charge sits one ulp above max_capacity), 135 `module_consumption <= 0`, 154 `module_production >= 0`
 * (a battery whose charge sits below min_capacity and is the first to produce).  `out` then holds the control built so far. */
int orc_populate_action(const orc_grid *g, const orc_state *s,
                        const orc_pl_element *plist, int32_t n_elements, orc_action *out)
{
    int32_t t = s->t;
    double total_load = 0.0;                                  /* _get_load :157-164 */
    for (int32_t j = 0; j < g->n_load; j++)
        total_load += -1 * g->load_ts[(int64_t)t * g->load_t_stride + (int64_t)j * g->load_m_stride];
    double pvs[ORC_MAX_ADDENDS];                              /* _get_renewable :166-167 (np.sum) */
    for (int32_t j = 0; j < g->n_pv; j++)
        pvs[j] = g->pv_ts[(int64_t)t * g->pv_t_stride + (int64_t)j * g->pv_m_stride];
    double renewable = orc_np_sum(pvs, g->n_pv);
    memset(out, 0, sizeof(*out));
    if (!(total_load >= 0 && renewable >= 0)) return 73;      /* :73 */
    double remaining = total_load - renewable;                /* :74 */

    int genset_set = 0, battery_set = 0, grid_set = 0;
    for (int32_t k = 0; k < n_elements; k++) {
        int32_t mod = plist[k].module, act = plist[k].action;
        if (mod == 0) { if (genset_set) continue; genset_set = 1; out->genset[0] = (double)act; }   /* :82-88 */
        else if (mod == 1) { if (battery_set) continue; battery_set = 1; }
        else { if (grid_set) continue; grid_set = 1; }
        double energy;
        if (fabs(remaining - 0.0) <= 1e-4 + 1e-5 * fabs(0.0)) {       /* np.isclose(.,0,atol=1e-4) :90 */
            energy = 0.0;
        } else if (remaining > 0) {                                   /* _produce_from_module :138-155 */
            double mx, mn;
            if (mod == 0) {
                int32_t ns = orc_genset_next_status(s, act);          /* next_max/min_production genset_module.py:392-424 */
                mx = ns * g->gen_running_max; mn = ns * g->gen_running_min;
            } else if (mod == 1) {
                mx = battery_max_production(g, s); mn = 0.0;
            } else {
                mx = g->grid_max_import * grid_comp(g, t, 3); mn = 0.0;
            }
            if (mn <= remaining && remaining <= mx) energy = remaining;
            else if (remaining < mn) energy = mn;
            else energy = mx;
            if (!(energy >= 0)) return 154;                           /* `assert module_production >= 0` :154 */
        } else {                                                      /* _consume_in_module :118-136 */
            if (!(remaining <= 0.0)) return 121;                      /* `assert remaining_load <= 0.0` :121 (NaN only) */
            if (mod == 0) energy = 0.0;                               /* not a sink */
            else {
                double mc = (mod == 1) ? battery_max_consumption(g, s) : g->grid_max_export * grid_comp(g, t, 3);
                if (!(mc >= 0)) return 124;                           /* `assert module_max_consumption >= 0` :124 */
                energy = (-1 * remaining > mc) ? -1.0 * mc : remaining;
            }
            if (!(energy <= 0)) return 135;                           /* `assert module_consumption <= 0` :135 */
        }
        if (mod == 0) out->genset[1] = energy;
        else if (mod == 1) out->battery = energy;
        else out->grid = energy;
        remaining -= energy;                                          /* :105 */
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* Several gensets / batteries / grids per microgrid: the same sweep with a loop per module list
 * (microgrid.py:262-275: `for name, modules in self.controllable.iterdict(): ... for module, _control in _zip`). */
static void mstate_view(const orc_mstate *s, const orc_state *inst, orc_state *v)
{
    *v = *inst; v->t = s->t;
}

int orc_mrun(const orc_mgrid *mg, orc_mstate *s, const double *actions, int normalized, orc_mstep_out *out)
{
    const orc_grid *g = &mg->base;
    int32_t t = s->t;
    if (t < 0 || t >= g->T) return -2;
    if (mg->n_genset > ORC_MAX_INST || mg->n_battery > ORC_MAX_INST || mg->n_grid > ORC_MAX_INST) return -2;
    if (g->n_load + g->n_pv + mg->n_genset + mg->n_battery + mg->n_grid + 2 > ORC_MAX_ADDENDS) return -2;
    memset(out, 0, sizeof(*out));
    orc_step_out *oc = &out->common;
    mstep m; m.n_provided = 0; m.n_absorbed = 0; m.reward = 0.0; m.done = 0;
    int done_ts = ts_done(g, t);

    for (int32_t j = 0; j < g->n_load; j++) {                         /* fixed modules, microgrid.py:255-257 */
        double cur = -1 * g->load_ts[(int64_t)t * g->load_t_stride + (int64_t)j * g->load_m_stride];
        oc->load_met += cur;
        mstep_append(&m, 0.0, done_ts, 0, cur);
    }
    double fixed_provided = orc_np_sum(m.provided, m.n_provided);
    double fixed_consumed = orc_np_sum(m.absorbed, m.n_absorbed);
    oc->fixed_provided = fixed_provided; oc->fixed_absorbed = fixed_consumed;

    /* controllable: pure sources (gensets), then the source-and-sink names in list order, each name's modules in order */
    const double *a_gen = actions, *a_bat = actions + 2 * mg->n_genset, *a_grid = a_bat + mg->n_battery;
    for (int32_t j = 0; j < mg->n_genset; j++) {
        orc_state v; mstate_view(s, &s->genset[j], &v);
        genset_step(&mg->genset[j], &v, a_gen + 2 * j, normalized, &m, &out->genset[j]);
        s->genset[j] = v;
    }
    for (int pass = 0; pass < 2; pass++) {
        int grids_now = (pass == 0) == (g->grid_before_battery != 0);
        if (grids_now) {
            for (int32_t j = 0; j < mg->n_grid; j++) {
                orc_state v; memset(&v, 0, sizeof(v)); v.t = s->t;
                grid_step(&mg->grid[j], &v, a_grid[j], normalized, &m, &out->grid[j]);
            }
        } else {
            for (int32_t j = 0; j < mg->n_battery; j++) {
                orc_state v; mstate_view(s, &s->battery[j], &v);
                if (battery_step(&mg->battery[j], &v, a_bat[j], normalized, &m, &out->battery[j]) != 0) return -3;
                s->battery[j] = v;
            }
        }
    }
    double provided = orc_np_sum(m.provided, m.n_provided);          /* microgrid.py:277 */
    double consumed = orc_np_sum(m.absorbed, m.n_absorbed);
    double difference = provided - consumed;
    oc->controllable_provided = provided - fixed_provided;
    oc->controllable_absorbed = consumed - fixed_consumed;

    if (difference > 0) {                                            /* :286-299 */
        double energy_excess = difference;
        for (int32_t j = 0; j < g->n_pv; j++) {
            double pv = g->pv_ts[(int64_t)t * g->pv_t_stride + (int64_t)j * g->pv_m_stride];
            oc->curtailment += pv - 0.0;
            mstep_append(&m, 0.0, done_ts, 1, 0.0);
            energy_excess += 0.0;
        }
        double sink_amt = -1.0 * energy_excess;
        double e = -1.0 * sink_amt;
        oc->overgeneration = e; oc->loss_load = 0.0;
        oc->unbalanced_reward = -1.0 * (g->overgeneration_cost * e);
        mstep_append(&m, oc->unbalanced_reward, 0, 0, e);
    } else {                                                         /* :301-314 */
        double energy_needed = -difference;
        for (int32_t j = 0; j < g->n_pv; j++) {
            double pv = g->pv_ts[(int64_t)t * g->pv_t_stride + (int64_t)j * g->pv_m_stride];
            double amt = (pv < energy_needed) ? pv : energy_needed;
            oc->renewable_used += amt;
            oc->curtailment += pv - amt;
            mstep_append(&m, 0.0, done_ts, 1, amt);
            energy_needed -= amt;
        }
        double e = energy_needed;
        oc->loss_load = e; oc->overgeneration = 0.0;
        oc->unbalanced_reward = -1.0 * (g->loss_load_cost * e);
        mstep_append(&m, oc->unbalanced_reward, 0, 1, e);
    }
    oc->overall_provided = orc_np_sum(m.provided, m.n_provided);
    oc->overall_absorbed = orc_np_sum(m.absorbed, m.n_absorbed);
    oc->reward = m.reward;
    oc->done = m.done;
    s->t = t + 1;
    if (!(fabs(oc->overall_provided - oc->overall_absorbed) <= 1e-8 + 1e-5 * fabs(oc->overall_absorbed)))
        return -1;
    return 0;
}

int32_t orc_mobs_dim(const orc_mgrid *mg)
{
    int32_t w = 1 + mg->base.horizon;
    return (mg->base.n_load + mg->base.n_pv) * w + 4 * mg->n_genset + 2 * mg->n_battery + 4 * w * mg->n_grid;
}

void orc_mobserve(const orc_mgrid *mg, const orc_mstate *s, double *obs)
{
    const orc_grid *g = &mg->base;
    int32_t H = g->horizon, w = 1 + H, k = 0;
    for (int32_t j = 0; j < g->n_load; j++, k += w)
        ts_window(g->load_ts + (int64_t)j * g->load_m_stride, g->load_t_stride, g->T, s->t, H,
                  g->load_lo[j], g->load_hi[j], obs + k, 1);
    for (int32_t j = 0; j < g->n_pv; j++, k += w)
        ts_window(g->pv_ts + (int64_t)j * g->pv_m_stride, g->pv_t_stride, g->T, s->t, H,
                  g->pv_lo[j], g->pv_hi[j], obs + k, 1);
    for (int32_t j = 0; j < mg->n_genset; j++) {
        const orc_grid *q = &mg->genset[j]; const orc_state *v = &s->genset[j];
        obs[k++] = space_normalize(0.0, 1.0, (double)v->gen_cur);
        obs[k++] = space_normalize(0.0, 1.0, (double)v->gen_goal);
        obs[k++] = space_normalize(0.0, (double)q->gen_start_up_time, (double)v->gen_up);
        obs[k++] = space_normalize(0.0, (double)q->gen_wind_down_time, (double)v->gen_down);
    }
    for (int32_t j = 0; j < mg->n_battery; j++) {
        const orc_grid *q = &mg->battery[j]; const orc_state *v = &s->battery[j];
        double min_soc = q->bat_min_capacity / q->bat_max_capacity;
        obs[k++] = space_normalize(min_soc, 1.0, v->soc);
        obs[k++] = space_normalize(q->bat_min_capacity, q->bat_max_capacity, v->charge);
    }
    for (int32_t j = 0; j < mg->n_grid; j++) {
        const orc_grid *q = &mg->grid[j];
        for (int c = 0; c < 4; c++)
            ts_window(q->grid_ts + (int64_t)c * q->grid_c_stride, q->grid_t_stride, g->T, s->t, H,
                      q->grid_lo[c], q->grid_hi[c], obs + k + c, 4);
        k += 4 * w;
    }
}

/* PriorityListAlgo._populate_action over module instances, priority_list.py:69-116 */
int orc_mpopulate_action(const orc_mgrid *mg, const orc_mstate *s, const orc_mpl_element *plist, int32_t n_elements,
                         double *actions)
{
    const orc_grid *g = &mg->base;
    int32_t t = s->t;
    double total_load = 0.0;
    for (int32_t j = 0; j < g->n_load; j++)
        total_load += -1 * g->load_ts[(int64_t)t * g->load_t_stride + (int64_t)j * g->load_m_stride];
    double pvs[ORC_MAX_ADDENDS];
    for (int32_t j = 0; j < g->n_pv; j++)
        pvs[j] = g->pv_ts[(int64_t)t * g->pv_t_stride + (int64_t)j * g->pv_m_stride];
    double renewable = orc_np_sum(pvs, g->n_pv);
    int set[3][ORC_MAX_INST];
    memset(set, 0, sizeof(set));
    double *a_gen = actions, *a_bat = actions + 2 * mg->n_genset, *a_grid = a_bat + mg->n_battery;
    for (int32_t j = 0; j < 2 * mg->n_genset + mg->n_battery + mg->n_grid; j++) actions[j] = 0.0;
    if (!(total_load >= 0 && renewable >= 0)) return 73;              /* the asserts: see orc_populate_action */
    double remaining = total_load - renewable;
    for (int32_t k = 0; k < n_elements; k++) {
        int32_t kind = plist[k].kind, j = plist[k].inst, act = plist[k].action;
        if (set[kind][j]) continue;                                   /* :82-88 */
        set[kind][j] = 1;
        if (kind == 0) a_gen[2 * j] = (double)act;
        orc_state v;
        const orc_grid *q;
        if (kind == 0) { q = &mg->genset[j]; mstate_view(s, &s->genset[j], &v); }
        else if (kind == 1) { q = &mg->battery[j]; mstate_view(s, &s->battery[j], &v); }
        else { q = &mg->grid[j]; memset(&v, 0, sizeof(v)); v.t = t; }
        double energy;
        if (fabs(remaining - 0.0) <= 1e-4 + 1e-5 * fabs(0.0)) {
            energy = 0.0;
        } else if (remaining > 0) {
            double mx, mn;
            if (kind == 0) {
                int32_t ns = orc_genset_next_status(&v, act);
                mx = ns * q->gen_running_max; mn = ns * q->gen_running_min;
            } else if (kind == 1) {
                mx = battery_max_production(q, &v); mn = 0.0;
            } else {
                mx = q->grid_max_import * grid_comp(q, t, 3); mn = 0.0;
            }
            if (mn <= remaining && remaining <= mx) energy = remaining;
            else if (remaining < mn) energy = mn;
            else energy = mx;
            if (!(energy >= 0)) return 154;                           /* :154 */
        } else {
            if (!(remaining <= 0.0)) return 121;                      /* :121 */
            if (kind == 0) energy = 0.0;
            else {
                double mc = (kind == 1) ? battery_max_consumption(q, &v) : q->grid_max_export * grid_comp(q, t, 3);
                if (!(mc >= 0)) return 124;                           /* :124 */
                energy = (-1 * remaining > mc) ? -1.0 * mc : remaining;
            }
            if (!(energy <= 0)) return 135;                           /* :135 */
        }
        if (kind == 0) a_gen[2 * j + 1] = energy;
        else if (kind == 1) a_bat[j] = energy;
        else a_grid[j] = energy;
        remaining -= energy;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
#define ORC_TILE 64

static void batch_init_grid(const orc_batch *b, int32_t i, int32_t t0, orc_grid *g, orc_state *s)
{
    const int32_t N = b->N;
    memset(g, 0, sizeof(*g)); memset(s, 0, sizeof(*s));
    g->has_genset = b->has_genset; g->has_battery = b->has_battery; g->has_grid = b->has_grid;
    g->grid_before_battery = b->grid_before_battery;
    g->n_load = 1; g->n_pv = 1; g->horizon = b->horizon; g->T = b->T; g->final_step = b->final_step;
    s->t = t0;
    if (b->has_battery) {
        g->bat_min_capacity = b->bat_min_capacity[i]; g->bat_max_capacity = b->bat_max_capacity[i];
        g->bat_max_charge = b->bat_max_charge[i]; g->bat_max_discharge = b->bat_max_discharge[i];
        g->bat_efficiency = b->bat_efficiency[i]; g->bat_cost_cycle = b->bat_cost_cycle[i];
        s->charge = b->charge[i]; s->soc = b->soc[i];
    }
    if (b->has_genset) {
        g->gen_running_min = b->gen_running_min[i]; g->gen_running_max = b->gen_running_max[i];
        g->gen_cost = b->gen_cost[i]; g->gen_co2_per_unit = b->gen_co2_per_unit[i];
        g->gen_cost_per_unit_co2 = b->gen_cost_per_unit_co2[i];
        g->gen_start_up_time = (int32_t)(b->gen_times[i] & 0xffu);
        g->gen_no_abortion = (int32_t)((b->gen_times[i] >> 8) & 1u);
        g->gen_wind_down_time = (int32_t)((b->gen_times[i] >> 16) & 0xffu);
        uint32_t st = b->gen_status[i];
        s->gen_cur = st & 0xff; s->gen_goal = (st >> 8) & 0xff; s->gen_up = (st >> 16) & 0xff; s->gen_down = st >> 24;
    }
    if (b->has_grid) {
        g->grid_max_import = b->grid_max_import[i]; g->grid_max_export = b->grid_max_export[i];
        g->grid_cost_per_unit_co2 = b->grid_cost_per_unit_co2[i];
        g->grid_ts = b->grid_ts + i; g->grid_t_stride = 4 * (int64_t)N; g->grid_c_stride = N;
    }
    g->loss_load_cost = b->loss_load_cost[i]; g->overgeneration_cost = b->overgeneration_cost[i];
    g->load_ts = b->load_ts + i; g->load_t_stride = N; g->load_m_stride = 0;
    g->pv_ts = b->pv_ts + i;     g->pv_t_stride = N;   g->pv_m_stride = 0;
}

/* Tiles of ORC_TILE grids are walked step-major so that the [T,N] series rows and the [K,N,A] action rows are
 * read along their contiguous axis; each grid still goes through the scalar orc_run() above. */
static uint8_t *g_failed = NULL;      /* optional [N] flags: grid hit a state where the reference raises */
void orc_set_failure_flags(uint8_t *flags) { g_failed = flags; }

int64_t orc_run_batch(const orc_batch *b, int32_t t0, int32_t K, const double *actions, int normalized,
                      double *reward, int32_t nthreads)
{
    const int32_t N = b->N;
    const int32_t A = 2 * b->has_genset + b->has_battery + b->has_grid;
    const int32_t n_tiles = (N + ORC_TILE - 1) / ORC_TILE;
    int64_t failures = 0;
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
    #pragma omp parallel for num_threads(nthreads) schedule(static) reduction(+:failures)
#else
    (void)nthreads;
#endif
    for (int32_t tile = 0; tile < n_tiles; tile++) {
        orc_grid g[ORC_TILE]; orc_state s[ORC_TILE];
        const int32_t i0 = tile * ORC_TILE;
        const int32_t n = (N - i0 < ORC_TILE) ? N - i0 : ORC_TILE;
        for (int32_t j = 0; j < n; j++) batch_init_grid(b, i0 + j, t0, &g[j], &s[j]);
        orc_step_out o;
        for (int32_t k = 0; k < K; k++) {
            for (int32_t j = 0; j < n; j++) {
                const int32_t i = i0 + j;
                const double *ap = actions + ((int64_t)k * N + i) * A;
                orc_action a; memset(&a, 0, sizeof(a));
                int32_t c = 0;
                if (b->has_genset)  { a.genset[0] = ap[c]; a.genset[1] = ap[c + 1]; c += 2; }
                if (b->has_battery) { a.battery = ap[c++]; }
                if (b->has_grid)    { a.grid = ap[c++]; }
                if (orc_run(&g[j], &s[j], &a, normalized, &o) != 0) {
                    if (g_failed) g_failed[i] = 1; else failures++;
                    s[j].t = t0 + k + 1;              /* keep walking the series; this grid is flagged */
                }
                if (reward) reward[(int64_t)k * N + i] = o.reward;
            }
        }
        for (int32_t j = 0; j < n; j++) {
            const int32_t i = i0 + j;
            if (b->has_battery) { b->charge[i] = s[j].charge; b->soc[i] = s[j].soc; }
            if (b->has_genset)
                b->gen_status[i] = (uint32_t)s[j].gen_cur | ((uint32_t)s[j].gen_goal << 8) |
                                   ((uint32_t)s[j].gen_up << 16) | ((uint32_t)s[j].gen_down << 24);
        }
    }
    return failures ? -failures : (int64_t)N * K;
}

/* K steps of `action = _populate_action(priority_list); microgrid.run(action, normalized=False)` per grid:
 * DiscreteMicrogridEnv.step (envs/discrete/discrete.py:109-143) with per-step ids [K,N], or
 * RuleBasedControl.run (algos/rbc/rbc.py:64-93) with one fixed list per grid (ids [N]). */
int64_t orc_rollout_batch(const orc_batch *b, int32_t t0, int32_t K, const uint8_t *ids, int per_step,
                          const int32_t *table, int32_t n_actions, double *reward, int32_t nthreads)
{
    const int32_t N = b->N;
    const int32_t n_tiles = (N + ORC_TILE - 1) / ORC_TILE;
    int64_t failures = 0;
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
    #pragma omp parallel for num_threads(nthreads) schedule(static) reduction(+:failures)
#else
    (void)nthreads;
#endif
    for (int32_t tile = 0; tile < n_tiles; tile++) {
        orc_grid g[ORC_TILE]; orc_state s[ORC_TILE];
        const int32_t i0 = tile * ORC_TILE;
        const int32_t n = (N - i0 < ORC_TILE) ? N - i0 : ORC_TILE;
        for (int32_t j = 0; j < n; j++) batch_init_grid(b, i0 + j, t0, &g[j], &s[j]);
        orc_step_out o;
        for (int32_t k = 0; k < K; k++) {
            for (int32_t j = 0; j < n; j++) {
                const int32_t i = i0 + j;
                int32_t id = per_step ? ids[(int64_t)k * N + i] : ids[i];
                if (id < 0 || id >= n_actions) { failures++; id = 0; }
                orc_pl_element pl[3]; int32_t n_el = 0;
                for (int32_t e = 0; e < 3; e++) {
                    const int32_t mod = table[(id * 3 + e) * 2], act = table[(id * 3 + e) * 2 + 1];
                    if (mod >= 0) { pl[n_el].module = mod; pl[n_el].action = act; n_el++; }
                }
                orc_action a;
                /* a state where the reference's _populate_action asserts (priority_list.py:73-154): the grid is flagged --
                 * the reference would have raised there -- and keeps going on the control built so far */
                int bad = orc_populate_action(&g[j], &s[j], pl, n_el, &a) != 0;
                if (orc_run(&g[j], &s[j], &a, 0, &o) != 0) { bad = 1; s[j].t = t0 + k + 1; }
                if (bad) { if (g_failed) g_failed[i] = 1; else failures++; }
                if (reward) reward[(int64_t)k * N + i] = o.reward;
            }
        }
        for (int32_t j = 0; j < n; j++) {
            const int32_t i = i0 + j;
            if (b->has_battery) { b->charge[i] = s[j].charge; b->soc[i] = s[j].soc; }
            if (b->has_genset)
                b->gen_status[i] = (uint32_t)s[j].gen_cur | ((uint32_t)s[j].gen_goal << 8) |
                                   ((uint32_t)s[j].gen_up << 16) | ((uint32_t)s[j].gen_down << 24);
        }
    }
    return failures ? -failures : (int64_t)N * K;
}
