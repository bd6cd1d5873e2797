"""Known-answer tests of the REFERENCE's own suite, restated for this engine.  Each test names the
reference test it restates; the expected values are the literals of those tests.  Every case runs through the CPU
oracle and (on a GPU) through the HIP engine.

  tests/microgrid/modules/module_tests/test_genset_module.py:64-163
  tests/microgrid/modules/module_tests/test_genset_long_status_changes.py:29-215
  tests/microgrid/modules/module_tests/test_genset_module_start_up_1_wind_down_1.py
  tests/microgrid/test_microgrid.py:263-320 (check_step, load/pv-only grids)
  tests/envs/test_discrete.py:73-80 (action_space.n)
  tests/microgrid/modules/module_tests/timeseries_modules.py:31-90, test_load_module.py:26-56,
  test_renewable_module.py:19-58 (observation bounds, first step, forecasts, negative-valued input series)
"""
import numpy as np
import torch
import pytest

T = 100
GENSET = dict(running_min_production=10.0, running_max_production=100.0, genset_cost=1.0, co2_per_unit=0.0,
              cost_per_unit_co2=0.0, start_up_time=0, wind_down_time=0, init_start_up=True)   # helpers/genset_module_testing_utils.py:4-11


def genset_grid(**kw):
    """A microgrid whose only controllable module is the test genset (load = pv = 0 rows do not touch it)."""
    return dict(load_ts=np.zeros(T), pv_ts=np.zeros(T), horizon=0, final_step=T, initial_step=0,
                unbalanced=dict(loss_load_cost=10.0, overgeneration_cost=2.0), genset=dict(GENSET, **kw))


class OracleBackend:
    name = "oracle"

    def __init__(self, oracle, params):
        self.m = oracle.OracleMicrogrid(params)

    def step(self, action, normalized=True):
        out = self.m.run(action, normalized).as_dict()
        out["status"] = list(self.m.status)
        out["obs"] = self.m.observe()
        return out


class DeviceBackend:
    name = "device"

    def __init__(self, device, params):
        from pymgrid_amd import MicrogridEnv
        self.env = MicrogridEnv(params, device=device)
        self.env.reset()

    def step(self, action, normalized=True):
        from pymgrid_amd import unpack_status
        ctrl = {k: [np.asarray(v)] if k == "genset" else [v] for k, v in action.items()}
        obs, reward, done, info = self.env.step(ctrl, normalized=normalized)
        out = dict(self.env.last_log); out["reward"], out["done"], out["obs"] = reward, int(done), obs
        out["info"] = info                                   # the reference-shaped {module: [info dict]}
        if "genset_status" in out:
            out["status"] = unpack_status(np.array([out["genset_status"]], dtype=np.uint32))[0].tolist()
        return out


@pytest.fixture(params=["oracle", pytest.param("device", marks=pytest.mark.gpu)])
def backend(request, oracle):
    if request.param == "oracle":
        return lambda p: OracleBackend(oracle, p)
    device = request.getfixturevalue("device")
    return lambda p: DeviceBackend(device, p)


def turn(b, goal, production):            # normalize_production: production / running_max_production (helpers :20-22)
    return b.step(dict(genset=[float(goal), production / 100.0]), True)


# ---- test_genset_module.py ----------------------------------------------------------------------------
def test_step_unnormalized_production(backend):                      # :64-76
    o = backend(genset_grid()).step(dict(genset=[1.0, 50.0]), normalized=False)
    assert o["genset_reward"] == -1.0 * 1.0 * 50 and o["status"] == [1, 1, 0, 0] and o["genset_production"] == 50


def test_step_normalized_production(backend):                        # :78-91
    o = turn(backend(genset_grid()), 1.0, 50)
    assert o["genset_reward"] == -50.0 and o["status"] == [1, 1, 0, 0] and o["genset_production"] == 50
    assert list(o["obs"][2:6]) == [1, 1, 0, 0]                       # obs == np.array([1, 1, 0, 0])


def test_step_immediate_status_change(backend):                      # :93-108
    o = turn(backend(genset_grid()), 0.0, 0)
    assert o["genset_reward"] == 0 and o["status"] == [0, 0, 0, 0] and o["genset_production"] == 0
    assert list(o["obs"][2:6]) == [0, 0, 0, 0] and not o["done"]


def test_step_genset_off_production_request_is_clipped(backend):     # :124-136 (raise_errors=False)
    o = turn(backend(genset_grid()), 0.0, 50)
    assert o["genset_reward"] == 0 and o["status"] == [0, 0, 0, 0] and o["genset_production"] == 0


def test_step_production_request_out_of_range_is_clipped(backend):   # :138-163
    rs = np.random.RandomState(0)
    for requested, possible in ((10 * rs.rand(), 10.0), (100 * (1 + rs.rand()), 100.0)):
        o = turn(backend(genset_grid()), 1.0, requested)
        assert o["genset_reward"] == -1.0 * possible and o["status"] == [1, 1, 0, 0]
        assert o["genset_production"] == possible


@pytest.mark.gpu
def test_raise_errors_true_raises_value_error(device):               # :110-122 (raise_errors=True -> ValueError)
    """Genset turned off (wind_down_time 0) and asked for 50: the reference raises ValueError with raise_errors=True;
    an out-of-range goal_status is an AssertionError there (:146-147) -- both surface as ValueError here."""
    from pymgrid_amd import MicrogridEnv
    env = MicrogridEnv(genset_grid(), device=device, raise_errors=True)
    env.reset()
    env.step({"genset": [np.array([1.0, 0.5])]})                     # in range: fine
    with pytest.raises(ValueError, match="Genset"):
        env.step({"genset": [np.array([0.0, 0.5])]})
    with pytest.raises(ValueError, match="goal_status"):
        env.step({"genset": [np.array([-0.5, 0.0])]})
    # the refusals came from the dry run (mgx_check_step): nothing was applied -- counter and genset status as before
    assert env.current_step == 1 and env.batch.cols["gen_status"].cpu().numpy().view(np.uint32)[0] & 0xffff == 0x0101
    _, _, _, info = env.step({"genset": [np.array([1.0, 0.25])]})
    assert env.current_step == 2 and env.last_log["violations"] == 0.0
    env.close()
    from pymgrid_amd import BatchedMicrogridEnv, MicrogridBatch
    bat = dict(load_ts=np.full(6, 10.0), pv_ts=np.zeros(6), horizon=0, final_step=6, initial_step=0,
               unbalanced=dict(loss_load_cost=10.0, overgeneration_cost=1.0),
               battery=dict(min_capacity=0.0, max_capacity=100.0, max_charge=20.0, max_discharge=20.0, efficiency=1.0,
                            battery_cost_cycle=0.0, init_soc=0.5))
    benv = BatchedMicrogridEnv(MicrogridBatch.from_grids([bat, bat], device=device), raise_errors=True, observations=False)
    benv.reset()
    ok = torch.tensor([[10.0], [-10.0]], dtype=torch.float64, device=device)
    bad = torch.tensor([[10.0], [-30.0]], dtype=torch.float64, device=device)      # grid 1: charge 30 > max_charge 20
    benv.step(ok, normalized=False)
    charge = benv.batch.cols["charge"].clone()
    with pytest.raises(ValueError, match=r"BatteryModule.*microgrid 1"):
        benv.step(bad, normalized=False)
    assert torch.equal(benv.batch.cols["charge"], charge) and benv.current_step == 1
    benv.close()
    quiet = MicrogridEnv(genset_grid(), device=device)               # raise_errors=False: silently clipped
    quiet.reset()
    _, _, _, info = quiet.step({"genset": [np.array([0.0, 0.5])]})
    assert quiet.last_log["violations"] == 1.0 and info["genset"][0]["provided_energy"] == 0
    quiet.close()


# ---- test_genset_long_status_changes.py: start_up_time 2, wind_down_time 3, on at start ------------------------
def long_genset(backend):
    return backend(genset_grid(start_up_time=2, wind_down_time=3))


def test_turn_off_steps_1_to_4(backend):                             # :29-93
    b = long_genset(backend)
    for expect in ([1, 0, 0, 2], [1, 0, 0, 1], [1, 0, 0, 0]):
        o = turn(b, 0.0, 50)
        assert o["status"] == expect and o["genset_reward"] == -50.0 and o["genset_production"] == 50
    o = turn(b, 0.0, 0)
    assert o["status"] == [0, 0, 2, 0] and o["genset_reward"] == 0 and o["genset_production"] == 0


def test_turn_on_after_turn_off(backend):                            # :95-178
    b = long_genset(backend)
    for _ in range(3):
        turn(b, 0.0, 50)
    assert turn(b, 0.0, 0)["status"] == [0, 0, 2, 0]
    o = turn(b, 1.0, 0)
    assert o["status"] == [0, 1, 1, 0] and o["genset_reward"] == 0
    o = turn(b, 1.0, 0)
    assert o["status"] == [0, 1, 0, 0] and o["genset_production"] == 0
    o = turn(b, 1.0, 50)
    assert o["status"] == [1, 1, 0, 3] and o["genset_reward"] == -50.0 and o["genset_production"] == 50


def test_turn_off_abortion(backend):                                 # :180-196
    b = long_genset(backend)
    turn(b, 0.0, 50)
    assert turn(b, 0.0, 50)["status"] == [1, 0, 0, 1]
    o = turn(b, 1.0, 50)
    assert o["status"] == [1, 1, 0, 3] and o["genset_reward"] == -50.0 and o["genset_production"] == 50


def test_turn_on_abortion(backend):                                  # :198-215
    b = backend(genset_grid(start_up_time=2, wind_down_time=3, init_start_up=False))
    assert turn(b, 1.0, 0)["status"] == [0, 1, 1, 0]
    o = turn(b, 0.0, 0)
    assert o["status"] == [0, 0, 2, 0] and o["genset_reward"] == 0 and o["genset_production"] == 0


def test_start_up_1_wind_down_1(backend):                            # test_genset_module_start_up_1_wind_down_1.py
    b = backend(genset_grid(start_up_time=1, wind_down_time=1))
    assert turn(b, 0.0, 50)["status"] == [1, 0, 0, 0]                # still running for one step
    o = turn(b, 0.0, 0)
    assert o["status"] == [0, 0, 1, 0] and o["genset_production"] == 0
    assert turn(b, 1.0, 0)["status"] == [0, 1, 0, 0]
    o = turn(b, 1.0, 50)
    assert o["status"] == [1, 1, 0, 1] and o["genset_production"] == 50


# ---- test_microgrid.py check_step: load/pv-only grids ----------------------------------------------------------
@pytest.mark.parametrize("case", ["equal", "excess_pv", "excess_load"])
def test_load_pv_only_check_step(backend, case):                     # :263-320 and the three set_ts variants
    rs = np.random.RandomState(3)
    load = 10 * rs.rand(T)
    pv = {"equal": load.copy(), "excess_pv": load + 5 * rs.rand(T), "excess_load": np.maximum(load - 5 * rs.rand(T), 0)}[case]
    p = dict(load_ts=load, pv_ts=pv, horizon=0, final_step=T, initial_step=0,
             unbalanced=dict(loss_load_cost=10.0, overgeneration_cost=2.0))
    b = backend(p)
    for k in range(T - 1):
        o = b.step({}, True)
        loss = load[k] - pv[k]
        assert 10.0 * max(loss, 0) == -1 * o["reward"]                # loss_load_cost * max(load - pv, 0) == -reward
        load_met = min(load[k], pv[k])
        assert o["load_met"] == load[k] and o["renewable_used"] == load_met
        assert o["curtailment"] == max(pv[k] - load_met, 0) and o["loss_load"] == max(load[k] - load_met, 0)
        assert o["overall_provided"] == load[k] and o["overall_absorbed"] == load[k]
        assert o["fixed_provided"] == 0.0 and o["fixed_absorbed"] == load[k]
        assert o["controllable_provided"] == 0.0 and o["controllable_absorbed"] == 0.0


# ---- test_discrete.py:73-80 --------------------------------------------------------------------------------
def test_discrete_action_space_size():
    from math import factorial
    from pymgrid_amd import get_priority_lists
    for has in ((1, 1, 1), (1, 1, 0), (0, 1, 1), (0, 1, 0), (1, 0, 0)):
        n_modules, n_gensets = sum(has), has[0]
        assert len(get_priority_lists(*map(bool, has))) == factorial(n_modules) * 2 ** n_gensets


# ---- timeseries_modules.py / test_load_module.py / test_renewable_module.py --------------------------------------
# The reference tests one module at a time; here the same series drive a load + renewable microgrid (no controllable
# module), where the load module behaves as in its own test and the renewable module is the flex source.
SERIES = 2 - np.cos(np.pi * np.arange(100) / 2)                       # timeseries_modules.py:24-26  (1, 2, 3, 2, ...)


@pytest.mark.parametrize("H", [0, 24])                                # NoForecasting / Forecasting
@pytest.mark.parametrize("flip", [1, -1])                             # ...NegativeVals: the module is handed -series
def test_timeseries_modules_first_step_and_bounds(backend, H, flip):
    from pymgrid_amd.batch import pack_grids
    p = dict(load_ts=flip * SERIES, pv_ts=flip * 0.5 * SERIES, horizon=H, final_step=100, initial_step=0,
             unbalanced=dict(loss_load_cost=10.0, overgeneration_cost=2.0))
    A, layout = pack_grids([p])
    # test_observation_space: Box(min(0, ts.min()), max(0, ts.max())) with ts in the module's sign convention
    assert A["load_lo"][0] == -3.0 and A["load_hi"][0] == 0.0 and A["pv_lo"][0] == 0.0 and A["pv_hi"][0] == 1.5
    assert layout.obs_dim == 2 * (1 + H)
    b = backend(p)
    o = b.step({}, True)
    assert o["done"] == 0 and o["load_met"] == SERIES[0]               # info["absorbed_energy"] == -1 * time_series[0]
    assert o["renewable_used"] == 0.5 * SERIES[0] and o["curtailment"] == 0   # the flex source covers what it can
    assert o["reward"] == -10.0 * (SERIES[0] - 0.5 * SERIES[0])        # the load module itself contributes reward 0
    obs = np.asarray(o["obs"], dtype=np.float64)
    W = 1 + H
    # from_normalized(obs, obs=True) == time_series[1 : forecast_horizon + 2]   (test_step; the reference's assertEqual
    # falls back to allclose(rtol=1e-7, atol=1e-10) for arrays, tests/helpers/test_case.py)
    np.testing.assert_allclose(-3.0 + 3.0 * obs[:W], -SERIES[1:H + 2], rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(0.0 + 1.5 * obs[W:2 * W], 0.5 * SERIES[1:H + 2], rtol=1e-7, atol=1e-10)
    # test_observations_in_observation_space: every observation of the episode lies in [0, 1]; done on the last row
    k = 1
    while not o["done"]:
        o = b.step({}, True)
        k += 1
        obs = np.asarray(o["obs"], dtype=np.float64)
        assert ((obs >= 0) & (obs <= 1)).all(), k
    assert k == 100                                                    # done on the step taken at t = final_step - 1


def test_mixed_sign_series_is_rejected():                              # base_timeseries_module.py:68-79 (_sign_check)
    from pymgrid_amd.batch import pack_grids
    p = dict(load_ts=np.array([1.0, -1.0, 2.0]), pv_ts=np.zeros(3), horizon=0, final_step=3, initial_step=0,
             unbalanced=dict(loss_load_cost=10.0, overgeneration_cost=2.0))
    with pytest.raises(ValueError, match="both positive and negative"):
        pack_grids([p])
