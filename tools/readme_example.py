import numpy as np, os, tempfile
from pymgrid_amd import Microgrid, RuleBasedControl
from pymgrid_amd.modules import BatteryModule, GensetModule, GridModule, LoadModule, RenewableModule
z = np.load("tests/golden/surface.npz")
load_ts, pv_ts, grid_ts = np.abs(z["c0_load_ts"][:, 0]), z["c0_pv_ts"][:, 0], z["c0_grid_ts"]
mg = Microgrid([("load", LoadModule(load_ts, forecaster="oracle", forecast_horizon=23)), ("pv", RenewableModule(pv_ts, forecaster="oracle")),
                GensetModule(running_min_production=10, running_max_production=90, genset_cost=0.4),
                BatteryModule(min_capacity=20, max_capacity=100, max_charge=25, max_discharge=25, efficiency=0.9, init_soc=0.5),
                GridModule(max_import=80, max_export=50, time_series=grid_ts)], loss_load_cost=10.0)
mg.reset()
obs, reward, done, info = mg.run(mg.sample_action())
print(type(obs), reward, done, list(info))
print(mg.modules.battery[0].soc, list(mg.state_dict(normalized=True)), mg.get_cost_info()["genset"], mg.to_normalized({"battery": [-12.0]}, act=True))
log = RuleBasedControl(mg).run(max_steps=24)
print(log.shape, log.columns.names)
d = tempfile.mkdtemp()
mg.dump(os.path.join(d, "microgrid.yaml")); mg2 = Microgrid.load(os.path.join(d, "microgrid.yaml"))
print(mg2.current_step, mg2.modules.battery[0].soc == mg.modules.battery[0].soc)
