"""Microgrids with several gensets / batteries / grids (the reference's container takes any number of modules per name,
module_container.py:355-413): the CPU oracle against fixtures made by the real reference (tests/golden/multi.npz), and the
device's general kernels against both."""
import numpy as np
import pytest

from conftest import multi_cases


def test_multi_instance_oracle_vs_reference(oracle):
    """orc_mrun / orc_mobserve / orc_mpopulate_action == the reference on 7 module mixes (up to 4 gensets + 3 batteries +
    2 grids + 3 loads + 2 pvs: lists of >= 8 addends go through numpy's pairwise sum), normalised and raw controls: reward,
    done, every log column of every module instance, observations, expanded priority lists."""
    for ci, p, mt, z in multi_cases():
        names = [str(s) for s in z[f"c{ci}_log_names"]]
        for tag, normalized in (("n", True), ("r", False)):
            om = oracle.OracleMultiMicrogrid(p)
            assert np.array_equal(om.reset(), z[f"c{ci}_{tag}_obs0"]), (ci, tag)
            a = z[f"c{ci}_{tag}_actions"]
            for k in range(a.shape[0]):
                out = om.run(a[k], normalized)
                for n, v, r in zip(names, om.log_row(out, names), z[f"c{ci}_{tag}_log"][k]):
                    assert v is not None and v == r, (ci, tag, k, n, v, r)
                assert out.common.reward == z[f"c{ci}_{tag}_reward"][k] and out.common.done == z[f"c{ci}_{tag}_done"][k]
                assert np.array_equal(om.observe(), z[f"c{ci}_{tag}_obs"][k]), (ci, tag, k)
            for j in range(om.counts["battery"]):
                assert om.s.battery[j].charge == z[f"c{ci}_{tag}_charge"][-1, j]
                assert om.s.battery[j].soc == z[f"c{ci}_{tag}_soc"][-1, j]
            for j in range(om.counts["genset"]):
                st = om.s.genset[j]
                assert [st.gen_cur, st.gen_goal, st.gen_up, st.gen_down] == list(z[f"c{ci}_{tag}_status"][-1, j])
        if mt["n_lists"]:
            om = oracle.OracleMultiMicrogrid(p)
            table, ids = z[f"c{ci}_pl_table"], z[f"c{ci}_ids"]
            for k in range(len(ids)):
                ctrl = om.populate_action([tuple(int(x) for x in e) for e in table[ids[k]] if e[0] >= 0])
                assert np.array_equal(ctrl, z[f"c{ci}_control"][k]), (ci, k)
                assert om.run(ctrl, False).common.reward == z[f"c{ci}_dreward"][k], (ci, k)
