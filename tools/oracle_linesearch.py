"""How far apart are the oracle's two Newton line searches?  ("so that when true-MuJoCo goldens arrive the first
mismatch is not the line search".)

The oracle's default line search is a safeguarded Newton iteration on p'(alpha); `orc_set_line_search(1)` switches to the bracketing
structure of MuJoCo 2.1's PrimalSearch (p0 / p1 initialisation, one-sided Newton steps until the derivative changes sign, then three
candidates per iteration) [3P-RECALL].  Both stop inside the same gradient tolerance (tolerance * 0.01 * |search| / scale).

Measured on the bench workload's 16 seeded action streams: the DEFAULT oracle walks each stream; before every env step a twin is put on
its state (qpos, qvel, act, time, warm start) and takes the same env step (10 substeps) with the bracketing search; the difference of
the results is what ONE env step's worth of line searches moves — plus, for reference, the whole-episode difference of two oracles
that each keep their own search.

    python tools/oracle_linesearch.py [--steps 200] [--out profiles/r05_oracle_linesearch.json]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def measure(streams, steps):
    from helpers import default_state, oracle_for
    from myochallenge_amd.envs.config import task_ids
    from myochallenge_amd.synth_hand import synthetic_hand
    from oracle import oracle as orc
    from oracle.oracle import OracleData, baoding_step, make_cfg
    cm, om, _ = oracle_for(synthetic_hand())
    cfg = make_cfg(task_ids(cm))
    fields = ("qpos", "qvel", "act", "qacc_warmstart")
    out = {"local_qpos_rel": [], "local_qvel_abs": [], "episode_qpos_rel": [], "solver_iter_mismatch_steps": 0, "env_steps": 0}
    for sg, seed in streams:
        rng = np.random.RandomState(seed)
        a_, b_, c_ = OracleData(om), OracleData(om), OracleData(om)      # default; twin (bracketing, re-synchronised); bracketing on its own
        sa, sb, sc = default_state(), default_state(), default_state()
        for d in (a_, c_):
            d.reset(); d.qpos[:23] = 0; d.qpos[0] = -1.57
        loc_q, loc_v, epi_q, split = 0.0, 0.0, 0.0, False
        for t in range(steps):
            act = np.clip(rng.normal(0, sg, 39), -1, 1).astype(np.float32)
            b_.reset()
            for f in fields:
                b_.arr(f)[:] = a_.arr(f)
            b_.arr("time")[0] = a_.arr("time")[0]
            for k in ("which_task", "counter"):
                setattr(sb, k, getattr(sa, k))
            for k in range(2):
                sb.start_angle[k] = sa.start_angle[k]
            sb.x_radius, sb.y_radius, sb.time_period = sa.x_radius, sa.y_radius, sa.time_period
            orc.set_line_search(False)
            _, ca = baoding_step(a_, cfg, sa, act)
            ita = a_.solver_iter
            orc.set_line_search(True)
            _, cb = baoding_step(b_, cfg, sb, act)
            out["solver_iter_mismatch_steps"] += int(b_.solver_iter != ita)
            if not split:
                _, cc = baoding_step(c_, cfg, sc, act)
            orc.set_line_search(False)
            scale = np.abs(np.asarray(a_.qpos)).max()
            loc_q = max(loc_q, float(np.abs(np.asarray(a_.qpos) - np.asarray(b_.qpos)).max() / scale))
            loc_v = max(loc_v, float(np.abs(np.asarray(a_.qvel) - np.asarray(b_.qvel)).max()))
            if not split:
                epi_q = max(epi_q, float(np.abs(np.asarray(a_.qpos) - np.asarray(c_.qpos)).max() / scale))
                if bool(ca[6]) != bool(cc[6]):
                    split = True
            out["env_steps"] += 1
            if ca[6]:                                            # episode over: both restart
                for d, s in ((a_, sa), (c_, sc)):
                    d.reset(); d.qpos[:23] = 0; d.qpos[0] = -1.57
                sa, sc = default_state(), default_state()
                split = False
        out["local_qpos_rel"].append(loc_q); out["local_qvel_abs"].append(loc_v); out["episode_qpos_rel"].append(epi_q)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    streams = [(sg, seed) for sg in (0.08, 0.135) for seed in range(8)]
    r = measure(streams, a.steps)
    rec = {"what": "oracle with the safeguarded-Newton line search vs the oracle with the PrimalSearch-style bracketing search (tools/oracle_linesearch.py)",
           "streams (action sigma, seed)": streams, "env_steps_per_stream": a.steps,
           "worst_local_qpos_rel (one env step from the same state)": max(r["local_qpos_rel"]),
           "worst_local_qvel_abs": max(r["local_qvel_abs"]),
           "local_qpos_rel_per_stream": [float("%.3g" % v) for v in r["local_qpos_rel"]],
           "worst_whole_episode_qpos_rel (each oracle keeps its own search)": max(r["episode_qpos_rel"]),
           "whole_episode_qpos_rel_per_stream": [float("%.3g" % v) for v in r["episode_qpos_rel"]],
           "env_steps_whose_last_solve_took_another_iteration_count": r["solver_iter_mismatch_steps"], "env_steps": r["env_steps"]}
    print(json.dumps(rec, indent=1))
    if a.out:
        json.dump(rec, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
