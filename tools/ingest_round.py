"""Copy the outputs of tools/profile_round.sh (gpurun_out/round/) into profiles/ and print the headline numbers.
Run on the build machine after `gpurun -- 'bash tools/profile_round.sh'`."""
import csv
import json
import os
import re
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R, P = os.path.join(ROOT, "gpurun_out", "round"), os.path.join(ROOT, "profiles")


def main():
    old = json.load(open(os.path.join(P, "r01_pmc.json")))
    vals, n = {}, None
    for line in open(os.path.join(R, "pmc_kstep.txt")):
        m = re.match(r"PMC (\S+) (\S+) (\d+) total", line)
        if m:
            vals[m.group(1)], n = float(m.group(2)), int(m.group(3))
    new = {k: old[k] for k in ("kernel", "envs_per_launch", "source", "note")}
    new["FETCH_SIZE_KB"], new["WRITE_SIZE_KB"] = vals.pop("FETCH_SIZE"), vals.pop("WRITE_SIZE")
    new["hbm_bytes_per_launch"] = (new["FETCH_SIZE_KB"] + new["WRITE_SIZE_KB"]) * 1024
    new["hbm_bytes_per_launch_fetch_doubled"] = (2 * new["FETCH_SIZE_KB"] + new["WRITE_SIZE_KB"]) * 1024
    new.update(vals)
    json.dump(new, open(os.path.join(P, "r01_pmc.json"), "w"), indent=1)
    print("k_step launches", n, "traffic MB", new["hbm_bytes_per_launch"] / 1e6, "VALU M", new["SQ_INSTS_VALU"] / 1e6,
          "wait", new["SQ_WAIT_ANY"] / new["SQ_WAVE_CYCLES"])
    for a, b in (("pmc_kstep.txt", "r01_pmc_kstep_f32_4096_final.txt"), ("pmc_gemm.txt", "r01_pmc_mfma_gemm.txt"),
                 ("stage_shares.txt", "r01_stage_shares_final.txt"), ("bench_kernel_stats.csv", "r01_bench_kernel_stats_final.csv")):
        shutil.copy(os.path.join(R, a), os.path.join(P, b))
    out = {}
    for k in ("f64", "rk4", "p2_8192", "rollout_only"):
        d = json.loads(open(os.path.join(R, f"bench_{k}.json")).read().strip().splitlines()[-1])
        out["bench_" + k] = {x: d[x] for x in ("value", "ms_per_step", "env_kernel_ms", "ppo_optimizer_steps_per_sec", "dtype", "config") if x in d}
    json.dump(out, open(os.path.join(P, "r01_other_configs.json"), "w"), indent=1)
    print({k: (round(v["value"]), round(v["env_kernel_ms"], 2), round(v.get("ppo_optimizer_steps_per_sec", 0))) for k, v in out.items()})
    d = json.loads(open(os.path.join(R, "bench_line.json")).read().strip().splitlines()[-1])
    open(os.path.join(P, "r01_bench_line.json"), "w").write(json.dumps(d) + "\n")
    print("bench", d["value"], d["ms_per_step"], d["env_kernel_ms"], d["ppo_optimizer_steps_per_sec"], d["roofline"]["traffic"],
          d["cpu_baseline"]["value"], d["cpu_baseline"]["all_cores"])
    rows = list(csv.DictReader(open(os.path.join(P, "r01_bench_kernel_stats_final.csv"))))
    print(rows[0]["Name"][:30], rows[0]["AverageNs"], rows[0]["Percentage"])


if __name__ == "__main__":
    main()
