"""Copy the outputs of tools/profile_round.sh (gpurun_out/round/) into profiles/ and print the headline numbers.
Run on the build machine after `gpurun -- 'bash tools/profile_round.sh'`."""
import csv
import json
import os
import re
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R, P = os.path.join(ROOT, "gpurun_out", "round"), os.path.join(ROOT, "profiles")
TAG = os.environ.get("ROUND", "r06")


def main():
    old = {"kernel": "k_step<double,false,16> (fp64 stepper, the bench default since round 4; 16 = contact-row capacity / 4 of the base scratch)", "envs_per_launch": 4096, "dtype": "f64",
           "source": "rocprofv3 --pmc <one set per pass> --kernel-trace --output-format csv -- python3 bench.py --no-cpu-baseline --no-variants --min-seconds 0 --steps 24 --warmup 8 (tools/profile_round.sh); per-launch averages over the k_step launches of the bench workload (first two dropped)",
           "note": "FETCH_SIZE is reported as counted; the doubled figure (the guide's gfx950 correction for 16-B/lane streaming reads) is given separately"}
    vals, n = {}, None
    for line in open(os.path.join(R, "pmc_kstep.txt")):
        m = re.match(r"PMC (\S+) (\S+) (\d+) total", line)
        if m:
            vals[m.group(1)], n = float(m.group(2)), int(m.group(3))
    new = dict(old)
    new["FETCH_SIZE_KB"], new["WRITE_SIZE_KB"] = vals.pop("FETCH_SIZE"), vals.pop("WRITE_SIZE")
    new["hbm_bytes_per_launch"] = (new["FETCH_SIZE_KB"] + new["WRITE_SIZE_KB"]) * 1024
    new["hbm_bytes_per_launch_fetch_doubled"] = (2 * new["FETCH_SIZE_KB"] + new["WRITE_SIZE_KB"]) * 1024
    new.update(vals)
    # the library build the counters were taken on (bench.py attaches them only to that build)
    try:
        new["build_id"] = open(os.path.join(R, "build_id.txt")).read().strip()
    except OSError:
        new["build_id"] = None
    if "SQ_THREAD_CYCLES_VALU" in new and "SQ_ACTIVE_INST_VALU" in new and new["SQ_ACTIVE_INST_VALU"] > 0:
        # active lanes per issued VALU instruction-cycle / 64
        new["lane_utilisation"] = new["SQ_THREAD_CYCLES_VALU"] / (64.0 * new["SQ_ACTIVE_INST_VALU"])
    json.dump(new, open(os.path.join(P, TAG + "_pmc.json"), "w"), indent=1)
    print("k_step launches", n, "traffic MB", new["hbm_bytes_per_launch"] / 1e6, "VALU M", new["SQ_INSTS_VALU"] / 1e6,
          "wait", new["SQ_WAIT_ANY"] / new["SQ_WAVE_CYCLES"])
    for a, b in (("pmc_kstep.txt", TAG + "_pmc_kstep_f64_4096.txt"), ("pmc_gemm.txt", TAG + "_pmc_mfma_gemm.txt"),
                 ("stage_shares.txt", TAG + "_stage_shares.txt"), ("bench_kernel_stats.csv", TAG + "_bench_kernel_stats.csv"),
                 ("drift.log", TAG + "_drift_32streams.log"), ("pmc_mfma_kstep_f64.txt", TAG + "_pmc_mfma_kstep_f64.txt"),
                 ("stage_shares_rk4.txt", TAG + "_stage_shares_rk4.txt"), ("wg_timeline_parts.log", TAG + "_wg_timeline_parts.log"),
                 ("wg_timeline_whole.log", TAG + "_wg_timeline_whole_steps.log"), ("gpu_tests.log", TAG + "_gpu_tests.log"),
                 ("pmc_kstep_whole_steps.txt", TAG + "_pmc_kstep_whole_steps.txt"), ("train_demo_p1_100m_f64.json", TAG + "_train_demo_p1_100m_f64.json"),
                 ("train_demo_reorient_lstm256_40m.json", TAG + "_train_demo_reorient_lstm256_40m.json"),
                 ("pmc_kstep_publish_fence.txt", TAG + "_pmc_kstep_publish_fence.txt"), ("config_e_kernel_stats.csv", TAG + "_config_e_kernel_stats.csv"),
                 ("lstm_seq_time.txt", TAG + "_lstm_seq_time.txt"), ("pmc_instruction_mix_f64.txt", TAG + "_pmc_instruction_mix_f64.txt")):
        if os.path.exists(os.path.join(R, a)):
            shutil.copy(os.path.join(R, a), os.path.join(P, b))
    out = {}
    for k in ("mixed", "rk4", "p2_8192", "rollout_only", "reorient_p2", "lstm128"):
        if not os.path.exists(os.path.join(R, f"bench_{k}.json")):
            continue
        d = json.loads(open(os.path.join(R, f"bench_{k}.json")).read().strip().splitlines()[-1])
        out["bench_" + k] = {x: d[x] for x in ("value", "ms_per_step", "env_kernel_ms", "ppo_optimizer_steps_per_sec", "dtype", "config") if x in d}
    if os.path.exists(os.path.join(R, "bench_publish_fence.json")):
        d = json.loads(open(os.path.join(R, "bench_publish_fence.json")).read().strip().splitlines()[-1])
        out["bench_publish_fence"] = {x: d[x] for x in ("value", "ms_per_step", "env_kernel_ms") if x in d}
    for key, fn in (("config_E_reorient_lstm256_light", "bench_reorient_lstm.json"), ("config_E_reorient_lstm256_reference_settings", "bench_reorient_lstm_reference.json"),
                    ("config_E_reorient_lstm256_reference_settings_step_kernels", "bench_reorient_lstm_reference_step_kernels.json")):
        try:
            out[key] = json.loads(open(os.path.join(R, fn)).read().strip().splitlines()[-1])
        except Exception as e:          # noqa: BLE001
            print("no record", fn, e)
    json.dump(out, open(os.path.join(P, TAG + "_other_configs.json"), "w"), indent=1)
    print({k: (round(v["value"]), round(v["env_kernel_ms"], 2), round(v.get("ppo_optimizer_steps_per_sec", 0))) for k, v in out.items() if "value" in v})
    d = json.loads(open(os.path.join(R, "bench_line.json")).read().strip().splitlines()[-1])
    open(os.path.join(P, TAG + "_bench_line.json"), "w").write(json.dumps(d) + "\n")
    print("bench", d["value"], d["ms_per_step"], d["env_kernel_ms"], d["ppo_optimizer_steps_per_sec"], d["roofline"]["traffic"],
          d["cpu_baseline"]["value"], d["cpu_baseline"]["all_cores"])
    rows = list(csv.DictReader(open(os.path.join(P, TAG + "_bench_kernel_stats.csv"))))
    print(rows[0]["Name"][:30], rows[0]["AverageNs"], rows[0]["Percentage"])


if __name__ == "__main__":
    main()
