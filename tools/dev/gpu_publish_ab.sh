#!/bin/bash
# A/B of the two ways a part of an env step publishes its record (MYO_PUBLISH=fence|wt): state checksums over a soak, k_step time,
# HBM traffic of the k_step launches.  Developer tool, run through gpurun.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/publish; mkdir -p $O
L=myochallenge_amd/libmyobatch.so
for dt in f64 mixed; do
  MYO_STEP_SPLIT=0 python tools/dev/kab.py $L --dtype $dt --rounds 1 --steps 1500 --env p2 > $O/soak_${dt}_whole.log 2>&1
  for m in fence wt; do
    MYO_PUBLISH=$m python tools/dev/kab.py $L --dtype $dt --rounds 1 --steps 1500 --env p2 > $O/soak_${dt}_$m.log 2>&1
  done
done
MYO_PUBLISH=wt python tools/dev/kab.py $L --dtype f64 --rounds 1 --steps 600 --env reorient > $O/soak_die_wt.log 2>&1
MYO_STEP_SPLIT=0 python tools/dev/kab.py $L --dtype f64 --rounds 1 --steps 600 --env reorient > $O/soak_die_whole.log 2>&1
MYO_PUBLISH=wt python tools/dev/kab.py $L --dtype f64 --integrator rk4 --rounds 1 --steps 300 > $O/soak_rk4_wt.log 2>&1
MYO_STEP_SPLIT=0 python tools/dev/kab.py $L --dtype f64 --integrator rk4 --rounds 1 --steps 300 > $O/soak_rk4_whole.log 2>&1
tail -n 2 $O/soak_*.log
for m in fence wt; do
  for r in 1 2; do MYO_PUBLISH=$m python bench.py --no-cpu-baseline --no-variants > $O/bench_${m}_$r.json 2>$O/bench_${m}_$r.err; done
  export MYO_PUBLISH=$m
  for set in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_${m}_$set -- python3 bench.py --no-cpu-baseline --no-variants --min-seconds 0 --steps 24 --warmup 8 > /dev/null 2>&1
  done
  unset MYO_PUBLISH
done
python - <<'P'
import csv, glob, json
for m in ("fence", "wt"):
    for r in (1, 2):
        try:
            d = json.loads(open("gpurun_out/publish/bench_%s_%d.json" % (m, r)).read().strip().splitlines()[-1])
            print(m, r, round(d["value"]), d["ms_per_step"], d.get("env_kernel_ms"))
        except Exception as e:
            print(m, r, "bench failed", e)
    for s in ("FETCH_SIZE", "WRITE_SIZE"):
        tot = n = 0
        for f in glob.glob("gpurun_out/publish/pmc_%s_%s/**/*counter_collection.csv" % (m, s), recursive=True):
            for row in csv.DictReader(open(f)):
                if row["Kernel_Name"].startswith("void k_step") and row["Counter_Name"] == s:
                    tot += float(row["Counter_Value"]); n += 1
        print(m, s, "KB per launch", tot / max(n, 1), "launches", n)
P
