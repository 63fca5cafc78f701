#!/bin/bash
# per-call durations of the pose encoder's kernels in ONE stage-2 iteration (launch order), from a kernel trace
R=$PWD; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof_uc
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_uc -o p -- python $R/bench.py --stage 2 --steps 6 --warmup 6 --no-cpu-baseline --no-kernel-events --no-fixed-batch --no-secondary --no-measure-traffic > /tmp/prof_uc.log 2>&1
f=$(find /tmp/prof_uc -name "*kernel_trace.csv" | head -1)
mkdir -p $R/gpurun_out/r05unet
python - "$f" > $R/gpurun_out/r05unet/calls.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# last complete iteration: between the last two adam_kernel launches
idx = [i for i, n in enumerate(names) if "adam_kernel" in n]
lo, hi = idx[-3], idx[-2]
t0 = int(rows[lo]["End_Timestamp"])
for r in rows[lo + 1: hi + 1]:
    n = r["Kernel_Name"]
    if any(k in n for k in ("ugemm", "uwgrad", "upack", "ubn_", "uconv1", "usum", "ucolsum", "ufill")):
        short = n.replace("(anonymous namespace)::", "").replace("ganet::", "").split("(")[0][-24:]
        print("%9.1f .. %9.1f us  %-24s %7.2f us  grid %s x %s x %s  wg %s  q %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, short,
              (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"], r.get("Queue_Id", "?")))
PY
cat $R/gpurun_out/r05unet/calls.txt
