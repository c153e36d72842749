# usage: bash tools/traffic_pmc.sh <name> [bench args...]  -- HBM-side bytes of one bench.py step (one launch of every kernel of the
# config) from rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes (MI355X_MICROARCH.md: they do not fit one pass;
# gfx950 tallies the 128-B requests of 16-B-per-lane streaming reads at 64 B, so FETCH_SIZE is doubled for such reads -- narrower
# reads are uncalibrated and the file says which kernels have them).  Output: gpurun_out/traffic/traffic_<name>.json
set -e
cd ${GRAFT_REPO_ROOT:-$(dirname "$0")/..}
export TMPDIR=/tmp
NAME=$1; shift
OUT=$PWD/gpurun_out/traffic
mkdir -p $OUT
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/tr_$C
  (cd /tmp && rocprofv3 --pmc $C --output-format csv -d /tmp/tr_$C -o tr -- python3 $OLDPWD/bench.py --steps 1 --warmup 0 --no-cpu --no-extra "$@" > $OUT/${NAME}_$C.log 2>&1)
done
python3 - $NAME $OUT "$*" <<'PY'
import csv, glob, json, sys, collections
name, out, args = sys.argv[1], sys.argv[2], sys.argv[3]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob("/tmp/tr_%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if "hd::" not in r["Kernel_Name"]:
                continue
            if "--mode decode" not in args and "k_inflate" in r["Kernel_Name"]:
                continue                      # bench.py's untimed full-size verification of an encode run, not the step
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if c == "FETCH_SIZE":
                calls[k] += 1
line = json.loads([l for l in open("%s/%s_FETCH_SIZE.log" % (out, name)).read().splitlines() if l.startswith('{"metric"')][-1])
kern = {k: {"launches": calls[k], "FETCH_SIZE_KiB": v.get("FETCH_SIZE", 0.0), "WRITE_SIZE_KiB": v.get("WRITE_SIZE", 0.0)} for k, v in acc.items()}
fetch = sum(v["FETCH_SIZE_KiB"] for v in kern.values()) * 1024
write = sum(v["WRITE_SIZE_KiB"] for v in kern.values()) * 1024
res = {"config": name, "bench_args": args, "input_bytes": line["config"].get("input_bytes_per_gpu"), "algorithmic_bytes_per_launch": line["roofline"]["algorithmic_bytes_per_launch"],
       "fetch_bytes_raw": fetch, "fetch_bytes_doubled": 2 * fetch, "write_bytes": write,
       "hbm_bytes_per_launch": 2 * fetch + write, "hbm_bytes_per_launch_fetch_as_counted": fetch + write, "kernels": kern,
       "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes over one bench.py step (--steps 1 --warmup 0); KiB units; "
                 "FETCH_SIZE doubled per MI355X_MICROARCH.md (exact for 16-B-per-lane streaming reads; dword reads -- token slabs, far-match "
                 "sources -- are uncalibrated and over-counted by the doubling); WRITE_SIZE as is"}
json.dump(res, open("%s/traffic_%s.json" % (out, name), "w"), indent=1)
print(json.dumps({k: res[k] for k in ("config", "algorithmic_bytes_per_launch", "fetch_bytes_raw", "write_bytes", "hbm_bytes_per_launch")}))
PY
