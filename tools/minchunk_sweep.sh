export TMPDIR=/tmp
for v in m128 m64 m32; do
  for seqs in 1 8; do
    rm -rf gpurun_out/mc; mkdir -p gpurun_out/mc
    NVK_LIB_PATH=build/variants/$v.so timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/mc -- python tools/bench_model.py --preset tinyllama --seqs $seqs --ctx 1024 --steps 32 > gpurun_out/mc.log 2>&1 || exit 1
    echo "== $v seqs $seqs: $(grep 'decode ' gpurun_out/mc.log | cut -c1-110)"
    python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/mc/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "nvk_decode_kernel" in r["Name"] or "nvk_merge_kernel" in r["Name"]:
            print("   ", r["Name"].split("(")[0][-36:], r["Calls"], "calls, avg", round(float(r["AverageNs"]) / 1e3, 2), "us")
PY
  done
done
