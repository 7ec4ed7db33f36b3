#!/bin/bash
# HBM traffic of the dominant GEMM kernel inside the bench command: two separate rocprofv3 --pmc passes
# (FETCH_SIZE, WRITE_SIZE) with --kernel-trace only, as MI355X_MICROARCH.md prescribes; FETCH_SIZE is corrected x2
# on gfx950 after calibrating on ln_kernel<2> (reads 65536 x 448 fp32 = 112 MiB) in the same pass.
# usage (GPU box, repo root):  tools/pmc_traffic.sh <out.json> [kernel-substring]
OUT=${1:-gpurun_out/traffic.json}; KSUB=${2:-"conv_gemm_f16x3_kernel<1, 7, 8, 1, true, 32,"}
REPO=$(pwd); D=$REPO/gpurun_out/pmc_traffic; mkdir -p "$D"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$D/$c" -o p -- \
    python "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extras > "$D/$c.log" 2>&1 || echo "pass $c failed"
done
cd "$REPO"
python - "$D" "$OUT" "$KSUB" <<'PY'
import csv, glob, json, sys
d, out, ksub = sys.argv[1:4]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{d}/{c}/**/*counter_collection.csv", recursive=True)[0]
    k_sum = k_n = l_sum = l_n = 0
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c:
            continue
        v = float(r["Counter_Value"])
        if ksub in r["Kernel_Name"]:
            k_sum += v; k_n += 1
        if "ln_kernel<2>" in r["Kernel_Name"]:
            l_sum += v; l_n += 1
    res[c] = {"launches": k_n, "sum_kb": k_sum, "per_launch_bytes": k_sum * 1024 / max(k_n, 1),
              "ln2_per_launch_mib": l_sum / max(l_n, 1) / 1024}
fetch_corr = 112.0 / res["FETCH_SIZE"]["ln2_per_launch_mib"] if res["FETCH_SIZE"]["ln2_per_launch_mib"] else 2.0
fb = res["FETCH_SIZE"]["per_launch_bytes"] * fetch_corr
wb = res["WRITE_SIZE"]["per_launch_bytes"]
json.dump({"kernel": ksub,
           "command": "tools/pmc_traffic.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) "
                      "--output-format csv -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline",
           "raw": res,
           "fetch_correction": f"x{fetch_corr:.3f}: calibrated in the same pass on ln_kernel<2>, which reads 65536x448 fp32 = "
                               f"112 MiB and is reported as {res['FETCH_SIZE']['ln2_per_launch_mib']:.2f} MiB (gfx950 FETCH_SIZE "
                               "undercounts wide coalesced reads, MI355X_MICROARCH.md)",
           "write_calibration": f"ln_kernel<2> writes 112 MiB and WRITE_SIZE reports {res['WRITE_SIZE']['ln2_per_launch_mib']:.1f} MiB",
           "hbm_bytes_per_launch": fb + wb, "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb},
          open(out, "w"), indent=1)
print(open(out).read())
PY
