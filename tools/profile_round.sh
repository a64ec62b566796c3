#!/bin/bash
# Round profile (run on the GPU box through gpurun): rocprofv3 kernel-trace stats of the default bench command, plus
# two separate PMC passes (FETCH_SIZE / WRITE_SIZE) on a short eager run.  Summaries land in gpurun_out/prof_round/.
#   tools/profile_round.sh [tag]     (BENCH_ARGS="--case ... --batch ..." selects another workload)
cd /tmp && export TMPDIR=/tmp
TAG=${1:-r04_case118_b128_train}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_round; rm -rf /tmp/pr; mkdir -p $O /tmp/pr
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d /tmp/pr/trace -o bench -- python $R/bench.py --no-cpu-baseline --no-live-traffic --no-dp-overhead --no-other-configs $BENCH_ARGS > $O/${TAG}_bench_under_rocprof.json 2> /tmp/pr/trace.err
find /tmp/pr/trace -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_kernel_stats.csv \;
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace -f csv -d /tmp/pr/$c -o pmc -- python $R/bench.py --child --no-cpu-baseline --no-live-traffic --no-dp-overhead --no-graph --steps 3 --warmup 1 --profile-steps 0 $BENCH_ARGS > /tmp/pr/$c.out 2> /tmp/pr/$c.err
  F=$(find /tmp/pr/$c -name "*counter_collection.csv" | head -1)
  python - "$F" "$c" > $O/${TAG}_pmc_$c.txt <<'PY'
import csv, sys, collections
path, ctr = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: [0, 0.0])
with open(path) as f:
    for row in csv.DictReader(f):
        if row.get("Counter_Name") != ctr: continue
        k = row["Kernel_Name"].split("(")[0][:60]
        agg[k][0] += 1; agg[k][1] += float(row["Counter_Value"])
print(f"# {ctr}: per-kernel launches, total counter value, value per launch (raw counter units as reported by rocprofv3: KiB)")
for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{k:60s} n={n:5d} total={v:16.1f} per_launch={v / n:14.1f}")
PY
done
# MFMA pipe occupancy of the GEMM kernels: SQ_VALU_MFMA_BUSY_CYCLES (per-SIMD busy cycles, summed over the chip) against the
# kernel's duration from the same pass: util = busy / (duration x 2.4 GHz x 1024 SIMDs)
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -f csv -d /tmp/pr/MFMA -o pmc -- python $R/bench.py --child --no-cpu-baseline --no-live-traffic --no-dp-overhead --no-graph --steps 3 --warmup 1 --profile-steps 0 $BENCH_ARGS > /tmp/pr/MFMA.out 2> /tmp/pr/MFMA.err
python - /tmp/pr/MFMA > $O/${TAG}_pmc_mfma.txt <<'PY'
import csv, sys, glob, collections
d = sys.argv[1]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
dur = {}
if kt:
    for row in csv.DictReader(open(kt[0])):
        try:
            dur[row["Dispatch_Id"]] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
        except Exception:
            pass
agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
if cc:
    for row in csv.DictReader(open(cc[0])):
        k = row["Kernel_Name"].split("(")[0][:60]
        a = agg[k]
        if row["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
            a[0] += 1; a[1] += float(row["Counter_Value"]); a[3] += dur.get(row.get("Dispatch_Id"), 0)
        elif row["Counter_Name"] == "GRBM_GUI_ACTIVE":
            a[2] += float(row["Counter_Value"])
print("# per kernel: launches, SQ_VALU_MFMA_BUSY_CYCLES per launch, GRBM_GUI_ACTIVE per launch, avg duration us (this pass), "
      "mfma_util = busy / (duration x 2.4e9 x 1024 SIMDs)")
for k, (n, busy, gui, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
    if n == 0: continue
    us = ns / n / 1e3
    util = busy / n / (us * 1e-6 * 2.4e9 * 1024) if us > 0 else float("nan")
    print(f"{k:60s} n={n:5d} busy={busy / n:14.0f} gui_active={gui / n:12.0f} dur_us={us:10.2f} mfma_util={util:6.3f}")
PY
# HBM-side bytes per launch of every kernel class: (2 x FETCH_SIZE + WRITE_SIZE) KiB (MI355X_MICROARCH.md: FETCH_SIZE
# reports half of a wide coalesced read on gfx950; WRITE_SIZE is taken as is)
python - $O $TAG "$BENCH_ARGS" > $O/${TAG}_pmc_traffic.json <<'PY'
import sys, re, json
o, tag, bargs = sys.argv[1], sys.argv[2], sys.argv[3]
def read(c):
    d = {}
    for line in open(f"{o}/{tag}_pmc_{c}.txt"):
        m = re.match(r"(.*?)\s+n=\s*(\d+) total=\s*([\d.]+) per_launch=\s*([\d.]+)", line)
        if m: d[m.group(1).strip()] = (int(m.group(2)), float(m.group(3)))
    return d
f, w = read("FETCH_SIZE"), read("WRITE_SIZE")
cls = {"gemm_nt": "gemm_nt_kernel", "gemm_tn": "gemm_tn_kernel", "hop_norm": "hop_kernel<true>", "edge_fwd": "edge_fwd_kernel",
       "edge_bwd": "edge_bwd_kernel", "fused_hops_fwd": "_hops_kernel$", "fused_hops_bwd": "_hops_kernel$",
       "ea_seg_fwd": "ea_seg_fwd_kernel", "ea_seg_bwd": "ea_seg_bwd_kernel", "seg_lin_hops_fwd": "seg_lin_hops_kernel<1>",
       "seg_lin_hops_bwd": "seg_lin_hops_kernel<2>", "front_seg_fwd": "front_seg_fwd_kernel"}
case = re.search(r"--case (\S+)", bargs); batch = re.search(r"--batch (\d+)", bargs); mode = re.search(r"--mode (\S+)", bargs)
cfg = re.search(r"--config (\S+)", bargs); hub = re.search(r"--hub-frac (\S+)", bargs)
key = f"{case.group(1) if case else '118v2'}:{batch.group(1) if batch else 128}:{mode.group(1) if mode else 'train'}"
if cfg and cfg.group(1) != "standard": key += ":" + cfg.group(1)
if hub and float(hub.group(1)) > 0: key += ":hub" + hub.group(1)
out = {"_note": "bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KiB * 1024, averaged over all launches of the class in a 3-step "
                "eager run; separate rocprofv3 --pmc passes.  The counters are fabric-side (L2 <-> Infinity Cache/HBM) and "
                "include Infinity-Cache hits; at case118v2 x 128 the whole working set sits in the 256 MiB Infinity Cache."}
def hit(pat, k):   # (names are cut at "(": a trailing $ anchors the pattern at the end -- seg_lin_hops_kernel<N> is its own class)
    return k.endswith(pat[:-1]) if pat.endswith("$") else pat in k
for c, pat in cls.items():
    nf = [v for k, v in f.items() if hit(pat, k)]
    if not nf: continue
    launches = sum(n for n, _ in nf)                      # all template instantiations of the class, launch-weighted
    fv = sum(t for _, t in nf) / launches
    wv = sum(t for k, (n, t) in w.items() if hit(pat, k)) / launches
    out[f"{c}:{key}"] = int((2 * fv + wv) * 1024)
print(json.dumps(out, indent=1))
PY
ls -la $O
