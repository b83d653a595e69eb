#!/usr/bin/env python3
"""gpurun_out/ artefacts of tools/refresh_profiles.sh -> profiles/rNN_* (run in the repo after the GPU call).
usage: python tools/make_profiles.py [round]"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
rnd = "r%02d" % (int(sys.argv[1]) if len(sys.argv) > 1 else 1)
import subprocess
try:
    COMMIT = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:
    COMMIT = "unknown"


sys.path.insert(0, ROOT)
from instaorder_amd._lib import csrc_digest      # noqa: E402  (what the profiled kernels were built from)
CSRC = csrc_digest()


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


for src, dst in (("bench_n1.json", "bench_n1.json"), ("bench_n1_bf16.json", "bench_n1_bf16.json")):
    if os.path.exists(os.path.join(G, src)):
        json.dump(last_json(os.path.join(G, src)), open(os.path.join(P, "%s_%s" % (rnd, dst)), "w"), indent=1)

for m in ("fwd", "infer"):
    for d in ("fp32", "bf16"):
        src = os.path.join(G, "bench_%s_%s.json" % (m, d))
        if os.path.exists(src):
            json.dump(last_json(src), open(os.path.join(P, "%s_bench_%s_%s.json" % (rnd, m, d)), "w"), indent=1)

for src, dst in (("bench_c2.json", "bench_config2_od_bf16_1024.json"), ("bench_c3.json", "bench_config3_od_bf16_images20_n1.json"),
                 ("bench_c2_384.json", "bench_config2_od_bf16_384_b256.json"),
                 ("bench_c4.json", "bench_config4_depthnet_od_bf16_b16.json"),
                 ("bench_c4_prof.json", "bench_config4_depthnet_od_bf16_b16_kernel_classes.json"),
                 ("bench_c4_fp32.json", "bench_config4_depthnet_od_fp32_b16.json")):
    if os.path.exists(os.path.join(G, src)):
        try:
            json.dump(last_json(os.path.join(G, src)), open(os.path.join(P, "%s_%s" % (rnd, dst)), "w"), indent=1)
        except Exception as ex:
            print("skipped", src, ex)

for src, dst in (("bench_b32_fp32.json", "bench_fp32_b32.json"), ("bench_b64_fp32.json", "bench_fp32_b64.json"),
                 ("bench_b32_bf16.json", "bench_bf16_b32.json"), ("ab_bf16_xop0.json", "ab_bf16_256_xop_off.json"),
                 ("ab_bf16_xop1.json", "ab_bf16_256_xop_on.json"), ("ab_c4_r5forms.json", "ab_config4_without_fork_and_direct_grads.json"),
                 ("bench_2ranks_gloo.json", "bench_o_fp32_b64_2ranks_gloo_one_gpu.json"),
                 ("bench_c4_2ranks_gloo.json", "bench_config4_depthnet_od_bf16_2ranks_gloo_one_gpu.json")):
    if os.path.exists(os.path.join(G, src)) and os.path.getsize(os.path.join(G, src)) > 100:
        try:
            json.dump(last_json(os.path.join(G, src)), open(os.path.join(P, "%s_%s" % (rnd, dst)), "w"), indent=1)
        except Exception as ex:
            print("skipped", src, ex)
src = os.path.join(G, "per_launch_fp32_b32.txt")
if os.path.exists(src) and os.path.getsize(src) > 1000:
    open(os.path.join(P, "%s_per_launch_fp32_b32.txt" % rnd), "w").write(open(src).read())

csv.field_size_limit(1 << 30)
for rawname, outname, cmd in (("kernel_stats_raw.csv", "_bench_kernel_stats.csv", "python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary"
                               "  [--no-secondary: the default line's two secondary legs (bf16 at 256 pairs, fp32 at 32 pairs) launch the SAME kernel "
                               "families at other sizes after the timed region and would blur the per-kernel averages]"),
                              ("kernel_stats_bf16_raw.csv", "_bench_bf16_kernel_stats.csv",
                               "python3 bench.py --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline"),
                              ("kernel_stats_depthnet_bf16_raw.csv", "_bench_depthnet_od_bf16_kernel_stats.csv",
                               "python3 bench.py --algo InstaDepthNet_od --size 384 --batch 16 --dtype bf16 --no-prof "
                               "--steps 5 --warmup 2 --no-cpu-baseline")):
    raw = os.path.join(G, rawname)
    if not os.path.exists(raw):
        continue
    rows = list(csv.DictReader(open(raw)))
    with open(os.path.join(P, rnd + outname), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- " + cmd + "  "
                "(1x MI355X; 13 steps of the step kernels: 2 warm-up + 5 timed (hipGraph replay) + 1 + 5 of the profiled pass; commit " + COMMIT + ")\n# kernel names shortened (namespaces / argument lists dropped); "
                "durations in ns\nName,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
        for r in rows:
            n = r["Name"]
            m = re.search(r"(\w+)(<[^(]*>)?\(", n)
            short = (m.group(1) + (m.group(2) or "")) if m else n
            short = re.sub(r"\(anonymous namespace\)::", "", short)[:90]
            f.write('"%s",%s,%s,%s,%s,%s,%s\n' % (short, r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                                                  r["MinNs"], r["MaxNs"]))

for d in ("fp32", "bf16"):
    src = os.path.join(G, "per_launch_%s.txt" % d)
    if os.path.exists(src) and os.path.getsize(src) > 1000:
        open(os.path.join(P, "%s_per_launch_%s.txt" % (rnd, d)), "w").write(open(src).read())

vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    p = os.path.join(G, "traffic_%s.txt" % c)
    if os.path.exists(p):
        m = re.search(r"%s: n=(\d+) mean=([0-9.e+]+)" % c, open(p).read())
        if m:
            vals[c] = (int(m.group(1)), float(m.group(2)))
if len(vals) == 2:
    b = last_json(os.path.join(G, "bench_n1.json"))
    fz, wz = vals["FETCH_SIZE"][1], vals["WRITE_SIZE"][1]
    json.dump({
        "_comment": "HBM traffic of the dominant kernel from rocprofv3 --pmc (tools/collect_traffic.sh: separate "
                    "FETCH_SIZE / WRITE_SIZE passes over `bench.py --steps 2 --warmup 1`, counters only). Units: the "
                    "counters report KiB. Per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 counts wide "
                    "coalesced reads at half their size, so bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024; the raw "
                    "(uncorrected) sum is kept alongside.",
        "commit": COMMIT, "csrc_sha": CSRC, "kernel": b["roofline"]["kernel"], "launches_profiled": vals["FETCH_SIZE"][0],
        "fetch_size_kib_mean": fz, "write_size_kib_mean": wz,
        "bytes_per_launch_corrected": int((2 * fz + wz) * 1024), "bytes_per_launch_raw": int((fz + wz) * 1024),
        "algorithmic_bytes_per_launch": int(b["kernel_classes"][b["roofline"]["kernel"]]["gbs"] * 1e6
                                            * b["roofline"]["avg_launch_ms"]),
    }, open(os.path.join(P, rnd + "_pmc_traffic.json"), "w"), indent=1)
# matrix-pipe occupancy (tools/collect_traffic.sh, second part)
mf = {}
for tag in ("nt", "wgrad", "wino4", "wgradwino4"):
    p = os.path.join(G, "mfma_busy_%s.txt" % tag)
    if os.path.exists(p):
        txt = open(p).read()
        c = {m.group(1): (int(m.group(2)), float(m.group(3)), float(m.group(4)))
             for m in re.finditer(r"(\w+): n=(\d+) mean=([0-9.e+]+) sum=([0-9.e+]+)", txt)}
        if "VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
            busy, gui, cu = c["VALU_MFMA_BUSY_CYCLES"][2], c["GRBM_GUI_ACTIVE"][2], c.get("BUSY_CU_CYCLES", (0, 0, 0))[2]
            mf[tag] = {"kernel_family": txt.splitlines()[0].strip(), "launches_profiled": c["VALU_MFMA_BUSY_CYCLES"][0],
                       "SQ_VALU_MFMA_BUSY_CYCLES_sum": busy, "GRBM_GUI_ACTIVE_sum": gui, "SQ_BUSY_CU_CYCLES_sum": cu,
                       # GRBM_GUI_ACTIVE is summed over the 8 XCDs: /8 = kernel cycles; 1024 SIMDs carry a matrix pipe each
                       "mfma_busy_fraction_of_all_simd_cycles": busy / (gui / 8.0 * 1024.0),
                       "mfma_busy_fraction_of_busy_cu_cycles": busy / (4.0 * cu) if cu else None,
                       "wave_cycles_wait_any_frac": c["WAIT_ANY"][2] / c["WAVE_CYCLES"][2] if "WAIT_ANY" in c else None,
                       "wave_cycles_wait_inst_any_frac": c["WAIT_INST_ANY"][2] / c["WAVE_CYCLES"][2] if "WAIT_INST_ANY" in c else None}
if mf:
    json.dump({"_comment": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE ... over `bench.py --steps 2 "
                           "--warmup 1` (fp32 headline configuration), counters only; sums over every launch of the family.  "
                           "SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD (64 per v_mfma_f32_32x32x2_f32).  nt / wgrad: the direct "
                           "128-wide kernels; wino4 / wgradwino4: the Winograd F(4,3) forward + data gradient / filter gradient.",
               "commit": COMMIT, "csrc_sha": CSRC, **mf}, open(os.path.join(P, rnd + "_pmc_mfma_busy.json"), "w"), indent=1)
# bf16 counters (tools/collect_traffic_bf16.sh): HBM traffic + matrix-pipe occupancy of the two dominant bf16 kernels
def _fam_blocks(path):
    """{family: {counter: (n, mean, sum)}} of a tools/pmc_summary.py output with several families"""
    out, cur = {}, None
    if not os.path.exists(path):
        return out
    for line in open(path):
        if not line.startswith(" "):
            cur = line.strip()
            out.setdefault(cur, {})
        elif cur:
            for m in re.finditer(r"(\w+): n=(\d+) mean=([0-9.e+]+) sum=([0-9.e+]+)", line):
                out[cur][m.group(1)] = (int(m.group(2)), float(m.group(3)), float(m.group(4)))
    return out


bf = {}
fz, wz, mb = (_fam_blocks(os.path.join(G, n)) for n in ("bf16_traffic_FETCH_SIZE.txt", "bf16_traffic_WRITE_SIZE.txt",
                                                         "bf16_mfma_busy.txt"))
ld = _fam_blocks(os.path.join(G, "bf16_lds.txt"))
for fam in fz:
    e = {}
    if "FETCH_SIZE" in fz.get(fam, {}) and "WRITE_SIZE" in wz.get(fam, {}):
        f_, w_ = fz[fam]["FETCH_SIZE"], wz[fam]["WRITE_SIZE"]
        e.update(launches_profiled=f_[0], fetch_size_kib_mean=f_[1], write_size_kib_mean=w_[1],
                 bytes_per_launch_corrected=(2 * f_[1] + w_[1]) * 1024)
    c = mb.get(fam, {})
    if "VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
        busy, gui, cu = c["VALU_MFMA_BUSY_CYCLES"][2], c["GRBM_GUI_ACTIVE"][2], c.get("BUSY_CU_CYCLES", (0, 0, 0))[2]
        e.update(mfma_busy_fraction_of_all_simd_cycles=busy / (gui / 8.0 * 1024.0),
                 mfma_busy_fraction_of_busy_cu_cycles=busy / (4.0 * cu) if cu else None,
                 wave_cycles_wait_any_frac=c["WAIT_ANY"][2] / c["WAVE_CYCLES"][2] if "WAIT_ANY" in c else None,
                 wave_cycles_wait_inst_any_frac=c["WAIT_INST_ANY"][2] / c["WAVE_CYCLES"][2] if "WAIT_INST_ANY" in c else None)
    c = ld.get(fam, {})
    if "LDS_BANK_CONFLICT" in c and "LDS_IDX_ACTIVE" in c and c["LDS_IDX_ACTIVE"][2] > 0:
        e.update(lds_bank_conflict_cycles_per_lds_active_cycle=c["LDS_BANK_CONFLICT"][2] / c["LDS_IDX_ACTIVE"][2],
                 lds_instructions_per_launch=c["INSTS_LDS"][1] if "INSTS_LDS" in c else None)
    if e:
        bf[fam] = e
bfb = os.path.join(G, "bench_n1_bf16.json")
if bf and os.path.exists(bfb):
    b = last_json(bfb)
    # The forward / data-gradient launches are spread over four kernel families (conv_p256, conv_halo3, stem_halo and what is
    # left on conv_nt_kernel) that bench.py's launch classes (128-wide, 64-wide, stem) cut differently: one traffic ratio for
    # all of them -- measured bytes of every such launch of the profiled run over the algorithmic bytes of the same launches.
    kcs = b.get("kernel_classes", {})
    nt_classes = [kcs[k] for k in ("conv_nt_kernel<128,false>", "conv_nt_kernel<64,false>", "conv_nt_kernel<64,true>") if k in kcs]
    nt_fams = [f for f in bf if f.startswith(("conv_p256", "conv_halo3", "stem_halo", "conv_nt_kernel"))
               and "bytes_per_launch_corrected" in bf[f]]
    if nt_classes and nt_fams:
        steps = b["profiled"]["steps"]
        alg_step = sum(k["gbs"] * 1e9 * k["ms_per_step"] * 1e-3 for k in nt_classes)
        launches_step = sum(k["launches"] for k in nt_classes) / steps
        meas = sum(bf[f]["launches_profiled"] * bf[f]["bytes_per_launch_corrected"] for f in nt_fams)
        nl = sum(bf[f]["launches_profiled"] for f in nt_fams)
        bf["_forward_and_data_gradient_launches"] = {
            "families": nt_fams, "launches_profiled": nl, "launches_per_step": launches_step,
            "measured_bytes_per_step": meas / (nl / launches_step), "algorithmic_bytes_per_step": alg_step,
            "traffic_ratio": meas / (nl / launches_step) / alg_step}
    wk = kcs.get("conv_wgrad_kernel")
    for fam, e in bf.items():
        if fam.startswith("conv_wgrad") and wk and "bytes_per_launch_corrected" in e:
            alg = wk["gbs"] * 1e9 * (wk["ms_per_step"] * 1e-3) / (wk["launches"] / b["profiled"]["steps"])
            e["algorithmic_bytes_per_launch_class_mean"] = alg
    json.dump({"_comment": "bf16 headline configuration (bench.py --dtype bf16 --steps 2 --warmup 1), rocprofv3 --pmc, counters only, "
                           "separate passes (tools/collect_traffic_bf16.sh).  FETCH_SIZE / WRITE_SIZE in KiB; bytes = (2*FETCH_SIZE + "
                           "WRITE_SIZE)*1024 (gfx950 correction, MI355X_MICROARCH.md).  SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD; "
                           "busy fraction = sum / (sum GRBM_GUI_ACTIVE / 8 * 1024).",
               "commit": COMMIT, "csrc_sha": CSRC, "kernels": bf}, open(os.path.join(P, rnd + "_pmc_bf16.json"), "w"), indent=1)
print("profiles refreshed:", sorted(os.listdir(P)))
