#!/usr/bin/env python
"""DESIGN.md = docs/DESIGN.template.md with its ⟨TOKENS⟩ replaced by the numbers of the committed end-state evidence (profiles/r06_*):
every figure of DESIGN.md sections 4-5 and 7 that describes the end state is read from those files, not typed.  Run from the repo root."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda f: os.path.join(ROOT, "profiles", f)      # noqa: E731
j = json.load(open(P("r06_bench_final.json")))
c3 = json.load(open(P("r06_bench_config3_final.json")))
pmc = json.load(open(P("r06_pmc_traffic.json")))


def kstat(pattern):
    for line in open(P("r06_final_kernel_stats.txt")):
        m = re.match(r"\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(.*)", line)
        if m and pattern in m.group(5):
            return float(m.group(3))
    raise KeyError(pattern)


def traffic(pattern):
    for k, v in pmc.items():
        if pattern in k and "fetch_bytes_per_launch" in v:
            return v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"]
    raise KeyError(pattern)


ffn, att, opj, qkv, lsc = (kstat(k) for k in ("ffn_fused_kernel", "attention_kernel<st::OpF16, false, false", "oproj_ws_kernel", "qkv_ws_kernel",
                                               "conv_gemm_phased3_kernel<st::OpF16, 1, true>"))
rg, tr = j["ragged"], j["train_step"]
b64 = "see profiles/r06_pytest_gpu_final.log"
for line in open(P("r06_pytest_gpu_final.log")):
    if "B=64 T=1000 ragged" in line:
        m = re.search(r"worst non-q/k ([\d.e+-]+); q/k end-to-end ([\d.e+-]+), min cosine ([\d.]+); d mu ([\d.e+-]+), d c ([\d.e+-]+)", line)
        if m:
            b64 = f"worst non-q/k tensor {m.group(1)}, d mu {m.group(4)}, d c {m.group(5)}; q / k end to end {m.group(2)}, cosine {m.group(3)}"
tok = {
    "FFN_US": f"{ffn:.1f}", "FFN_FRAC": f"{201.3266 / ffn / 2.5:.3f}", "ATT_US": f"{att:.1f}", "ATT_FRAC": f"{65.536 / att / 2.5:.3f}",
    "OPJ_US": f"{opj:.1f}", "OPJ_TBS": f"{traffic('oproj_ws_kernel') / opj / 1e6:.1f}", "QKV_US": f"{qkv:.1f}", "QKV_TBS": f"{traffic('qkv_ws_kernel') / qkv / 1e6:.1f}",
    "LSC_US": f"{lsc:.1f}", "LSC_FRAC": f"{50.33 / lsc / 2.5:.3f}",
    "HEAD_MS": f"{j['ms_per_step']:.2f}", "HEAD_MFS": f"{j['value'] / 1e6:.3f}", "HEAD_TF": f"{j['whole_solve_tflops']:.0f}",
    "HEAD_FRAC": f"{j['whole_solve_tflops'] / 2500:.3f}", "HBM_TBS": f"{j['whole_solve_hbm']['achieved'] / 1e3:.2f}" if j.get("whole_solve_hbm") else "n/a",
    "RF_US": f"{j['roofline']['avg_launch_us']:.1f}", "RF_FRAC": f"{j['roofline']['frac']:.3f}", "BF16_MS": f"{j['other_dtype']['ms_per_step']:.2f}",
    "RAG_MS": f"{rg['ms_per_pass']:.1f}", "RAG_MFS": f"{rg['value'] / 1e6:.3f}", "RAG_MIN": f"{min(rg['ms_per_bucket']):.1f}", "RAG_MAX": f"{max(rg['ms_per_bucket']):.1f}",
    "RAG_CEIL": f"{rg['measured_ceiling_one_bucket_per_gpu']['speedup']:.2f}", "C3_MS": f"{c3['ms_per_step']:.1f}", "C1_MS": f"{j['config1_latency']['ms_per_solve']:.2f}",
    "SPLIT_MS": f"{j['attention_precision_split']['ms_per_step']:.2f}", "TR_F": f"{tr['ms_forward']:.2f}", "TR_B": f"{tr['ms_backward']:.2f}",
    "TR_O": f"{tr['ms_optimizer_incl_repack']:.2f}", "TR_MS": f"{tr['ms_step_back_to_back']:.2f}", "TR_MS_FUSED": f"{tr.get('ms_step_back_to_back_fused_adamw', float('nan')):.2f}", "TR_TF": f"{tr['tflops_fwd_bwd_3x_forward']:.0f}",
    "TR_FRAC": f"{tr['frac_of_mfma_peak']:.3f}", "VOC_MS": f"{j['vocoder']['ms_per_batch']:.2f}", "CPU_FS": f"{j['cpu_baseline']['value']:.0f}", "B64_TXT": b64,
}
s = open(os.path.join(ROOT, "docs", "DESIGN.template.md")).read()
missing = set(re.findall(r"⟨([A-Z0-9_]+)⟩", s)) - set(tok)
assert not missing, missing
for k, v in tok.items():
    s = s.replace("⟨" + k + "⟩", v)
open(os.path.join(ROOT, "DESIGN.md"), "w").write(s)
print("DESIGN.md written;", len(s.splitlines()), "lines")
