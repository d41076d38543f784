"""Condense the output of tools/profile.sh (gpurun_out/<tag>/) into tracked files under profiles/:
  r01_<run>_stats.csv / _bench.json   rocprofv3 --kernel-trace --stats tables and the bench line of the same command
  r01_pmc_means.csv                   PMC counters: per kernel, mean per dispatch AND sum per bench step
  r01_summary.md                      the tables + derived HBM traffic / MFMA-busy / clock figures
usage: python tools/summarize_profiles.py <gpurun_out tag> [<round prefix, default r02>]"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
RP = sys.argv[2] if len(sys.argv) > 2 else "r06"
src = f"gpurun_out/{tag}"
STEPS = 22   # tools/profile.sh runs bench.py with --warmup 2 --steps 10; bench.py then repeats 10 searches with per-launch events (roofline pass)


def short(n):
    m = re.search(r'(scan_topk_kernel|scan_screen_lean3_kernel|scan_screen_lean2_kernel|scan_screen_lean_kernel|scan_screen_kernel|k_rescore|k_split_rows|k_seed_thr|k_img_err|merge_keys_partial_kernel|merge_keys_kernel|'
                  r'merge_lists_kernel|merge_select_kernel|merge_wg_kernel|k_gather_flagged|k_ffn3|k_ffn2|k_ffn_fused|k_gemm3|k_qkv_attn_small|k_gemm_small|k_gemm<\d, \d, \d+, \d(?:, \w+)?>|k_attn3|k_attention|k_layernorm|k_embed_ln|k_pool|k_tokens_out|k_cls_head)', n)
    s = m.group(1) if m else n[:40]
    if s in ('scan_topk_kernel',):
        c = re.search(r'Cfg<([^>]*)>', n)
        s += '<Cfg<%s>>' % c.group(1).replace(', ', ';') if c else ''
    return s


rows = []
for f in sorted(glob.glob(src + '/*_counters.csv')):
    p = os.path.basename(f).replace('_counters.csv', '')
    acc = collections.defaultdict(list)
    meta = {}
    for r in csv.DictReader(open(f)):
        k = (short(r['Kernel_Name']), r['Counter_Name'])
        acc[k].append(float(r['Counter_Value']))
        meta[k[0]] = (r['Workgroup_Size'], r['LDS_Block_Size'], r['Scratch_Size'], r['VGPR_Count'], r['Accum_VGPR_Count'], r['SGPR_Count'])
    for (kn, cn), v in sorted(acc.items()):
        rows.append((p, kn, cn, len(v), sum(v) / len(v), sum(v) / STEPS) + meta[kn])
with open(f'profiles/{RP}_pmc_means.csv', 'w') as o:
    o.write('pass,kernel,counter,dispatches,mean_per_dispatch,sum_per_bench_step,wg,lds_block,scratch,vgpr,agpr,sgpr\n')
    for r in rows:
        o.write(','.join(str(x) for x in r) + '\n')
for t in ['scan', 'exact', 'scan_b1', 'exact_b1', 'scan_b32', 'scan_b16', 'embed']:
    if not os.path.exists(f'{src}/{t}_stats.csv'):
        continue
    shutil.copy(f'{src}/{t}_stats.csv', f'profiles/{RP}_{t}_stats.csv')
    open(f'profiles/{RP}_{t}_bench.json', 'w').write(open(f'{src}/{t}_bench.json').read().strip().splitlines()[-1] + '\n')


def stats(t, title):
    out = [f"## kernel-trace --stats, {title}\n", "| kernel | calls | total ms | avg us | % |", "|---|---|---|---|---|"]
    for r in csv.DictReader(open(f'profiles/{RP}_{t}_stats.csv')):
        n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')
        out.append(f"| `{n[:78]}` | {r['Calls']} | {int(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e3:.1f} | {r['Percentage']} |")
    b = json.loads(open(f'profiles/{RP}_{t}_bench.json').read())
    if t == 'embed' and b.get('secondary'):
        out.append(f"\nbench leg of the same command (`profiles/{RP}_{t}_bench.json`): " + json.dumps({k: v for k, v in b['secondary'][0].items() if k != 'config'}))
    elif b.get('roofline'):
        rf = b['roofline']
        out.append(f"\nbench line of the same command (`profiles/{RP}_{t}_bench.json`): value {b['value']} {b['unit']}, ms_per_step {b['ms_per_step']}, "
                   f"hipEvent kernel_ms {rf['kernel_ms']} over {rf['launch']['launches']} launch(es), roofline {rf['bound']} {rf['achieved']} / {rf['peak']} {rf['unit']} = {rf['frac']}")
    else:
        out.append(f"\nbench line (`profiles/{RP}_{t}_bench.json`): {json.dumps(b)}")
    return "\n".join(out) + "\n"


pm = collections.defaultdict(dict)
for r in rows:
    d = pm[(r[0], r[1])]
    d[r[2]] = (r[4], r[5])
    d['_n'] = r[3]
    d['_meta'] = f"wg {r[6]} lds {r[7]} scratch {r[8]} vgpr {r[9]} agpr {r[10]} sgpr {r[11]}"


def line(p, k):
    d = pm[(p, k)]
    return (f"- `{p}` `{k}` ({d['_n']} dispatches in {STEPS} bench steps; {d['_meta']}), per bench step: "
            + ", ".join(f"{c}={v[1]:.4g}" for c, v in d.items() if not c.startswith('_')))


SK1 = 'scan_screen_lean3_kernel'        # one query tile (batch <= 128): the 4-wave nt instantiation of the same kernel
SK = 'scan_screen_lean3_kernel'         # full query tiles: the headline's kernel (round 4)


def find(p, prefix):
    ks = [k for (pp, k) in pm if pp == p and k.startswith(prefix)]
    return ks[0]


EK = find('exact_pmc_a', 'scan_topk_kernel<Cfg<384;4;')
B1 = find('exact_pmc_b1', 'scan_topk_kernel<Cfg<384;1;')
f = lambda p, k, c: pm[(p, k)][c][1]      # per bench step
sb = json.loads(open(f'profiles/{RP}_scan_bench.json').read())
eb = json.loads(open(f'profiles/{RP}_exact_bench.json').read())
k_ms, e_ms = sb['roofline']['kernel_ms'], eb['roofline']['kernel_ms']
scr_fetch = f('pmc_b', SK, 'FETCH_SIZE') * 2048
scr_write = f('pmc_c', SK, 'WRITE_SIZE') * 1024
hit = f('pmc_c', SK, 'TCC_HIT_sum') / (f('pmc_c', SK, 'TCC_HIT_sum') + f('pmc_c', SK, 'TCC_MISS_sum'))
clk_s = f('pmc_a', SK, 'GRBM_GUI_ACTIVE') / 8 / (k_ms * 1e-3) / 1e9
clk_e = f('exact_pmc_a', EK, 'GRBM_GUI_ACTIVE') / 8 / (e_ms * 1e-3) / 1e9
txt = [
    f"# Round {int(RP[1:])} rocprofv3 summary (MI355X, gfx950, ROCm 7.2)",
    f"Produced by `tools/profile.sh {tag}` on the GPU box, condensed by `tools/summarize_profiles.py {tag}`; per-kernel tables:",
    f"`{RP}_*_stats.csv`; counters (mean per dispatch and sum per bench step): `{RP}_pmc_means.csv`.  The screening path launches its",
    "kernel once per row range of the threshold ladder (7 launches per 10M-row batch of 1024 queries, 5 for <= 128 queries), so its counters are summed per bench step.\n",
    stats('scan', 'headline bench (10M x 384 fp32, batch 1024, top-10), default path = fp16 screening ladder + exact fp32 re-score'),
    stats('exact', 'same workload forced onto the exact fp32 scan (`RMU_SCREEN=0`)'),
    stats('scan_b1', 'HBM-bound regime: batch 1, default path (fp16 image, 768 B per row, screening ladder with ratio 8)'),
    stats('exact_b1', 'batch 1 forced onto the exact fp32 scan (`RMU_SCREEN=0`, WQ=1 geometry)'),
    stats('scan_b32', 'north-star regime: batch 32 over 10M rows, default path (fp16 image, nt stream, ladder ratio 8)'),
    stats('scan_b16', 'north-star regime: batch 16 over 10M rows (the leg bench.py reports as roofline.north_star)') if os.path.exists(f'profiles/{RP}_scan_b16_stats.csv') else '',
    stats('embed', 'encoder: 8192-chunk calls x ~128 tokens (BERT-6x384, bf16 MFMA; k_ffn3 / k_attn3 / k_gemm3 / k_gemm)'),
    "## PMC passes (separate runs, `--kernel-trace --pmc ...` only)\n",
    line('pmc_a', SK), line('pmc_b', SK), line('pmc_c', SK), line('exact_pmc_a', EK), line('exact_pmc_b', EK), line('pmc_b1', SK1), line('exact_pmc_b1', B1), line('pmc_b32', SK1), line('pmc_b16', SK1) if ('pmc_b16', SK1) in pm else '',
    "\n## Derived (FETCH_SIZE is in KiB and under-reports by 2x on gfx950 per MI355X_MICROARCH.md -> bytes = FETCH_SIZE x 1024 x 2)\n",
    f"- screening launches, per batch: HBM fetch {scr_fetch/1e9:.2f} GB + write {scr_write/1e6:.1f} MB; the fp16 image is 7.68 GB and each of the 4 "
    f"query-tile workgroups of a row chunk streams it (L2 hit {hit:.3f}; ideal 0.75), i.e. x{scr_fetch/7.68e9:.2f} the image, x{scr_fetch/15.36e9:.2f} the "
    f"fp32 corpus the exact scan reads; MFMA busy {f('pmc_a',SK,'SQ_VALU_MFMA_BUSY_CYCLES')/(f('pmc_a',SK,'GRBM_GUI_ACTIVE')*128):.3f} of SIMD-cycles "
    f"(SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs x 256 CUs x 4 SIMDs)); mean clock over the launches "
    f"{clk_s:.2f} GHz (GRBM_GUI_ACTIVE/8 / {k_ms} ms): the part down-clocks from 2.4 GHz under the combined MFMA + LDS + L2 load",
    f"- exact kernel: HBM fetch {f('exact_pmc_b',EK,'FETCH_SIZE')*2048/1e9:.2f} GB per launch vs 15.36 GB algorithmic (x{f('exact_pmc_b',EK,'FETCH_SIZE')*2048/15.36e9:.3f}); "
    f"MFMA busy {f('exact_pmc_a',EK,'SQ_VALU_MFMA_BUSY_CYCLES')/(f('exact_pmc_a',EK,'GRBM_GUI_ACTIVE')*128):.3f} of SIMD-cycles; mean clock {clk_e:.2f} GHz",
    f"- batch 1, default path: HBM fetch {f('pmc_b1',SK1,'FETCH_SIZE')*2048/1e9:.3f} GB per query over all launches vs the 7.680 GB image (x{f('pmc_b1',SK1,'FETCH_SIZE')*2048/7.68e9:.4f})",
    f"- batch 32, default path: HBM fetch {f('pmc_b32',SK1,'FETCH_SIZE')*2048/1e9:.3f} GB per batch over all launches vs the 7.680 GB image (x{f('pmc_b32',SK1,'FETCH_SIZE')*2048/7.68e9:.4f})",
    (f"- batch 16, default path: HBM fetch {f('pmc_b16',SK1,'FETCH_SIZE')*2048/1e9:.3f} GB per batch over all launches vs the 7.680 GB image (x{f('pmc_b16',SK1,'FETCH_SIZE')*2048/7.68e9:.4f})" if ('pmc_b16', SK1) in pm else ''),
    f"- batch 1, exact scan: HBM fetch {f('exact_pmc_b1',B1,'FETCH_SIZE')*2048/1e9:.3f} GB per launch vs 15.360 GB algorithmic (x{f('exact_pmc_b1',B1,'FETCH_SIZE')*2048/15.36e9:.4f}) -- no wasted re-reads",
]
open(f'profiles/{RP}_summary.md', 'w').write("\n".join(txt) + "\n")
for t in ('c2', 'shard8'):      # round 5: BASELINE configs[1] and the 8-way shard size (optional outputs of tools/profile.sh)
    if os.path.exists(f'{src}/{t}_stats.csv'):
        shutil.copy(f'{src}/{t}_stats.csv', f'profiles/{RP}_{t}_stats.csv')
        open(f'profiles/{RP}_{t}_bench.json', 'w').write(open(f'{src}/{t}_bench.json').read().strip().splitlines()[-1] + '\n')
        open(f'profiles/{RP}_summary.md', 'a').write("\n" + stats(t, {'c2': 'BASELINE.json configs[1]: 1M x 384, batch 1024', 'shard8': 'the 8-way shard size: 1.25M x 384, batch 1024'}[t]))
# C5's dense part and the encoder PMC pass (optional outputs of tools/profile.sh)
if os.path.exists(f'{src}/top100_stats.csv'):
    shutil.copy(f'{src}/top100_stats.csv', f'profiles/{RP}_top100_stats.csv')
    if os.path.exists(f'{src}/top100_probe.txt'):
        shutil.copy(f'{src}/top100_probe.txt', f'profiles/{RP}_top100_probe.txt')
if os.path.exists(f'{src}/enc_pmc_counters.csv'):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f'{src}/enc_pmc_counters.csv')):
        n = r['Kernel_Name']
        kn = next((k for k in ('k_ffn3', 'k_ffn2', 'k_gemm3', 'k_attn3', 'k_gemm_small', 'k_gemm') if k in n), n[:30])
        acc[kn][r['Counter_Name']].append(float(r['Counter_Value']))
    cn = ['GRBM_GUI_ACTIVE', 'SQ_INSTS_MFMA', 'SQ_INSTS_VALU', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_WAVE_CYCLES', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_BUSY_CYCLES']
    ks = [k for k in ('k_ffn3', 'k_ffn2', 'k_gemm3', 'k_gemm', 'k_attn3') if k in acc]
    m = {k: {c: sum(v) / len(v) for c, v in acc[k].items()} for k in ks}
    out = [f"# Encoder kernels: PMC pass (round {int(RP[1:])})\n",
           "`rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU "
           "SQ_INSTS_MFMA --kernel-include-regex \"k_ffn3|k_gemm3|k_attn3|k_gemm\" -- python tools/enc_smoke.py 2048` (tools/profile.sh); 2048 chunks (~262 k tokens), "
           "means over the 6 launches (one per layer) of each kernel; raw sums over the chip (GRBM_GUI_ACTIVE summed over the 8 XCDs; SQ_WAVE_CYCLES / SQ_WAIT_INST_ANY / "
           "SQ_ACTIVE_INST_ANY in quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES in cycles: MI355X_MICROARCH.md).\n",
           "| counter | " + " | ".join(f"`{k}`" for k in ks) + " |", "|---|" + "---|" * len(ks)]
    for c in cn:
        out.append(f"| {c} | " + " | ".join(f"{m[k].get(c, float('nan')):,.0f}".replace(",", " ") for k in ks) + " |")
    out += ["\nDerived (1024 SIMDs; kernel cycles = GRBM_GUI_ACTIVE / 8):\n", "| | " + " | ".join(f"`{k}`" for k in ks) + " |", "|---|" + "---|" * len(ks)]
    cyc = {k: m[k]['GRBM_GUI_ACTIVE'] / 8 for k in ks}
    out.append("| kernel cycles | " + " | ".join(f"{cyc[k] / 1e6:.3f} M" for k in ks) + " |")
    out.append("| MFMA pipe busy = MFMA_BUSY / 1024 / cycles | " + " | ".join(f"**{m[k]['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc[k]:.3f}**" for k in ks) + " |")
    out.append("| VALU instructions per MFMA (INSTS_VALU counts the MFMAs too) | " + " | ".join(f"{m[k]['SQ_INSTS_VALU'] / max(m[k]['SQ_INSTS_MFMA'], 1):.1f}" for k in ks) + " |")
    out.append("| share of wave time waiting (WAIT_INST_ANY / WAVE_CYCLES) | " + " | ".join(f"{m[k]['SQ_WAIT_INST_ANY'] / m[k]['SQ_WAVE_CYCLES']:.2f}" for k in ks) + " |")
    out.append("| share of wave time with an instruction active | " + " | ".join(f"{m[k]['SQ_ACTIVE_INST_ANY'] / m[k]['SQ_WAVE_CYCLES']:.2f}" for k in ks) + " |")
    out.append("| resident waves per SIMD = WAVE_CYCLES x 4 / 1024 / cycles | " + " | ".join(f"{m[k]['SQ_WAVE_CYCLES'] * 4 / 1024 / cyc[k]:.2f}" for k in ks) + " |")
    open(f'profiles/{RP}_encoder_pmc.md', 'w').write("\n".join(out) + "\n")
# encoder HBM traffic (round 4): one forward of the embed / rerank workloads, every encoder kernel: 2 x FETCH_SIZE (gfx950 correction for
# wide coalesced reads -- the LDS-DMA / 16-byte loads these kernels use) + WRITE_SIZE (uncalibrated: taken at face value), KiB -> bytes
enc_entries, enc_lines = [], []
for name, units in (("embed", 8192), ("rerank", 6400)):
    ff, fw = f'{src}/enc_{name}_f_counters.csv', f'{src}/enc_{name}_w_counters.csv'
    if not (os.path.exists(ff) and os.path.exists(fw)):
        continue
    per = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for path, col in ((ff, 0), (fw, 1)):
        for r in csv.DictReader(open(path)):
            if r['Counter_Name'] in ('FETCH_SIZE', 'WRITE_SIZE'):
                k = short(r['Kernel_Name'])
                per[k][col] += float(r['Counter_Value'])
                if col == 0:
                    per[k][2] += 1
    fetch = sum(v[0] for v in per.values()) * 2048
    write = sum(v[1] for v in per.values()) * 1024
    enc_entries.append({"path": name, "rows": units, "batch": 0, "bytes_per_step": fetch + write, "fetch_bytes": fetch, "write_bytes": write})
    enc_lines.append(f"\n### {name}: one forward of {units} {'chunks' if name == 'embed' else 'pairs'} (`tools/enc_smoke.py`), all encoder kernels: "
                     f"fetch {fetch / 1e9:.2f} GB (2 x FETCH_SIZE) + write {write / 1e9:.2f} GB = **{(fetch + write) / 1e9:.2f} GB**\n")
    enc_lines += ["| kernel | launches | fetch GB | write GB |", "|---|---|---|---|"]
    for k, v in sorted(per.items(), key=lambda kv: -kv[1][0]):
        enc_lines.append(f"| `{k}` | {v[2]} | {v[0] * 2048 / 1e9:.3f} | {v[1] * 1024 / 1e9:.3f} |")
if enc_lines:
    open(f'profiles/{RP}_encoder_traffic.md', 'w').write(
        f"# Encoder HBM traffic (round {int(RP[1:])}): `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, tools/profile.sh step 7)\n"
        + "\n".join(enc_lines) + "\n")
# what bench.py reports as roofline.traffic: HBM bytes per bench step of exactly these configurations, from THIS round's passes
json.dump({"source": f"tools/profile.sh {tag} -> tools/summarize_profiles.py: sum per bench step over the launches of (2 x FETCH_SIZE [gfx950 correction] "
                     f"+ WRITE_SIZE where collected) x 1024, profiles/{RP}_pmc_means.csv",
           "entries": [{"path": "screen", "rows": 10_000_000, "batch": 1024, "bytes_per_step": scr_fetch + scr_write},
                       {"path": "screen", "rows": 10_000_000, "batch": 32, "bytes_per_step": f('pmc_b32', SK1, 'FETCH_SIZE') * 2048},
                       {"path": "screen", "rows": 10_000_000, "batch": 1, "bytes_per_step": f('pmc_b1', SK1, 'FETCH_SIZE') * 2048}]
                      + ([{"path": "screen", "rows": 10_000_000, "batch": 16, "bytes_per_step": f('pmc_b16', SK1, 'FETCH_SIZE') * 2048}] if ('pmc_b16', SK1) in pm else []) + [
                       {"path": "exact", "rows": 10_000_000, "batch": 1024, "bytes_per_step": f('exact_pmc_b', EK, 'FETCH_SIZE') * 2048},
                       {"path": "exact", "rows": 10_000_000, "batch": 1, "bytes_per_step": f('exact_pmc_b1', B1, 'FETCH_SIZE') * 2048}] + enc_entries},
          open(f'profiles/{RP}_traffic.json', 'w'), indent=1)
print("\n".join(txt[-4:]))
print("SCREEN_TRAFFIC =", scr_fetch + scr_write, "B1 screen", f('pmc_b1',SK1,'FETCH_SIZE')*2048, "B32 screen", f('pmc_b32',SK1,'FETCH_SIZE')*2048, "exact", f('exact_pmc_b',EK,'FETCH_SIZE')*2048, "exact B1", f('exact_pmc_b1',B1,'FETCH_SIZE')*2048)
