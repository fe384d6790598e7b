"""Turns gpurun_out/prof_<tag>/ (scripts/profile_round.sh) into the committed summaries under profiles/."""
import csv, collections, json, shutil, sys
tag = sys.argv[1]
d = 'gpurun_out/prof_%s' % tag
shutil.copy(d + '/kernel_stats.csv', 'profiles/r01_bench_kernel_stats.csv')
rows = collections.defaultdict(list)
for f in ('fetch', 'write', 'sq1', 'sq2', 'ta1', 'ta2'):
    for r in csv.DictReader(open('%s/%s_counter_collection.csv' % (d, f))):
        k = r['Kernel_Name']
        if 'srd::' in k and ('ris_kernel<0>' in k or 'final_kernel<0>' in k):
            rows[(k.split('(')[0].replace('void srd::', ''), r['Counter_Name'])].append(float(r['Counter_Value']))
lines = ["# rocprofv3 --pmc, mean per launch over the launches of `python bench.py --no-cpu-baseline --steps 5 --warmup 2` (1x MI355X).",
         "# One counter group per run (FETCH_SIZE | WRITE_SIZE | SQ group 1 | SQ group 2 | TA_TA_BUSY+GRBM_GUI_ACTIVE | TA stalls); SQ_* cycle counters are in quad-cycles,",
         "# *_sum counters are summed over the 256 CUs, GRBM_GUI_ACTIVE over the 8 XCDs. Build: quantised BVH4 (leaves <= 2 triangles), folded plane test,",
         "# nearest-first branch-free pushes, 64-thread workgroups, cost-aware tile sweep, speculative traversal, work stealing inside the wave, predicated final pass."]
m = {}
for (k, c), v in sorted(rows.items()):
    m[(k, c)] = sum(v) / len(v)
    lines.append("%-18s %-34s n=%-3d mean=%.4g" % (k, c, len(v), sum(v) / len(v)))
lines.append("")
for k in ('final_kernel<0>', 'ris_kernel<0>'):
    g = m[(k, 'GRBM_GUI_ACTIVE')] / 8
    lines.append("%s derived: lane utilisation %.1f %% (SQ_THREAD_CYCLES_VALU / 64 / SQ_ACTIVE_INST_VALU); wave time waiting %.0f %% (SQ_WAIT_ANY / SQ_WAVE_CYCLES); "
                 "mean resident waves per SIMD %.2f (4*SQ_WAVE_CYCLES / 1024 / kernel cycles); TA busy %.0f %% (TA_TA_BUSY_sum / 256 / kernel cycles), of which stalled by TC %.0f %%"
                 % (k, 100 * m[(k, 'SQ_THREAD_CYCLES_VALU')] / 64 / m[(k, 'SQ_ACTIVE_INST_VALU')], 100 * m[(k, 'SQ_WAIT_ANY')] / m[(k, 'SQ_WAVE_CYCLES')],
                    4 * m[(k, 'SQ_WAVE_CYCLES')] / 1024 / g, 100 * m[(k, 'TA_TA_BUSY_sum')] / 256 / g,
                    100 * (m[(k, 'TA_ADDR_STALLED_BY_TC_CYCLES_sum')] + m[(k, 'TA_DATA_STALLED_BY_TC_CYCLES_sum')]) / m[(k, 'TA_TA_BUSY_sum')]))
open('profiles/r01_pmc_sq_tcc_summary.txt', 'w').write("\n".join(lines) + "\n")
traffic = {"_comment": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on `python bench.py --steps 5 --warmup 2 --no-cpu-baseline`, 1x MI355X, mean per launch. FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-B requests at 64 B (MI355X_MICROARCH.md §HBM), so read bytes are doubled: traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024. The doubling is calibrated for wide streaming reads; these kernels gather 16 B per lane, so the read side is an upper bound.", "kernels": {}}
for k, name in (('ris_kernel<0>', 'ris_kernel'), ('final_kernel<0>', 'final_kernel')):
    f, w = m[(k, 'FETCH_SIZE')], m[(k, 'WRITE_SIZE')]
    traffic["kernels"][name] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "traffic_bytes_per_launch": (2 * f + w) * 1024}
json.dump(traffic, open('profiles/r01_pmc_hbm_traffic.json', 'w'), indent=1)
print("\n".join(lines[-3:]))
print(traffic["kernels"])
