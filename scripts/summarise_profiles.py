"""Turns gpurun_out/prof_<tag>/ (scripts/profile_round.sh) into the committed summaries under profiles/:
<tag>_bench_kernel_stats.csv (+ _sequential), <tag>_pmc_summary.txt, <tag>_pmc_hbm_traffic.json."""
import csv, collections, glob, json, shutil, sys
tag = sys.argv[1]
d = 'gpurun_out/prof_%s' % tag
shutil.copy(d + '/kernel_stats.csv', 'profiles/%s_bench_kernel_stats.csv' % tag)
shutil.copy(d + '/kernel_stats_sequential.csv', 'profiles/%s_bench_kernel_stats_sequential.csv' % tag)
rows = collections.defaultdict(list)
for f in sorted(glob.glob(d + '/pmc*_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'srd::' in k and ('ris_kernel<0>' in k or 'final_kernel<0>' in k):
            rows[(k.split('(')[0].replace('void srd::', ''), r['Counter_Name'])].append(float(r['Counter_Value']))
lines = ["# rocprofv3 --pmc, mean per launch over the launches of `python3 bench.py --no-cpu-baseline --no-pmc --steps 5 --warmup 3` (1x MI355X), one counter",
         "# group per run (scripts/profile_round.sh). SQ_* cycle counters are in quad-cycles, *_sum counters are summed over the 256 CUs, GRBM_GUI_ACTIVE over the 8 XCDs.",
         "# Under --pmc the dispatches run one after the other (no overlap of the two frames in flight), so these are per-kernel figures."]
m = {}
for (k, c), v in sorted(rows.items()):
    m[(k, c)] = sum(v) / len(v)
    lines.append("%-18s %-34s n=%-3d mean=%.4g" % (k, c, len(v), sum(v) / len(v)))
lines.append("")
for k in ('final_kernel<0>', 'ris_kernel<0>'):
    g = m[(k, 'GRBM_GUI_ACTIVE')] / 8
    lines.append("%s derived: kernel cycles %.3g; VALU issue %.3f of peak (SQ_INSTS_VALU x 2 / (1024 SIMDs x cycles)); lane utilisation %.1f %% (SQ_THREAD_CYCLES_VALU / 64 / "
                 "SQ_ACTIVE_INST_VALU); wave time: waiting %.0f %% (SQ_WAIT_ANY), issue-stalled %.0f %% (SQ_WAIT_INST_ANY), issuing %.0f %% (SQ_ACTIVE_INST_ANY) of SQ_WAVE_CYCLES; "
                 "mean resident waves per SIMD %.2f; TA busy %.0f %% (of which stalled by TC %.0f %%); L2 hit rate %.3f (TCC_HIT / (TCC_HIT + TCC_MISS))"
                 % (k, g, m[(k, 'SQ_INSTS_VALU')] * 2 / (1024 * g), 100 * m[(k, 'SQ_THREAD_CYCLES_VALU')] / 64 / m[(k, 'SQ_ACTIVE_INST_VALU')],
                    100 * m[(k, 'SQ_WAIT_ANY')] / m[(k, 'SQ_WAVE_CYCLES')], 100 * m[(k, 'SQ_WAIT_INST_ANY')] / m[(k, 'SQ_WAVE_CYCLES')],
                    100 * m[(k, 'SQ_ACTIVE_INST_ANY')] / m[(k, 'SQ_WAVE_CYCLES')], 4 * m[(k, 'SQ_WAVE_CYCLES')] / 1024 / g, 100 * m[(k, 'TA_TA_BUSY_sum')] / 256 / g,
                    100 * (m[(k, 'TA_ADDR_STALLED_BY_TC_CYCLES_sum')] + m[(k, 'TA_DATA_STALLED_BY_TC_CYCLES_sum')]) / m[(k, 'TA_TA_BUSY_sum')],
                    m[(k, 'TCC_HIT_sum')] / (m[(k, 'TCC_HIT_sum')] + m[(k, 'TCC_MISS_sum')])))
open('profiles/%s_pmc_summary.txt' % tag, 'w').write("\n".join(lines) + "\n")
traffic = {"_comment": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of the bench command, 1x MI355X, mean per launch. FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 128-B requests at 64 B (MI355X_MICROARCH.md, HBM), so read bytes are doubled: traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024. The doubling is calibrated for wide streaming reads; these kernels gather 16 B per lane, so the read side is an upper bound. bench.py measures the same figures live in every N = 1 run (roofline.traffic).", "kernels": {}}
for k, name in (('ris_kernel<0>', 'ris_kernel'), ('final_kernel<0>', 'final_kernel')):
    f, w = m[(k, 'FETCH_SIZE')], m[(k, 'WRITE_SIZE')]
    traffic["kernels"][name] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "traffic_bytes_per_launch": (2 * f + w) * 1024}
json.dump(traffic, open('profiles/%s_pmc_hbm_traffic.json' % tag, 'w'), indent=1)
print("\n".join(lines[-3:]))
print(traffic["kernels"])
