#!/usr/bin/env python
"""MFMA utilisation of the convolution kernels of the bench step from a rocprofv3 --pmc pass:

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d DIR -o pmc -- python bench.py ...
    python tools/pmc_mfma.py DIR/pmc_results.db --steps 3 --csv out.csv

Only dispatches of the last `--steps` steps are used (window between the matching prefilter_scan
launches: excludes MIOpen's find-mode exploration).  Counter rows come per dispatch and per hardware
instance; per dispatch: mfma = SUM over instances, gui = MAX over instances (cycles the dispatch was
active), util = mfma / (gui * 1024 SIMDs) -- the average fraction of the chip's 1024 matrix pipes that
was busy (counter semantics: MI355X_MICROARCH.md, per-instruction cycle constants)."""
import argparse, sqlite3, collections

ap = argparse.ArgumentParser()
ap.add_argument('db'); ap.add_argument('--steps', type=int, default=3); ap.add_argument('--csv')
a = ap.parse_args()
cur = sqlite3.connect(a.db).cursor()
marks = cur.execute("select d.start from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id "
                    "where s.kernel_name like '%prefilter_scan%' order by d.start").fetchall()
t1 = marks[-1][0]
t0 = marks[-a.steps - 1][0] if len(marks) > a.steps else 0
rows = cur.execute("select d.id, s.kernel_name, d.end-d.start, p.name, sum(e.value), max(e.value), count(*) from rocpd_pmc_event e "
                   "join rocpd_info_pmc p on e.pmc_id=p.id join rocpd_kernel_dispatch d on e.event_id=d.event_id "
                   "join rocpd_info_kernel_symbol s on d.kernel_id=s.id where d.start > ? and d.start <= ? "
                   "group by d.id, p.name", (t0, t1)).fetchall()
disp = collections.defaultdict(dict)
for did, name, ns, pmc, vsum, vmax, n in rows:
    disp[did].update(name=name, ns=ns)
    disp[did][pmc] = (vsum, vmax, n)
agg = collections.defaultdict(lambda: [0, 0, 0.0, 0.0, 0])
for d in disp.values():
    if 'SQ_VALU_MFMA_BUSY_CYCLES' not in d or 'GRBM_GUI_ACTIVE' not in d:
        continue
    g = agg[d['name']]
    g[0] += 1; g[1] += d['ns']; g[2] += d['SQ_VALU_MFMA_BUSY_CYCLES'][0]; g[3] += d['GRBM_GUI_ACTIVE'][1]
    g[4] = max(g[4], d['SQ_VALU_MFMA_BUSY_CYCLES'][2])
tot = sum(g[1] for g in agg.values()) or 1
lines = ['kernel,calls,total_ns,percent_of_gpu_time,mfma_busy_cycles_sum,gui_active_cycles_max_sum,instances,mfma_util']
out = sorted(agg.items(), key=lambda kv: -kv[1][1])
conv_m = conv_g = conv_ns = 0
for name, (c, ns, m, g, inst) in out:
    util = m / (g * 1024.0) if g else 0.0
    lines.append('"%s",%d,%d,%.2f,%d,%d,%d,%.4f' % (name[:100].replace(',', ';'), c, ns, 100.0 * ns / tot, m, g, inst, util))
    if 'conv' in name or 'igemm' in name:
        conv_m += m; conv_g += g; conv_ns += ns
for l in lines[:16]:
    print(l[:200])
print('window: %d steps, %.2f ms of kernels; conv kernels %.1f %% of GPU time, time-weighted MFMA util %.3f'
      % (a.steps, tot / 1e6, 100.0 * conv_ns / tot, conv_m / (conv_g * 1024.0) if conv_g else 0))
if a.csv:
    open(a.csv, 'w').write('\n'.join(lines) + '\n')
