"""Summarise the rocprofv3 counter passes of scripts/pmc_kernels.sh: mean counter values per dispatch for every kernel (template
arguments kept, argument lists dropped), plus derived figures:
  hbm_traffic_bytes = 2 * FETCH_SIZE[KB] * 1024 + WRITE_SIZE[KB] * 1024     (gfx950: FETCH_SIZE reports half of a 16-byte-per-lane
                                                                              streaming read, MI355X_MICROARCH.md "HBM")
  mfma_busy_frac    = SQ_VALU_MFMA_BUSY_CYCLES / (128 * GRBM_GUI_ACTIVE)    busy cycles are summed over the chip's 1024 SIMDs (64 per
                      v_mfma_f32_32x32x2_f32: BUSY / SQ_INSTS_MFMA = 64.0 in every pass), GRBM_GUI_ACTIVE over the 8 XCDs, so
                      kernel cycles = GUI_ACTIVE / 8 and the SIMD-cycle budget is 1024 * GUI_ACTIVE / 8
usage: python scripts/pmc_summarize.py <dir with p1..pN>  ->  <dir>/kernel_pmc.json
"""
import collections, csv, glob, json, os, re, sys

out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(out, 'p*', '*', '*_counter_collection.csv'))):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name']
        name = name.replace('(anonymous namespace)::', '')
        name = re.sub(r'\(.*$', '', name).replace('void ', '').replace('ide3d::', '').strip()
        if name.startswith('at::') or name.startswith('Cijk') or 'rocclr' in name:
            name = re.sub(r'<.*', '', name)
        acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
res = {}
for k, cs in sorted(acc.items()):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    m['dispatches'] = max(len(v) for v in cs.values())
    if 'FETCH_SIZE' in m and 'WRITE_SIZE' in m:
        m['hbm_traffic_bytes'] = 2 * m['FETCH_SIZE'] * 1024 + m['WRITE_SIZE'] * 1024
    if m.get('GRBM_GUI_ACTIVE') and 'SQ_VALU_MFMA_BUSY_CYCLES' in m:
        m['mfma_busy_frac'] = m['SQ_VALU_MFMA_BUSY_CYCLES'] / (128.0 * m['GRBM_GUI_ACTIVE'])
        m['kernel_cycles'] = m['GRBM_GUI_ACTIVE'] / 8.0
    if m.get('SQ_LDS_IDX_ACTIVE'):
        m['lds_bank_conflict_frac'] = m.get('SQ_LDS_BANK_CONFLICT', 0.0) / m['SQ_LDS_IDX_ACTIVE'] if 'SQ_LDS_BANK_CONFLICT' in m else None
    if m.get('SQ_WAVE_CYCLES'):
        for c in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS'):
            if c in m: m[c.lower() + '_frac_of_wave_cycles'] = m[c] / m['SQ_WAVE_CYCLES']
    res[k] = m
json.dump(res, open(os.path.join(out, 'kernel_pmc.json'), 'w'), indent=1)
for k, m in res.items():
    line = f'{k[:90]:90s}'
    if 'hbm_traffic_bytes' in m: line += f" hbm {m['hbm_traffic_bytes'] / 1e6:9.1f} MB"
    if 'mfma_busy_frac' in m and m.get('SQ_INSTS_MFMA', 0) > 0: line += f" mfma_busy {m['mfma_busy_frac']:.2f}"
    print(line)
