#!/usr/bin/env python3
"""Summarise rocprofv3 outputs under gpurun_out/ into small committed files under profiles/<round>/.

  python profiles/summarize_pmc.py gpurun_out profiles/r01

Round 2: python profiles/summarize_pmc.py gpurun_out/r02c profiles/r02   (also copies the bench lines / sweep / text artefacts)
Reads  <in>/pmc_*/**/_counter_collection.csv (+ kernel_trace.csv for durations) and <in>/prof*/**/_kernel_stats.csv.
Writes <out>/pmc_summary.json (per kernel: mean duration, counters, derived MFMA utilisation and clock) and
       profiles/traffic.json (HBM bytes per launch per pipeline stage; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes
       for wide coalesced reads on gfx950, WRITE_SIZE as is; units KiB -> bytes).
"""
import collections
import csv
import glob
import json
import os
import sys

STAGE_OF = {'traj_chain_kernel': 'trajectory_chain', 'mlp_block0_kernel': 'mlp_block0', 'mlp_block1_kernel': 'mlp_block1', 'post_attn_kernel': 'post_attn',
            'embed_qkv_kernel': 'embed_qkv', 'embed_qkv_lat_kernel': 'embed_qkv', 'agents_fused_kernel': 'agents_fused', 'mhgsa_attn_kernel': 'mhgsa_attn', 'linear_cols_kernel': 'agent_preact', 'gru_cols_kernel': 'gru_cols'}


def main(src, dst):
    os.makedirs(dst, exist_ok=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    newest = {}
    for f in glob.glob(os.path.join(src, 'pmc_*', '*', '*_counter_collection.csv')):
        d = os.path.dirname(f)
        if d not in newest or os.path.getmtime(f) > os.path.getmtime(newest[d]):
            newest[d] = f   # gpurun merges runs into the same directory: keep only the latest pass of each counter set
    for f in newest.values():
        trace = f.replace('_counter_collection.csv', '_kernel_trace.csv')
        dur = {r['Dispatch_Id']: int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in csv.DictReader(open(trace))}
        seen = set()
        for r in csv.DictReader(open(f)):
            name = r['Kernel_Name'].split('(')[0].replace('void ', '')
            # gru_cols runs twice per step (per agent / per trajectory) with the same persistent grid: split by duration
            if 'gru_cols' in name:
                name += '[trajectories]' if dur[r['Dispatch_Id']] > 300000 else '[agents]'
            agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
            key = (f, r['Dispatch_Id'])
            if key not in seen:
                seen.add(key)
                agg[name]['_dur_ns'].append(dur[r['Dispatch_Id']])
    out = {}
    for name, c in agg.items():
        if not any(k in name for k in STAGE_OF):
            continue
        e = {k: sum(v) / len(v) for k, v in c.items()}
        e['mean_us'] = e.pop('_dur_ns') / 1e3
        if 'GRBM_GUI_ACTIVE' in e and 'SQ_VALU_MFMA_BUSY_CYCLES' in e:
            cyc = e['GRBM_GUI_ACTIVE'] / 8.0                      # summed over the 8 XCDs
            e['clock_GHz'] = cyc / (e['mean_us'] * 1e3)
            e['mfma_busy_frac'] = e['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * cyc)   # 1024 SIMDs
        if 'FETCH_SIZE' in e or 'WRITE_SIZE' in e:
            e['hbm_bytes_per_launch'] = (2.0 * e.get('FETCH_SIZE', 0.0) + e.get('WRITE_SIZE', 0.0)) * 1024.0
        out[name] = e
    json.dump(out, open(os.path.join(dst, 'pmc_summary.json'), 'w'), indent=1, sort_keys=True)
    traffic = {}
    for name, e in out.items():
        if 'hbm_bytes_per_launch' not in e:
            continue
        base = name.split('<')[0].split('[')[0]
        st = STAGE_OF.get(base)
        targs = [t.strip() for t in name.split('<', 1)[1].rstrip('>').split(',')] if '<' in name else []
        if base == 'traj_chain_kernel' and len(targs) > 1 and targs[1] in ('1', '2', 'true'):   # FUSE != 0: roles + groups in one launch
            if targs[1] != '2' and 'agents+trajectory_chain[fused launch]' in traffic:
                continue                                          # (the lagged launch of the pipelined path is the one bench.py times)
            st = 'agents+trajectory_chain[fused launch]'
        if st == 'gru_cols':
            st = 'gru_cols[block1,trajectories]' if 'trajectories' in name else 'gru_cols[block0,agents]'
        if st:
            traffic[st] = e['hbm_bytes_per_launch']
    # bench.py looks the dominant kernel's traffic up as traffic.json[leg][stage]; the PMC passes run the headline leg
    tp = os.path.join(os.path.dirname(dst.rstrip('/')), 'traffic.json')
    allt = json.load(open(tp)) if os.path.exists(tp) else {}
    if allt and not all(isinstance(v, dict) for v in allt.values()):
        allt = {'round1_three_kernel_form': allt}
    allt['eth_512'] = traffic
    json.dump(allt, open(tp, 'w'), indent=1, sort_keys=True)
    stats = {}
    for f in glob.glob(os.path.join(src, 'prof*', '*', '*_kernel_stats.csv')):   # gpurun merges runs into one directory: newest pass per tag
        tag = f.split(os.sep)[-3]
        if tag not in stats or os.path.getmtime(f) > os.path.getmtime(stats[tag]):
            stats[tag] = f
    for tag, f in stats.items():
        rows = list(csv.DictReader(open(f)))[:14]
        for r in rows:                                   # torch's template instantiations run to kilobytes: keep the head of the name
            if len(r['Name']) > 140:
                r['Name'] = r['Name'][:137] + '...'
        with open(os.path.join(dst, f'{tag}_kernel_stats.csv'), 'w') as fo:
            w = csv.DictWriter(fo, fieldnames=rows[0].keys())
            w.writeheader()
            w.writerows(rows)
    print(json.dumps({k: {kk: round(vv, 4) for kk, vv in v.items() if kk in ('mean_us', 'clock_GHz', 'mfma_busy_frac', 'hbm_bytes_per_launch')}
                      for k, v in out.items()}, indent=1))


def copy_lines(src, dst):
    import shutil
    sweep = {}
    for f in sorted(glob.glob(os.path.join(src, '*.json'))):
        name = os.path.basename(f)
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if name.startswith('sweep_s'):
            r = d.get('roofline') or {}
            sweep[name[len('sweep_s'):-5]] = {'scenes': d['config']['scenes_per_gpu'], 'trajectories': d['config']['trajectories_rank0'],
                                              'ms_per_step': d['ms_per_step'], 'M_traj_per_s': d['value'] / 1e6, 'dominant_kernel': r.get('kernel'),
                                              'frac': r.get('frac'), 'tflops': {k: v.get('tflops') for k, v in d['kernels'].items() if 'tflops' in v}}
        else:
            shutil.copy(f, os.path.join(dst, name))
    if sweep:
        json.dump(dict(sorted(sweep.items(), key=lambda kv: int(kv[0]))), open(os.path.join(dst, 'batch_sweep.json'), 'w'), indent=1)
    for f in glob.glob(os.path.join(src, '*.txt')):
        shutil.copy(f, os.path.join(dst, os.path.basename(f)))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
    copy_lines(sys.argv[1], sys.argv[2])
