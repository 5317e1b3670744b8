"""(rounds 5 and 6; `python tools/collect_profiles.py r06` after tools/gpu_r6.sh) copy the evidence of one `tools/gpu_r5e.sh` round trip from gpurun_out/ (scratch) into profiles/ (tracked) and derive the two small JSON files
bench.py quotes: r05_traffic.json (HBM bytes per launch of the NT GEMM family from the FETCH_SIZE / WRITE_SIZE passes) and r05_parity.json (the suite-wide
parity figures the GPU tests measured).   python tools/collect_profiles.py"""
import json
import os
import re
import shutil
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(R, 'gpurun_out'), os.path.join(R, 'profiles')
# python tools/collect_profiles.py            -> the first session's trip (gpurun_out/*r05e* -> profiles/r05_*)
# python tools/collect_profiles.py r05f       -> the second session's trip (tools/gpu_r5f.sh; gpurun_out/*r05f* -> profiles/r05f_*; r05_traffic.json / r05_parity.json,
#                                                    which bench.py quotes, are rewritten from it)
TAG = sys.argv[1] if len(sys.argv) > 1 else 'r05e'
PRE = 'r05' if TAG == 'r05e' else TAG
RND = PRE[:3]                                       # r05 / r06: the small JSON files bench.py quotes are named by round
COPY_ = {
    'ev_r05e_bench_line.json': 'r05_bench_line.json', 'ev_r05e_cfg2_kernel_summary.txt': 'r05_cfg2_kernel_summary.txt', 'ev_r05e_cfg2_gaps.txt': 'r05_cfg2_gaps.txt',
    'ev_r05e_cfg3_kernel_summary.txt': 'r05_cfg3_kernel_summary.txt', 'ev_r05e_cfg3_gaps.txt': 'r05_cfg3_gaps.txt',
    'ev_r05e_cfg4_kernel_summary.txt': 'r05_cfg4_kernel_summary.txt', 'ev_r05e_cfg4_gaps.txt': 'r05_cfg4_gaps.txt',
    'ev_r05e_traffic_gemm_nt.txt': 'r05_traffic_gemm_nt.txt', 'ev_r05e_traffic_tokenwise.txt': 'r05_traffic_tokenwise.txt',
    'ev_r05e_pmc_sq_attn_tokenwise.txt': 'r05_pmc_sq_attn_tokenwise.txt', 'r05e_shapes.txt': 'r05_shapes.txt', 'r05e_gemm_vs_lib.txt': 'r05_gemm_vs_lib.txt',
    'r05e_ab_qknr.txt': 'r05_ab_qknr.txt', 'r05e_parity_measured.json': 'r05_parity_measured.json',
}
COPY = {k.replace('r05e', TAG): v.replace('r05_', PRE + '_', 1) for k, v in COPY_.items()}
if TAG.startswith('r06'):
    COPY.update({f'{TAG}_attn_probe.txt': f'{TAG}_attn_probe.txt', f'{TAG}_attn_kernel_times.txt': f'{TAG}_attn_kernel_times.txt', f'{TAG}_ab_dq.txt': f'{TAG}_ab_dq.txt',
                 f'{TAG}_decode_kernels_cfg5.txt': f'{TAG}_decode_kernels_cfg5.txt', f'{TAG}_decode_prefetch.txt': f'{TAG}_decode_prefetch.txt'})
    COPY.pop(f'{TAG}_ab_qknr.txt', None)
if TAG == 'r05f':
    COPY.update({'r05f_ab_ow.txt': 'r05f_ab_ow.txt', 'r05f_cfg_ab.txt': 'r05f_cfg_ab.txt', 'r05f_probe_cmp.txt': 'r05f_probe_cmp.txt', 'r05f_probe_pp.txt': 'r05f_probe_pp.txt',
                 'r05f_probe_ow.txt': 'r05f_probe_ow.txt', 'r05f_probe_steady.txt': 'r05f_probe_steady.txt'})
    for case in ('sq4096', 'n512k512'):
        for v in ('base', 'pp'):
            COPY[f'r05f_pmc_{v}_{case}.txt'] = f'r05f_pmc_sq_{v}_{case}.txt'
    COPY.pop('r05f_ab_qknr.txt', None)
for src, dst in COPY.items():
    if os.path.exists(os.path.join(G, src)):
        shutil.copy(os.path.join(G, src), os.path.join(P, dst))
    else:
        print('missing', src)
with open(os.path.join(P, PRE + '_cfg3_cfg4_lines.txt'), 'w') as f:
    for c in (3, 4):
        log = os.path.join(G, f'ev_{TAG}_cfg{c}.log')
        if os.path.exists(log):
            f.writelines(ln for ln in open(log) if ln.startswith(f'config {c}'))

# ---- HBM traffic of the NT family (the file holds one FETCH_SIZE and one WRITE_SIZE table; its "calls/step" column divides the run's 3 steps by 2)
rows = {'FETCH_SIZE': [], 'WRITE_SIZE': []}
cur = None
for ln in open(os.path.join(P, PRE + '_traffic_gemm_nt.txt')):
    if ln.startswith('kernel'):
        cur = ln.split()[-1]
    elif cur and ln.strip():
        m = re.match(r'(\S.*?)\s+([\d.]+)\s+([\d.e+]+)\s*$', ln)
        if m and 'gemm_nt' in m.group(1):
            rows[cur].append((float(m.group(2)), float(m.group(3))))
calls = sum(c for c, _ in rows['FETCH_SIZE']) / 1.5
fetch_kb, write_kb = sum(v for _, v in rows['FETCH_SIZE']) / 1.5, sum(v for _, v in rows['WRITE_SIZE']) / 1.5
json.dump({'kernel_family': 'tfx_gemm_nt',
           'source': 'profiles/' + PRE + '_traffic_gemm_nt.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, bench.py --steps 1 --warmup 1 --family-steps 0 + the '
                     'structure-miss step = 3 steps; the file\'s per-step columns divide by 2)',
           'launches_per_step': calls, 'fetch_kb_raw_per_step': fetch_kb, 'write_kb_per_step': write_kb,
           'correction': 'FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B for wide coalesced reads; MI355X_MICROARCH.md HBM section); WRITE_SIZE uncalibrated, taken as reported',
           'bytes_per_launch': (2 * fetch_kb + write_kb) * 1024 / calls}, open(os.path.join(P, RND + '_traffic.json'), 'w'), indent=1)

# ---- suite-wide parity figures
pm = json.load(open(os.path.join(P, PRE + '_parity_measured.json')))
cases = [k for k in pm if 'logits_rel' in pm[k]]
out = {'north_star': 'outputs within 1e-3 bf16 tolerance, token argmax bit-exact',
       'loss': {'gate': 1e-3, 'met': all(pm[k].get('loss_rel', 0) <= 1e-3 for k in pm), 'worst_measured': max(pm[k].get('loss_rel', 0) for k in pm),
                'note': 'every loss of every golden case (tests/test_model_gpu.py LOSS_TOL)'},
       'logits_rel_frobenius': {'gate': 1e-2, 'met_1e-3': False, 'measured': {k: round(pm[k]['logits_rel'], 5) for k in cases},
                                'note': "bf16 activations: the reference's own bf16-autocast run is at 4.9e-3 (SURVEY section 6); 1e-3 is not reachable with bf16 storage"},
       'argmax': {'bit_exact_where_reference_top2_margin_gt_0.05': all(pm[k].get('argmax_margin_gt_0p05', 1.0) == 1.0 for k in cases),
                  'unfiltered_agreement': {k: round(pm[k]['argmax_unfiltered'], 4) for k in cases if 'argmax_unfiltered' in pm[k]},
                  'note': 'flips only at near-ties of random-init logits (recorded margins < 0.05)'},
       'gradients': {'norm_weighted_mean_rel': {k: round(pm[k]['grad_mean_rel'], 5) for k in pm if 'grad_mean_rel' in pm[k]}},
       'source': 'tests/test_model_gpu.py on MI355X (round ' + RND[2] + ', gpurun_out/parity_measured.json -> profiles/' + PRE + '_parity_measured.json); bench_shape is replaced by the in-run measurement'}
json.dump(out, open(os.path.join(P, RND + '_parity.json'), 'w'), indent=1)
print('ok', json.load(open(os.path.join(P, RND + '_traffic.json')))['bytes_per_launch'], out['loss'], len(cases), 'cases')
