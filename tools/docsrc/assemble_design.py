"""DESIGN.md = its section sources (design_*.md, this directory) with the @PLACEHOLDERS@ filled from a bench.py line and the
gradient A/B: python tools/docsrc/assemble_design.py profiles/r06_bench.json profiles/r06_ab_group_grad.txt <gpu tests passed>"""
import json, sys, re
bj = json.loads(open(sys.argv[1]).read())
ab = open(sys.argv[2]).read() if len(sys.argv) > 2 else ""
import os
HERE = os.path.dirname(os.path.abspath(__file__))
rd_ = lambda n: open(os.path.join(HERE, n)).read()
head, s5, s6, tail78, s9 = rd_('design_1_4.md'), rd_('design_5.md'), rd_('design_6.md'), rd_('design_7_8.md'), rd_('design_9.md')
k = bj['kernels_ms_per_step']; r = bj['roofline']; rd = bj['roofline_dense']; rb = bj['roofline_backward']
e = bj['emd']; er = e['roofline']; ek = er['kernels_ms']; c3 = bj['c3']; c5 = bj['c5']; po = bj['per_op_roofline']
def f(x, n=1): return f"{x:.{n}f}"
vals = {
 'SORT': f(k['nnp_sort']*1e3), 'SWEEP': f(k['nnp_sweep']*1e3), 'GRAD': f(k['nnp_grad_sorted']*1e3),
 'SWEEPTF': f(r['achieved']), 'SWEEPFRAC': f(r['frac'],3), 'SWEEPALG': f(r['algorithmic_achieved'],0),
 'DENSE': f(rd['avg_launch_ms']*1e3,0), 'DENSETF': f(rd['achieved'],0), 'DENSEFRAC': f(rd['frac'],2),
 'GRADTB': f(rb['achieved']/1e3,1), 'GRADFRAC': f(rb['frac'],2),
 'FPS': f(c3['farthest_point_sample']['kernel_ms'],2), 'FPSIT': f(c3['farthest_point_sample']['us_per_iteration'],2),
 'BALL': f(c3['query_ball_point']['kernels_ms'].get('query_ball_boxes',0)*1e3), 'BALLMS': f(c3['query_ball_point']['ms'],3),
 'MATCH': f(ek['am_match']*1e3,0), 'MATCHTB': f(536.9e6/(ek['am_match']*1e-3)/1e12,1), 'MATCHFRAC': f(536.9e6/(ek['am_match']*1e-3)/8e12,2),
 'MC': f(po['match_cost']['avg_launch_ms']*1e3,0), 'MCTB': f(po['match_cost']['achieved']/1e3,1), 'MCFRAC': f(po['match_cost']['frac'],2),
 'MCG': f(po['match_cost_grad']['avg_launch_ms']*1e3,0), 'MCGTB': f(po['match_cost_grad']['achieved']/1e3,1), 'MCGFRAC': f(po['match_cost_grad']['frac'],2),
 'FUSED': f(e['fused']['ms_per_call'],3), 'FUSEDCPS': f(e['fused']['value'],0),
 'TNN': f(c3['three_nn']['ms'],3),
 'STEP': f(bj['ms_per_step'],4), 'VALUE': f"{bj['value']:.3g}", 'TRUE': f(bj['reference_true_shape']['ms_per_step'],4), 'ROT': f(bj['rotating_inputs']['ms_per_step'],4),
 'EMDCPS': f(e['value'],0), 'EMDMS': f(e['ms_per_call'],3), 'C3ONE': f(c3['ms_per_pass_one_call'],3), 'C3FOUR': f(c3['ms_per_pass'],3),
 'C5': f(c5['value'],0), 'TRAIN': f(c5['train_step']['ms_per_step'],1),
 'EMDVSCPU': f(e.get('vs_cpu_baseline',0),0), 'EXT50MS': f(e['extended_50']['ms_per_call'],2),
 'SKIP': f((ek['am_p1'])*1e3 + 72, 0),  # p1 + the three skipping launches (rocprofv3 timeline: 33 + 22 + 17)
 'LIVE': f((ek['am_p2'] + ek['am_p3p1'] + ek.get('am_compact',0))*1e3 - 72, 0),
}
# from the A/B tool: group_point_grad and three_interpolate_grad lines
m = re.search(r"^\s*32\s+16384\s+1024\s+32\s+64\s+atomic\s+[\d.]+ us\s+auto\s+([\d.]+) us.*?'group_point_grad_sort': ([\d.]+), 'group_point_grad': ([\d.]+)", ab, re.M)
if m:
    vals.update({'GPG': f(float(m.group(1)),0), 'GPGSORT': m.group(2), 'GPGGATHER': m.group(3), 'GPGTB': f(415.2e6/(float(m.group(1))*1e-6)/1e12,1), 'GPGFRAC': f(415.2e6/(float(m.group(1))*1e-6)/8e12,2)})
m = re.search(r"^\s*32\s+16384\s+4096\s+64\s+inline\s+[\d.]+ us\s+auto\s+([\d.]+) us", ab, re.M)
if m:
    vals.update({'TIG': f(float(m.group(1)),0), 'TIGTB': f(180.4e6/(float(m.group(1))*1e-6)/1e12,1), 'TIGFRAC': f(180.4e6/(float(m.group(1))*1e-6)/8e12,2)})
vals['NGPU'] = sys.argv[3] if len(sys.argv) > 3 else '651'
# rf_earth_mover over sizes (tools/ab_emd_cull.py base): profiles/r06_emd_sizes.txt
try:
    es = open(os.path.join(HERE, '..', '..', 'profiles', 'r06_emd_sizes.txt')).read()
    for key, shape in (('BIGMS', '4x16384'), ('BIG8', '8x8192'), ('BIG4', '16x4096'), ('BIG2', '32x2048')):
        mm = re.search(shape + r"\s+\{'total_ms': ([\d.]+)", es)
        vals[key] = f(float(mm.group(1)), 2 if key != 'BIG2' else 3) if mm else '?'
except OSError:
    pass
doc = head + s5.replace("History — every variant", "Source comments that cite \"DESIGN.md 5.x\" mean the section numbers of rounds 1–5, which DESIGN_NOTES.md keeps (Part II).\nHistory — every variant") + "\n" + s6 + "\n" + tail78 + s9
for k_, v in vals.items():
    doc = doc.replace('@' + k_ + '@', str(v))
left = set(re.findall(r'@[A-Z0-9]+@', doc))
print("unfilled:", left)
open(os.path.join(HERE, '..', '..', 'DESIGN.md'), 'w').write(doc)
print(len(doc.splitlines()), "lines")
