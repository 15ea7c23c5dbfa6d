"""Regenerate every measured number the documents quote from the measurement files (not product code).

    python tools/render_tables.py [--round r06] [--check]

Inputs (all under profiles/): <round>_bench.json (the bench line of tools/profile_round_<round>.sh), parity_<round>.json (worst errors recorded by the GPU suite),
<round>_kernel_stats_bench.txt (rocprofv3 kernel stats of the same bench command, with its launch-order table).
Outputs: profiles/TABLES_<round>.md (every block) and, in DESIGN.md / BASELINE.md / README.md, the text between

    <!-- BEGIN GENERATED <block> -->   ...   <!-- END GENERATED <block> -->

is replaced by the freshly rendered block of that name (blocks: headline, legs, roofline, step_table, parity_full, parity_misc).  Nothing between such markers is ever
edited by hand: round 4 spent 15 of 37 commits re-typing these numbers into five files.  --check: exit 1 if a document would change (CI-style guard)."""
import argparse
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_json(path):
    try:
        with open(path) as f:
            return json.load(f)
    except Exception:
        return {}


def g(d, *keys, default=None):
    for k in keys:
        if not isinstance(d, dict) or k not in d:
            return default
        d = d[k]
    return d


def f(v, fmt='%.4f', na='n/a'):
    return (fmt % v) if isinstance(v, (int, float)) else na


def sci(v):
    return ('%.1e' % v) if isinstance(v, (int, float)) else 'n/a'


def headline(b, tag, label):
    s = g(b, 'ms_per_step_samples', 'all', default=[])
    return ('Measured state (%s, one MI355X, `profiles/%s_bench.json`): **%s ms per denoising step = %s M frame-steps/s** over whole 1000-step samples with their 11 correction '
            'steps (median of %d samples: %s); no correction %s ms/step; denoiser forward (22 launches as one graph replay) %s µs; the timed route equals the eager route bit '
            'for bit (%s).' % (label, tag, f(b.get('ms_per_step')), f(b.get('value', 0) / 1e6 if b.get('value') else None, '%.2f'), len(s), ' / '.join('%.4f' % x for x in s),
                               f(g(b, 'no_correction', 'ms_per_step')), f(g(b, 'denoiser_forward', 'us'), '%.1f'), 'asserted in the run' if g(b, 'timed_route_equals_eager_route', 'ok') else 'NOT confirmed'))


def legs(b):
    rows = [('#2 B=16, T=100, correction (**headline**)', b.get('ms_per_step'), b.get('value')),
            ('#2 the same clips, no correction', g(b, 'no_correction', 'ms_per_step'), g(b, 'no_correction', 'value')),
            ('#3 B=32, T=100, correction', g(b, 'config3_B32_correction', 'ms_per_step'), g(b, 'config3_B32_correction', 'value')),
            ("reference's own default shape B=32, T=35, correction", g(b, 'reference_default_B32_T35', 'ms_per_step'), g(b, 'reference_default_B32_T35', 'value')),
            ('#4 per-GPU share: 8 clips, 5 windows incl. conditioning', g(b, 'config4_long_horizon', 'ms_per_step'), g(b, 'config4_long_horizon', 'value'))]
    out = ['| configuration | ms per step | frame-steps/s |', '|---|---|---|']
    for name, ms, val in rows:
        out.append('| %s | %s | %s |' % (name, f(ms), ('%.2f M' % (val / 1e6)) if isinstance(val, (int, float)) else 'n/a'))
    po = b.get('post_optimisation') or {}
    if po:
        out.append('| #5 post-optimisation, 16 clips x 20 frames, 200 Adam iterations | %s ms per iteration | %s clips/s |' % (f(po.get('ms_per_iteration'), '%.3f'), f(po.get('clips_per_sec'), '%.1f')))
    cb = b.get('cpu_baseline') or {}
    if cb:
        out.append('')
        out.append('CPU baseline (`cpu_baseline.kind = "%s"`, %s threads): plain denoising steps %s frame-steps/s -- **GPU / CPU %sx, the quotable ratio** --; blended over the 989 + 11 step mix %s frame-steps/s '
                   '(%sx: dominated by the oracle\'s brute-force nearest-neighbour search).  Port vs the reference\'s own source on the same cores (recorded, build container, %s threads): %s (plain step) / %s (correction call).'
                   % (cb.get('kind'), cb.get('cores'), f(g(cb, 'plain_only', 'value'), '%.0f'), f(g(cb, 'plain_only', 'gpu_over_cpu'), '%.0f'), f(cb.get('value'), '%.0f'), f(cb.get('gpu_over_cpu_blended'), '%.0f'),
                      g(cb, 'port_vs_reference', 'threads'), f(g(cb, 'port_vs_reference', 'plain_step'), '%.2f'), f(g(cb, 'port_vs_reference', 'correction_call'), '%.2f')))
    return '\n'.join(out)


def roofline(b):
    r = b.get('roofline') or {}
    sr = b.get('step_roofline') or {}
    out = ['| quantity | value |', '|---|---|',
           '| dominant kernel | `%s` (`%s`), %s µs per launch live (best burst %s) |' % (r.get('kernel'), r.get('kernel_id'), f(r.get('us_per_launch'), '%.2f'), f(r.get('us_per_launch_best'), '%.2f')),
           '| bound / peak | %s, %s %s (f16 dense MFMA peak / 3: the roof of an fp32-grade product on the pipe the kernel issues on) |' % (r.get('bound'), f(r.get('peak'), '%.0f'), r.get('unit')),
           '| achieved (ALGORITHMIC fp32 FLOP per launch / duration) | %s %s = **frac %s** |' % (f(r.get('achieved'), '%.1f'), r.get('unit'), f(r.get('frac'), '%.3f')),
           '| rocprofv3 in-situ (recorded) | %s µs -> frac %s |' % (f(g(r, 'rocprofv3_in_situ', 'us_per_launch'), '%.2f'), f(g(r, 'rocprofv3_in_situ', 'frac'), '%.3f')),
           '| f16 FLOP issued (3 products + zero padding; secondary) | %s TFLOP/s = %s of the f16 dense peak; recorded PMC matrix-pipe busy share %s |'
           % (f(g(r, 'issued_f16', 'achieved_tflops'), '%.1f'), f(g(r, 'issued_f16', 'frac_of_f16_dense_peak'), '%.3f'), f(g(r, 'mfma_busy_share', 'recorded_pmc'), '%.3f')),
           '| binding resource | weight stream %s GB/s per CU of a ~%s GB/s L2-fed DMA ceiling + fixed phases |' % (f(g(r, 'binding_resource', 'weight_stream_gb_per_s_per_cu'), '%.1f'), f(g(r, 'binding_resource', 'l2_fed_dma_ceiling_gb_per_s_per_cu'), '%.0f')),
           '| traffic per launch (PMC, recorded) | %s B vs %s B algorithmic |' % (r.get('traffic'), g(r, 'traffic_vs_algorithmic', 'algorithmic_bytes')),
           '| exact-fp32 kernel, same process | %s µs, frac %s of the fp32-MFMA peak |' % (f(g(r, 'exact_fp32_kernel', 'us_per_launch'), '%.2f'), f(g(r, 'exact_fp32_kernel', 'frac'), '%.3f')),
           '| whole step vs the chip | %s µs per step; matrix roof %s µs (%s), memory roof %s µs (%s): **step frac %s** |'
           % (f(sr.get('us_per_step'), '%.1f'), f(g(sr, 'matrix_roof', 'us'), '%.1f'), f(g(sr, 'matrix_roof', 'frac_of_step'), '%.3f'), f(g(sr, 'memory_roof', 'us'), '%.1f'), f(g(sr, 'memory_roof', 'frac_of_step'), '%.3f'), f(sr.get('frac'), '%.3f'))]
    worst = max([v for v in [r.get('frac'), g(r, 'one_layer_burst', 'frac'), g(r, 'two_chain_form', 'frac'), g(r, 'small_batch_16_row_tile', 'frac'), g(r, 'large_batch_64_row_tile', 'frac'),
                             g(r, 'exact_fp32_kernel', 'frac'), g(r, 'issued_f16', 'frac_of_f16_dense_peak'), sr.get('frac')] if isinstance(v, (int, float))] or [0])
    out.append('| largest fraction anywhere in the block | %s (none may exceed 1) |' % f(worst, '%.3f'))
    ex = b.get('exclusive_cu') or {}
    if ex:
        out.append('| f16-MFMA kernels that do NOT own their CU | %s of %d |' % (ex.get('kernels_not_exclusive'), len(ex.get('table') or [])))
    ar = b.get('arithmetic_by_layer') or {}
    if ar:
        out.append('| arithmetic by layer | %s |' % ('every contraction of every layer split-f16, self-attention included' if ar.get('all_split') else
                                                   '; '.join('L%d %s' % (d['layer'], '/'.join('%s=%s' % (k, v) for k, v in d.items() if k not in ('layer', 'kind'))) for d in ar.get('layers', []))))
    return '\n'.join(out)


def step_table(path):
    if not os.path.exists(path):
        return '(no kernel stats file: %s)' % os.path.relpath(path, ROOT)
    lines = open(path).read().splitlines()
    out, grab = [], False
    for ln in lines:
        if re.match(r'\s*#\s+kernel\s+avg_us', ln):
            grab = True
            out += ['| # | kernel | µs (mean, in situ) | gap before (µs) |', '|---|---|---|---|']
            continue
        if grab:
            m = re.match(r'\s*(\d+)\s+(.*?)\s+([\d.]+)\s+([\d.]+)\s*$', ln)
            if m:
                out.append('| %s | `%s` | %s | %s |' % (m.group(1), m.group(2).strip(), m.group(3), m.group(4)))
            elif 'sum of kernels' in ln:
                out.append('')
                out.append(ln.strip())
                break
    top = ['| kernel | launches | µs mean | % of GPU time |', '|---|---|---|---|']
    for ln in lines[1:14]:
        m = re.match(r'(.*?)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$', ln)
        if m:
            top.append('| `%s` | %s | %s | %s |' % (m.group(1).strip(), m.group(2), m.group(4), m.group(7)))
    return '\n'.join(top) + '\n\n' + ('\n'.join(out) if out else '(no launch-order table in the stats file)')


def parity_full(p):
    out = []
    for key, title in (('full_size_end_to_end_well_conditioned_split', 'well-conditioned fixture (`fullwc.npz`), shipped split-f16 arithmetic'),
                       ('full_size_end_to_end_well_conditioned_exact', 'well-conditioned fixture (`fullwc.npz`), exact fp32 MFMA'),
                       ('full_size_end_to_end', 'random-init fixture (`full.npz`), shipped arithmetic')):
        e = p.get(key)
        if not e:
            out.append('*%s: not recorded*' % title)
            continue
        out.append('**%s** (B=16, T=100, P=2048, 1000 steps, 11 corrections, injected noise; max|Δ|/max|ref|):' % title)
        out += ['', '| quantity | HIP vs reference | HIP vs fp64 | reference vs fp64 |', '|---|---|---|---|']
        for idx, d in sorted((e.get('sampler_state_rel_err_by_loop_index') or {}).items(), key=lambda kv: int(kv[0])):
            out.append('| sampler state, loop index %s | %s | %s | %s |' % (idx, sci(d.get('hip_vs_reference')), sci(d.get('hip_vs_fp64')), sci(d.get('reference_vs_fp64'))))
        fr, f64, y = e.get('final_outputs_rel_err_vs_reference') or {}, e.get('final_outputs_hip_vs_fp64') or {}, e.get('final_outputs_reference_vs_fp64') or {}
        for k in fr:
            out.append('| final %s | %s | %s | %s |' % (k.replace('_', ' '), sci(fr.get(k)), sci(f64.get(k)), sci(y.get(k))))
        m = e.get('metrics_rel_err_vs_reference') or {}
        out.append('| six metrics (worst; penetration ratio) | %s; %s | | |' % (sci(max([v for k, v in m.items() if k != 'penetrate'] or [0])), sci(m.get('penetrate'))))
        out.append('| hook decisions flipped (condition / contact marker, of 176) | %s / %s | | %s / %s |' % (e.get('condition_flips_vs_reference'), e.get('contact_marker_flips_vs_reference'),
                                                                                                          e.get('reference_vs_fp64_condition_flips'), e.get('reference_vs_fp64_marker_flips')))
        out.append('')
    return '\n'.join(out)


def parity_misc(p):
    out = ['| check (key of the parity file) | worst relative error / result |', '|---|---|']
    for k in sorted(p):
        if k.startswith('full_size_end_to_end') or k == 'exclusive_cu_report':
            continue
        e = p[k]
        vals = ', '.join('%s %s' % (kk, sci(v) if isinstance(v, float) else v) for kk, v in e.items() if isinstance(v, (int, float)))
        out.append('| `%s` | %s |' % (k, vals[:260]))
    return '\n'.join(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--round', default='r06')
    ap.add_argument('--check', action='store_true')
    a = ap.parse_args()
    tag = a.round
    prof = os.path.join(ROOT, 'profiles')
    b, p = load_json(os.path.join(prof, '%s_bench.json' % tag)), load_json(os.path.join(prof, 'parity_%s.json' % tag))
    blocks = dict(headline=headline(b, tag, 'round ' + tag[1:].lstrip('0')), legs=legs(b), roofline=roofline(b), step_table=step_table(os.path.join(prof, '%s_kernel_stats_bench.txt' % tag)),
                  parity_full=parity_full(p), parity_misc=parity_misc(p))
    tables = ['# Measured tables of %s (generated by tools/render_tables.py from profiles/%s_bench.json, parity_%s.json, %s_kernel_stats_bench.txt -- do not edit)\n' % (tag, tag, tag, tag)]
    for name, body in blocks.items():
        tables.append('## %s\n\n%s\n' % (name, body))
    changed = []
    targets = [(os.path.join(prof, 'TABLES_%s.md' % tag), '\n'.join(tables), True)]
    for doc in ('DESIGN.md', 'BASELINE.md', 'README.md'):
        path = os.path.join(ROOT, doc)
        txt = open(path).read()
        new = txt
        for name, body in blocks.items():
            pat = re.compile(r'(<!-- BEGIN GENERATED %s -->\n)(.*?)(<!-- END GENERATED %s -->)' % (name, name), re.S)
            new = pat.sub(lambda m, body=body: m.group(1) + body + '\n' + m.group(3), new)
        targets.append((path, new, False))
    for path, new, whole in targets:
        old = open(path).read() if os.path.exists(path) else None
        if old != new:
            changed.append(os.path.relpath(path, ROOT))
            if not a.check:
                with open(path, 'w') as fh:
                    fh.write(new)
    print(('would change: ' if a.check else 'updated: ') + (', '.join(changed) if changed else 'nothing'))
    return 1 if (a.check and changed) else 0


if __name__ == '__main__':
    sys.exit(main())
