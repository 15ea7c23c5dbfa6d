"""profiles/traffic.json from the counter passes and the kernel stats of one measurement set (not product code):

    python tools/make_traffic_json.py <tag> [--dir gpurun_out]      reads <dir>/<tag>_fetch_size_kbench_pmc.txt, <tag>_write_size_kbench_pmc.txt, <tag>_sq_kbench_pmc.txt,
                                                                    <tag>_bench_kernel_stats.txt; writes profiles/traffic.json (+ a copy <dir>/<tag>_traffic.json)

What bench.py's `roofline` block reads back as RECORDED values of the kernel build named in `kernel_id` (bench.py DOMINANT_KERNEL_ID): per-launch fabric-side bytes of the feed-forward
kernel (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16-B/lane streaming reads on gfx950 + WRITE_SIZE as reported, KiB -> bytes), its rocprofv3 in-situ mean duration, and its
matrix-pipe busy share (SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs x launch cycles)).  tools/profile_round_r06.sh runs it on the GPU box BEFORE the bench line is taken, so the line and the file agree."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = 'idf_ffn_h2::ffn_h2_kernel<2, 4, 0, 8>'
KERNEL_ID = 'idf_ffn_h2::ffn_h2_kernel r06 (8 computing + 8 loader waves)'
CLOCK_GHZ, SIMDS = 2.4, 1024


def pmc(path, counter):
    if not os.path.exists(path):
        return None
    for ln in open(path):
        if ln.startswith(KERNEL) and (' ' + counter + ' ') in ln:
            m = re.search(r'(\d+)\s+([\d.]+)\s*$', ln)
            if m:
                return float(m.group(2)), int(m.group(1))
    return None


def in_situ(path):
    if not os.path.exists(path):
        return None
    for ln in open(path):
        if ln.startswith(KERNEL):
            m = re.match(r'.*?\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s*$', ln)
            if m:
                return float(m.group(3)), int(m.group(1))
    return None


def main():
    tag = sys.argv[1]
    d = sys.argv[3] if len(sys.argv) > 3 and sys.argv[2] == '--dir' else 'gpurun_out'
    d = os.path.join(ROOT, d)
    fetch = pmc(os.path.join(d, tag + '_fetch_size_kbench_pmc.txt'), 'FETCH_SIZE')
    write = pmc(os.path.join(d, tag + '_write_size_kbench_pmc.txt'), 'WRITE_SIZE')
    busy = pmc(os.path.join(d, tag + '_sq_kbench_pmc.txt'), 'SQ_VALU_MFMA_BUSY_CYCLES')
    conf = pmc(os.path.join(d, tag + '_sq_kbench_pmc.txt'), 'SQ_LDS_BANK_CONFLICT')
    ldsa = pmc(os.path.join(d, tag + '_sq_kbench_pmc.txt'), 'SQ_LDS_IDX_ACTIVE')
    dur = in_situ(os.path.join(d, tag + '_bench_kernel_stats.txt'))
    out = dict(kernel_id=KERNEL_ID, kernel=KERNEL,
               _how='rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE / SQ_* (three separate passes, tools/gpu_pmc.sh) on `python tools/kbench.py --reps 3`, per-dispatch means of %s; '
                    'in-situ duration from rocprofv3 --kernel-trace --stats of `python bench.py` (INTERDIFF_CHAINS=1); FETCH_SIZE doubled (MI355X_MICROARCH.md: wide coalesced reads on gfx950 are '
                    'tallied at half their bytes), WRITE_SIZE as reported; KiB -> bytes.  Written by tools/make_traffic_json.py %s' % (KERNEL, tag))
    if fetch and write:
        out['ffn_fused'] = int(round((2 * fetch[0] + write[0]) * 1024))
        out['ffn_fused_detail'] = dict(FETCH_SIZE_KiB=fetch[0], WRITE_SIZE_KiB=write[0], dispatches=fetch[1], hbm_side_bytes='2 x FETCH_SIZE + WRITE_SIZE',
                                       algorithmic_bytes=5493824)
    if dur:
        out['rocprofv3_in_situ_us'] = dur[0]
        out['rocprofv3_in_situ_launches'] = dur[1]
    if busy and dur:
        out['mfma_busy_share'] = busy[0] / (SIMDS * dur[0] * 1e-6 * CLOCK_GHZ * 1e9)
        out['SQ_VALU_MFMA_BUSY_CYCLES'] = busy[0]
    if conf and ldsa:
        out['lds_bank_conflict_share'] = conf[0] / max(ldsa[0], 1.0)
    txt = json.dumps(out, indent=1)
    for path in (os.path.join(ROOT, 'profiles', 'traffic.json'), os.path.join(d, tag + '_traffic.json')):
        with open(path, 'w') as f:
            f.write(txt + '\n')
    print(txt)


if __name__ == '__main__':
    main()
