#!/usr/bin/env python3
"""Rewrites the round-3 tables of DESIGN.md section 7 from the bench lines and counter profiles under profiles/
(r03_bench_*.json, r03_extend_pmc_*.json), so that the document quotes what the committed files hold."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(t):
    return json.loads(open(os.path.join(ROOT, "profiles", "r03_bench_%s.json" % t)).read().strip().split("\n")[-1])


def main():
    names = ['c4', '1m', 'c5', 'c3', 'c2', 'c4_device_build', 'c4_wide_nodes', 'c5_origins_inside', 'c1_rtiow_640x480_128spp',
             'c3_1024spp_strong_n1', '100Mtri_4k_device_build', 'launcher_2ranks_gloo_1Mtri']
    b = {t: load(t) for t in names}
    w = lambda t: b[t]['value']
    e = lambda t: b[t]['roofline']['extend_Mrays_per_s']
    rows = []
    c4 = b['c4']
    rows.append("| C4 10 M triangles, 4K, 16 spp/pass, 8 bounces (default `bench.py`) | `python bench.py` | **%.0f Mrays/s**, %.4f s per 4K pass (492 M rays) | **%.0f Mrays/s**, 32 launches, %.1f ms each | round 2: 3104 / 3566; CPU oracle %.1f Mrays/s on 16 host threads (`r03_bench_c4.json`, `r03_extend_pmc_c4.json`). Steps of the round in A/B runs on one box: kernel restructuring, extend 3575 → 3930; 128-byte shade records, shade 18.8 → 17.9 ms per pass; camera rays refilled a whole wave at a time, extend +0.9 %%; the samples of a pixel in one wave, extend 3955 → 4100 and whole job 3421 → 3536; sixteen work counters and runs of 128 rays, extend +2 %% |" % (w('c4'), c4['config']['s_per_4k_frame'], e('c4'), c4['roofline']['mean_launch_ms'], c4['cpu_baseline']['value']))
    rows.append("| the same over the 128-byte nodes (exact boxes) | `RTCORE_NODES=wide python bench.py` | %.0f Mrays/s | %.0f Mrays/s, %.1f ms per launch | `r03_bench_c4_wide_nodes.json` (round 2: 2733 / 3085) |" % (w('c4_wide_nodes'), e('c4_wide_nodes'), b['c4_wide_nodes']['roofline']['mean_launch_ms']))
    rows.append("| the same, tree built on the GPU | `--builder device` | %.0f Mrays/s | %.0f Mrays/s | `r03_bench_c4_device_build.json`; 28.8 instead of 27.4 node visits but 7.9 instead of 9.0 primitive tests per ray |" % (w('c4_device_build'), e('c4_device_build')))
    m = b['100Mtri_4k_device_build']
    rows.append("| 100 M triangles, 4K, 16 spp, tree built on the GPU | `--tris 100000000 --builder device` | %.0f Mrays/s | %.0f Mrays/s | `r03_bench_100Mtri_4k_device_build.json`: %.0f GB of scene built in %.2f s on the device, %.1f node visits per ray (round 2: 2559 / 2905) |" % (w('100Mtri_4k_device_build'), e('100Mtri_4k_device_build'), m['config']['scene_MB'] / 1000, m['config']['scene_build_ms_library'] / 1000, m['roofline']['node_visits_per_ray']))
    rows.append("| 1 M triangles, 4K, 16 spp | `--tris 1000000` | %.0f Mrays/s | %.0f Mrays/s | `r03_bench_1m.json`, `r03_extend_pmc_1m.json` (round 2: 4160 / 4922); north-star target ≥ 1 Grays/s |" % (w('1m'), e('1m')))
    rows.append("| C3 tea set (17,228 tris, glass, HDR sky), 4K, 16 spp, 1 GPU | `--workload teaset` | %.0f Mrays/s | %.1f Grays/s | `r03_bench_c3.json`, `r03_extend_pmc_c3.json` (round 2: 7576 / 10.3) |" % (w('c3'), e('c3') / 1000))
    s = b['c3_1024spp_strong_n1']
    rows.append("| C3 as stated: 1024 spp in all, N = 1 anchor of the strong-scaling line | `--workload teaset --spp 16 --total-passes 64 --gpus 1` | %.0f Mrays/s, 64 passes in %.2f s | %.1f Grays/s | `r03_bench_c3_1024spp_strong_n1.json`; with `--gpus 8` the same command is the 8-GPU run (the driver's) |" % (w('c3_1024spp_strong_n1'), s['ms_per_step'] * 64 / 1000, e('c3_1024spp_strong_n1') / 1000))
    rows.append("| C2 Cornell, 1080p, 64 spp/pass | `--workload cornell --width 1920 --height 1080 --spp 64` | %.1f Grays/s | %.1f Grays/s | `r03_bench_c2.json`, `r03_extend_pmc_c2.json` (round 2: 11.6 / 18.4; 12.0 / 19.4 with one work counter) |" % (w('c2') / 1000, e('c2') / 1000))
    c1 = b['c1_rtiow_640x480_128spp']
    rows.append("| C1 procedural spheres scene, 640 × 480, 128 spp | `--workload rtiow --width 640 --height 480 --spp 128` | %.1f Grays/s, %.1f ms per pass | %.1f Grays/s | `r03_bench_c1_rtiow_640x480_128spp.json`; CPU oracle %.1f Mrays/s on 16 threads |" % (w('c1_rtiow_640x480_128spp') / 1000, c1['ms_per_step'], e('c1_rtiow_640x480_128spp') / 1000, c1['cpu_baseline']['value']))
    rows.append("| C5 100 M random rays vs 1 M triangles | `--workload closest_hit --tris 1000000` | %.1f Grays/s | %.1f Grays/s | `r03_bench_c5.json`, `r03_extend_pmc_c5.json`; shallow (3.8 node visits per ray: origins in the 1.5× inflated box); 13.6 / 19.5 with one work counter |" % (w('c5') / 1000, e('c5') / 1000))
    rows.append("| the same with origins among the meshes | `--workload closest_hit --tris 1000000 --inflate 0.7` | %.1f Grays/s | %.1f Grays/s | `r03_bench_c5_origins_inside.json`: 15.9 node visits, 5.1 primitive tests per ray (round 2: 5.0 / 6.0) |" % (w('c5_origins_inside') / 1000, e('c5_origins_inside') / 1000))
    l = b['launcher_2ranks_gloo_1Mtri']
    rows.append("| launcher rehearsal, 2 ranks sharing the one GPU, gloo | `RT_BENCH_BACKEND=gloo python bench.py --gpus 2 --tris 1000000` | (not a rate: one card) | — | `r03_bench_launcher_2ranks_gloo_1Mtri.json`: `n_gpus` %d, `ranks_seen` %d, sum through the host %.0f ms |" % (l['n_gpus'], l['ranks_seen'], l['config']['reduce_ms']))
    krows = []
    label = {'c4': 'C4', '1m': '1 M tris', 'c5': 'C5', 'c3': 'C3', 'c2': 'C2'}
    for t in ['c4', '1m', 'c5', 'c3', 'c2']:
        r = b[t]['roofline']
        l1, f, v, h = r['l1'], r['fill'], r['valu'], r['hbm']
        bound = r['bound']
        fill = '**%.2f**' % f['frac'] if bound == 'fill' else '%.2f' % f['frac']
        iss = '**%.2f**' % v['issue_frac'] if bound == 'valu' else '%.2f' % v['issue_frac']
        l1s = '%.2f (**%.2f**)' % (l1['accesses_per_clk_and_cu'], l1['frac']) if bound == 'l1' else '%.2f (%.2f)' % (l1['accesses_per_clk_and_cu'], l1['frac'])
        krows.append("| %s | extend | %.2f | %s | %s | %.2f | %.2f | %s × %.2f = %.2f | %.2f | %.2f | %.1f TB/s |" % (label[t], r['profile']['mean_launch_ms_under_profile'], l1s, fill, l1['ta_busy'], l1['td_busy'], iss, v['valu_lane_utilisation'], v['frac'], v['scalar_per_vector_inst'], v['wait_any_frac_of_wave_cycles'], h['achieved_GBps'] / 1000))
    p = json.load(open(os.path.join(ROOT, 'profiles', 'r03_extend_pmc_c4.json')))['shade']
    cyc = p['clock_GHz_under_profile'] * 1e9 * p['mean_launch_ms'] / 1e3 * 256
    tt = p['mean_launch_ms'] / 1e3
    krows.append("| C4 | shade | %.2f (2.35 with 96-byte strides and 8 × 8 tiles) | %.2f | — | %.2f | %.2f | — × %.2f | — | — | %.1f TB/s of reads (%.1f GB per launch, was 8.1); with its writes and the state it streams the stage moves ≈ 114 GB per pass in 17.5 ms, 6.5 TB/s |" % (p['mean_launch_ms'], p['TCP_TOTAL_CACHE_ACCESSES_sum_per_launch'] / cyc, p['TA_TA_BUSY_sum_per_launch'] / cyc, p['TD_TD_BUSY_sum_per_launch'] / cyc, p['valu_lane_utilisation'], p['hbm_bytes_per_launch_raw'] * 1.941 / tt / 1e12, p['hbm_bytes_per_launch_raw'] * 1.941 / 1e9))
    r = c4['roofline']
    tail = ("`bench.py`'s line for the default configuration: `roofline.bound = \"%s\"`, `frac = %.2f` (`achieved` %.1f of `peak` %.1f G\nline fills/s, the measured fill rates mixed at the launch's L2 hit rate), `roofline.l1.frac = %.2f`, `roofline.valu.frac = %.2f`\n(`issue_frac` %.2f, ceiling named in the line), `roofline.hbm.frac = %.2f` (`traffic` %.1f GB per launch against 148.6 GB\nalgorithmic). rocprofv3's mean extend launch time (%.2f ms, `r03_rocprofv3_kernel_stats_c4.csv`) agrees with the HIP-event\nfigure of the bench line (%.2f ms).\n" % (r['bound'], r['frac'], r['achieved'], r['peak'], r['l1']['frac'], r['valu']['frac'], r['valu']['issue_frac'], r['hbm']['frac'], r['traffic'] / 1e9, r['profile']['mean_launch_ms_under_profile'], r['mean_launch_ms']))
    path = os.path.join(ROOT, 'DESIGN.md')
    d = open(path).read()
    i0 = d.index('| C4 10 M triangles, 4K, 16 spp/pass, 8 bounces (default `bench.py`) | `python bench.py` | **3')
    i1 = d.index('\n\nPer kernel, from the round-3 counter profiles')
    d = d[:i0] + "\n".join(rows) + d[i1:]
    i0 = d.index('| C4 | extend | 1')
    i1 = d.index('\n\n(bold = the unit `bench.py` names as `bound`.)')
    d = d[:i0] + "\n".join(krows) + d[i1:]
    i0 = d.index("`bench.py`'s line for the default configuration: `roofline.bound")
    i1 = d.index('### Round 2 (`profiles/r02_*`)')
    d = d[:i0] + tail + "\n" + d[i1:]
    open(path, 'w').write(d)
    print("\n".join(rows))
    print("\n".join(krows))
    print(tail)


if __name__ == "__main__":
    main()
