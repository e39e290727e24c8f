"""bench.py's host logic that needs no GPU: the legs of the default run derive their arguments (and their `metric`) from the
headline's instead of hard-coding them (round 2's records all carried the headline's metric string)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import bench


def _args(**kw):
    base = dict(backbone='ResNet50FPN', batch=8, height=800, width=1280, dtype='bf16', rotated_bbox=False, unit_rotation=False,
                steps=50, warmup=10, cpu_seconds=20.0, no_eager_leg=False, other_steps=20, mode='infer')
    base.update(kw)
    return argparse.Namespace(**base)


def test_short_names():
    assert bench.short_name('ResNet50FPN') == 'RN50FPN'
    assert bench.short_name('ResNet101FPN') == 'RN101FPN'
    assert bench.short_name('ResNeXt50_32x4dFPN') == 'RNX50_32x4dFPN'


def test_legs_override_only_what_they_name():
    head = _args()
    leg = bench.leg_args(head, backbone='ResNet101FPN', batch=16)
    assert (leg.backbone, leg.batch, leg.steps, leg.cpu_seconds, leg.no_eager_leg) == ('ResNet101FPN', 16, 20, 0.0, True)
    assert leg.warmup == 5 and leg.height == 800 and leg.dtype == 'bf16'
    assert (head.backbone, head.batch, head.steps, head.cpu_seconds) == ('ResNet50FPN', 8, 50, 20.0)      # the headline's own arguments are untouched
    rot = bench.leg_args(head, rotated_bbox=True, unit_rotation=True)
    assert rot.rotated_bbox and rot.unit_rotation and not head.rotated_bbox


def test_spec_constants_match_the_survey():
    # sparse-realistic = 23 224 of 11.52 M scores of P3 clear the threshold
    assert abs(bench.SPEC_FRACTION - 23224 / 11520000.0) < 1e-12
    assert bench.SPEC_CANDIDATES == [23224, 5646, 1456, 391, 93]
    assert bench.HBM_PEAK_GBS == 8000.0 and bench.MFMA_BF16_PEAK_TFLOPS == 2500.0


# ---- the stdout contract (round 4's line carried a bare NaN and 20 KB of legs: the driver could not parse it) ----
import json
import subprocess


def _strict_loads(line):
    def bad(const):
        raise ValueError('non-JSON constant %s on the line' % const)
    return json.loads(line, parse_constant=bad)


def _full_record(legs=7):
    nan, inf = float('nan'), float('inf')
    return {
        'metric': 'images/sec end-to-end (incl. decode+NMS), RN50FPN 800px bs=8', 'value': 1100.12, 'unit': 'images/s', 'n_gpus': 1,
        'steps': 20, 'warmup': 5, 'ms_per_step': 7.27, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'bf16', 'data': 'synthetic ' * 30,
        'config': {'workload': 'ResNet50FPN bf16 inference, bs=8 per GPU at 800x1280, HIP decode x5 + NMS', 'global_batch': 8,
                   'parallelism': 'replicas x1 (no data-path collective)', 'entry': 'Model.forward (eval)', 'postproc': 'fused',
                   'graph': 'g' * 200},
        'roofline': {'kernel': 'prefilter_scan_kernel', 'bound': 'hbm', 'achieved': 5300.0, 'peak': 8000.0, 'unit': 'GB/s', 'frac': 0.6625,
                     'traffic': None, 'alg_bytes_per_launch': 245721600, 'avg_us': 46.3, 'launches': 20, 'timing': 't' * 200},
        'latency_bound': {'select': {'us_per_step': 31.0, 'lower_bound_us': 4.0, 'ratio': 7.8, 'model': 'm' * 300},
                          'nms_kernel': {'us_per_launch': 25.0, 'lower_bound_us': 7.9, 'ratio': 3.2, 'model': 'm' * 300}},
        'conv_roofline': {'bound': 'mfma', 'achieved': 690.0, 'peak': 2500.0, 'unit': 'TFLOP/s', 'frac': 0.276},
        'kernels': {'prefilter_scan_kernel': {'avg_us': 46.3, 'launches': 20}, 'select_decode_kernel': {'avg_us': nan, 'launches': 20}},
        'epilogue_roofline': {'bias_act_kernel': {'x': 'y' * 900}},
        'cpu_baseline': {'value': 0.61, 'unit': 'images/s', 'cores': 32, 'kind': 'port', 'sample': 's' * 400,
                         'postproc': {'value': 9.1, 'unit': 'images/s', 'ms_per_image': 110.0, 'cores': 8, 'kind': 'port',
                                      'gpu_us_per_image': 12.8, 'gpu_vs_cpu': 8600.0, 'sample': 's' * 400}},
        'other_configs': [{'key': 'leg%d' % i, 'leg': 'l' * 80, 'value': nan if i % 2 else 900.0 + i, 'unit': 'images/s', 'ms_per_step': inf,
                           'loss': {'focal': nan, 'box': -inf, 'finite': False}, 'roofline': {'frac': nan},
                           'latency_bound': {'select': {'ratio': 9.0}}, 'bulk': 'z' * 3000} for i in range(legs)],
    }


def test_headline_is_strict_json_under_4k_even_with_nan_legs():
    line = bench.headline_line(_full_record())
    assert '\n' not in line and len(line) < 4096
    d = _strict_loads(line)                                   # a bare NaN / Infinity raises here
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['value'] == 1100.12 and d['roofline']['frac'] == 0.6625 and d['cpu_baseline']['postproc']['gpu_vs_cpu'] == 8600.0
    assert d['other_configs']['leg1']['value'] is None and d['other_configs']['leg1']['loss'] == {'focal': None, 'box': None, 'finite': False}
    assert d['other_configs']['leg0']['value'] == 900.0
    assert 'bulk' not in line and 'epilogue_roofline' not in d       # legs' records and quoted figures stay in the detail file


def test_headline_sheds_optional_objects_rather_than_grow():
    rec = _full_record(legs=60)
    rec['data'] = 'd' * 3000
    line = bench.headline_line(rec)
    d = _strict_loads(line)
    assert len(line) < 4096 and d['value'] == 1100.12 and d['roofline'] and d['cpu_baseline'] and d['dropped_to_fit']


def test_sanitize_handles_tensors_and_numpy():
    import numpy as np
    import torch
    out = bench.sanitize({'a': torch.tensor(float('nan')), 'b': np.float32(2.5), 'c': (1, float('inf')), 4: torch.tensor(3)})
    assert out == {'a': None, 'b': 2.5, 'c': [1, None], '4': 3}


def _run_bench(tmp_path, *flags):
    detail = str(tmp_path / 'detail.json')
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    env['OMP_NUM_THREADS'] = '2'
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--device', 'cpu', '--tiny', '--steps', '2', '--warmup', '1',
                        '--detail-out', detail] + list(flags), env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                          # ONE stdout line, whatever the ranks and legs did
    return _strict_loads(lines[0]), json.load(open(detail))


def test_bare_gpus_2_spawns_its_own_ranks_infer(tmp_path):
    """`python bench.py --gpus 2` with no launcher (reference odtk/main.py:246-250 spawns its own workers): the whole N>1 path
    -- rendezvous, barrier + MAX-over-ranks bracket, rank 0's one line -- on gloo."""
    d, detail = _run_bench(tmp_path, '--gpus', '2')
    assert d['n_gpus'] == 2 and d['steps'] == 2 and d['warmup'] == 1 and d['config']['global_batch'] == 2
    assert d['config']['backend'] == 'gloo' and d['config']['device'] == 'cpu' and d['scaling'] == 'weak'
    assert abs(d['value'] - 2 * 2 / (d['ms_per_step'] * 2e-3)) / d['value'] < 0.01      # value = all ranks' images / the MAX time
    assert detail['value'] == d['value']


def test_bare_gpus_2_train_reports_exposed_allreduce(tmp_path):
    d, _ = _run_bench(tmp_path, '--gpus', '2', '--mode', 'train')
    assert d['n_gpus'] == 2 and 'training' in d['metric'] and d['config']['parallelism'].startswith('ddp x2 (gloo')
    assert d['exposed_allreduce_ms'] is not None and d['loss']['finite'] is True


def test_launcher_world_mismatch_is_an_error_not_an_assert(tmp_path):
    env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--device', 'cpu', '--tiny', '--gpus', '2', '--steps', '1'],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and 'one rank per GPU' in r.stderr and r.stdout.strip() == ''


def test_committed_pmc_traffic_is_for_these_kernel_sources():
    """`roofline.traffic` goes on the bench line only when the newest committed PMC file was measured on THIS tree's kernel
    sources (bench.kernel_src_hash over csrc/*.hpp + *.hip).  A kernel edit without a refreshed `tools/profile_round.sh` run
    would silently turn the field into null on the driver's line: fail here instead."""
    newest = next(n for n in ('r06_pmc_traffic.json', 'r05_pmc_traffic.json') if os.path.isfile(os.path.join(bench.ROOT, 'profiles', n)))
    pmc = json.load(open(os.path.join(bench.ROOT, 'profiles', newest)))
    assert pmc['kernel_src_sha16'] == bench.kernel_src_hash(), (newest, 'refresh with tools/profile_round.sh and copy the summaries into profiles/')
    key = pmc['bf16_logits_channels_last']
    assert key['scores_per_launch'] == 122860800                     # 15 357 600 scores per image x 8
    assert 1.0 <= key['traffic_bytes'] / float(key['algorithmic_bytes']) <= 1.05
